#!/bin/bash
# extra PMC passes for a stall picture of the file-path kernels: instruction cache, branches, LDS, scalar unit
export TMPDIR=/tmp
out=gpurun_out/prof_stall
mkdir -p $out
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --no-extras $@"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/icache -- $P > $out/a.json 2> $out/a.err || echo "icache pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/lds -- $P > $out/b.json 2> $out/b.err || echo "lds pass failed"
python3 - <<'PY'
import csv, glob, collections
for sub in ('icache', 'lds'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('gpurun_out/prof_stall/%s/*/*counter_collection.csv' % sub):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:48]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for f in glob.glob('gpurun_out/prof_stall/%s/*/*kernel_trace.csv' % sub):
        for r in csv.DictReader(open(f)):
            cnt[r['Kernel_Name'][:48]] += 1
    for k, v in acc.items():
        if any(s in k for s in ('quad', 'multi', 'finalize')):
            n = max(cnt[k], 1)
            print(sub, k, 'launches', n, {c: round(x / n) for c, x in sorted(v.items())})
PY
find $out -name "*counter_collection.csv" -size +3M -delete
