#!/bin/bash
# PMC picture of the i_reinterp = 1 kernels (tools/reinterp_time.py under rocprofv3): VALU / wait shares, LDS, traffic
export TMPDIR=/tmp
out=gpurun_out/prof_reinterp
mkdir -p $out
P="python3 tools/reinterp_time.py"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -- $P > $out/sq.json 2> $out/sq.err || echo "sq failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/lds -- $P > $out/lds.json 2> $out/lds.err || echo "lds failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $P > $out/f.json 2> $out/f.err || echo "fetch failed"
python3 - <<'PY'
import csv, glob, collections
for sub in ('sq', 'lds', 'fetch'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('gpurun_out/prof_reinterp/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']] += float(r['Counter_Value'])
    for f in glob.glob('gpurun_out/prof_reinterp/%s/**/*kernel_trace.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r['Kernel_Name'][:40]] += 1
    for k, v in acc.items():
        if any(s in k for s in ('reinterp', 'humidity', 'adjust')):
            n = max(cnt[k], 1)
            print(sub, k, 'launches', n, {c: round(x / n) for c, x in sorted(v.items())})
PY
find $out -name "*counter_collection.csv" -size +3M -delete
find $out -name "*kernel_trace.csv" -size +3M -delete
