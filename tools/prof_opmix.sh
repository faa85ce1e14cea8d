#!/bin/bash
# VALU instruction mix of the file-path kernels (wave-instructions per launch by class) - which arithmetic a VALU-bound
# kernel spends its issue slots on.  usage: bash tools/prof_opmix.sh [bench args, e.g. --storage f32]
export TMPDIR=/tmp
out=gpurun_out/prof_opmix
mkdir -p $out
# OPMIX_PROG: another program than bench.py's timed region, e.g. OPMIX_PROG="python3 tools/reinterp_time.py"
P="${OPMIX_PROG:-python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-streams 0 --no-extras} $@"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $out/f64 -- $P > $out/a.json 2> $out/a.err || echo "f64 pass failed"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/f32 -- $P > $out/b.json 2> $out/b.err || echo "f32 pass failed"
python3 - <<'PY'
import csv, glob, collections
for sub in ('f64', 'f32'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('gpurun_out/prof_opmix/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
    for f in glob.glob('gpurun_out/prof_opmix/%s/**/*kernel_trace.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r['Kernel_Name'][:60]] += 1
    for k, v in acc.items():
        if any(s in k for s in ('quad', 'multi', 'finalize', 'reinterp')):
            n = max(cnt[k], 1)
            print(sub, k, 'launches', n, {c: round(x / n / 1e6, 1) for c, x in sorted(v.items())}, '(M wave-instructions per launch)')
PY
find $out -name "*counter_collection.csv" -size +3M -delete
