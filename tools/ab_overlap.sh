#!/bin/bash
# A/B of the two-files-in-flight region of bench.py (two contexts / streams / host threads) with alternative builds.
mkdir -p gpurun_out/ab
i=0
for lib in "$@"; do
  i=$((i+1)); name=ov_$(basename $lib .so)_$i
  PGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --overlap-streams ${OV:-2} --steps 12 --warmup 2 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || echo "$name failed"
  python - <<PY
import json
d=json.load(open('gpurun_out/ab/$name.json'))
print('$lib', d['ms_per_step'], d['overlap'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
PY
done
