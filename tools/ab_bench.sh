#!/bin/bash
# A/B: bench.py's timed region with alternative builds of the library (PGW_LIB), same box, back to back.
# usage: bash tools/ab_bench.sh lib_a.so lib_b.so ...   (paths relative to the repo root; results in gpurun_out/ab/)
mkdir -p gpurun_out/ab
for lib in "$@"; do
  name=$(basename $lib .so)
  PGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --overlap-streams 0 --steps 10 --warmup 2 ${AB_ARGS} > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || echo "$name failed"
  python - <<PY
import json
d=json.load(open('gpurun_out/ab/$name.json'))
print('$name', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
PY
done
