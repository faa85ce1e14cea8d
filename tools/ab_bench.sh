#!/bin/bash
# A/B: bench.py's timed region with alternative builds of the library (PGW_LIB), same box, back to back.
# usage: bash tools/ab_bench.sh [ENV=VALUE:]lib_a.so lib_b.so ...   (paths relative to the repo root; results in gpurun_out/ab/)
mkdir -p gpurun_out/ab
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=""; lib=$spec
  if [[ "$spec" == *:* ]]; then envs="${spec%%:*}"; lib="${spec#*:}"; fi
  name=$(basename $lib .so)_$i
  env $envs PGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --overlap-streams 0 --steps 10 --warmup 2 ${AB_ARGS} > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || echo "$name failed"
  python - <<PY
import json
d=json.load(open('gpurun_out/ab/$name.json'))
print('$spec', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
PY
done
