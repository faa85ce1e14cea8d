#!/bin/bash
# A/B of the reference's other two modes (tools/modes_run.py: p_ref_inp = None, i_reinterp = 1) with alternative builds (PGW_LIB)
# usage: bash tools/ab_modes.sh lib_a.so lib_b.so ...   [AB_ARGS="--storage f32"]
mkdir -p gpurun_out/ab
for lib in "$@"; do
  name=$(basename $lib .so)
  PGW_LIB=$PWD/$lib python tools/modes_run.py ${AB_ARGS} > gpurun_out/ab/modes_$name.json 2> gpurun_out/ab/modes_$name.err || echo "$name failed"
  python - <<PY
import json
d=json.load(open('gpurun_out/ab/modes_$name.json'))
for k,v in d.items():
    print('$lib', k, v['ms_per_file'], v['iterations'], {a:b['avg_launch_ms'] for a,b in v['kernels'].items() if b['avg_launch_ms']>0.1})
PY
done
