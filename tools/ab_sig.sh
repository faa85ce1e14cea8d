#!/bin/bash
# A/B of the signature-faithful and step_02 kernels (bench.py's side measurements) with alternative library builds.
# usage: bash tools/ab_sig.sh lib_a.so lib_b.so ...
mkdir -p gpurun_out/ab
i=0
for lib in "$@"; do
  i=$((i+1)); name=sig_$(basename $lib .so)_$i
  PGW_LIB=$PWD/$lib python bench.py --no-cpu-baseline --overlap-streams 0 --e2e-files 0 --steps 4 --warmup 1 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || echo "$name failed"
  python - <<PY
import json
d=json.load(open('gpurun_out/ab/$name.json'))
m={k:(v.get('avg_ms') if isinstance(v,dict) else v) for k,v in d['signature_kernels'].items()}
e=d['extras']
print('$lib', m, {k:e[k].get('kernel_ms') for k in ('regrid_one_var_12_months','harmonic_smooth_daily_19lev','byteswap_one_field')})
PY
done
