#!/bin/bash
# Same-box A/B of settings.placement: bench.py's timed region with plain allocations and with the level arrays spread over the
# card's memory regions (device.SpreadPool), alternating.  usage (GPU box): bash tools/ab_placement.sh [rounds] [bench flags]
n=${1:-2}; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $n); do for m in plain spread; do
  PGW_PLACEMENT=$m python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline "$@" > gpurun_out/ab/pl_${m}_$i.json 2> gpurun_out/ab/pl_${m}_$i.err || { tail -5 gpurun_out/ab/pl_${m}_$i.err; exit 1; }
  python3 - <<PY
import json
j=json.loads([l for l in open("gpurun_out/ab/pl_${m}_$i.json") if l.startswith("{")][-1])
r=j["roofline"]; k=j["kernels"]
print("$m", "ms/file", j["ms_per_step"], "quad", r["avg_launch_ms"], r["frac"], "bare", (r.get("bare_pattern_same_arrays") or {}).get("GBps"), "loop", k["ps_loop_multi"]["avg_ms"], "final", k["finalize"]["avg_ms"], j["placement"])
PY
done; done
