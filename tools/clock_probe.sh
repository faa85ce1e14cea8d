#!/bin/bash
# The quad kernel's 10 % spread between boxes and runs against what the card reports about itself:
# bench.py lines of 10, 300 and 3000 files back to back, each with its `device_state` (gfx clock per XCD, socket power against
# the cap, temperatures, power-cap residency; tools/smi).  Usage (GPU box): bash tools/clock_probe.sh [tag] [bench flags]
tag=${1:-probe}; shift
mkdir -p gpurun_out
for k in 10 300 3000 10; do
  python3 bench.py --steps $k --warmup 2 --no-extras --no-cpu-baseline "$@" > gpurun_out/clock_${tag}_$k.json 2> gpurun_out/clock_${tag}_$k.err || { tail -5 gpurun_out/clock_${tag}_$k.err; exit 1; }
  python3 - <<PY
import json
j=json.loads([x for x in open("gpurun_out/clock_${tag}_$k.json") if x.startswith("{")][-1])
d=j.get("device_state") or {}
print("steps", $k, "ms/file", j["ms_per_step"], "quad ms", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"], "| cap", d.get("power_cap_w"), "timed", d.get("timed_region"))
PY
done
