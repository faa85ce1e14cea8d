#!/usr/bin/env python
"""What is the best the bare 4 read + 4 write pattern gets from ANY choice of eight arrays out of a pool?  Random subsets of a
classified pool (two-write-stream probe against one representative per class), the probe's rate per subset, the best and the
worst with their class compositions.  Exploration for device.SpreadPool.  Usage: POOL=120 python tools/placement_search.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd.device import default_context

ctx = default_context()
P = int(os.environ.get('POOL', 120))
shape = (1, 137, 721, 1440)
pool = [ctx.empty(shape, np.float64) for _ in range(P)]
unclassified = list(range(P)); classes = []
while unclassified and len(classes) < 6:
    r = unclassified[0]
    rates = {k: ctx.placement_probe([], [pool[r], pool[k]], rows=64) for k in unclassified[1:]}
    if not rates or max(rates.values()) - min(rates.values()) < 0.10 * max(rates.values()):
        classes.append(unclassified); unclassified = []; break
    mid = 0.5 * (min(rates.values()) + max(rates.values()))
    members = [r] + [k for k, v in rates.items() if v < mid]
    classes.append(members); unclassified = [k for k in unclassified if k not in members]
if unclassified:
    classes.append(unclassified)
cls_of = {k: i for i, c in enumerate(classes) for k in c}
print('classes', [len(c) for c in classes]); print(''.join(str(cls_of[k]) for k in range(P)))
rng = np.random.default_rng(0)
rows = []
for trial in range(int(os.environ.get('TRIALS', 60))):
    idx = [int(i) for i in rng.choice(P, size=8, replace=False)]
    g = ctx.placement_probe([pool[i] for i in idx[:4]], [pool[i] for i in idx[4:]], reps=3)
    rows.append((g, idx))
rows.sort(reverse=True)
for g, idx in rows[:8] + rows[-5:]:
    print('%5.0f GB/s  in %s out %s   pool indices %s' % (g, [cls_of[i] for i in idx[:4]], [cls_of[i] for i in idx[4:]], idx))
# adjacent runs of eight consecutive allocations for comparison
for start in (0, P // 3, 2 * P // 3):
    idx = list(range(start, start + 8))
    g = ctx.placement_probe([pool[i] for i in idx[0::2]], [pool[i] for i in idx[1::2]], reps=3)
    print('consecutive %3d..%3d alternating in/out: %5.0f GB/s classes %s' % (start, start + 7, g, [cls_of[i] for i in idx]))
