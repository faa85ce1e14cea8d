#!/usr/bin/env python
"""HBM ceilings of the box at hand, for the `bound` column of DESIGN.md section 4: device fill (write only), device copy
(read + write) and a read-only reduction, 2 GB operands, torch kernels timed with events.  Not part of the product."""
import json
import torch

n = 1 << 28                                   # 2^28 doubles = 2.1 GB
a = torch.empty(n, dtype=torch.float64, device='cuda')
b = torch.empty(n, dtype=torch.float64, device='cuda')
out = {}


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 8 / 1e9
ms = timed(lambda: a.fill_(1.5))
out['fill_write_only'] = dict(ms=round(ms, 4), GBps=round(gb / ms * 1e3, 1))
ms = timed(lambda: b.copy_(a))
out['copy_read_write'] = dict(ms=round(ms, 4), GBps=round(2 * gb / ms * 1e3, 1))
ms = timed(lambda: a.sum())
out['sum_read_only'] = dict(ms=round(ms, 4), GBps=round(gb / ms * 1e3, 1))
print(json.dumps(out))
