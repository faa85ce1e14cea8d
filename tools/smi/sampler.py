"""Background sampler of the card's own telemetry (gfx clock per XCD, socket power, temperatures, power-cap residency) over a
measurement: `bench.py` attaches the result to its line as `device_state`, `tools/clock_probe.sh` prints it.  Measurement
infrastructure only: the product path never imports this.  Needs `tools/smi/libsmi_probe.so` (built by `__graft_entry__.build()`
from `smi_probe.c`); without it, or on a box whose SMI does not answer, `Sampler.available` is False and everything is a no-op.
"""
import ctypes
import os
import threading
import time

FIELDS = ('gfx_mhz', 'gfx_mhz_min_xcd', 'gfx_mhz_max_xcd', 'uclk_mhz', 'power_w', 't_hotspot_c', 't_mem_c', 'acc_counter',
          'ppt_acc', 'socket_thm_acc', 'hbm_thm_acc', 'prochot_acc', 'vr_thm_acc', 'power_cap_w', 'gfx_activity', 'umc_activity')
_HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    path = os.path.join(_HERE, 'libsmi_probe.so')
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.smi_sample.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
        if lib.smi_nfields() != len(FIELDS) or lib.smi_open() != 0:
            return None
        return lib
    except OSError:
        return None


class Sampler:
    def __init__(self, device=0, period_s=0.004):
        self.lib = _load()
        self.device = device
        self.period = period_s
        self.samples = []                  # (t_monotonic, {field: value})
        self._stop = threading.Event()
        self._thread = None
        self.available = self.lib is not None and self.sample() is not None

    def sample(self):
        if self.lib is None:
            return None
        buf = (ctypes.c_double * len(FIELDS))()
        if self.lib.smi_sample(self.device, buf) != 0:
            return None
        return {k: buf[i] for i, k in enumerate(FIELDS)}

    def _run(self):
        while not self._stop.is_set():
            s = self.sample()
            if s is not None:
                self.samples.append((time.monotonic(), s))
            self._stop.wait(self.period)

    def start(self):
        if self.available and self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        return self

    def window(self, t0, t1):
        """Summary of the samples with t0 <= t <= t1 (time.monotonic()), or of all of them when both are None."""
        rows = [s for t, s in self.samples if (t0 is None or t >= t0) and (t1 is None or t <= t1)]
        if not rows:
            return None
        def col(k):
            return [r[k] for r in rows if r[k] >= 0]
        def mean(v):
            return round(sum(v) / len(v), 1) if v else None
        out = {'samples': len(rows)}
        g = col('gfx_mhz')
        out['gfx_mhz_mean'] = mean(g)
        out['gfx_mhz_min'] = min(col('gfx_mhz_min_xcd'), default=None)
        out['gfx_mhz_max'] = max(col('gfx_mhz_max_xcd'), default=None)
        out['uclk_mhz'] = mean(col('uclk_mhz'))
        out['power_w_mean'] = mean(col('power_w'))
        out['power_w_max'] = max(col('power_w'), default=None)
        out['t_hotspot_c_max'] = max(col('t_hotspot_c'), default=None)
        out['t_mem_c_max'] = max(col('t_mem_c'), default=None)
        first, last = rows[0], rows[-1]
        d = last['acc_counter'] - first['acc_counter']
        if first['acc_counter'] >= 0 and d > 0:
            for k in ('ppt', 'socket_thm', 'hbm_thm', 'prochot', 'vr_thm'):
                a, b = first[k + '_acc'], last[k + '_acc']
                if a >= 0 and b >= 0:
                    out[k + '_residency'] = round((b - a) / d, 3)   # share of the window the limiter was active
        return out

    def summary(self, t0=None, t1=None):
        if not self.available:
            return None
        s = self.sample() or {}
        return {'source': 'ROCm SMI gpu_metrics (tools/smi)', 'period_ms': self.period * 1e3,
                'power_cap_w': s.get('power_cap_w'),
                'timed_region': self.window(t0, t1), 'whole_run': self.window(None, None)}
