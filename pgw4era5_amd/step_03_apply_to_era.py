"""
step_03: impose the climate deltas on ERA5 files and re-balance surface pressure, on MI355X.

Mirror of the reference's `step_03_apply_to_era.py` (same `pgw_for_era5` signature, same
command-line flags, `-p` = number of worker ranks) with the per-file compute path running as
HIP kernels on device-resident arrays:

    RELHUM = q->RH(QV, pa(PS), T)                       reference step_03:64-94
    sea ice / skin / soil temperature riders            :103-146
    ta, hur, ua, va: time-lerp + surface insert + ln-p interpolation + add      :155-173
    fixed-point loop on delta_ps until max|phi error| <= 0.15                   :182-319
    PS, T, QV, U, V, T_SKIN, T_SO, FR_SEA_ICE written back                      :369-378

The twelve monthly records of every delta live in HBM for the whole run (`DeltaSet`; ~10 GB
fp64 at 0.25 deg for ta,hur,ua,va,zg on plev19 - the reference re-reads them from disk for
every file and `zg` once per iteration, functions.py:203 / step_03:292).
"""
import ctypes as C
import datetime as _dt
import os

import numpy as np

from . import _lib
from . import settings as S
from .constants import CON_G, CON_RD   # noqa: F401
from .device import DeviceArray, default_context, dtype_tag, ptr

_dp = C.POINTER(C.c_double)


# ----------------------------------------------------------------------------------------
# time bracketing of load_delta (reference functions.py:224-283), host control logic
# ----------------------------------------------------------------------------------------
def _to_dt64(t):
    return np.datetime64(t).astype('datetime64[s]')


def _reyear(times, year):
    """`dt64_to_dt(t).replace(year=year)` of functions.py:235-238 for an array of stamps: month, day and time of day kept."""
    y = times.astype('datetime64[Y]')
    m = times.astype('datetime64[M]')
    month = (m - y.astype('datetime64[M]')).astype(np.int64)                     # 0 .. 11
    rest = times - m.astype('datetime64[s]')                                       # day of month and time of day
    first = (np.datetime64('%04d' % year, 'Y').astype('datetime64[M]') + month).astype('datetime64[s]')
    return first + rest


def delta_time_bracket(delta_times, target):
    """Indices and (re-yeared) stamps of the records bracketing `target`, periodic in the
    year; Feb-29 dropped first.  Returns (ind_before, ind_after, x_hi, x_new, keep) with
    x_* = float nanoseconds relative to the 'before' stamp (what xarray hands to scipy).
    Array arithmetic on datetime64 (a daily axis has 365 stamps and this runs in front of every file)."""
    times = np.asarray(delta_times).astype('datetime64[s]')
    target = _to_dt64(target)
    m = times.astype('datetime64[M]')
    month = (m - times.astype('datetime64[Y]').astype('datetime64[M]')).astype(np.int64)
    day = ((times - m.astype('datetime64[s]')).astype(np.int64)) // 86400        # 0-based day of the month
    is_leap_day = np.nonzero((month == 1) & (day == 28))[0]                       # :224-230: the LAST Feb 29 found is dropped
    leap = int(is_leap_day[-1]) if len(is_leap_day) else None
    keep = np.array([i for i in range(len(times)) if i != leap], dtype=np.int64)
    year = int(str(target)[:4])
    ty = _reyear(times[keep], year)                                               # :235-238
    before = ty <= target                                                         # :242-243
    if before.any():
        ib = int(np.nonzero(before)[0][-1]); tb = ty[ib]
    else:                                                                         # :253-258
        ib = len(ty) - 1; tb = _reyear(ty[ib:ib + 1], year - 1)[0]
    after = ty >= target                                                          # :262-263
    if after.any():
        ia = int(np.nonzero(after)[0][0]); ta = ty[ia]
    else:                                                                         # :273-278
        ia = 0; ta = _reyear(ty[:1], year + 1)[0]
    ns = 'datetime64[ns]'
    x_hi = float((ta.astype(ns) - tb.astype(ns)).astype(np.int64))
    x_new = float((target.astype(ns) - tb.astype(ns)).astype(np.int64))
    if ib == ia:                                                                  # :282-283
        x_hi = 0.0; x_new = 0.0
    return ib, ia, x_hi, x_new, keep


# ----------------------------------------------------------------------------------------
# deltas resident in HBM
# ----------------------------------------------------------------------------------------
class DeltaSet:
    """The climate deltas of one run on the device.

    arrays: dict var -> [nrec, (nplev,) nlat, nlon] for ta,hur,ua,va,zg (4-D) and tas,hurs,ts,tos,siconc,ps_hist (3-D): a
    host array, a DeviceArray, or a record provider (`ncio.RecordReader`: .nrec, .rec_shape, .read_record(r)).  `plev` in
    file order (descending for CMIP).

    Time axes: the reference loads every delta file on its own (load_delta, functions.py:195-303), so each variable has
    its own time axis - `times_by_var[var]`, default `delta_times` - and is bracketed on it (monthly tos / siconc beside
    daily 3-D deltas work).

    HBM budget rule: all records of all variables stay resident when they fit into `PGW_DELTA_HBM_FRACTION` (default
    0.35) of the device's memory - 12 monthly records of everything at 0.25 deg are 10 GB in float64.  Otherwise (365
    daily records of five 19-level variables: 288 GB in float64, 144 GB in float32) a variable keeps a window of
    `WINDOW` records on the device: consecutive ERA5 files need the same two, a new record is uploaded when the instant
    moves past one (least recently used record dropped).  resident=True / False forces either."""

    VARS_3D = ('ta', 'hur', 'ua', 'va', 'zg')
    VARS_2D = ('tas', 'hurs', 'ts', 'tos', 'siconc', 'ps_hist')
    WINDOW = 4

    def __init__(self, ctx, arrays, delta_times, plev, dtype, times_by_var=None, resident=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.times = None if delta_times is None else np.asarray(delta_times).astype('datetime64[s]')
        self.plev = np.ascontiguousarray(plev, dtype=np.float64)
        self.plev_zg = self.plev                                    # zg_delta.nc may bring its own plev axis (load_delta_set)
        self.times_by_var = {}
        self.dev, self._src, self._cache, self._lock = {}, {}, {}, __import__('threading').Lock()
        names = [k for k in self.VARS_3D + self.VARS_2D if k in arrays]
        for k in names:
            t = (times_by_var or {}).get(k)
            t = self.times if t is None else np.asarray(t).astype('datetime64[s]')
            if t is None:
                raise ValueError('no time axis for %s' % k)
            if self._nrec(arrays[k]) != len(t):
                raise ValueError('time axis of %s has %d stamps, the array %d records' % (k, len(t), self._nrec(arrays[k])))
            self.times_by_var[k] = t
        if self.times is None and 'ta' in self.times_by_var:
            self.times = self.times_by_var['ta']
        # variables with equal time axes share one bracket computation per instant (the bracket is host work in front of
        # every file's one C call: a dozen of them per file cost 0.2 ms of a 3.4 ms file)
        self._axis_of, axes = {}, []
        for k, t in self.times_by_var.items():
            for i, u in enumerate(axes):
                if len(u) == len(t) and bool(np.all(u == t)):
                    self._axis_of[k] = i
                    break
            else:
                self._axis_of[k] = len(axes)
                axes.append(t)
        self._axes = axes
        self._bracket_cache = {}
        total = sum(self._nrec(arrays[k]) * self._rec_elems(arrays[k]) for k in names) * self.dtype.itemsize
        if resident is None:
            forced = os.environ.get('PGW_DELTA_RESIDENT')
            if forced in ('0', '1'):
                resident = forced == '1'
            else:
                resident = total <= self.budget_bytes(ctx)
        self.resident, self.total_bytes = bool(resident), int(total)
        for k in names:
            a = arrays[k]
            if isinstance(a, DeviceArray):
                self.dev[k] = a
            elif self.resident:
                if hasattr(a, 'read_record'):
                    a = np.stack([a.read_record(r) for r in range(a.nrec)])
                self.dev[k] = ctx.to_device(np.ascontiguousarray(a, dtype=self.dtype), self.dtype)
            else:
                self._src[k] = a
                self._cache[k] = {}
        # annual-mean skin-temperature delta (step_03:134-136): depends on the delta file only
        if 'ts' in arrays:
            _, _, _, _, keep = delta_time_bracket(self.times_by_var['ts'], self.times_by_var['ts'][0])
            ts = arrays['ts']
            if isinstance(ts, DeviceArray):
                mean = ts.numpy()[keep].astype(np.float64).mean(axis=0)
            elif hasattr(ts, 'read_record'):
                mean = np.zeros(ts.rec_shape, dtype=np.float64)
                for r in keep:                                      # one record at a time: the file may not fit in host memory
                    mean += ts.read_record(int(r)).astype(np.float64)
                mean /= len(keep)
            else:
                mean = np.asarray(ts)[keep].astype(np.float64).mean(axis=0)
            self.ts_clim = ctx.to_device(mean.astype(self.dtype), self.dtype)
        else:
            self.ts_clim = None

    @staticmethod
    def _nrec(a):
        return int(a.nrec) if hasattr(a, 'read_record') else int(a.shape[0])

    @staticmethod
    def _rec_elems(a):
        shape = a.rec_shape if hasattr(a, 'read_record') else a.shape[1:]
        return int(np.prod(shape, dtype=np.int64))

    @staticmethod
    def budget_bytes(ctx):
        import ctypes
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        ctx._check(ctx.lib.pgw_mem_info(ctx.handle, ctypes.byref(free), ctypes.byref(total)))
        return int(float(os.environ.get('PGW_DELTA_HBM_FRACTION', '0.35')) * total.value)

    def __contains__(self, var):
        return var in self.times_by_var

    def rec(self, var, r):
        """Record `r` of `var` on the device: a slab of the resident array, or the window's copy (uploaded now if absent)."""
        r = int(r)
        if var in self.dev:
            return self.dev[var].slab(r)
        with self._lock:
            cache = self._cache[var]
            if r in cache:
                cache[r] = cache.pop(r)                             # most recently used last
                return cache[r]
            src = self._src[var]
            host = src.read_record(r) if hasattr(src, 'read_record') else src[r]
            if len(cache) >= self.WINDOW:
                old = next(iter(cache))
                arr = cache.pop(old)                                # reuse the buffer of the least recently used record
                self.ctx.sync()                                     # no kernel may still read it
                arr.copy_from(np.ascontiguousarray(host, dtype=self.dtype))
            else:
                arr = self.ctx.to_device(np.ascontiguousarray(host, dtype=self.dtype), self.dtype)
            cache[r] = arr
            return arr

    def same_axis(self, var, ref='ta'):
        return self._axis_of[var] == self._axis_of[ref]

    def bracket(self, target, var=None):
        if var is None:
            return delta_time_bracket(self.times, target)
        key = (self._axis_of[var], target)
        hit = self._bracket_cache.get(key)
        if hit is None:
            if len(self._bracket_cache) > 64:
                self._bracket_cache.clear()
            hit = self._bracket_cache[key] = delta_time_bracket(self._axes[key[0]], target)
        return hit

    def pair(self, var, target, scratch):
        """(record before, record after, x_hi, x_new) of `var` for the instant, bracketed on its OWN time axis
        (functions.py:240-283).  x_hi == 0: the instant is a record (the 'after' record is the same array)."""
        ib, ia, x_hi, x_new, keep = self.bracket(target, var)
        return self.rec(var, keep[ib]), self.rec(var, keep[ia]), x_hi, x_new

    def pair_on_axis_of(self, var, target, main_x, scratch):
        """The pair of `var` for kernels that interpolate a group of variables with ONE (x_hi, x_new) = main_x, the
        bracket of `ta`: the records themselves when `var` shares ta's time axis, else `var` interpolated here on its own
        axis and handed over as both records (b + (b - b) / x_hi * x_new = b).  `scratch(name, shape)` supplies the buffer.
        (A float32 run holds that one field rounded to float32, where the fused path keeps the float64 value.)"""
        if self.same_axis(var):
            ib, ia, _, _, keep = self.bracket(target, 'ta')
            return self.rec(var, keep[ib]), self.rec(var, keep[ia])
        b, a, x_hi, x_new = self.pair(var, target, scratch)
        if x_hi == 0.0:
            return b, b
        out = scratch('_pre_' + var, b.shape)
        self.ctx._check(self.ctx.lib.pgw_time_lerp(self.ctx.handle, dtype_tag(self.dtype), b.size, b.ptr, a.ptr, x_hi, x_new,
                                                   out.ptr))
        return out, out

    def lerp2d(self, name, target, out=None):
        """Time-interpolated delta record (load_delta, functions.py:195-303) -> (1,) + record shape."""
        b, a, x_hi, x_new = self.pair(name, target, None)
        if out is None:
            out = self.ctx.empty((1,) + b.shape, self.dtype)
        if x_hi == 0.0:
            self.ctx._check(self.ctx.lib.pgw_memcpy_d2d(self.ctx.handle, out.ptr, b.ptr, b.nbytes))
        else:
            self.ctx._check(self.ctx.lib.pgw_time_lerp(self.ctx.handle, dtype_tag(self.dtype), b.size, b.ptr, a.ptr,
                                                       x_hi, x_new, out.ptr))
        return out

    def free(self):
        for v in list(self.dev.values()) + [x for c in self._cache.values() for x in c.values()] + [self.ts_clim]:
            if v is not None:
                v.free()
        self.dev, self._cache, self.ts_clim = {}, {}, None


# Placement classes of the level arrays (device.SpreadPool; only their balance matters): the write streams
# of the quad kernel - T, e (the library's workspace: class 1, Context.enable_placement), U, V on the hybrid levels - two and two;
# the final kernel reads e (1) and writes QV (0); the inputs two and two.
LEVEL_CLASS_IN = {'T': 0, 'QV': 1, 'U': 0, 'V': 1}
LEVEL_CLASS_OUT = {'T': 0, 'QV': 0, 'U': 0, 'V': 1}


def _upload_era(ctx, era, dtype):
    out = {}
    for k in ('PS', 'FIS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_LAND', 'FR_SEA_ICE'):
        v = era[k]
        if isinstance(v, DeviceArray):
            out[k] = v
        elif k in ('T', 'QV', 'U', 'V'):            # level fields: placed by the context (Context.level_array)
            h = np.ascontiguousarray(v, dtype=dtype)
            out[k] = ctx.level_array(h.shape, dtype, LEVEL_CLASS_IN[k]).copy_from(h)
        else:
            out[k] = ctx.to_device(np.ascontiguousarray(v, dtype=dtype), dtype)
    return out


def ref_dtype_mode(dtype, ref_dtype=None):
    """Whether a file of storage `dtype` runs in reference-dtype mode (settings.f32_file_mode; float32 files only)."""
    if np.dtype(dtype) != np.float32:
        if ref_dtype:
            raise ValueError('reference-dtype mode is the float32-file mode (float64 files already compute as the reference does)')
        return False
    if ref_dtype is None:
        if S.f32_file_mode not in ('reference', 'fast'):
            raise ValueError("settings.f32_file_mode must be 'reference' or 'fast'")
        return S.f32_file_mode == 'reference'
    return bool(ref_dtype)


def process_file_device(ctx, era, coeffs, deltas, target_dt, ignore_top_pressure_error=False,
                        p_ref=None, out=None, keep_hur=False, ref_dtype=None, i_reinterp=False):
    """The per-file compute path of pgw_for_era5 (reference step_03:62-346, i_reinterp = 0,
    fixed p_ref) on device arrays: ONE call into the C-ABI (`pgw_step03_file`).

    era: dict of DeviceArrays PS,FIS,(T_SKIN,FR_LAND,FR_SEA_ICE) (1,nlat,nlon); T,QV,U,V
    (1,N,nlat,nlon); T_SO (1,nsoil,nlat,nlon).  coeffs: dict ak,bk,[akm,bkm],soil1 (host).
    deltas: DeltaSet.  out: optional dict of preallocated output DeviceArrays (reused
    across files).  ref_dtype (float32 storage only; default settings.f32_file_mode): reference-dtype mode, T, QV, U, V
    come back as float64 arrays like the reference's `era + delta`.  i_reinterp: settings.i_reinterp = 1 (step_03:202-216,
    330-343: ta / hur and their deltas re-interpolated onto the current levels in every pass, ua / va after convergence) - the
    same call with `pgw_file_args.i_reinterp` set, every combination with the reference level and the storage modes.
    Returns (dict of DeviceArrays, info)."""
    lib, h = ctx.lib, ctx.handle
    if p_ref is None:
        p_ref = S.p_ref_inp
    local_p_ref = (p_ref is None) or (p_ref == 'local')          # settings.p_ref_inp = None, step_03:219-253
    dt = deltas.dtype
    ctx.set_levels(coeffs['ak'], coeffs['bk'], coeffs.get('akm'), coeffs.get('bkm'))
    T, PS = era['T'], era['PS']
    nt, N, nlat, nlon = T.shape
    if nt != 1:
        # the deltas are time-interpolated to ONE instant (functions.py:288-296); with more than one time step in
        # the file the reference fails in vert_interp_delta -> interp_logp_4d (functions.py:457-459)
        raise ValueError('Time dimension of input files is inconsistent!')
    out = {} if out is None else out
    ref = ref_dtype_mode(dt, ref_dtype)
    dt4 = np.dtype('float64') if ref else dt                        # era (float32) + delta (float64), step_03:170-173

    def buf(name, shape, dtype=dt):
        if name not in out or out[name].shape != tuple(shape) or out[name].dtype != dtype:
            out[name] = ctx.level_array(shape, dtype, LEVEL_CLASS_OUT[name]) if name in LEVEL_CLASS_OUT else ctx.empty(shape, dtype)
        return out[name]

    _, _, x_hi, x_new, _ = deltas.bracket(target_dt, 'ta')          # functions.py:224-283, the axis of the quad group
    plev = deltas.plev
    a = _lib.FileArgs()
    a.per_var_time = 1                                              # every delta file is bracketed on its own time axis
    a.i_reinterp = 1 if i_reinterp else 0
    zb, za, a.zg_x_hi, a.zg_x_new = deltas.pair('zg', target_dt, buf)
    if local_p_ref:
        if len(deltas.plev_zg) != len(plev) or np.any(deltas.plev_zg != plev):
            raise NotImplementedError('p_ref_inp = None with a zg delta on other pressure levels than ta / hur / ua / va')
        a.local_p_ref = 1
        a.zg3_b, a.zg3_a = zb.ptr, za.ptr
    else:
        kref = np.nonzero(deltas.plev_zg == p_ref)[0]               # .sel(plev=p_ref), step_03:294
        if len(kref) != 1:
            raise KeyError(p_ref)
        a.zg_b = zb.slab(int(kref[0])).ptr
        a.zg_a = za.slab(int(kref[0])).ptr
        a.p_ref = float(p_ref)
    a.dtype, a.ntime, a.nlev, a.nplev, a.ncol = dtype_tag(dt), nt, N, len(plev), nlat * nlon
    a.ignore_top = 1 if ignore_top_pressure_error else 0
    a.ref_dtype = 1 if ref else 0
    a.max_n_iter = int(S.max_n_iter)
    a.adj_factor, a.thresh = float(S.adj_factor), float(S.thresh_phi_ref_max_error)
    a.x_hi, a.x_new = x_hi, x_new
    for k in ('PS', 'FIS', 'T', 'QV', 'U', 'V'):
        setattr(a, k, era[k].ptr)
    a.plev = plev.ctypes.data_as(_dp)
    held = []                                                       # keeps the record arrays alive until the call returns
    for var, name in (('ta', 'ta'), ('hur', 'hur'), ('ua', 'ua'), ('va', 'va'), ('tas', 'tas'), ('hurs', 'hurs'),
                      ('ps_hist', 'pshist')):
        b_, a_ = deltas.pair_on_axis_of(var, target_dt, (x_hi, x_new), buf)
        held += [b_, a_]
        setattr(a, name + '_b', b_.ptr)
        setattr(a, name + '_a', a_.ptr)
    soil = None
    if 'FR_SEA_ICE' in era and 'siconc' in deltas:                  # surface riders, step_03:103-146
        soil = np.ascontiguousarray(coeffs['soil1'], dtype=np.float64)
        a.nsoil = len(soil)
        a.soil_depth = soil.ctypes.data_as(_dp)
        for k in ('T_SKIN', 'T_SO', 'FR_LAND', 'FR_SEA_ICE'):
            setattr(a, k, era[k].ptr)
        for var in ('siconc', 'ts', 'tos'):
            b_, a_, vx_hi, vx_new = deltas.pair(var, target_dt, buf)
            held += [b_, a_]
            setattr(a, var + '_b', b_.ptr)
            setattr(a, var + '_a', a_.ptr)
            setattr(a, var + '_x_hi', vx_hi)
            setattr(a, var + '_x_new', vx_new)
        a.ts_clim = deltas.ts_clim.ptr
        a.T_SKIN_out = buf('T_SKIN', era['T_SKIN'].shape).ptr
        a.T_SO_out = buf('T_SO', era['T_SO'].shape).ptr
        a.FR_SEA_ICE_out = buf('FR_SEA_ICE', era['FR_SEA_ICE'].shape).ptr
    a.PS_out = buf('PS', PS.shape).ptr
    for k in ('T', 'QV', 'U', 'V'):
        setattr(a, k + '_out', buf(k, T.shape, dt4).ptr)
    if keep_hur:
        a.hur_pgw_out = buf('_hur_pgw', T.shape, dt4).ptr
    ctx._check(lib.pgw_step03_file(h, C.byref(a)))
    info = dict(n_iter=a.n_iter, max_err=[a.max_err_hist[i] for i in range(min(a.n_iter, 32))],
                levels_touched=int(a.levels_touched), passes_launched=int(a.passes_launched))
    return out, info


def process_file_device_reinterp(ctx, era, coeffs, deltas, target_dt, ignore_top_pressure_error=False,
                                 p_ref=None, out=None, ref_dtype=None, keep_hur=True):
    """settings.i_reinterp = 1 (reference step_03:202-216, 330-343): process_file_device with `i_reinterp` - ONE call into
    the C-ABI; fixed or local reference level, float64 / float32 files, reference-dtype mode."""
    return process_file_device(ctx, era, coeffs, deltas, target_dt, ignore_top_pressure_error, p_ref=p_ref, out=out,
                               keep_hur=keep_hur, ref_dtype=ref_dtype, i_reinterp=True)


def process_file_device_reinterp_composed(ctx, era, coeffs, deltas, target_dt, ignore_top_pressure_error=False,
                                          p_ref=None, out=None):
    """The same path composed on the host from the function-level C-ABI entries (pgw_reinterp_pass per pass,
    pgw_reinterp_pair for ua / va, the humidity entries) - round 2's form, fixed p_ref, float64 arithmetic on the storage
    type; kept as an independent composition the one-call path is tested against bit for bit."""
    lib, h = ctx.lib, ctx.handle
    p_ref = S.p_ref_inp if p_ref is None else p_ref
    if p_ref is None or p_ref == 'local':
        raise NotImplementedError('the host-composed form knows the fixed reference level only')
    dt = deltas.dtype
    tag = dtype_tag(dt)
    ctx.set_levels(coeffs['ak'], coeffs['bk'], coeffs.get('akm'), coeffs.get('bkm'))
    T, QV, PS, FIS = era['T'], era['QV'], era['PS'], era['FIS']
    nt, N, nlat, nlon = T.shape
    if nt != 1:
        raise ValueError('Time dimension of input files is inconsistent!')      # functions.py:457-459
    ncol, n2 = nlat * nlon, nt * nlat * nlon
    out = {} if out is None else out
    f64 = np.dtype('float64')

    def buf(name, shape, dtype=dt):
        if name not in out or out[name].shape != tuple(shape) or out[name].dtype != dtype:
            out[name] = ctx.empty(shape, dtype)
        return out[name]

    _, _, x_hi, x_new, _ = deltas.bracket(target_dt, 'ta')
    plev = deltas.plev
    # records of the group the kernels interpolate with ta's (x_hi, x_new); a variable on another time axis comes
    # interpolated on its own (DeltaSet.pair_on_axis_of)
    R = {var: deltas.pair_on_axis_of(var, target_dt, (x_hi, x_new), buf)
         for var in ('ta', 'hur', 'ua', 'va', 'tas', 'hurs', 'ps_hist')}
    kref = np.nonzero(deltas.plev_zg == p_ref)[0]
    if len(kref) != 1:
        raise KeyError(p_ref)
    # ERA state: RELHUM (step_03:87-94), phi_ref_era, g*dzg (the 4-D pressure fields live in registers, k_reinterp_field)
    relhum = buf('_RELHUM', T.shape)
    ctx._check(lib.pgw_specific_to_relative_humidity_hybrid(h, tag, nt, ncol, QV.ptr, PS.ptr, T.ptr, relhum.ptr))
    phi_era = buf('_phi_era', PS.shape, f64)
    ctx._check(lib.pgw_phi_ref_hybrid(h, tag, nt, ncol, T.ptr, QV.ptr, PS.ptr, FIS.ptr, float(p_ref), None, phi_era.ptr))
    dzg = buf('_dzg', PS.shape)
    zb, za, zx_hi, zx_new = deltas.pair('zg', target_dt, buf)                   # zg_delta.nc on its own time axis
    zb, za = zb.slab(int(kref[0])), za.slab(int(kref[0]))
    if zx_hi == 0.0:
        ctx._check(lib.pgw_memcpy_d2d(h, dzg.ptr, zb.ptr, zb.nbytes))
    else:
        ctx._check(lib.pgw_time_lerp(h, tag, zb.size, zb.ptr, za.ptr, zx_hi, zx_new, dzg.ptr))
    # loop state in buffers that live across files (a hipMalloc / hipFree per file synchronises the device)
    dphi = buf('_dphi', PS.shape, f64)
    dphi.copy_from(dzg.numpy().astype(np.float64) * CON_G)                        # step_03:292-293 (2-D, once)
    delta_ps, adj_ps = buf('_delta_ps', PS.shape, f64), buf('_adj_ps', PS.shape, f64)
    for x in (delta_ps, adj_ps):                                                  # :182-184
        ctx._check(lib.pgw_memset(h, x.ptr, 0, x.nbytes))
    ps_pgw = buf('PS', PS.shape)

    def reinterp_pair(var0, var1, era0, era1, target0, target1):
        """interp_logp_4d(era, pa_era, pa_pgw, 'constant') + load_delta_interp(var, pa_pgw)  :209-216 for two variables on
        the same axes (ta + hur: one ps_hist; ua + va), one kernel"""
        def arr(*ptrs):
            return (C.c_void_p * 2)(*[p.ptr if hasattr(p, 'ptr') else p for p in ptrs])
        if var0 in ('ta', 'hur'):
            sb = arr(R[var0 + 's'][0], R[var1 + 's'][0])
            sa = arr(R[var0 + 's'][1], R[var1 + 's'][1])
            pb, pa_ = R['ps_hist'][0].ptr, R['ps_hist'][1].ptr
        else:
            sb = sa = pb = pa_ = None
        ctx._check(lib.pgw_reinterp_pair(h, tag, nt, len(plev), ncol, plev.ctypes.data_as(_dp),
                                         arr(R[var0][0], R[var1][0]), arr(R[var0][1], R[var1][1]),
                                         x_hi, x_new, sb, sa, pb, pa_, arr(era0, era1), PS.ptr, ps_pgw.ptr,
                                         1 if ignore_top_pressure_error else 0, arr(target0, target1)))

    ta_pgw, hur_pgw = buf('T', T.shape), buf('_hur_pgw', T.shape)
    err = np.inf
    it = 1
    hist = []
    max_err = C.c_double()
    def arr(*xs):
        return (C.c_void_p * 2)(*[x.ptr for x in xs])
    thermo = (arr(R['ta'][0], R['hur'][0]), arr(R['ta'][1], R['hur'][1]), x_hi, x_new,
              arr(R['tas'][0], R['hurs'][0]), arr(R['tas'][1], R['hurs'][1]),
              R['ps_hist'][0].ptr, R['ps_hist'][1].ptr)
    while err > S.thresh_phi_ref_max_error:                                       # :189
        # one call and one host round trip per pass: delta_ps += adj_ps, ps_pgw (:192-193); ta / hur re-interpolated onto
        # the new levels (:202-216); the pass on them (:262-308)
        ctx._check(lib.pgw_reinterp_pass(h, tag, nt, len(plev), ncol, plev.ctypes.data_as(_dp), *thermo, T.ptr, relhum.ptr,
                                         PS.ptr, FIS.ptr, phi_era.ptr, dphi.ptr, delta_ps.ptr, adj_ps.ptr, float(p_ref),
                                         float(S.adj_factor), 1 if ignore_top_pressure_error else 0, ps_pgw.ptr, ta_pgw.ptr,
                                         hur_pgw.ptr, C.byref(max_err)))
        err = max_err.value                                                       # :308
        hist.append(err)
        it += 1
        if it > S.max_n_iter:                                                     # :313-319
            raise ValueError('ERROR! Pressure adjustment did not converge')
    reinterp_pair('ua', 'va', era['U'], era['V'], buf('U', T.shape), buf('V', T.shape))   # :330-343
    ctx._check(lib.pgw_relative_to_specific_humidity_hybrid(h, tag, nt, ncol, hur_pgw.ptr, ps_pgw.ptr, ta_pgw.ptr,
                                                            buf('QV', T.shape).ptr))   # hus of the last pass, :262-266,370
    if 'FR_SEA_ICE' in era and 'siconc' in deltas:                                # surface riders :103-146
        s3 = era['T_SKIN'].shape
        soil = np.ascontiguousarray(coeffs['soil1'], dtype=np.float64)
        ctx._check(lib.pgw_surface_update(
            h, tag, nt, ncol, len(soil), soil.ctypes.data_as(_dp), era['FR_SEA_ICE'].ptr,
            deltas.lerp2d('siconc', target_dt, buf('_dsic', s3)).ptr, deltas.lerp2d('tos', target_dt, buf('_dtos', s3)).ptr,
            deltas.lerp2d('ts', target_dt, buf('_dts', s3)).ptr, era['FR_LAND'].ptr, deltas.ts_clim.ptr,
            era['T_SKIN'].ptr, era['T_SO'].ptr, buf('FR_SEA_ICE', s3).ptr, buf('_dts_comb', s3).ptr,
            buf('T_SKIN', s3).ptr, buf('T_SO', era['T_SO'].shape).ptr))
    return out, dict(n_iter=it - 1, max_err=hist, levels_touched=0)


def _band_of(arrays, j0, j1):
    """Latitude rows [j0, j1) of every field of a dict (lat is the second-to-last axis; tables such as ak / bk pass through)."""
    out = {}
    for k, v in arrays.items():
        a = np.asarray(v) if v is not None else None
        out[k] = a[..., j0:j1, :] if (a is not None and a.ndim >= 3) else v
    return out


def pgw_for_era5_arrays(era, deltas, delta_times, plev, target_dt, ignore_top_pressure_error=False,
                        p_ref=None, dtype=None, i_reinterp=False, ref_dtype=None, band=None, reduce_max=None, resident=None):
    """Whole-file path on in-memory host arrays (upload, compute on the GPU, download).

    band = (rank, world): this process handles latitude band `rank` of `world` of the file (parallel.band_rows) and
    returns that band's rows; `reduce_max` (parallel.band_max_hook) makes the loop's stopping test global, so the bands
    put together are bit for bit the single-process result, pass count and max|err| history included."""
    ctx = default_context()
    if band is not None:
        from .parallel import band_rows
        if i_reinterp:
            raise ValueError('latitude-band sharding of one file needs the multi-pass loop (settings.i_reinterp = 0)')
        if reduce_max is None and band[1] > 1:
            raise ValueError('band sharding over more than one rank needs reduce_max (parallel.band_max_hook)')
        j0, j1 = band_rows(np.asarray(era['T']).shape[-2], band[0], band[1])
        era, deltas = _band_of(era, j0, j1), _band_of(deltas, j0, j1)
    if reduce_max is not None:
        ctx.set_reduce_hook(reduce_max)
        try:
            return pgw_for_era5_arrays(era, deltas, delta_times, plev, target_dt, ignore_top_pressure_error, p_ref, dtype,
                                       i_reinterp, ref_dtype, resident=resident)
        finally:
            ctx.set_reduce_hook(None)
    if dtype is None:
        dtype = np.asarray(era['T']).dtype
    dtype = np.dtype(dtype)
    try:
        if isinstance(delta_times, dict):                 # one time axis per delta file, like the reference's load_delta
            ds = DeltaSet(ctx, deltas, delta_times.get('ta'), plev, dtype, times_by_var=delta_times, resident=resident)
        else:
            ds = DeltaSet(ctx, deltas, delta_times, plev, dtype, resident=resident)
        e = _upload_era(ctx, era, dtype)
        coeffs = dict(ak=era['ak'], bk=era['bk'], akm=era.get('akm'), bkm=era.get('bkm'), soil1=era['soil1'])
        if os.environ.get('PGW_TEST_FAIL_SETUP') == os.environ.get('RANK', '0') and ctx.has_reduce_hook():
            raise MemoryError('PGW_TEST_FAIL_SETUP: forced failure of the host-side set-up of this band')
    except BaseException:
        # latitude-band mode: the other bands are on their way into pgw_step03_file and would wait for this one in their
        # first reduce until the backend's timeout (gloo: 30 min) - tell them
        if ctx.has_reduce_hook():
            ctx.band_abort()
        raise
    if i_reinterp:
        out, info = process_file_device_reinterp(ctx, e, coeffs, ds, target_dt, ignore_top_pressure_error, p_ref,
                                                 ref_dtype=ref_dtype)
    else:
        out, info = process_file_device(ctx, e, coeffs, ds, target_dt, ignore_top_pressure_error, p_ref, keep_hur=True,
                                        ref_dtype=ref_dtype)
    res = {k: v.numpy() for k, v in out.items() if not k.startswith('_')}
    res['RELHUM_pgw'] = out['_hur_pgw'].numpy()
    res.update(info)
    return res


# ----------------------------------------------------------------------------------------
# file-level driver: pgw_for_era5 and the command line
# ----------------------------------------------------------------------------------------
_DELTASETS = {}


class _BandOfRecords:
    """A record provider cut to latitude rows [j0, j1) (band-wise runs with the record window)."""

    def __init__(self, reader, j0, j1):
        self.reader, self.j0, self.j1 = reader, j0, j1
        self.nrec = reader.nrec
        self.rec_shape = tuple(reader.rec_shape[:-2]) + (j1 - j0, reader.rec_shape[-1])

    def read_record(self, r):
        return self.reader.read_record(r)[..., self.j0:self.j1, :]


def load_delta_set(ctx, delta_input_dir, dtype, band=None):
    """All delta files of a directory -> DeltaSet on the device, cached per process (every ERA5 file of a run uses the
    same deltas).  Like the reference's load_delta (functions.py:195-303) every file brings its own time axis; the four
    model-level variables must share one plev axis (the quad kernel interpolates them together), zg may have its own.
    When all records fit the HBM budget (DeltaSet) the files are read whole; otherwise they are opened record-wise and
    the DeltaSet keeps a window of records per variable on the device."""
    from . import ncio
    key = (os.path.abspath(delta_input_dir), np.dtype(dtype).str, ctx.device, band)
    if key in _DELTASETS:
        return _DELTASETS[key]
    readers = {}
    for var in DeltaSet.VARS_3D + ('tas', 'hurs', 'ts', 'tos', 'siconc'):
        readers[var] = ncio.RecordReader(os.path.join(delta_input_dir, S.file_name_bases['SCEN-HIST'].format(var)), var)
    readers['ps_hist'] = ncio.RecordReader(os.path.join(delta_input_dir, S.file_name_bases['HIST'].format('ps')), 'ps')
    times, plevs = {}, {}
    for var, r in readers.items():
        if not r.dims or r.dims[0] != S.TIME_GCM or S.TIME_GCM not in r.coords:
            raise ValueError('%s: the first dimension of %s must be %s with a coordinate variable' % (r.path, r.var, S.TIME_GCM))
        times[var] = np.asarray(r.coords[S.TIME_GCM])
        if var in DeltaSet.VARS_3D:
            plevs[var] = np.asarray(r.coords[S.PLEV_GCM], dtype=np.float64)
    plev = plevs['ta']
    for var in ('hur', 'ua', 'va'):
        if len(plevs[var]) != len(plev) or np.any(plevs[var] != plev):
            raise ValueError('plev axis of %s differs from that of ta (the four model-level deltas are interpolated together)' % var)
    total = sum(r.nrec * int(np.prod(r.rec_shape, dtype=np.int64)) for r in readers.values()) * np.dtype(dtype).itemsize
    if band is not None:                                                        # (j0, j1): this rank's latitude rows
        nlat = readers['ta'].rec_shape[-2]
        total = total * (band[1] - band[0]) // max(nlat, 1)
    forced = os.environ.get('PGW_DELTA_RESIDENT')
    resident = (forced == '1') if forced in ('0', '1') else total <= DeltaSet.budget_bytes(ctx)
    if resident:
        arrays = {}
        for var, r in readers.items():
            a = ncio.open_dataset(r.path)[r.var].values                         # whole file, threaded reads
            arrays[var] = a if band is None else np.ascontiguousarray(a[..., band[0]:band[1], :])
            r.close()
    else:
        arrays = readers if band is None else {k: _BandOfRecords(r, band[0], band[1]) for k, r in readers.items()}
    dset = DeltaSet(ctx, arrays, times['ta'], plev, dtype, times_by_var=times, resident=resident)
    dset.plev_zg = plevs['zg']
    _DELTASETS[key] = dset
    return dset


_POOLS = {}
_BIG_OUT = ('T', 'QV', 'U', 'V')


def _io_raw():
    """PGW_IO_RAW=0 converts the byte order of the 4-D fields on the host (first version of the drivers);
    default: the file's big-endian bytes go to the GPU as they are, through pinned buffers, and come back
    big-endian, so no host pass touches a field between `pread` and `pwrite`."""
    return os.environ.get('PGW_IO_RAW', '1') != '0'


def _pinned_pool(ctx):
    from .device import PinnedPool
    with _BUFFER_LOCK:                              # the reader threads of the first two files arrive here together
        if id(ctx) not in _POOLS:
            _POOLS[id(ctx)] = PinnedPool(ctx)
        return _POOLS[id(ctx)]


def _release_pinned(pool, buf):
    """Give a pinned buffer back to the pool it came from; a buffer the pool does not know is a bug of the pipeline (it
    would never be recycled: 4.6 GB of pinned memory per float64 file), not something to pass over."""
    if not pool.release(buf):
        raise RuntimeError('a pinned buffer was released to a pool that does not own it')


def reset_after_abort():
    """parallel.run_shard calls this when a stage of the pipeline has failed and every stage thread has stopped: stage
    chains that were cancelled between two stages may have held a device buffer set or pinned buffers - all of them are free
    again, so that the next run in this process (a long-lived driver, a retry over the remaining files) finds them."""
    with _BUFFER_LOCK:
        for sets in _DEVICE_BUFFERS.values():
            sets.reset()
        for pool in _POOLS.values():
            pool.reclaim_all()
    _ABORT.clear()


def _stage_load(inp_era_file_path, out_era_file_path, delta_input_dir, era_step_dt,
                ignore_top_pressure_error, debug_mode=None):
    """Stage 1 (host, I/O): read one ERA5 file and lay its fields out as C-order host arrays."""
    from . import ncio
    if debug_mode is not None:
        raise NotImplementedError('debug_mode (step_03_apply_to_era.py:350-361, 387-414) is a validation aid of '
                                  'the reference and not part of the MI355X hot path')
    if S.f32_out_dtype not in ('float64', 'float32'):                  # checked before any buffer is taken
        raise ValueError("settings.f32_out_dtype must be 'float64' or 'float32'")
    if S.i_debug >= 0:
        print('Start working on input file {}'.format(inp_era_file_path))
    raw = _io_raw()
    pinned = []
    alloc = None
    if raw:
        pool = _pinned_pool(default_context())

        def alloc(nbytes):
            buf = pool.acquire(nbytes)
            pinned.append(buf)
            return buf
    try:
        era_file = ncio.open_dataset(inp_era_file_path, decode_times=False, raw_big=raw, alloc=alloc)   # step_03:60
    except BaseException:
        for b_ in pinned:
            _release_pinned(pool, b_)
        raise
    vm = S.var_name_map
    dtype = np.dtype('float64') if era_file[vm['ta']].dtype.itemsize == 8 else np.dtype('float32')
    dims4 = (S.TIME_ERA, S.LEV_ERA, S.LAT_ERA, S.LON_ERA)
    dims3 = (S.TIME_ERA, S.LAT_ERA, S.LON_ERA)
    dims_so = (S.TIME_ERA, S.SOIL_HLEV_ERA, S.LAT_ERA, S.LON_ERA)

    def get(name, dims):
        v = era_file[name].transpose(*dims).values
        if (not v.dtype.isnative) and v.dtype.kind == 'f' and v.dtype.itemsize == dtype.itemsize and v.flags.c_contiguous:
            return v                                                         # file byte order; converted on the GPU
        return np.ascontiguousarray(v, dtype=dtype)

    era = dict(PS=get(vm['ps'], dims3), FIS=get(vm['zgs'], dims3), T=get(vm['ta'], dims4), QV=get(vm['hus'], dims4),
               U=get(vm['ua'], dims4), V=get(vm['va'], dims4), T_SKIN=get(vm['ts'], dims3),
               T_SO=get(vm['st'], dims_so), FR_LAND=get(vm['sftlf'], dims3), FR_SEA_ICE=get(vm['sic'], dims3))
    coeffs = dict(ak=np.asarray(era_file['ak'].values, dtype=np.float64), bk=np.asarray(era_file['bk'].values, dtype=np.float64),
                  soil1=np.asarray(era_file[S.SOIL_HLEV_ERA].values, dtype=np.float64))
    if 'akm' in era_file:                                                    # step_03:68-70
        coeffs['akm'] = np.asarray(era_file['akm'].values, dtype=np.float64)
        coeffs['bkm'] = np.asarray(era_file['bkm'].values, dtype=np.float64)
    return dict(era_file=era_file, era=era, coeffs=coeffs, dtype=dtype, out_path=out_era_file_path, inp_path=inp_era_file_path,
                delta_input_dir=delta_input_dir, era_step_dt=era_step_dt, ignore_top=ignore_top_pressure_error,
                dims=dict(d3=dims3, d4=dims4, so=dims_so), pinned=pinned)


class _BufferSets:
    """Device buffer sets of one (shape, dtype): `n` input sets and `n` output sets, handed round between the stages.
    With two sets the upload of file i+1 (stream 'h2d') and the download of file i-1 (stream 'd2h') run while the kernels
    of file i do - 288 GB of HBM hold many 0.25 deg L137 files (fp64: 4.6 GB in, 4.6 GB out each)."""

    def __init__(self, n):
        import queue
        self.inp, self.out = queue.Queue(), queue.Queue()
        self.all_inp, self.all_out = [{} for _ in range(n)], [{} for _ in range(n)]
        for a, b in zip(self.all_inp, self.all_out):
            self.inp.put(a)
            self.out.put(b)

    def reset(self):
        """all sets free again (after an aborted run: a cancelled stage chain may have held one)"""
        import queue
        for q, sets in ((self.inp, self.all_inp), (self.out, self.all_out)):
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            for x in sets:
                q.put(x)


_DEVICE_BUFFERS = {}
_BUFFER_LOCK = __import__('threading').Lock()
_ABORT = __import__('threading').Event()          # set by parallel.run_shard when a stage failed: waiting stages give up


def _buffer_sets(key):
    with _BUFFER_LOCK:
        if key not in _DEVICE_BUFFERS:
            _DEVICE_BUFFERS[key] = _BufferSets(max(1, int(os.environ.get('PGW_DEVICE_SETS', '2'))))
        return _DEVICE_BUFFERS[key]


def _take(q):
    """A free buffer set.  Gives up when another stage has failed (_ABORT) and - so that a set lost to a bug can never hang
    a run - after PGW_BUFFER_WAIT_S seconds (default 600; a stage of a 0.25 deg file takes well under a second)."""
    import queue
    import time
    t0 = time.time()
    limit = float(os.environ.get('PGW_BUFFER_WAIT_S', '600'))
    while True:
        try:
            return q.get(timeout=0.2)
        except queue.Empty:
            if _ABORT.is_set():
                raise RuntimeError('pipeline aborted: another stage failed')
            if time.time() - t0 > limit:
                raise RuntimeError('no device buffer set became free within %.0f s (PGW_BUFFER_WAIT_S)' % limit)


def _stage_upload(item):
    """Stage 2 (stream 'h2d'): host -> device copies of one file into a free input buffer set; the file's big-endian
    bytes are converted on the device."""
    ctx = default_context()
    up = ctx.side('h2d')
    dtype = item['dtype']
    item['deltas'] = load_delta_set(ctx, item['delta_input_dir'], dtype)
    sets = _buffer_sets((item['era']['T'].shape, dtype.str))
    with _BUFFER_LOCK:                              # once per process and grid: where the level arrays of the buffer sets lie
        ctx.enable_placement(int(np.prod(item['era']['T'].shape, dtype=np.int64)) * 8, 8 * len(sets.all_inp) + 1)
    inp = _take(sets.inp)
    item['sets'], item['inp_set'] = sets, inp
    try:
        for k, v in item['era'].items():
            if k not in inp:
                inp[k] = ctx.level_array(v.shape, dtype, LEVEL_CLASS_IN[k]) if k in LEVEL_CLASS_IN else ctx.empty(v.shape, dtype)
            inp[k].copy_from(v, sync=False, ctx=up)
        up.sync()                                           # the host copies may go, and the compute stream may read
    except BaseException:
        sets.inp.put(inp)
        raise
    item['era'] = None                                      # release the host copies of the inputs
    return item


def _stage_compute(item):
    """Stage 3 (the context's own stream): one pgw_step03_file call on the uploaded buffers into a free output set."""
    ctx = default_context()
    sets, inp = item['sets'], item['inp_set']
    try:
        out_set = _take(sets.out)
    except BaseException:
        sets.inp.put(inp)
        raise
    try:
        out, info = process_file_device(ctx, inp, item['coeffs'], item['deltas'], item['era_step_dt'], item['ignore_top'],
                                        p_ref='local' if S.p_ref_inp is None else S.p_ref_inp, out=out_set,
                                        i_reinterp=bool(S.i_reinterp))
        ctx.sync()                                          # outputs complete before the 'd2h' stream reads them
    except ValueError as e:
        sets.out.put(out_set)
        if getattr(e, 'status', None) == _lib.PGW_ERR_NOT_CONVERGED or str(e).startswith('ERROR! Pressure adjustment did not converge'):
            raise ValueError('ERROR! Pressure adjustment did not converge ' +                   # step_03:315-319, text and file name
                             'for file {}. '.format(item.get('inp_path')) +
                             'Consider increasing the value for "max_n_iter" in ' +
                             'settings.py') from None
        raise
    except BaseException:
        sets.out.put(out_set)
        raise
    finally:
        sets.inp.put(inp)                                   # the next upload may overwrite the inputs
        item['inp_set'] = None
    item['out_set'], item['out'], item['info'], item['deltas'] = out_set, out, info, None
    if S.i_debug >= 2:
        for it, err in enumerate(info['max_err']):
            print('### iteration {:03d}, phi max error: {}'.format(it + 1, err))
    return item


def _stage_download(item):
    """Stage 4 (stream 'd2h'): results to the host - the large fields converted to the file's byte order on the device
    and DMA-ed into pinned buffers the writer `pwrite`s from."""
    from . import ncio
    ctx = default_context()
    dn = ctx.side('d2h')
    sets, out_set, out = item['sets'], item['out_set'], item['out']
    raw = _io_raw()
    pool = _pinned_pool(ctx) if raw else None
    result, pinned_out = {}, []
    # settings.f32_out_dtype = 'float32': the float64 T, QV, U, V of a float32 file leave the device as float32
    narrow = (S.f32_out_dtype == 'float32' and item['dtype'] == np.dtype('float32'))
    try:
        for k in ('PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE'):
            o = out[k]
            if narrow and k in _BIG_OUT and o.dtype == np.dtype('float64'):
                sk = '_f32_' + k
                if sk not in out_set or out_set[sk].size != o.size:
                    out_set[sk] = ctx.empty(o.shape, np.float32)
                if raw and out_set[sk].nbytes >= ncio.BIG_VARIABLE:
                    hb = pool.acquire(out_set[sk].nbytes)
                    pinned_out.append(hb)
                    result[k] = o.download_narrow_f32(hb, out_set[sk], big_endian=True, ctx=dn)
                else:
                    hb = np.empty(out_set[sk].nbytes, dtype=np.uint8)
                    result[k] = o.download_narrow_f32(hb, out_set[sk], big_endian=False, ctx=dn)
            elif raw and k in _BIG_OUT and o.nbytes >= ncio.BIG_VARIABLE:
                hb = pool.acquire(o.nbytes)
                pinned_out.append(hb)
                result[k] = o.download_foreign(hb, ctx=dn)      # big-endian on the device, DMA into pinned memory
            else:
                result[k] = o.numpy(ctx=dn)
        dn.sync()
    except BaseException:
        for b_ in pinned_out:
            _release_pinned(pool, b_)
        raise
    finally:
        sets.out.put(out_set)
        item['out_set'] = item['out'] = None
    item['result'] = result
    # the four replaced 4-D variables leave the dataset now, so their pinned input buffers can be recycled
    # before the writer gets to this file
    era_file, vm = item['era_file'], S.var_name_map
    if pool is not None:
        for k in _BIG_OUT:
            old = era_file[vm[dict(T='ta', QV='hus', U='ua', V='va')[k]]]
            base = old.values
            old.values = np.empty((0,) * base.ndim, dtype=base.dtype)
            keep = []
            for b_ in item['pinned']:
                if b_.ctypes.data == base.ctypes.data:
                    _release_pinned(pool, b_)
                else:
                    keep.append(b_)
            item['pinned'] = keep
    item['pinned_out'] = pinned_out
    return item


def _stage_store(item):
    """Stage 3 (host, I/O): write the modified file (step_03:369-378: PS, T, QV, U, V; T_SKIN, T_SO and
    FR_SEA_ICE were updated in place at :105-144; RELHUM is never added)."""
    from . import ncio
    era_file, vm, d = item['era_file'], S.var_name_map, item['dims']
    names = dict(PS=(vm['ps'], d['d3']), T=(vm['ta'], d['d4']), QV=(vm['hus'], d['d4']), U=(vm['ua'], d['d4']),
                 V=(vm['va'], d['d4']), T_SKIN=(vm['ts'], d['d3']), T_SO=(vm['st'], d['so']), FR_SEA_ICE=(vm['sic'], d['d3']))
    try:
        for key, (name, dims) in names.items():
            old = era_file[name]
            era_file[name] = ncio.Field(item['result'][key], dims, {k: old.coords[k] for k in dims if k in old.coords}, old.attrs)
        ncio.to_netcdf(era_file, item['out_path'])
    finally:
        bufs = item.get('pinned', []) + item.get('pinned_out', [])
        if bufs:
            pool = _pinned_pool(default_context())
            item['result'] = None
            for b_ in bufs:
                _release_pinned(pool, b_)
            item['pinned'], item['pinned_out'] = [], []
    if S.i_debug >= 1:
        print('Done. Saved to file {}.'.format(item['out_path']))
    return item['info']['n_iter']


def pgw_for_era5(inp_era_file_path, out_era_file_path, delta_input_dir, era_step_dt,
                 ignore_top_pressure_error, debug_mode=None):
    """Apply the PGW deltas to one ERA5 file (reference step_03_apply_to_era.py:44-381).
    Returns the number of loop passes.  When many files are processed through IterMP the three
    stages below run as a pipeline (read of file i+1 and write of file i-1 overlap the GPU work of
    file i; SURVEY.md section 8 f rank 1)."""
    return _stage_store(_stage_download(_stage_compute(_stage_upload(_stage_load(
        inp_era_file_path, out_era_file_path, delta_input_dir, era_step_dt, ignore_top_pressure_error, debug_mode)))))


def pgw_for_era5_banded(inp_era_file_path, out_era_file_path, delta_input_dir, era_step_dt, ignore_top_pressure_error,
                        rank, world, reduce_max, barrier):
    """ONE ERA5 file over `world` ranks in latitude bands (the latency mode) from file to file:
    every rank `pread`s its rows of the fields (ncio.read_band: one byte range per (time, level) plane of the classic
    layout), runs the per-file path on its band with the loop's stopping test made global by `reduce_max`
    (parallel.band_max_hook: MAX all-reduce of the per-pass figures), and `pwrite`s its rows into the shared output file
    whose header and remaining variables rank 0 wrote (ncio.BandedWriter).  The file equals the one-rank file byte for
    byte.  Every rank returns the pass count.  `barrier`: a callable all ranks meet in."""
    from . import ncio
    from .parallel import band_rows
    if S.i_reinterp:
        raise ValueError('latitude-band sharding of one file needs the multi-pass loop (settings.i_reinterp = 0)')
    if S.f32_out_dtype not in ('float64', 'float32'):
        raise ValueError("settings.f32_out_dtype must be 'float64' or 'float32'")
    vm = S.var_name_map
    big = dict(T=vm['ta'], QV=vm['hus'], U=vm['ua'], V=vm['va'], PS=vm['ps'], T_SKIN=vm['ts'], T_SO=vm['st'], FR_SEA_ICE=vm['sic'])
    ctx = default_context()
    ctx.set_reduce_hook(reduce_max)
    try:
        try:
            if S.i_debug >= 0 and rank == 0:
                print('Start working on input file {} in {} latitude bands'.format(inp_era_file_path, world))
            ds = ncio.open_dataset(inp_era_file_path, decode_times=False, skip=tuple(big.values()))      # step_03:60
            dims4 = (S.TIME_ERA, S.LEV_ERA, S.LAT_ERA, S.LON_ERA)
            dims3 = (S.TIME_ERA, S.LAT_ERA, S.LON_ERA)
            for k, name in big.items():
                want = dims4 if k in ('T', 'QV', 'U', 'V') else ((S.TIME_ERA, S.SOIL_HLEV_ERA, S.LAT_ERA, S.LON_ERA) if k == 'T_SO' else dims3)
                if ds[name].dims != want:
                    raise NotImplementedError('band-wise I/O reads %s in the order %s, the file stores %s' % (name, want, ds[name].dims))
            dtype = np.dtype('float64') if ds[big['T']].dtype.itemsize == 8 else np.dtype('float32')
            nlat = ds[big['T']].shape[-2]
            j0, j1 = band_rows(nlat, rank, world)
            era = {k: np.ascontiguousarray(ncio.read_band(inp_era_file_path, name, j0, j1), dtype=dtype) for k, name in big.items()}
            for k, name in (('FIS', vm['zgs']), ('FR_LAND', vm['sftlf'])):                               # small: read whole
                era[k] = np.ascontiguousarray(ds[name].transpose(*dims3).values[..., j0:j1, :], dtype=dtype)
            coeffs = dict(ak=np.asarray(ds['ak'].values, dtype=np.float64), bk=np.asarray(ds['bk'].values, dtype=np.float64),
                          soil1=np.asarray(ds[S.SOIL_HLEV_ERA].values, dtype=np.float64))
            if 'akm' in ds:                                                                              # step_03:68-70
                coeffs['akm'] = np.asarray(ds['akm'].values, dtype=np.float64)
                coeffs['bkm'] = np.asarray(ds['bkm'].values, dtype=np.float64)
            deltas = load_delta_set(ctx, delta_input_dir, dtype, band=(j0, j1))
            e = _upload_era(ctx, era, dtype)
        except BaseException:
            ctx.band_abort()                  # the other bands are on their way into the loop's first reduce: tell them
            raise
        out, info = process_file_device(ctx, e, coeffs, deltas, era_step_dt, ignore_top_pressure_error,
                                        p_ref='local' if S.p_ref_inp is None else S.p_ref_inp)
        res = {k: out[k].numpy() for k in big}
        for v in list(e.values()) + [x for x in out.values()]:
            v.free()
    finally:
        ctx.set_reduce_hook(None)
    narrow = (S.f32_out_dtype == 'float32' and dtype == np.dtype('float32'))
    F = ncio.Field
    for k, name in big.items():
        old = ds[name]
        dt_out = res[k].dtype
        if narrow and k in _BIG_OUT:
            res[k] = res[k].astype(np.float32)                                  # the reference's float64 field rounded once
            dt_out = np.dtype('float32')
        ds[name] = F(ncio.placeholder(old.shape, dt_out), old.dims, old.coords, old.attrs)
    writer = ncio.BandedWriter(ds, out_era_file_path, tuple(big.values()))
    if rank == 0:
        writer.create()
    barrier()
    for k, name in big.items():
        writer.write_band(name, j0, j1, res[k])
    barrier()
    if S.i_debug >= 1 and rank == 0:
        print('Done. Saved to file {}.'.format(out_era_file_path))
    return info['n_iter']


pgw_for_era5.stages = (_stage_load, _stage_upload, _stage_compute, _stage_download, _stage_store)
pgw_for_era5.abort = _ABORT
pgw_for_era5.reset = reset_after_abort


def _cli(argv=None):
    import argparse
    from pathlib import Path
    from .parallel import IterMP
    p = argparse.ArgumentParser(description='Perturb ERA5 files with PGW climate deltas on MI355X GPUs '
                                            '(flags of the reference step_03_apply_to_era.py:505-568; settings in settings.py).')
    p.add_argument('-i', '--input_dir', type=str, default=None, help='directory with the ERA5 input files')
    p.add_argument('-o', '--output_dir', type=str, default=None, help='directory for the processed ERA5 files')
    p.add_argument('-f', '--first_era_step', type=str, default='2006080200', help='first time step, YYYYMMDDHH')
    p.add_argument('-l', '--last_era_step', type=str, default='2006080300', help='last time step (inclusive), YYYYMMDDHH')
    p.add_argument('-H', '--hour_inc_step', type=int, default=3, help='hours between time steps')
    p.add_argument('-d', '--delta_input_dir', type=str, default=None,
                   help='directory with the regridded climate deltas (output of step_02) and ps_historical.nc')
    p.add_argument('-p', '--n_par', type=int, default=1, help='number of worker ranks = GPUs; files are dealt round-robin')
    p.add_argument('-t', '--ignore_top_pressure_error', action='store_true',
                   help='do not fail if ERA5 reaches higher than the climate deltas')
    p.add_argument('-D', '--debug_mode', type=str, default=None,
                   help='interpolate_time | interpolate_full: the reference dumps the deltas instead of the ERA5 files '
                        '(step_03_apply_to_era.py:350-361, 387-414) - a validation aid, NOT built here: the run stops with '
                        'NotImplementedError')
    p.add_argument('--bands', action='store_true',
                   help='latency mode: EVERY file is split over all ranks in latitude bands (each rank reads, computes and '
                        'writes its rows; one MAX all-reduce per loop launch) instead of file i -> rank i mod W.  Needs the '
                        'ranks of `python -m torch.distributed.run --nproc-per-node W -m pgw4era5_amd.step_03_apply_to_era ...`; '
                        'settings.i_reinterp = 0.  No counterpart in the reference.')
    args = p.parse_args(argv)
    if args.input_dir is None:
        raise ValueError('Input directory (-i) is required.')
    if args.output_dir is None:
        raise ValueError('Output directory (-o) is required.')
    if args.delta_input_dir is None:
        raise ValueError('Delta input directory (-d) is required.')
    if args.debug_mode is not None and args.debug_mode not in ['interpolate_time', 'interpolate_full']:
        raise ValueError('Invalid input for argument --debug_mode! Valid arguments are: "interpolate_time" or "interpolate_full"')
    first = _dt.datetime.strptime(args.first_era_step, '%Y%m%d%H')
    last = _dt.datetime.strptime(args.last_era_step, '%Y%m%d%H')
    inc = _dt.timedelta(hours=args.hour_inc_step)
    steps = []
    t = first
    while t < last + inc:                                   # np.arange(first, last + inc, inc), step_03:594-596
        steps.append(t)
        t += inc
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    fargs = dict(delta_input_dir=args.delta_input_dir, ignore_top_pressure_error=args.ignore_top_pressure_error,
                 debug_mode=args.debug_mode)
    step_args = [dict(inp_era_file_path=os.path.join(args.input_dir, S.era5_file_name_base.format(s)),
                      out_era_file_path=os.path.join(args.output_dir, S.era5_file_name_base.format(s)),
                      era_step_dt=s) for s in steps]
    if args.bands:
        return _run_banded(fargs, step_args)
    imp = IterMP(njobs=args.n_par, run_async=True)
    imp.run(pgw_for_era5, fargs, step_args)
    return imp.output


def _run_banded(fargs, step_args):
    """`--bands`: all ranks of the torch.distributed.run launch work on ONE file at a time, in latitude bands."""
    from .parallel import _dist_env, band_max_hook, bind_rank_to_numa
    rank, world = _dist_env()
    if fargs.get('debug_mode') is not None:
        raise NotImplementedError('debug_mode is a validation aid of the reference and not part of the MI355X hot path')
    if world == 1:
        return [pgw_for_era5(**dict(fargs, **s)) for s in step_args]
    import torch                          # before any HIP call of libpgw_hip.so (see _lib.py)
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        backend = os.environ.get('PGW_BANDS_BACKEND') or ('nccl' if torch.cuda.is_available() and torch.cuda.device_count() >= world else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(backend)
        created = True
    bind_rank_to_numa()
    hook = band_max_hook()
    out, err = [], None
    try:
        for s in step_args:
            out.append(pgw_for_era5_banded(s['inp_era_file_path'], s['out_era_file_path'], fargs['delta_input_dir'], s['era_step_dt'],
                                           fargs['ignore_top_pressure_error'], rank, world, hook, dist.barrier))
    except Exception as e:                # noqa: BLE001 - a data error reached every band through the reduce: all ranks are here
        err = e
    if created:
        try:
            dist.destroy_process_group()
        except Exception:                 # noqa: BLE001
            pass
    if err is not None:
        raise err
    return out


if __name__ == '__main__':
    _cli()
