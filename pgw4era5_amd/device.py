"""
Device context and device-resident arrays for the HIP path.

`Context` wraps one `pgw_ctx` (one HIP device + stream); `DeviceArray` is a shape/dtype-tagged
device buffer.  Arrays stay in HBM between calls (288 GB per MI355X: a whole 0.25 deg L137 file,
its PGW state and all twelve months of every delta fit many times over), host<->device copies
happen only at the edges (`Context.to_device`, `DeviceArray.numpy`).
"""
import ctypes as C
import os

import numpy as np

from . import _lib

_FOREIGN = '>' if np.little_endian else '<'      # numpy byteorder code of the non-native order

_DTYPE_TAG = {np.dtype('float32'): _lib.PGW_F32, np.dtype('float64'): _lib.PGW_F64}


def dtype_tag(dtype):
    try:
        return _DTYPE_TAG[np.dtype(dtype)]
    except KeyError:
        raise TypeError('field arrays must be float32 or float64, got %s' % dtype)


class DeviceArray:
    """A C-order array in device memory owned by a Context."""

    __slots__ = ('ctx', 'ptr', 'shape', 'dtype', 'nbytes', '_owner', 'placement_class', '__weakref__')

    def __init__(self, ctx, shape, dtype, ptr=None, owner=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owner = owner
        self.placement_class = None              # set by SpreadPool.take
        if ptr is None:
            p = C.c_void_p()
            ctx._check(ctx.lib.pgw_malloc(ctx.handle, self.nbytes, C.byref(p)))
            self.ptr = p.value
            self._owner = self
            ctx._live += self.nbytes
        else:
            self.ptr = ptr

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def ndim(self):
        return len(self.shape)

    def numpy(self, ctx=None):
        c = ctx or self.ctx
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            c._check(c.lib.pgw_memcpy_d2h(c.handle, out.ctypes.data, self.ptr, self.nbytes))
            c.sync()
        return out

    def copy_from(self, host, sync=True, ctx=None):
        """Upload a host array.  An array in the file's byte order (e.g. '>f4' from `ncio.open_dataset(raw_big=True)`)
        of the same element size is uploaded as it is and converted on the device (`pgw_byteswap`), so the host
        never touches the values.  sync=False: the caller synchronises the context before `host` may be reused
        (meant for pinned sources, where the copy is a real asynchronous DMA).  ctx: another context of the same device
        whose stream carries the transfer (Context.side), so that it overlaps the kernels of this array's own context; the
        caller orders the two streams (a sync of `ctx` before the array is used)."""
        c = ctx or self.ctx
        host = np.asarray(host)
        swapped = (host.dtype.byteorder == _FOREIGN and host.dtype.kind == self.dtype.kind
                   and host.dtype.itemsize == self.dtype.itemsize and host.flags.c_contiguous)
        if not swapped:
            host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != self.shape:
            raise ValueError('shape mismatch %s vs %s' % (host.shape, self.shape))
        if self.nbytes:
            c._check(c.lib.pgw_memcpy_h2d(c.handle, self.ptr, host.ctypes.data, self.nbytes))
            if swapped:
                c._check(c.lib.pgw_byteswap(c.handle, self.dtype.itemsize, self.size, self.ptr, self.ptr))
            if sync:
                c.sync()             # pageable source: keep it alive until the copy is done
        return self

    def download_foreign(self, host_bytes, scratch=None, ctx=None):
        """Enqueue: convert to the file's (big-endian) byte order on the device - into `scratch` if given, else in
        place, which leaves this array byte-reversed - and copy to the writable uint8 host buffer.  No
        synchronisation: the caller calls `ctx.sync()` before reading `host_bytes`.  Returns the host buffer viewed
        with the big-endian dtype and this array's shape.  ctx: see copy_from."""
        c = ctx or self.ctx
        if host_bytes.nbytes < self.nbytes:
            raise ValueError('host buffer too small')
        dst = self.ptr if scratch is None else scratch.ptr
        if self.nbytes:
            c._check(c.lib.pgw_byteswap(c.handle, self.dtype.itemsize, self.size, self.ptr, dst))
            c._check(c.lib.pgw_memcpy_d2h(c.handle, host_bytes.ctypes.data, dst, self.nbytes))
        return host_bytes[:self.nbytes].view(self.dtype.newbyteorder(_FOREIGN)).reshape(self.shape)

    def download_narrow_f32(self, host_bytes, scratch, big_endian=True, ctx=None):
        """Enqueue: this float64 array -> float32 (`pgw_narrow_f64_f32`, in the file's byte order if big_endian) into the float32
        device array `scratch`, then copy to the uint8 host buffer.  No synchronisation (see download_foreign).  Returns the
        host buffer viewed as float32 of this array's shape."""
        c = ctx or self.ctx
        if self.dtype != np.dtype('float64') or scratch.dtype != np.dtype('float32') or scratch.size != self.size:
            raise ValueError('download_narrow_f32: float64 source and a float32 scratch of the same size')
        if host_bytes.nbytes < scratch.nbytes:
            raise ValueError('host buffer too small')
        if self.nbytes:
            c._check(c.lib.pgw_narrow_f64_f32(c.handle, self.size, self.ptr, scratch.ptr, 1 if big_endian else 0))
            c._check(c.lib.pgw_memcpy_d2h(c.handle, host_bytes.ctypes.data, scratch.ptr, scratch.nbytes))
        dt = np.dtype('float32').newbyteorder(_FOREIGN) if big_endian else np.dtype('float32')
        return host_bytes[:scratch.nbytes].view(dt).reshape(self.shape)

    def view(self, shape):
        """Reshaped alias of the same buffer."""
        shape = tuple(int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError('cannot view %s as %s' % (self.shape, shape))
        return DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr, owner=self._owner)

    def slab(self, index):
        """Alias of sub-array [index] along axis 0 (contiguous)."""
        sub = self.shape[1:]
        n = int(np.prod(sub, dtype=np.int64)) * self.dtype.itemsize
        if not (0 <= index < self.shape[0]):
            raise IndexError(index)
        return DeviceArray(self.ctx, sub, self.dtype, ptr=self.ptr + index * n, owner=self._owner)

    def free(self):
        if self._owner is self and self.ptr:
            self.ctx.lib.pgw_free(self.ctx.handle, self.ptr)
            self.ctx._live -= self.nbytes
            self.ptr = None

    def __del__(self):
        try:
            if self._owner is self and self.ptr and self.ctx.handle:
                self.free()
        except Exception:
            pass

    # duck-typing used by the functions.py mirror
    @property
    def values(self):
        return self.numpy()


class _PooledArray(DeviceArray):
    """A stock array of a SpreadPool: when the last view of it is gone it goes back into the pool's stock (same memory, same
    class) instead of back to hipMalloc - so that a long run keeps its placement.  `Context.ws_adopt` and `free()` take it
    out of this cycle."""

    __slots__ = ('_pool', '_cls')

    def __del__(self):
        try:
            pool = self._pool
            if self._owner is self and self.ptr and self.ctx.handle and pool is not None and not pool.closed:
                pool._recycle(self.ptr, self.nbytes, self._cls)
                self.ptr = None                      # the memory lives on in a new stock object
                return
        except Exception:
            pass
        DeviceArray.__del__(self)


class SpreadPool:
    """Level arrays of one size placed over the card's memory regions (no counterpart in the reference: numpy arrays live
    wherever malloc put them).  On an MI355X the arrays a column kernel reads and writes at the same time run 12 % faster
    when they do NOT all lie in one stretch of physical memory (`k_delta_quad` 2.14 ms with all nine
    level arrays in one stretch, 1.87-1.89 ms with both the inputs and the outputs spread over two or three) - and
    consecutive hipMallocs come from one stretch for the first 60-90 GB.  The pool therefore DRAWS: a reference array, then
    candidates (with spacers between them) until `count` arrays are in stock, half of them in the reference's stretch and
    half outside it - told apart by `Context.placement_probe` of two write streams, reference + candidate: 5.0-5.45 TB/s when
    both lie in one stretch, 6.6-6.9 when they do not - and frees the rest.  `take()` hands the stock out by class (or
    alternating); an array whose last view is dropped comes back into the stock (`_PooledArray`).  When the draw finds no
    second class within its budget (a small or busy card) the stock is plain arrays and `info['classes']` says 1."""

    SPACER = 8 << 30

    def __init__(self, ctx, nbytes, count, budget_bytes=None):
        self.ctx, self.nbytes, self.count = ctx, int(nbytes), int(count)
        free, _total = ctx.mem_info()
        if budget_bytes is None:
            # most of what is free (the reference's stretch has been seen to span 60 ... 140 GB; everything drawn beyond the
            # stock goes back within the second), but a rank that shares its card (more local ranks than devices: a
            # rehearsal) takes its share of under half only - the draw must never starve a neighbour's ordinary allocations
            if os.environ.get('PGW_CARD_SHARE'):        # parallel.card_share: the ranks' PCI addresses, gathered
                share = max(1, int(os.environ['PGW_CARD_SHARE']))
            else:                                       # no process group: more local ranks than visible devices = sharing
                n_dev = C.c_int(0)
                ctx.lib.pgw_device_count(C.byref(n_dev))
                local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('PGW_LOCAL_WORLD', '1')) or 1)
                share = max(1, -(-local_world // max(n_dev.value, 1)))
            budget_bytes = int(0.85 * free) // share if share == 1 else int(0.45 * free) // share
        self.stock = [[], []]                       # class 0 = the reference's stretch, class 1 = outside it
        self.closed = False
        self._next = 0
        self.info = {'classes': 1, 'drawn_GB': 0.0, 'kept': 0, 'probe_GBps_inside': None, 'probe_GBps_outside': None}
        import time
        t0 = time.perf_counter()
        self._draw(budget_bytes)
        self.info['seconds'] = round(time.perf_counter() - t0, 2)

    def _probe(self, a, b):
        # a float64 view of both: rows x 1 Mi columns (the probe needs whole rows)
        ncol = 1 << 20
        rows = min(self.nbytes // (8 * ncol), 64)
        if rows < 2:
            return None
        va = DeviceArray(self.ctx, (1, rows, 1, ncol), np.float64, ptr=a.ptr, owner=a)
        vb = DeviceArray(self.ctx, (1, rows, 1, ncol), np.float64, ptr=b.ptr, owner=b)
        return self.ctx.placement_probe([], [va, vb], reps=3)       # two write streams: ~5 TB/s inside one stretch, ~6.7 across two

    def _draw(self, budget):
        ctx, n = self.ctx, self.nbytes
        want0, want1 = (self.count + 1) // 2, self.count // 2
        held, spent = [], 0

        def alloc(nbytes):
            nonlocal spent
            if spent + nbytes > budget:
                return None
            try:
                a = ctx.empty((nbytes // 8,), np.float64)
            except _lib.PGWHipError:
                return None
            spent += nbytes
            return a
        ref = alloc(n)
        if ref is None:
            raise _lib.PGWHipError('SpreadPool: no memory for the first array')
        cands = []                                  # (array, class); rates[i] = probe rate of cands[i] with the reference
        rates = []
        if n >= 2 * 8 * (1 << 20):                  # arrays of at least two probe rows: worth placing
            spacer_next = False
            split = False
            while True:
                c0 = 1 + sum(1 for _, cls in cands if cls == 0)
                c1 = sum(1 for _, cls in cands if cls == 1)
                if c0 >= want0 and c1 >= want1:
                    break
                if c0 >= want0 and spacer_next and not split:   # enough of the reference's stretch: stride through it in big steps
                    sp = alloc(self.SPACER)
                    if sp is None:
                        break
                    held.append(sp)
                spacer_next = True
                a = alloc(n)
                if a is None:
                    break
                r = self._probe(ref, a)
                rates.append(r)
                lo, hi = min(rates), max(rates)
                # two populations 20-25 % apart; until both have been seen everything counts as the reference's stretch
                split = hi > 1.10 * lo
                cls = 1 if (split and r > 0.5 * (lo + hi)) else 0
                cands.append((a, cls))
                if split:                           # re-label what was seen before the second population showed up
                    cands = [(x, 1 if rr > 0.5 * (lo + hi) else 0) for (x, _), rr in zip(cands, rates)]
        self.stock[0] = [ref] + [x for x, cls in cands if cls == 0][:max(want0 - 1, 0)]
        self.stock[1] = [x for x, cls in cands if cls == 1][:want1]
        kept = set(id(x) for x in self.stock[0] + self.stock[1])
        short = self.count - len(kept)
        for x, _ in cands:                          # not enough of one class: fill up with what there is
            if short <= 0:
                break
            if id(x) not in kept:
                self.stock[0].append(x); kept.add(id(x)); short -= 1
        for x, _ in cands:
            if id(x) not in kept:
                x.free()
        for sp in held:
            sp.free()
        while short > 0:                            # the budget ended the draw early: plain arrays for the rest
            self.stock[0].append(ctx.empty((n // 8,), np.float64)); short -= 1
        if rates and self.stock[1]:
            lo, hi = min(rates), max(rates)
            mid = 0.5 * (lo + hi)
            ins, outs = [r for r in rates if r <= mid], [r for r in rates if r > mid]
            self.info.update(classes=2, probe_GBps_inside=round(sum(ins) / len(ins)) if ins else None,
                             probe_GBps_outside=round(sum(outs) / len(outs)) if outs else None)
        for c in (0, 1):                            # stock objects that come back when their last view is dropped
            self.stock[c] = [self._pooled(a, c) for a in self.stock[c]]
        self.info.update(drawn_GB=round(spent / 1e9, 1), kept=len(self.stock[0]) + len(self.stock[1]),
                         kept_per_class=[len(self.stock[0]), len(self.stock[1])])

    def _pooled(self, a, cls):
        """the memory of the plain array `a` as a _PooledArray of class `cls`"""
        p = _PooledArray(self.ctx, a.shape, a.dtype, ptr=a.ptr)
        p._owner, p._pool, p._cls = p, self, cls
        a._owner, a.ptr = None, None                # `a` no longer owns anything (its bytes stay counted in ctx._live)
        return p

    def _recycle(self, ptr, nbytes, cls):
        p = _PooledArray(self.ctx, (nbytes // 8,), np.float64, ptr=ptr)
        p._owner, p._pool, p._cls = p, self, cls
        self.stock[cls].append(p)

    def close(self):
        """free the stock; arrays still in use are freed (not recycled) when their last view goes"""
        self.closed = True
        for c in (0, 1):
            for a in self.stock[c]:
                a.free()
            self.stock[c] = []

    def take(self, shape, dtype, cls=None):
        """One array of the stock as a DeviceArray of `shape` / `dtype` (at most `nbytes`), classes alternating from call
        to call (or the class asked for, while it lasts); plain memory once the stock is used up."""
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if need > self.nbytes:
            raise ValueError('SpreadPool of %d-byte arrays asked for %d bytes' % (self.nbytes, need))
        order = [self._next % 2, (self._next + 1) % 2] if cls is None else [cls % 2, (cls + 1) % 2]
        if not (self.stock[0] or self.stock[1]):
            import gc
            gc.collect()                            # an owning array refers to itself: dropped ones come back with the collector
        for c in order:
            if self.stock[c]:
                if cls is None:
                    self._next += 1
                a = self.stock[c].pop(0)
                v = DeviceArray(self.ctx, shape, dtype, ptr=a.ptr, owner=a)
                v.placement_class = c if self.info['classes'] > 1 else 0
                return v
        v = self.ctx.empty(shape, dtype)
        v.placement_class = None
        return v

    def take_owner(self, cls=0):
        """A whole stock array (it owns its memory; for `Context.ws_adopt`), or None when the stock is used up."""
        for c in (cls % 2, (cls + 1) % 2):
            if self.stock[c]:
                return self.stock[c].pop(0)
        return None


class Context:
    """One HIP device + stream (`pgw_ctx`)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
        n = C.c_int(0)
        self.lib.pgw_device_count(C.byref(n))
        if n.value <= 0:
            raise _lib.PGWHipError('no HIP device visible: pgw4era5_amd needs an MI355X (no CPU fallback)')
        h = C.c_void_p()
        rc = self.lib.pgw_ctx_create(int(device) % n.value, C.byref(h))
        if rc != 0:
            raise _lib.PGWHipError('pgw_ctx_create(device=%d) failed with status %d' % (device, rc))
        self.handle = h
        self.device = int(device) % n.value
        self._live = 0
        self._levels_key = None
        self._reduce_cb = None                  # keeps the ctypes callback of set_reduce_hook alive
        self._side = {}
        self._reduce_exc = None

    def _check(self, rc):
        if rc == _lib.PGW_ERR_REDUCE and self._reduce_exc is not None:
            exc, self._reduce_exc = self._reduce_exc, None
            raise exc                           # what the hook itself raised (e.g. a torch.distributed error)
        _lib.check(self.handle, rc)

    def set_reduce_hook(self, fn):
        """`pgw_set_reduce_hook`: `fn(vals)` replaces the float64 array `vals` IN PLACE by its element-wise maximum over the ranks that
        hold the other latitude bands of the file being processed (parallel.band_max_hook); None removes the hook."""
        if fn is None:
            self._check(self.lib.pgw_set_reduce_hook(self.handle, _lib.REDUCE_MAX_FN(), None))
            self._reduce_cb = None
            return

        def cb(vals, n, _user):
            try:
                fn(np.ctypeslib.as_array(vals, shape=(n,)))
                return 0
            except BaseException as e:          # noqa: BLE001 - re-raised by _check on the Python side of the call
                self._reduce_exc = e
                return 1
        cfn = _lib.REDUCE_MAX_FN(cb)
        self._check(self.lib.pgw_set_reduce_hook(self.handle, cfn, None))
        self._reduce_cb = cfn

    def has_reduce_hook(self):
        return getattr(self, '_reduce_cb', None) is not None

    def band_abort(self, code=None, max_n_iter=None):
        """`pgw_band_abort`: this band cannot take part in the file the other bands are about to process (its set-up raised) -
        meet them in their first reduce with an error status so that they fail too instead of waiting (needs the hook)."""
        from . import settings as S
        self.lib.pgw_band_abort(self.handle, int(_lib.PGW_ERR_REDUCE if code is None else code),
                                int(S.max_n_iter if max_n_iter is None else max_n_iter))

    def side(self, name):
        """A second context on the same device (its own HIP stream), created on first use: 'h2d' / 'd2h' carry the file
        transfers of the pipelined driver so that uploads, kernels and downloads of consecutive files overlap (PCIe is
        full duplex; step_03_apply_to_era.py stages)."""
        if name not in self._side:
            self._side[name] = Context(self.device)
        return self._side[name]

    def close(self):
        for c in self._side.values():
            c.close()
        self._side = {}
        if getattr(self, '_spread', None) is not None:
            self._spread.close()
            self._spread = None
        if self.handle:
            self.lib.pgw_ctx_destroy(self.handle)
            self.handle = None

    def sync(self):
        self._check(self.lib.pgw_sync(self.handle))

    def set_option(self, name, value):
        """Per-context option of include/pgw_hip.h `enum pgw_option` ('quad', 'full_column', 'force_vec1', 'multipass', 'loop_guess', 'force_off64', 'test_fail');
        returns the previous value."""
        old = self.get_option(name)
        self._check(self.lib.pgw_set_option(self.handle, _lib.OPTIONS[name], int(value)))
        return old

    def get_option(self, name):
        v = C.c_int(0)
        self._check(self.lib.pgw_get_option(self.handle, _lib.OPTIONS[name], C.byref(v)))
        return v.value

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.pgw_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        self._check(self.lib.pgw_mem_info(self.handle, C.byref(f), C.byref(t)))
        return f.value, t.value

    # ---- arrays --------------------------------------------------------------------------
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        self._check(self.lib.pgw_memset(self.handle, a.ptr, 0, a.nbytes))
        return a

    def to_device(self, host, dtype=None):
        host = np.asarray(host)
        if dtype is None:
            dtype = host.dtype if host.dtype in (np.float32, np.float64) else np.float64
        a = DeviceArray(self, host.shape, dtype)
        return a.copy_from(host)

    def placement_probe(self, src=(), dst=(), reps=3, rows=None):
        """GB/s of the column kernels' access pattern, no arithmetic, over the read streams `src` and the write streams
        `dst` (DeviceArrays of one size, <= 4 each; the dst arrays are OVERWRITTEN) - `pgw_placement_probe`: what this set
        of arrays gets where hipMalloc put them.  `rows`: the leading rows to touch (default: all of them)."""
        src, dst = list(src), list(dst)
        if len(src) > 4 or len(dst) > 4 or not (src or dst):
            raise ValueError('placement_probe takes 1 to 4 + 4 streams')
        arrs = src + dst
        nbytes = min(a.nbytes for a in arrs)
        ncol = int(np.prod(arrs[0].shape[-2:], dtype=np.int64)) if arrs[0].ndim >= 2 else 1 << 20
        n_rows = nbytes // (8 * ncol)
        if rows is not None:
            n_rows = min(n_rows, rows)
        if n_rows < 1:
            raise ValueError('arrays too small for a probe')
        ps = (C.c_void_p * 4)(*[a.ptr for a in src])
        pd = (C.c_void_p * 4)(*[a.ptr for a in dst])
        g = C.c_double()
        self._check(self.lib.pgw_placement_probe(self.handle, len(src), ps, len(dst), pd, n_rows, ncol, reps, C.byref(g)))
        return g.value

    def enable_placement(self, field_bytes, count, budget_bytes=None):
        """Place the level arrays of this context's files over the card's memory regions (`SpreadPool`; settings.placement =
        'spread', the default; `PGW_PLACEMENT=plain` or settings.placement = 'plain' turn it off): draws `count` arrays of
        `field_bytes`, hands one of class 1 to the library as its vapour-pressure workspace and serves `level_array()` from
        the rest.  Returns the pool's `info` (None when placement is off).  Idempotent per size."""
        from . import settings as S
        mode = os.environ.get('PGW_PLACEMENT', getattr(S, 'placement', 'spread'))
        if mode not in ('spread', 'plain'):
            raise ValueError("settings.placement / PGW_PLACEMENT must be 'spread' or 'plain'")
        if mode == 'plain':
            return None
        cur = getattr(self, '_spread', None)
        if cur is not None and cur.nbytes >= field_bytes:
            return cur.info
        try:
            pool = SpreadPool(self, field_bytes, count, budget_bytes)
            ws = pool.take_owner(1)                 # class 1: with T, U (0) and V (1) the quad kernel's writes are two and two
            if ws is not None:
                self.ws_adopt(0, ws)
        except Exception as e:                      # noqa: BLE001 - placement is an optimisation: never a reason to lose a run
            import sys
            sys.stderr.write('pgw4era5_amd: placement of the level arrays skipped (%s: %s)\n' % (type(e).__name__, e))
            return {'classes': 1, 'error': '%s: %s' % (type(e).__name__, e)}
        self._spread = pool
        return pool.info

    def level_array(self, shape, dtype, cls=None):
        """A device array for one level field: from the placement pool when `enable_placement` has set one up and the array is
        of its size class (more than a third of its arrays: the float32 inputs of a float64 pool count), else `empty()`."""
        pool = getattr(self, '_spread', None)
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if pool is not None and pool.nbytes // 3 < need <= pool.nbytes:
            return pool.take(shape, dtype, cls)
        return self.empty(shape, dtype)

    def ws_adopt(self, slot, arr):
        """Hand `arr` (a DeviceArray that owns its memory) to the library as workspace `slot` (`pgw_ws_adopt`; 0 = the
        vapour-pressure field of the file path).  The library owns the memory from here on."""
        if arr._owner is not arr or not arr.ptr:
            raise ValueError('ws_adopt needs an array that owns its memory')
        self._check(self.lib.pgw_ws_adopt(self.handle, slot, arr.ptr, arr.nbytes))
        self._live -= arr.nbytes
        arr._owner = None                  # no longer ours to free
        return arr

    # ---- profiling -----------------------------------------------------------------------
    def profile(self, on=True):
        self._check(self.lib.pgw_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.pgw_profile_reset(self.handle))

    def profile_get(self, kernel):
        n, ms = C.c_longlong(), C.c_double()
        self._check(self.lib.pgw_profile_get(self.handle, _lib.KERNEL_IDS[kernel], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def timer_start(self):
        self._check(self.lib.pgw_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_double()
        self._check(self.lib.pgw_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    # ---- vertical grid -------------------------------------------------------------------
    def set_levels(self, ak, bk, akm=None, bkm=None):
        ak = np.ascontiguousarray(ak, dtype=np.float64)
        bk = np.ascontiguousarray(bk, dtype=np.float64)
        key = (ak.tobytes(), bk.tobytes(),
               None if akm is None else np.asarray(akm, dtype=np.float64).tobytes(),
               None if bkm is None else np.asarray(bkm, dtype=np.float64).tobytes())
        if key == self._levels_key:
            return
        dp = C.POINTER(C.c_double)
        if akm is not None:
            akm = np.ascontiguousarray(akm, dtype=np.float64)
            bkm = np.ascontiguousarray(bkm, dtype=np.float64)
            pm, pb = akm.ctypes.data_as(dp), bkm.ctypes.data_as(dp)
        else:
            pm = pb = None
        self._check(self.lib.pgw_set_levels(self.handle, len(ak) - 1, ak.ctypes.data_as(dp),
                                            bk.ctypes.data_as(dp), pm, pb))
        self._levels_key = key
        self.nlev = len(ak) - 1


class PinnedPool:
    """Reusable pinned (page-locked) host buffers of one context: the reader threads `pread` file bytes into
    them, the GPU stage DMAs from / to them, the writer threads `pwrite` from them.  Pinning 2 GB costs far
    more than a copy, so buffers are recycled by size; thread-safe."""

    def __init__(self, ctx):
        import threading
        self.ctx = ctx
        self._free = {}
        self._owned = {}                  # address -> (rounded size, base array)
        self._lock = threading.Lock()
        self.allocated = 0

    def acquire(self, nbytes):
        size = max((int(nbytes) + (1 << 20) - 1) >> 20 << 20, 1 << 20)
        with self._lock:
            lst = self._free.get(size)
            if lst:
                return lst.pop()
        p = C.c_void_p()
        self.ctx._check(self.ctx.lib.pgw_host_alloc(self.ctx.handle, size, C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_ubyte * size).from_address(p.value))
        with self._lock:
            self._owned[p.value] = (size, arr)
            self.allocated += size
        return arr

    def release(self, arr):
        """Give a buffer (or any view that starts at its first byte) back."""
        addr = arr.ctypes.data
        with self._lock:
            ent = self._owned.get(addr)
            if ent is None:
                return False
            self._free.setdefault(ent[0], []).append(ent[1])
        return True

    def owns(self, arr):
        return arr.ctypes.data in self._owned

    def reclaim_all(self):
        """Every buffer this pool ever handed out is free again - for use after an aborted pipeline run, when nothing
        holds one any more (step_03_apply_to_era.reset_after_abort)."""
        with self._lock:
            self._free = {}
            for size, arr in self._owned.values():
                self._free.setdefault(size, []).append(arr)

    def close(self):
        with self._lock:
            for addr in list(self._owned):
                self.ctx.lib.pgw_host_free(self.ctx.handle, addr)
            self._owned.clear()
            self._free.clear()
            self.allocated = 0


_default = None
_default_lock = __import__('threading').Lock()


def default_context():
    """Process-wide context (device = LOCAL_RANK, one process per GPU); safe to call from the I/O threads."""
    global _default
    if _default is None:
        with _default_lock:
            if _default is None:
                _default = Context()
    return _default


def ptr(a):
    return None if a is None else a.ptr
