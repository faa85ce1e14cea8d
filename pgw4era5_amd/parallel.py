"""
File-level fan-out: one process per GPU (replaces the reference's `parallel.py`).

The reference distributes ERA5 files over a `multiprocessing.Pool`
(reference parallel.py:18-32, 53-68: `IterMP(njobs, run_async).run(func, fargs, step_args)`,
results in `.output`).  Files are independent units (step_03_apply_to_era.py:611-638), so the
MI355X equivalent is a rank-per-GPU launcher: task `i` goes to rank `i mod W`, every rank binds
to GPU `LOCAL_RANK`, there is no data-path collective, and the only communication is one
barrier plus an object gather of the (small) return values:

  * under `torchrun` / `python -m torch.distributed.run` (WORLD_SIZE > 1 in the environment)
    each rank runs its shard; `torch.distributed` with backend "nccl" (= RCCL) on GPUs, "gloo"
    on CPU-only hosts (tests);
  * started as a plain process with njobs > 1, `run` spawns njobs workers itself
    (`multiprocessing` spawn context, LOCAL_RANK = worker index) - same interface as the
    reference's `-p N`;
  * njobs == 1: serial loop in this process, like the reference.
"""
import multiprocessing as mp
import os
import sys


def shard_indices(ntasks, rank, world):
    """Round-robin deal of task indices to ranks."""
    return list(range(rank, ntasks, world))


def band_rows(nlat, rank, world):
    """Latitude rows [j0, j1) of rank `rank` when ONE file is split over `world` ranks (the
    latency mode).  Contiguous bands, sizes differing by at most one row; the columns of a file are independent but for
    the loop's stopping test (step_03_apply_to_era.py:189, 308), which band_max_hook makes global."""
    base, extra = divmod(int(nlat), int(world))
    j0 = rank * base + min(rank, extra)
    return j0, j0 + base + (1 if rank < extra else 0)


def band_max_hook(group=None):
    """The exchange step of the latency mode: element-wise MAX all-reduce of the loop's per-pass figures (at most 25
    doubles per loop launch) over the ranks of `group` - RCCL when the process group's backend is nccl (the values make
    one hop through a device tensor), gloo otherwise.  For Context.set_reduce_hook."""
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == 'nccl'

    def hook(vals):
        t = torch.from_numpy(vals)              # shares memory with `vals`
        if on_gpu:
            d = t.cuda()
            dist.all_reduce(d, op=dist.ReduceOp.MAX, group=group)
            t.copy_(d)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return hook


# ---- rank placement on the host -----------------------------------------------------------------------------------
# A rank moves 2.3 GB in and 2.3-4.6 GB out per file through pinned host buffers and runs reader / writer threads; on a
# two-socket host with 8 GPUs both should sit on the socket (NUMA node) its GPU hangs off, and ranks should not share
# cores.  The reference's Pool workers are unplaced (parallel.py:18-32).

def parse_cpulist(text):
    """'0-3,8,10-11' (the kernel's cpulist format) -> set of ints"""
    out = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(ndev):
    """NUMA node of HIP devices 0 .. ndev-1 (-1 where the kernel reports none): /sys/bus/pci/devices/<PCI address>/numa_node,
    the address from the library (pgw_device_pci_bus_id)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    have = C.c_int(0)
    lib.pgw_device_count(C.byref(have))
    nodes = []
    for d in range(ndev):
        buf = C.create_string_buffer(32)
        node = -1
        # more ranks than devices (a rehearsal on a box with fewer GPUs): rank r uses device r mod count, like Context
        if have.value > 0 and lib.pgw_device_pci_bus_id(d % have.value, buf, 32) == 0:
            try:
                with open('/sys/bus/pci/devices/%s/numa_node' % buf.value.decode().lower()) as f:
                    node = int(f.read().strip())
            except (OSError, ValueError):
                node = -1
        nodes.append(node)
    return nodes


def node_cpus(node):
    try:
        with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
            return parse_cpulist(f.read())
    except OSError:
        return set()


def rank_cpu_set(local_rank, local_world, allowed, gpu_nodes=None, cpus_of_node=node_cpus):
    """The CPUs rank `local_rank` of `local_world` ranks on this host binds to: the allowed CPUs (the process's affinity
    mask: a container may own a slice of the host) of its GPU's NUMA node, divided evenly - disjoint, in rank order - among
    the ranks whose GPUs share that node.  If any rank's node is unknown (-1) or has no allowed CPU, the allowed CPUs are
    dealt evenly to all ranks instead (still disjoint).  Pure function of its arguments (tests/test_host_logic.py)."""
    allowed = sorted(allowed)
    gpu_nodes = list(gpu_nodes) if gpu_nodes is not None else [-1] * local_world
    gpu_nodes += [-1] * (local_world - len(gpu_nodes))

    def deal(pool, ranks, sets):
        n = len(ranks)
        if len(pool) < n:                       # fewer CPUs than ranks: they share them
            for r in ranks:
                sets[r] = list(pool)
            return
        base, extra = divmod(len(pool), n)
        at = 0
        for k, r in enumerate(ranks):
            size = base + (1 if k < extra else 0)
            sets[r] = pool[at:at + size]
            at += size
    by_node, sets = {}, {}
    for r in range(local_world):
        node = gpu_nodes[r]
        pool = [c for c in allowed if c in cpus_of_node(node)] if node >= 0 else []
        if not pool:
            by_node = None
            break
        by_node.setdefault(node, (pool, []))[1].append(r)
    if by_node is None:
        deal(allowed, list(range(local_world)), sets)
    else:
        for pool, ranks in by_node.values():
            deal(pool, ranks, sets)
    return set(sets[local_rank])


def card_share(dist=None, local_rank=None):
    """How many ranks of this job use the SAME card as this one: every rank's (host, PCI address of its device) gathered over
    `torch.distributed` when a process group is up - right whether the ranks see all devices or one each
    (`HIP_VISIBLE_DEVICES` per rank) - else 1.  Exported as PGW_CARD_SHARE for `device.SpreadPool`, whose draw takes most of
    the card's free memory only when nobody shares it."""
    import ctypes as C
    import socket
    share = 1
    try:
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import _lib
            lib = _lib.load()
            have = C.c_int(0)
            lib.pgw_device_count(C.byref(have))
            d = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else int(local_rank)
            buf = C.create_string_buffer(32)
            addr = buf.value.decode() if (have.value > 0 and lib.pgw_device_pci_bus_id(d % have.value, buf, 32) == 0) else ''
            mine = (socket.gethostname(), addr)
            everyone = [None] * dist.get_world_size()
            dist.all_gather_object(everyone, mine)
            share = max(1, sum(1 for x in everyone if x == mine)) if addr else 1
    except Exception:                          # noqa: BLE001 - an optimisation's input: fall back to the environment's answer
        share = 1
    os.environ['PGW_CARD_SHARE'] = str(share)
    return share


def bind_rank_to_numa(local_rank=None, local_world=None):
    """Bind this process (and the threads it starts from now on: stage threads, NetCDF reader / writer pools) to its
    rank's CPU set, before pinned host buffers are allocated so that they are placed on that node.  PGW_NUMA_BIND=0
    switches it off.  Returns what was done, for logs and bench.py."""
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else int(local_rank)
    if local_world is None:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('PGW_LOCAL_WORLD', '1')))
    info = {'local_rank': local_rank, 'local_world': local_world, 'bound': False}
    if os.environ.get('PGW_NUMA_BIND', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        info['note'] = 'switched off' if hasattr(os, 'sched_setaffinity') else 'no sched_setaffinity on this platform'
        return info
    try:
        allowed = os.sched_getaffinity(0)
        try:
            nodes = gpu_numa_nodes(local_world)
        except Exception:                      # noqa: BLE001 - no library / no device: split the allowed CPUs evenly
            nodes = [-1] * local_world
        cpus = rank_cpu_set(local_rank, local_world, allowed, nodes)
        os.sched_setaffinity(0, cpus)
        info.update(bound=True, gpu_numa_node=nodes[local_rank] if local_rank < len(nodes) else -1, n_cpus=len(cpus),
                    cpus=_cpulist(cpus), allowed_cpus=len(allowed))
    except OSError as e:
        info['note'] = 'sched_setaffinity failed: %s' % e
    return info


def _cpulist(cpus):
    """set of ints -> '0-3,8' """
    cpus = sorted(cpus)
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else '%d-%d' % (cpus[i], cpus[j]))
        i = j + 1
    return ','.join(parts)


def _merge(fargs, step_args):
    tasks = []
    for s in step_args:
        kw = dict(fargs)
        kw.update(s)
        tasks.append(kw)
    return tasks


def run_shard(func, tasks, indices, depth=None, io_threads=None):
    """Run `func(**tasks[i])` for i in indices on this rank.  If `func.stages = (load, ..., store)` exists (three or more
    callables, each taking the previous one's result), the stages run as a pipeline over the files: `io_threads` reader
    threads stay up to `depth` files ahead (host I/O; `pread` and numpy release the GIL), every middle stage has ONE
    thread and works in task order (pgw_for_era5: upload on the 'h2d' stream, kernels, download on the 'd2h' stream - so
    the transfers of neighbouring files overlap the kernels and each other, PCIe being full duplex), `io_threads` writer
    threads store results.  The reference does all of this serially per worker (step_03_apply_to_era.py:44-381).
    A stage that raises stops the run: `func.abort` (a threading.Event, optional) is set so that stages waiting for a
    resource give up, and the exception of the earliest failing file is re-raised."""
    stages = getattr(func, 'stages', None)
    if not stages or len(stages) < 3 or len(indices) < 2:
        return [(i, func(**tasks[i])) for i in indices]
    from concurrent.futures import ThreadPoolExecutor
    # stage threads; the NetCDF reader / writer are themselves multi-threaded per file (ncio.py)
    io_threads = int(os.environ.get('PGW_IO_THREADS', '2')) if io_threads is None else io_threads
    depth = io_threads + len(stages) - 2 if depth is None else depth
    load, middle, store = stages[0], list(stages[1:-1]), stages[-1]
    abort = getattr(func, 'abort', None)
    if abort is not None:
        abort.clear()
    pools = [ThreadPoolExecutor(max_workers=io_threads)] + [ThreadPoolExecutor(max_workers=1) for _ in middle] + \
            [ThreadPoolExecutor(max_workers=io_threads)]
    results, chains = [], []                             # chains: (task index, future of the last stage), in task order

    def after(stage, prev):
        return lambda: stage(prev.result())              # waits for the previous stage of the SAME file (another pool)

    def submit(i):
        fut = pools[0].submit(load, **tasks[i])
        for k, stage in enumerate(middle):
            fut = pools[1 + k].submit(after(stage, fut))
        chains.append((i, pools[-1].submit(after(store, fut))))

    it = iter(indices)
    failed = False
    try:
        for _ in range(depth):                           # bound the files in flight (host buffers, device sets)
            i = next(it, None)
            if i is None:
                break
            submit(i)
        while chains:
            j, f = chains.pop(0)
            results.append((j, f.result()))              # re-raises a stage's exception
            i = next(it, None)
            if i is not None:
                submit(i)
    except BaseException:
        failed = True
        if abort is not None:
            abort.set()
        for p in pools:
            p.shutdown(wait=False, cancel_futures=True)
        raise
    finally:
        for p in pools:
            p.shutdown(wait=True)
        # a chain cancelled between two stages never hands back what its finished stages took (a device buffer set, pinned
        # buffers): once every stage thread has stopped, `func.reset` makes them all available again
        if failed and callable(getattr(func, 'reset', None)):
            func.reset()
    results.sort(key=lambda r: indices.index(r[0]))
    return results


def _worker(rank, world, func, tasks, queue):
    os.environ['LOCAL_RANK'] = str(rank)
    os.environ['PGW_RANK'] = str(rank)
    os.environ['PGW_LOCAL_WORLD'] = str(world)
    try:
        bind_rank_to_numa(rank, world)                  # before the context, its pinned pools and the stage threads exist
        out = run_shard(func, tasks, shard_indices(len(tasks), rank, world))
        queue.put((rank, out, None))
    except BaseException as e:   # noqa: BLE001 - reported to the parent, which re-raises
        queue.put((rank, [], '%s: %s' % (type(e).__name__, e)))


def _dist_env():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    return rank, world


def starmap_helper(tup):
    """reference parallel.py:12-15."""
    tup = dict(tup)
    func = tup.pop('func')
    return func(**tup)


def run_starmap(func, fargs={}, njobs=1, run_async=False):
    """`run_starmap(func, fargs, njobs, run_async)` of the reference (parallel.py:18-32): `fargs` is the list IterMP.run builds -
    one kwargs dict per task (wrapped in a 1-tuple with the function under 'func' when njobs > 1).  Tasks go to the ranks
    of this build (one process per GPU) instead of a multiprocessing.Pool; results in task order."""
    tasks = []
    for a in fargs:
        kw = dict(a[0] if isinstance(a, tuple) else a)
        kw.pop('func', None)
        tasks.append(kw)
    imp = IterMP(njobs=njobs, run_async=run_async)
    imp.run(func, {}, tasks)
    return imp.output


def test_IMP(iter_arg, fixed_arg):
    """The reference's self-test task (parallel.py:72-77): returns its first argument."""
    return iter_arg


class IterMP:
    """`IterMP(njobs=None, run_async=False).run(func, fargs={}, step_args=None)`; results of
    all tasks, in task order, in `.output` (reference parallel.py:36-68)."""

    def __init__(self, njobs=None, run_async=False, backend=None):
        self.run_async = run_async
        if njobs is None:
            njobs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
        self.njobs = int(njobs)
        self.backend = backend
        self.output = None
        print('IterMP: njobs = ' + str(self.njobs))

    def run(self, func, fargs={}, step_args=None):
        tasks = _merge(fargs, step_args or [])
        rank, world = _dist_env()
        if world > 1:
            self.output = self._run_distributed(func, tasks, rank, world)
        elif self.njobs > 1 and len(tasks) > 1:
            self.output = self._run_spawn(func, tasks, min(self.njobs, len(tasks)))
        else:
            self.output = [r for _, r in run_shard(func, tasks, list(range(len(tasks))))]
        return self.output

    # one process per GPU, launched by torchrun
    def _run_distributed(self, func, tasks, rank, world):
        import torch                      # before any HIP call of libpgw_hip.so (see _lib.py)
        import torch.distributed as dist
        created = False
        if not dist.is_initialized():
            backend = self.backend or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            dist.init_process_group(backend)
            created = True
        err = None
        mine = []
        try:
            bind_rank_to_numa()                          # LOCAL_RANK / LOCAL_WORLD_SIZE of torch.distributed.run
            card_share(dist)                             # PGW_CARD_SHARE for the placement draw (device.SpreadPool)
            mine = run_shard(func, tasks, shard_indices(len(tasks), rank, world))
        except Exception as e:            # noqa: BLE001 - every rank must reach the gather
            err = '%s: %s' % (type(e).__name__, e)
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, err))
        dist.barrier()
        if created:
            dist.destroy_process_group()
        errs = [e for _, e in gathered if e]
        if errs:
            raise RuntimeError('worker failed: ' + '; '.join(errs))
        out = [None] * len(tasks)
        for part, _ in gathered:
            for i, r in part:
                out[i] = r
        return out

    # self-spawned workers (plain `python step_03... -p N`)
    def _run_spawn(self, func, tasks, world):
        import queue as _queue
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, func, tasks, q)) for r in range(world)]
        for p in procs:
            p.start()
        out = [None] * len(tasks)
        errs = []
        pending = set(range(world))
        # A worker that dies without posting (a fault or abort inside the HIP library, an OOM kill, a result the queue's
        # feeder thread cannot pickle) must not leave this process waiting forever: poll the queue and the workers' exit
        # codes; a worker that has exited gets a short grace period for a message still in the pipe.
        dead_since = {}
        grace = float(os.environ.get('PGW_WORKER_GRACE_S', '5'))
        import time as _time
        while pending:
            try:
                rank, part, err = q.get(timeout=0.2)
            except _queue.Empty:
                now = _time.time()
                for r in list(pending):
                    if procs[r].exitcode is not None:
                        dead_since.setdefault(r, now)
                        if now - dead_since[r] > grace:
                            pending.discard(r)
                            errs.append('rank %d: worker process exited with code %s without delivering its results'
                                        % (r, procs[r].exitcode))
                if errs and pending:              # one rank is lost: stop the others, the run has failed
                    for r in pending:
                        if procs[r].is_alive():
                            procs[r].terminate()
                    break
                continue
            pending.discard(rank)
            if err:
                errs.append('rank %d: %s' % (rank, err))
            for i, r in part:
                out[i] = r
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        if errs:
            raise RuntimeError('worker failed: ' + '; '.join(errs))
        return out
