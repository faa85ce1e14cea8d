#!/usr/bin/env python
"""
bench.py - ERA5 files/hour of the step_03 hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole per-file compute path (reference
step_03_apply_to_era.py:62-346: RH of the ERA state, surface riders, the four delta
interpolations, the surface-pressure fixed-point loop, final PS/QV) over one synthetic ERA5
file of BASELINE.json configs[1]: global 0.25 deg (1440 x 721), L137, deltas on plev19.
Inputs (the ERA5 fields and all 12 monthly delta records) are resident in HBM when the timed
region starts; outputs stay in HBM.  One process per GPU; ranks work on independent files
(weak scaling, no data-path collective; RCCL only for the barriers / max-reduction of the
timing).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--storage f64|f32]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`
as a CHILD process, relays the child's JSON line and exits with its code (reference `-p N`,
parallel.py:18-32 / step_03_apply_to_era.py:630-638).  Every rank asserts WORLD_SIZE == --gpus.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant
kernel, HIP-event timed inside this run, beside the figure recomputed from the committed rocprofv3
summary), `cpu_baseline` (the numpy oracle on the host cores, rank 0, N = 1 only: one process and
file-parallel processes, on bounded latitude bands of the same file) and, at N = 1, `extras`
(PCIe-inclusive rate, end-to-end rate through the command line incl. NetCDF I/O, step_02 regridding of
BASELINE.json configs[3], float32-storage rates) - reported, never `value`; at N > 1 also `latency_mode`: ONE file over
all ranks in latitude bands, the loop's stopping test made global by an all-reduce MAX (RCCL) of the per-pass maxima.
"""
import argparse
import datetime as dt
import datetime as dt_mod
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--storage', choices=['f64', 'f32'], default='f64')
    p.add_argument('--f32-mode', choices=['reference', 'fast'], default='reference',
                   help="--storage f32 only: reference-dtype mode (float64 4-D outputs) or float64 arithmetic with float32 outputs")
    p.add_argument('--nlat', type=int, default=721)
    p.add_argument('--nlon', type=int, default=1440)
    p.add_argument('--nlev', type=int, default=137)
    p.add_argument('--cpu-rows', type=int, default=160, help='latitude rows of the one-process CPU-baseline sample')
    p.add_argument('--cpu-procs', type=int, default=0,
                   help='processes of the file-parallel CPU-baseline leg (0: min(host cores, 16), the CPU share of a 1-GPU box)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--overlap-streams', type=int, default=2,
                   help='extra (untimed-for-value) region with this many files in flight per GPU on separate HIP streams; 0 = skip')
    p.add_argument('--no-extras', action='store_true',
                   help='skip the N = 1 side measurements (PCIe-inclusive rate, end-to-end command line, step_02 regridding, '
                        'float32 storage)')
    p.add_argument('--extras', action='store_true', help='(default at N = 1; kept for older command lines)')
    p.add_argument('--e2e-files', type=int, default=4, help='files of the end-to-end command-line measurement (0 = skip)')
    p.add_argument('--e2e-dir', default=None, help='scratch directory for the end-to-end files (default: a temp dir)')
    p.add_argument('--full-column', action='store_true',
                   help='pass kernel reads every level (input-independent traffic) instead of stopping above p_ref')
    p.add_argument('--dry-run', action='store_true',
                   help='no GPU work: ranks rendezvous, count each other and print the line with value null '
                        '(CPU rehearsal of the multi-rank launch path; use with PGW_BENCH_BACKEND=gloo)')
    return p.parse_args(argv)


def spawn_ranks(a, argv):
    """`--gpus N` given to a plain process: start N ranks as a child `torch.distributed.run` (never exec: this
    process may be watched by a profiler that has already initialised the GPU) and relay its line and exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if r.stderr:
        sys.stderr.write(r.stderr[-8000:])
    if line is None:
        sys.stderr.write('bench.py: the %d-rank child printed no result line (exit code %d)\n%s\n'
                         % (a.gpus, r.returncode, r.stdout[-2000:]))
        return r.returncode or 1
    d = json.loads(line)
    if d.get('n_gpus') != a.gpus:
        sys.stderr.write('bench.py: asked for %d ranks, the line reports %r\n' % (a.gpus, d.get('n_gpus')))
        return 1
    print(line, flush=True)
    return r.returncode


def kernel_bytes(name, N, S, ncol, s, info):
    """Algorithmic bytes one launch of kernel `name` moves.  s = storage bytes of the ERA5 fields
    and delta records, info['so'] = bytes of the PGW level arrays (T_pgw, e, QV, U, V out: 8 in reference-dtype mode)."""
    so = info.get('so', s)
    if name == 'integ_geopot':        # pa_hl, ta, hus, zgs in; phi_ref out          (3N+3) ncol
        return (3 * N + 3) * ncol * s
    if name == 'adjust_ps_step':      # ta_pgw, e per level read; PS,FIS (storage) + 6 fp64 state words
        lv = info.get('levels_per_launch', N * ncol)
        return 2 * lv * so + ncol * (2 * s + 6 * 8)
    if name == 'ps_loop_multi':       # first launch of a file: T, QV of the ERA state below p_ref + passes_per_launch passes,
        # each re-reading ta_pgw, e below p_ref; PS, FIS, zg records in, 4 fp64 state words + one delta_ps per pass out
        lv = info.get('levels_per_launch', N * ncol)
        npass = info.get('passes_per_launch', 1.0)
        return 2 * lv * s + npass * 2 * lv * so + ncol * (4 * s + 4 * 8 + npass * 8)
    if name == 'quad_delta':          # T, QV, U, V in; T_pgw, e, U_pgw, V_pgw out; 2 records x S for 4 variables; 8 2-D fields
        return (4 * N * s + 4 * N * so + (8 * S + 8) * s) * ncol
    if name == 'thermo_delta':        # T, QV in; T_pgw, e out; 2 records x S for ta and hur; 7 2-D fields
        return (4 * N + 4 * S + 7) * ncol * s
    if name == 'wind_delta':          # U, V in; U_pgw, V_pgw out; 2 records x S for ua and va; PS
        return (4 * N + 4 * S + 1) * ncol * s
    if name == 'phi_ref_hybrid':      # T, QV below p_ref; PS, FIS in; phi_ref (fp64) out
        lv = info.get('levels_per_launch', N * ncol)
        return 2 * lv * s + ncol * (2 * s + 8)
    if name == 'vert_interp_delta':   # 2 records x S, add_to N in, N out, ps + surface pairs
        return (2 * S + 2 * N) * ncol * s + 5 * ncol * s
    if name == 'q_to_rh':             # QV, T in, RH out, PS
        return (3 * N + 1) * ncol * s
    if name == 'rh_to_q':             # hur, ta in, e out (vapour pressure pre-pass)
        return 3 * N * ncol * s
    if name == 'finalize':            # e in, QV out (levels the quad kernel has not finalised already), PS in/out, delta_ps
        return 2 * (N - info.get('qv_done_levels', 0)) * ncol * so + ncol * (2 * s + 8)
    if name == 'pressure':            # ps in, pa_hl out
        return (N + 2) * ncol * s
    return 0


def gather_objects(dist, obj, world):
    """list of every rank's `obj` (one-element list at N = 1)"""
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def scratch_ok(path, need_bytes):
    """(ok, note): does `path` have room for `need_bytes` of files without endangering the host?  Twice the bytes free on
    the file system; on a memory-backed file system (tmpfs / ramfs) also a third of MemAvailable at most."""
    import shutil
    os.makedirs(path, exist_ok=True)
    free = shutil.disk_usage(path).free
    if free < 2 * need_bytes:
        return False, 'scratch %s has %.0f GB free, the leg writes %.0f GB' % (path, free / 1e9, need_bytes / 1e9)
    fstype, best = None, ''
    try:
        real = os.path.realpath(path)
        for ln in open('/proc/mounts'):
            f = ln.split()
            if len(f) >= 3 and (real == f[1] or real.startswith(f[1].rstrip('/') + '/')) and len(f[1]) >= len(best):
                best, fstype = f[1], f[2]
        if fstype in ('tmpfs', 'ramfs'):
            avail = 0
            for ln in open('/proc/meminfo'):
                if ln.startswith('MemAvailable:'):
                    avail = int(ln.split()[1]) * 1024
            if need_bytes > avail / 3:
                return False, 'scratch %s is %s: %.0f GB of files against %.0f GB of available memory' % (path, fstype, need_bytes / 1e9, avail / 1e9)
    except OSError:
        pass
    return True, fstype


def cli_leg_all_ranks(a, rank, world, dist, barrier, files_per_rank=3):
    """End to end through the step_03 command line with ALL ranks at once, every rank writing into the same scratch
    directory: rank 0 writes two synthetic float32 0.25 deg L137 ERA5 files + the delta directory and links them under
    `files_per_rank x world` hourly names; every rank then runs `step_03_apply_to_era._cli(... -p W)` IN THIS PROCESS -
    under torch.distributed.run that is parallel.IterMP._run_distributed on the existing process group: file i -> rank
    i mod W, each rank the five-stage pipeline (pread -> h2d -> kernels -> d2h -> pwrite) on its GPU.  Reported: wall seconds
    per file over all ranks (slowest rank, start-up = delta upload + pinning included), the steady-state rate from the
    output files' modification times, for the reference's float64 T, QV, U, V and for settings.f32_out_dtype = 'float32'.
    This is what shared host memory bandwidth, PCIe and the scratch file system can make worse than 1 / N (reference
    `-p N`: parallel.py:18-32, step_03_apply_to_era.py:590-638)."""
    import glob
    import shutil
    import tempfile
    import numpy as np
    from pgw4era5_amd import synthetic, settings as S, step_03_apply_to_era as s3
    nfiles = files_per_rank * world
    one_in = 4 * a.nlat * a.nlon * a.nlev * 4 + 20 * a.nlat * a.nlon * 4
    res = {'files': nfiles, 'ranks': world}
    plan = None
    if rank == 0:
        base = a.e2e_dir or tempfile.mkdtemp(prefix='pgw_e2e_ranks_')
        ok64, note64 = scratch_ok(base, 2 * one_in + nfiles * 2 * one_in)
        ok32, note32 = scratch_ok(base, 2 * one_in + nfiles * one_in)
        plan = {'dir': base, 'modes': ([('float64', None)] if ok64 else [('float64', note64)]) +
                ([('float32', None)] if ok32 else [('float32', note32)])}
        if ok32:
            try:
                case = synthetic.make_case(a.nlat, a.nlon, a.nlev, seed=1, dtype=np.float32)
                first = dt.datetime(2006, 8, 2, 0)
                real = []
                for i in range(min(2, nfiles)):
                    case['target_dt'] = first + dt.timedelta(hours=i)
                    real.append(synthetic.write_case_files(case, os.path.join(base, 'era'), os.path.join(base, 'deltas')))
                for i in range(len(real), nfiles):       # further hourly names: links to the two files (reads are 0.05 s of a file's 0.3)
                    os.symlink(real[i % len(real)], os.path.join(base, 'era', S.era5_file_name_base.format(first + dt.timedelta(hours=i))))
                del case
            except Exception as e:      # noqa: BLE001
                plan = {'error': '%s: %s' % (type(e).__name__, e)}
    plan = gather_objects(dist, plan, world)[0]
    if plan is None or 'error' in plan:
        return {'error': (plan or {}).get('error', 'no plan')}
    base = plan['dir']
    first = dt.datetime(2006, 8, 2, 0)
    last = first + dt.timedelta(hours=nfiles - 1)
    old_debug, old_out = S.i_debug, S.f32_out_dtype
    S.i_debug = -1
    try:
        for mode, why_not in plan['modes']:
            key = 'float64_out' if mode == 'float64' else 'float32_out'
            if why_not is not None:
                res[key] = {'skipped': why_not}
                continue
            S.f32_out_dtype = mode
            outdir = os.path.join(base, 'out_' + mode)
            argv = ['-i', os.path.join(base, 'era'), '-o', outdir, '-d', os.path.join(base, 'deltas'),
                    '-f', first.strftime('%Y%m%d%H'), '-l', last.strftime('%Y%m%d%H'), '-H', '1', '-p', str(world), '-t']
            err = None
            barrier()
            t0 = time.time()
            try:
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):     # the command line's progress lines must not reach the stdout
                    n_iter = s3._cli(argv)                       # that carries this program's ONE result line
            except Exception as e:      # noqa: BLE001
                err, n_iter = '%s: %s' % (type(e).__name__, e), None
            barrier()
            wall = time.time() - t0
            errs = [e for e in gather_objects(dist, err, world) if e]
            if rank == 0:
                if errs:
                    res[key] = {'error': errs[0]}
                else:
                    done = sorted(os.path.getmtime(f) for f in glob.glob(os.path.join(outdir, '*.nc')))
                    k = min(world, len(done) - 1)       # the first file of every rank: pipeline fill, delta upload, pinning
                    steady = (done[-1] - done[k - 1]) / (len(done) - k) if (k >= 1 and len(done) > k and done[-1] > done[k - 1]) else None
                    size = os.path.getsize(sorted(glob.glob(os.path.join(outdir, '*.nc')))[0])
                    res[key] = dict(n_iter=sorted(set(n_iter)) if n_iter else None, out_file_GB=round(size / 1e9, 3),
                                    wall_s=round(wall, 2), wall_s_per_file=round(wall / nfiles, 3),
                                    steady_state_s_per_file=None if steady is None else round(steady, 4),
                                    files_per_hour_all_ranks=round(3600.0 / (steady if steady else wall / nfiles), 1))
                shutil.rmtree(outdir, ignore_errors=True)
            barrier()
    finally:
        S.i_debug, S.f32_out_dtype = old_debug, old_out
        if rank == 0 and not a.e2e_dir:
            shutil.rmtree(base, ignore_errors=True)
        elif rank == 0:
            for sub in ('era', 'deltas'):
                shutil.rmtree(os.path.join(base, sub), ignore_errors=True)
    return res


def pcie_f32_leg(ctx, case, coeffs, np):
    """The pipelined PCIe-inclusive rate of this rank for a float32 file in reference-dtype mode (2.3 GB in, 4.55 GB out)
    and with settings.f32_out_dtype = 'float32' (2.3 GB out) - the leg `extras.f32_storage` carries at N = 1."""
    import ctypes as C
    from pgw4era5_amd import step_03_apply_to_era as s3
    f32 = np.float32
    era32 = {k: (v.astype(f32) if isinstance(v, np.ndarray) and v.ndim >= 3 else v) for k, v in case['era'].items()}
    d32 = {k: v.astype(f32) for k, v in case['deltas'].items()}
    deltas = s3.DeltaSet(ctx, d32, case['delta_times'], case['plev'], f32)
    era = s3._upload_era(ctx, era32, f32)
    names = ('T', 'QV', 'U', 'V')
    n4 = era['T'].nbytes
    hp = []
    res = {}
    try:
        for k in names:
            p = C.c_void_p()
            ctx._check(ctx.lib.pgw_host_alloc(ctx.handle, n4, C.byref(p)))
            C.memmove(p, era32[k].ctypes.data, n4)
            hp.append(p)
        del era32, d32
        res['float64_out'] = pcie_pipelined(ctx, era, coeffs, deltas, case, True, names, hp, n4)
        res['float32_out'] = pcie_pipelined(ctx, era, coeffs, deltas, case, True, names, hp, n4, narrow=True)
    finally:
        for p in hp:
            ctx.lib.pgw_host_free(ctx.handle, p)
        for v in era.values():
            v.free()
        deltas.free()
    return res


def per_rank_object(aff_all, pcie_all, cli_all):
    """What an N-rank line reports beside the HBM-resident `value` - the legs that can fail to scale: every rank's host
    placement, every rank's pipelined PCIe-inclusive rate for a float32 file in reference-dtype mode (all ranks measuring at
    once), and the end-to-end command-line rate with all ranks writing into one scratch directory."""
    out = {'affinity': aff_all,
           'affinity_disjoint': None,
           'pcie_inclusive_f32_reference': None, 'end_to_end_cli_all_ranks': cli_all}
    sets = []
    try:
        from pgw4era5_amd.parallel import parse_cpulist
        sets = [parse_cpulist(x['cpus']) for x in aff_all if x and x.get('bound')]
        if len(sets) == len(aff_all) and sets:
            out['affinity_disjoint'] = all(not (sets[i] & sets[j]) for i in range(len(sets)) for j in range(i))
    except Exception:                              # noqa: BLE001
        pass
    if any(p for p in pcie_all):
        legs = {}
        for key in ('float64_out', 'float32_out'):
            vals = [(p or {}).get(key) if isinstance(p, dict) else None for p in pcie_all]
            ms = [v.get('ms_per_file') if isinstance(v, dict) else None for v in vals]
            good = [m for m in ms if m]
            legs[key] = {'ms_per_file_by_rank': ms,
                         'files_per_hour_all_ranks': round(sum(3.6e6 / m for m in good), 1) if good else None,
                         'GB_in': next((v.get('GB_in') for v in vals if isinstance(v, dict) and 'GB_in' in v), None),
                         'GB_out': next((v.get('GB_out') for v in vals if isinstance(v, dict) and 'GB_out' in v), None),
                         'errors': [v.get('error') for v in vals if isinstance(v, dict) and 'error' in v] or None}
        errs = [p['error'] for p in pcie_all if isinstance(p, dict) and 'error' in p]
        if errs:
            legs['errors'] = errs
        out['pcie_inclusive_f32_reference'] = legs
    return out


def roofline_object(dom, kern, a):
    """The `roofline` object of the line for kernel `dom` of a run described by `a` (storage, f32_mode, shape): HIP-event
    figure of this run, PMC traffic and rocprofv3 average from the committed summaries of the same command."""
    traffic, tsrc = pmc_traffic(dom, a)
    roof = dict(bound='hbm', kernel=dom, achieved=kern[dom]['GBps'], peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(kern[dom]['GBps'] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_unit='GB/launch',
                traffic_source=tsrc, avg_launch_ms=kern[dom]['avg_ms'],
                algorithmic_GB_per_launch=kern[dom]['algo_GB'],
                timing='HIP events on the launching stream, this run (frac); frac_rocprof = the same algorithmic bytes over '
                       'the committed rocprofv3 --kernel-trace average of the timed launches')
    roof.update(rocprof_frac(dom, a, kern[dom]['algo_GB']))
    valu = pmc_valu(dom, a, kern[dom]['avg_ms'])
    if valu:
        roof['valu_fp64'] = valu
    return roof


def dom_streams_ok(era, out):
    return all(k in era for k in ('T', 'QV', 'U', 'V')) and all(k in out for k in ('T', 'QV', 'U', 'V'))


def kernel_table(prof, N, S, ncol, s, kinfo):
    """per-kernel launches / average ms / algorithmic GB / GB/s from the context's HIP-event profile"""
    kern = {}
    for k, (cnt, ms) in prof.items():
        if cnt == 0:
            continue
        avg_ms = ms / cnt
        b = kernel_bytes(k, N, S, ncol, s, kinfo)
        kern[k] = dict(launches=cnt, avg_ms=round(avg_ms, 4), total_ms=round(ms, 3),
                       algo_GB=round(b / 1e9, 4), GBps=round(b / 1e9 / (avg_ms / 1e3), 1) if b else None)
    return kern


FILE_KERNELS = ('quad_delta', 'thermo_delta', 'wind_delta', 'phi_ref_hybrid', 'adjust_ps_step', 'ps_loop_multi', 'finalize',
                'surface', 'integ_geopot', 'vert_interp_delta', 'q_to_rh', 'rh_to_q', 'pressure', 'time_lerp')


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(a, argv)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE is %d: start it as `python bench.py --gpus N` or as '
                         '`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`\n' % (a.gpus, world))
        return 2
    dist = None
    backend = None
    # PGW_BENCH_FORCE_DIST=1: initialise the process group even for one rank, so that the RCCL code path (init, barrier,
    # all-reduce on device memory, torch's HIP runtime beside the library's) can be exercised on a one-GPU box
    if world > 1 or os.environ.get('PGW_BENCH_FORCE_DIST') == '1':
        # torch first: libpgw_hip.so then binds to the HIP runtime torch loaded (pgw4era5_amd/_lib.py)
        import torch
        import torch.distributed as dist
        # 'nccl' IS RCCL on ROCm.  PGW_BENCH_BACKEND=gloo only exists to rehearse the multi-rank code path
        # on a box with fewer GPUs than ranks (ranks then share devices: local % device_count) or without one (--dry-run).
        backend = os.environ.get('PGW_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            ndev = max(torch.cuda.device_count(), 1)
            local = local % ndev
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
            if not a.dry_run:
                local = local % max(torch.cuda.device_count(), 1)

    # host placement of this rank: its GPU's NUMA node, a CPU set disjoint from the other ranks' (before any pinned
    # allocation and before the stage threads exist)
    from pgw4era5_amd.parallel import bind_rank_to_numa, card_share
    try:
        affinity = bind_rank_to_numa(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    except Exception as e:                  # noqa: BLE001 - placement is an optimisation, never a reason to lose the line
        affinity = {'bound': False, 'note': '%s: %s' % (type(e).__name__, e)}
    affinity['card_share'] = card_share(dist, local)       # ranks on this rank's card (PGW_CARD_SHARE for the placement draw)

    def reduce(x, op):
        """float all-reduce over the ranks (RCCL on device memory, gloo on host memory)"""
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    if a.dry_run:
        if dist is not None:
            dist.barrier()
        seen = int(round(reduce(1.0, 'SUM')))
        el = reduce(1e-3 * (rank + 1), 'MAX')
        aff_all = gather_objects(dist, affinity, world)
        per_rank = per_rank_object(aff_all, [None] * world, None)
        if rank == 0:
            print(json.dumps({'metric': 'ERA5 files/hour (0.25deg L137), step_03 hot path, inputs resident in HBM',
                              'value': None, 'unit': 'files/hour', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                              'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                              'dtype': 'f64', 'data': 'synthetic', 'dry_run': True, 'per_rank': per_rank,
                              'collective': {'backend': backend, 'ranks_counted_by_all_reduce': seen, 'max_reduced': el},
                              'config': {'workload': 'dry run: launch path only, no GPU work'}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    import numpy as np
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
    from pgw4era5_amd.device import Context

    dtype = np.float64 if a.storage == 'f64' else np.float32
    ref = (a.storage == 'f32' and a.f32_mode == 'reference')
    s = np.dtype(dtype).itemsize
    so = 8 if (ref or a.storage == 'f64') else 4         # bytes per element of the PGW level arrays (T_pgw, e, QV, U, V out)
    t0 = time.time()
    case = synthetic.make_case(nlat=a.nlat, nlon=a.nlon, nlev=a.nlev, seed=1 + rank, dtype=dtype)
    t_gen = time.time() - t0
    # CPU baseline first: this process has not touched the GPU yet, so the file-parallel leg can fork its workers
    cpu = None
    if not a.no_cpu_baseline and world == 1 and rank == 0:
        try:
            cpu = cpu_baseline(case, a, np)
        except Exception as e:                      # noqa: BLE001 - reported in the line, the GPU measurement goes on
            cpu = {'error': '%s: %s' % (type(e).__name__, e)}
    ctx = Context(local)
    if a.full_column:
        ctx.set_option('full_column', 1)
    from pgw4era5_amd import device as _device
    _device._default = ctx                 # the functions.py mirror uses the process-wide context
    ncol = a.nlat * a.nlon
    N, S = a.nlev, len(case['plev'])
    # level arrays over the card's memory regions (settings.placement = 'spread'; PGW_PLACEMENT=plain for the A/B): inputs,
    # outputs and the vapour-pressure workspace of the file path, 9 arrays of one float64 field (+ 8 for the float32 leg's file, + 2 for the signature kernels' pa_hl and pa)
    nt_, N_, nlat_, nlon_ = case['era']['T'].shape
    placement = ctx.enable_placement(nt_ * (N_ + 1) * nlat_ * nlon_ * 8,            # the half-level field pa_hl is the largest
                                     (9 if a.no_extras else 17) + (2 if (rank == 0 and world == 1) else 0))
    deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], dtype)
    era = s3._upload_era(ctx, case['era'], dtype)
    coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
    # one ERA5 file per step: hourly stamps (config 3 of BASELINE.json is 24 hourly files)
    stamps = [case['target_dt'] + dt.timedelta(hours=i) for i in range(a.steps + a.warmup)]
    out = {}

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            if backend == 'nccl':
                import torch
                torch.cuda.synchronize()

    infos = []
    smi = None
    if rank == 0 and not os.environ.get('PGW_BENCH_NOSMI'):
        # the card's own telemetry over the run (gfx clock, socket power against its cap, temperatures): the file path drives
        # an MI355X to its power cap, and the clock the firmware then allows is what differs between boxes
        try:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'smi'))
            from sampler import Sampler
            smi = Sampler(device=local).start()
        except Exception:                           # noqa: BLE001 - telemetry is optional
            smi = None
    if not os.environ.get('PGW_BENCH_NOPROF'):      # A/B knob: cost of the per-launch HIP events
        ctx.profile(True)                           # on during warm-up too, so the event pool is populated
    for i in range(a.warmup):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, stamps[i], True, out=out, ref_dtype=ref)
    ctx.profile_reset()
    barrier()
    tm0 = time.monotonic()
    t0 = time.perf_counter()
    for i in range(a.steps):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, stamps[a.warmup + i], True, out=out, ref_dtype=ref)
        infos.append(info)
    barrier()
    elapsed = time.perf_counter() - t0
    tm1 = time.monotonic()
    prof = {k: ctx.profile_get(k) for k in FILE_KERNELS}
    solo = (rank == 0 and world == 1)
    bare = None
    if rank == 0 and dom_streams_ok(era, out):
        # the quad kernel's access pattern with no arithmetic (4 read + 4 write streams, Context.placement_probe) on the very
        # arrays of this run: what this box gives the pattern WHERE these arrays lie (5.0-6.2 TB/s between allocations, DESIGN.md
        # section 4).  Overwrites the outputs of the last file; every later leg writes them anew.
        try:
            ctx.profile(False)
            g = ctx.placement_probe([era[k] for k in ('T', 'QV', 'U', 'V')], [out[k] for k in ('T', 'QV', 'U', 'V')], reps=5)
            bare = {'GBps': round(g, 1), 'frac_of_peak': round(g / HBM_PEAK_GBS, 4),
                    'what': 'k_placement_probe: 4 read + 4 write streams over the ERA inputs and the outputs of this run, '
                            'one thread per column, two levels per step, streaming loads / stores, no arithmetic'}
            if era['T'].dtype.itemsize != 8:
                bare['note'] = ('float32 inputs: the probe moves their BYTES (as half as many float64 rows) and writes as many rows '
                                '- half the kernel\'s output rows; a looser yardstick than for float64 files')
            ctx.profile(not os.environ.get('PGW_BENCH_NOPROF'))
        except Exception as e:                      # noqa: BLE001 - a side measurement
            bare = {'error': '%s: %s' % (type(e).__name__, e)}
    micro = microbench(ctx, era, coeffs, a, np) if solo else {}       # N = 1 only: keeps multi-rank runs short
    for k, v in micro.items():
        # beside the HIP-event figure of this run: the fraction from the committed rocprofv3 trace of the same command.
        # They differ most for `pressure`, a pure-write kernel whose event time depends on what the kernel before it left
        # in the caches (0.33-0.39 ms); the rocprofv3 average is the figure to quote.
        rp = rocprof_frac(k, a, v['algo_GB'])
        if rp:
            v['frac_rocprof'] = rp['frac_rocprof']
            v['rocprof_timed_avg_launch_ms'] = rp['rocprof_timed_avg_launch_ms']
            v['rocprof_source'] = rp['rocprof_source']
    overlap = overlap_region(local, era, coeffs, deltas, stamps[a.warmup:], a, ref) if (solo and a.overlap_streams > 1) else None
    ctx.profile(False)
    latency = None
    bk_host = case['era']['bk']
    pcie_rank, cli_all = None, None
    if dist is not None:                            # N > 1 (and the one-rank RCCL rehearsal, PGW_BENCH_FORCE_DIST=1)
        if world > 1:
            if not a.no_extras:
                try:                                # every rank at once: PCIe and host memory bandwidth are shared
                    barrier()
                    pcie_rank = pcie_f32_leg(ctx, case, coeffs, np)
                except Exception as e:              # noqa: BLE001
                    pcie_rank = {'error': '%s: %s' % (type(e).__name__, e)}
            case = None                             # host copy of this rank's file (~15 GB): free it before the shared file is built
        try:
            latency = latency_mode(ctx, a, rank, world, dist, backend, dtype, ref)
        except Exception as e:                      # noqa: BLE001 - a side measurement, never costs the headline line
            latency = {'error': '%s: %s' % (type(e).__name__, e)}
    if dist is not None and world > 1 and not a.no_extras and a.e2e_files > 0:
        try:
            cli_all = cli_leg_all_ranks(a, rank, world, dist, barrier)
        except Exception as e:                      # noqa: BLE001
            cli_all = {'error': '%s: %s' % (type(e).__name__, e)}
    aff_all = gather_objects(dist, affinity, world)
    pcie_all = gather_objects(dist, pcie_rank, world)
    elapsed = reduce(elapsed, 'MAX')
    ranks_seen = int(round(reduce(1.0, 'SUM')))
    iters_min = int(round(reduce(float(min(i['n_iter'] for i in infos)), 'MIN')))
    iters_max = int(round(reduce(float(max(i['n_iter'] for i in infos)), 'MAX')))

    if rank == 0:
        n_iter = [i['n_iter'] for i in infos]
        passes = sum(n_iter)
        lv_per_launch = sum(i['levels_touched'] for i in infos) / max(passes, 1)
        bk = bk_host
        n_pure = 0
        while n_pure < N and (0.5 * (bk[n_pure + 1] - bk[n_pure]) + bk[n_pure]) == 0.0:
            n_pure += 1
        # pure-pressure levels: their final QV is written by k_delta_quad (stop-above-p_ref passes)
        quad = not a.full_column and ctx.get_option('quad') != 0
        launches_multi = prof['ps_loop_multi'][0]
        kinfo = dict(levels_per_launch=lv_per_launch, qv_done_levels=n_pure if quad else 0, so=so,
                     passes_per_launch=(sum(i.get('passes_launched', 0) for i in infos) / launches_multi) if launches_multi else 1.0)
        kern = kernel_table(prof, N, S, ncol, s, kinfo)
        cand = [k for k in kern if kern[k]['GBps']]
        if not cand:                       # PGW_BENCH_NOPROF: no per-kernel timings
            print(json.dumps({'ms_per_step': round(elapsed / a.steps * 1e3, 3), 'note': 'launch profiling disabled'}), flush=True)
            return 0
        dom = max(cand, key=lambda k: kern[k]['total_ms'])
        roof = roofline_object(dom, kern, a)
        if bare is not None:
            roof['bare_pattern_same_arrays'] = bare
            if bare.get('GBps') and kern[dom]['GBps']:
                roof['frac_of_bare_pattern'] = round(kern[dom]['GBps'] / bare['GBps'], 4)
        files = a.steps * world
        res = {
            'metric': 'ERA5 files/hour (0.25deg L137), step_03 hot path, inputs resident in HBM',
            'value': round(files / elapsed * 3600.0, 1), 'unit': 'files/hour',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'Single ERA5 file, global 0.25deg (%dx%d) L%d, plev%d monthly deltas, per MI355X'
                                   % (a.nlon, a.nlat, a.nlev, S),
                       'storage': a.storage if a.storage == 'f64' else 'f32 (%s mode)' % a.f32_mode,
                       'files_per_rank': a.steps, 'sharding': 'file i -> rank i mod W, no data-path collective',
                       'iterations_per_file': n_iter[0] if len(set(n_iter)) == 1 else n_iter,
                       'iterations_per_file_over_ranks': [iters_min, iters_max],
                       'pass_kernel': 'full_column' if a.full_column else 'stops_above_p_ref',
                       'passes_launched_per_file': round(sum(i.get('passes_launched', 0) for i in infos) / len(infos), 2),
                       'loop_launches_per_file': round((prof['ps_loop_multi'][0] + prof['adjust_ps_step'][0]) / len(infos), 2),
                       'mean_levels_read_per_column_per_pass': round(lv_per_launch / ncol, 2)},
            'collective': {'backend': ('rccl (torch nccl)' if backend == 'nccl' else backend) if dist is not None else None,
                           'ranks_counted_by_all_reduce': ranks_seen,
                           'use': 'barrier + MAX of the elapsed time; files are independent'},
            'roofline': roof,
            'kernels': kern,
            'signature_kernels': micro,
            'extras': None,
            'overlap': overlap,
            'latency_mode': latency,
            'device': ctx.device_name(),
            'placement': placement if placement is not None else {'mode': 'plain'},
            'setup_s': round(t_gen, 1),
        }
        if solo and not a.no_extras:
            try:
                res['extras'] = extras(ctx, case, era, coeffs, deltas, a, np)
            except Exception as e:          # noqa: BLE001 - side measurements never cost the headline line
                res['extras'] = {'error': '%s: %s' % (type(e).__name__, e)}
        f32s = (res['extras'] or {}).get('f32_storage') or {}
        if world == 1 and res['extras']:            # the same legs, measured by `extras` at N = 1
            ex = res['extras']
            pcie_all = [{'float64_out': f32s.get('pcie_inclusive_reference'), 'float32_out': f32s.get('pcie_inclusive_reference_f32_out')}]
            cli_all = {'files': a.e2e_files, 'ranks': 1, 'float64_out': ex.get('end_to_end_cli'), 'float32_out': ex.get('end_to_end_cli_f32_out')}
        res['per_rank'] = per_rank_object(aff_all, pcie_all, cli_all)
        if 'roofline' in f32s.get('reference', {}):
            # float32 files in reference-dtype mode - what real ERA5 files run by default - beside the float64 object
            res['roofline_f32ref'] = dict(f32s['reference']['roofline'], ms_per_file=f32s['reference']['ms_per_file'],
                                          iterations=f32s['reference']['iterations'])
        if smi is not None:
            smi.stop()
            res['device_state'] = smi.summary(tm0, tm1)
        if cpu is not None:
            res['cpu_baseline'] = cpu
        assert res['n_gpus'] == a.gpus == ranks_seen, (res['n_gpus'], a.gpus, ranks_seen)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def latency_mode(ctx, a, rank, world, dist, backend, dtype, ref, files=6, warmup=2):
    """SURVEY.md section 8e, row 2, measured beside `value` at N > 1: ONE file split into N latitude bands, one per rank;
    the loop's stopping test is made global by an all-reduce MAX of the per-pass maxima (pgw_set_reduce_hook; RCCL when the
    backend is nccl).  Every rank builds the same synthetic file (seed 1) and keeps its band.  ms per file = wall time of
    the slowest rank between two barriers / files; a strong-scaling figure (time to one file), not files/hour."""
    import torch
    import numpy as np
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
    from pgw4era5_amd.parallel import band_rows, band_max_hook
    dev = 'cuda' if backend == 'nccl' else 'cpu'
    ready, err = 1.0, None
    try:
        case = synthetic.make_case(nlat=a.nlat, nlon=a.nlon, nlev=a.nlev, seed=1, dtype=dtype)
        j0, j1 = band_rows(a.nlat, rank, world)
        era = s3._upload_era(ctx, s3._band_of(case['era'], j0, j1), dtype)
        deltas = s3.DeltaSet(ctx, s3._band_of(case['deltas'], j0, j1), case['delta_times'], case['plev'], dtype)
        coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
        stamps = [case['target_dt'] + dt.timedelta(hours=i) for i in range(files + warmup)]
    except Exception as e:                          # noqa: BLE001
        ready, err = 0.0, e
    # every rank must enter the band loop or none: its reduces are collective
    t = torch.tensor([ready], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if float(t.item()) < 1.0:
        raise RuntimeError('set-up failed on a rank%s' % ('' if err is None else ' (this one: %s: %s)' % (type(err).__name__, err)))
    out, n_iter = {}, []
    ctx.set_reduce_hook(band_max_hook())
    try:
        for i in range(warmup):
            s3.process_file_device(ctx, era, coeffs, deltas, stamps[i], True, out=out, ref_dtype=ref)
        ctx.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(files):
            _, info = s3.process_file_device(ctx, era, coeffs, deltas, stamps[warmup + i], True, out=out, ref_dtype=ref)
            n_iter.append(info['n_iter'])
        ctx.sync()
        dist.barrier()
        el = time.perf_counter() - t0
    finally:
        ctx.set_reduce_hook(None)
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    for v in out.values():
        v.free()
    return {'bands': world, 'rows_of_rank0': j1 - j0, 'files': files, 'ms_per_file': round(float(t.item()) / files * 1e3, 3),
            'iterations_per_file': n_iter[0] if len(set(n_iter)) == 1 else n_iter,
            'exchange': 'all-reduce MAX of 1 + 3 K doubles per loop launch (K passes), %s' % ('rccl' if backend == 'nccl' else backend),
            'scaling': 'strong (one file over all ranks)'}


def overlap_region(device, era, coeffs, deltas, stamps, a, ref=False):
    """Same K files, but `--overlap-streams` files in flight per GPU: one host thread + one
    pgw_ctx (HIP stream, workspaces, outputs) per in-flight file, so the fp64-VALU-bound delta
    kernels of one file overlap the HBM-bound loop / finalize kernels of another and the host-side
    status reads of one stream hide behind the other's kernels.  Reported beside `value`, not as it
    (per-kernel event timings of overlapped kernels would not be clean roofline inputs)."""
    import threading
    from pgw4era5_amd import step_03_apply_to_era as s3
    from pgw4era5_amd.device import Context
    n = a.overlap_streams
    ctxs = [Context(device) for _ in range(n)]
    outs = [{} for _ in range(n)]
    dsets = []
    for c in ctxs:                     # same device arrays, other context (arrays are plain device pointers)
        d = s3.DeltaSet.__new__(s3.DeltaSet)
        d.__dict__.update(deltas.__dict__)
        d.ctx = c
        dsets.append(d)
    for i, c in enumerate(ctxs):       # warm-up: workspaces + level tables of every context
        s3.process_file_device(c, era, coeffs, dsets[i], stamps[0], True, out=outs[i], ref_dtype=ref)
        c.sync()
    errs = []

    def work(i):
        try:
            for k in range(i, len(stamps), n):
                s3.process_file_device(ctxs[i], era, coeffs, dsets[i], stamps[k], True, out=outs[i], ref_dtype=ref)
            ctxs[i].sync()
        except Exception as e:        # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    for o in outs:
        for v in o.values():
            v.free()
    for c in ctxs:
        c.close()
    if errs:
        return {'error': errs[0]}
    return {'streams': n, 'files': len(stamps), 'ms_per_file': round(el / len(stamps) * 1e3, 3),
            'files_per_hour_per_gpu': round(len(stamps) / el * 3600.0, 1)}


def extras(ctx, case, era, coeffs, deltas, a, np):
    """Measurements that are reported but are not `value`: (1) step_02 bilinear regridding of one
    19-level variable x 12 months from a 192x384 Gaussian-like grid to the ERA5 grid
    (BASELINE.json configs[3]); (2) the PCIe-inclusive rate of the step_03 path: T, QV, U, V, PS
    from pinned host memory to the device, the path, and T, QV, U, V, PS back."""
    import ctypes as C
    from pgw4era5_amd import functions as F, synthetic, step_03_apply_to_era as s3
    out = {}
    dt = era['T'].dtype
    s = dt.itemsize
    g = synthetic.make_gcm_grid_case(nlat_src=192, nlon_src=384, nlat=a.nlat, nlon=a.nlon, nplev=19, ntime=12, seed=4,
                                     dtype=dt)
    src = ctx.to_device(g['field'], dt)
    F.regrid_field(src, g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon']).free()
    ctx.profile(True); ctx.profile_reset()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        r = F.regrid_field(src, g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon'])
        ctx.sync()
        r.free()
    wall = (time.perf_counter() - t0) / reps
    cnt, ms = ctx.profile_get('regrid')
    ctx.profile(False); ctx.profile_reset()
    nbytes = (12 * 19 * a.nlat * a.nlon + g['field'].size) * s
    out['regrid_one_var_12_months'] = dict(kernel_ms=round(ms / cnt, 3), wall_ms=round(wall * 1e3, 3), algo_GB=round(nbytes / 1e9, 3),
                                           GBps=round(nbytes / 1e9 / (ms / cnt / 1e3), 1))
    src.free()
    # step_02 smoothing: one daily 19-level delta on the 192x384 GCM grid (366 x 19 x 192 x 384), values irrelevant
    lt, inner = 366, 19 * 192 * 384
    cos_t, sin_t = F.harmonic_tables(lt)
    d_in, d_out = ctx.empty((lt, inner), dt), ctx.empty((lt, inner), dt)
    ctx._check(ctx.lib.pgw_memset(ctx.handle, d_in.ptr, 0x3f, d_in.nbytes))
    dp = C.POINTER(C.c_double)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(4):
        ctx._check(ctx.lib.pgw_harmonic_smooth(ctx.handle, 1 if s == 8 else 0, lt, inner, cos_t.ctypes.data_as(dp),
                                               sin_t.ctypes.data_as(dp), d_in.ptr, d_out.ptr))
    ctx.sync()
    cnt, ms = ctx.profile_get('harmonic')
    ctx.profile(False); ctx.profile_reset()
    out['harmonic_smooth_daily_19lev'] = dict(kernel_ms=round(ms / cnt, 3), algo_GB=round(2 * d_in.nbytes / 1e9, 3),
                                              GBps=round(2 * d_in.nbytes / 1e9 / (ms / cnt / 1e3), 1))
    d_in.free(); d_out.free()
    # byte-order conversion of one 4-D field in place (the I/O path runs it on T, QV, U, V both ways)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        ctx._check(ctx.lib.pgw_byteswap(ctx.handle, s, era['U'].size, era['U'].ptr, era['U'].ptr))
        ctx._check(ctx.lib.pgw_byteswap(ctx.handle, s, era['U'].size, era['U'].ptr, era['U'].ptr))
    ctx.sync()
    cnt, ms = ctx.profile_get('byteswap')
    ctx.profile(False); ctx.profile_reset()
    out['byteswap_one_field'] = dict(kernel_ms=round(ms / cnt, 4), algo_GB=round(2 * era['U'].nbytes / 1e9, 3),
                                     GBps=round(2 * era['U'].nbytes / 1e9 / (ms / cnt / 1e3), 1))
    # PCIe-inclusive: pinned staging buffers, one stream (copies and kernels serialised; the overlapped
    # variant is bounded by the same PCIe time, which dominates)
    names = ('T', 'QV', 'U', 'V')
    n4 = era['T'].nbytes
    hp = []
    for _ in range(4):
        p = C.c_void_p()
        ctx._check(ctx.lib.pgw_host_alloc(ctx.handle, n4, C.byref(p)))
        hp.append(p)
    for k, p in zip(names, hp):
        C.memmove(p, case['era'][k].ctypes.data, n4)
    res = {}
    ref = (a.storage == 'f32' and a.f32_mode == 'reference')
    s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=res, ref_dtype=ref)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        for k, p in zip(names, hp):
            ctx._check(ctx.lib.pgw_memcpy_h2d(ctx.handle, era[k].ptr, p, n4))
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=res, ref_dtype=ref)
        for k, p in zip(names, hp):
            ctx._check(ctx.lib.pgw_memcpy_d2h(ctx.handle, p, res[k].ptr, n4))      # n4 bytes of each output (all of it unless ref)
        ctx.sync()
    el = (time.perf_counter() - t0) / reps
    out['pcie_inclusive'] = dict(ms_per_file=round(el * 1e3, 2), files_per_hour=round(3600.0 / el, 1),
                                 GB_moved_each_way=round(4 * n4 / 1e9, 3),
                                 note='pinned H2D of T,QV,U,V + path + D2H of T,QV,U,V on one stream')
    for v in res.values():
        v.free()
    try:
        out['pcie_inclusive']['pipelined'] = pcie_pipelined(ctx, era, coeffs, deltas, case, ref, names, hp, n4)
    except Exception as e:                  # noqa: BLE001
        out['pcie_inclusive']['pipelined'] = {'error': '%s: %s' % (type(e).__name__, e)}
    for p in hp:
        ctx.lib.pgw_host_free(ctx.handle, p)
    # step_02 for tos / siconc: 12 months of an ocean-grid delta (802 x 404 curvilinear points, NaN over land) onto the ERA5 grid
    try:
        oc = synthetic.make_ocean_grid_case(nj=404, ni=802, ntime=12, seed=6, land_patches=6)
        lat = np.linspace(-90.0, 90.0, a.nlat); lon = np.arange(a.nlon) * (360.0 / a.nlon)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        r = F.gauss_interp_fields(np.zeros((a.nlat, a.nlon)), lat, lon, oc['latitude'], oc['longitude'], list(oc['values']), 1.0e6, 4.0)
        wall = time.perf_counter() - t0
        cnt, ms = ctx.profile_get('gauss_interp')
        ctx.profile(False); ctx.profile_reset()
        out['gauss_interp_tos_12_months'] = dict(kernel_ms=round(ms / max(cnt, 1), 3), wall_s_incl_host_geometry=round(wall, 2),
                                                 source_points=int((~np.isnan(oc['values'][0])).sum()) * 3, targets=a.nlat * a.nlon,
                                                 valid_targets=int((~np.isnan(r[0])).sum()))
        del r, oc
    except Exception as e:      # noqa: BLE001
        out['gauss_interp_tos_12_months'] = {'error': '%s: %s' % (type(e).__name__, e)}
    # the reference's two other operating modes on the same resident file (settings.p_ref_inp = None: step_03:219-253;
    # settings.i_reinterp = 1: step_03:202-216, 330-343), HBM-resident like `value`
    for key, which in (('local_p_ref', 'local'), ('i_reinterp', 'reinterp')):
        try:
            out[key] = mode_leg(ctx, era, coeffs, deltas, case, a, which)
        except Exception as e:      # noqa: BLE001
            out[key] = {'error': '%s: %s' % (type(e).__name__, e)}
    # float32 storage (what real ERA5 files hold), HBM-resident like `value`: both modes of settings.f32_file_mode
    if dt == np.float64:
        try:
            out['f32_storage'] = f32_storage(ctx, case, coeffs, a, np)
        except Exception as e:      # noqa: BLE001
            out['f32_storage'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if a.e2e_files > 0:
        try:
            out['end_to_end_cli'] = end_to_end(a)
        except Exception as e:      # noqa: BLE001
            out['end_to_end_cli'] = {'error': '%s: %s' % (type(e).__name__, e)}
        try:                        # same run with settings.f32_out_dtype = 'float32' (half the download and the file)
            r = end_to_end(a, 'float32')
            out['end_to_end_cli_f32_out'] = {k: r[k] for k in ('n_iter', 'steady_state_s_per_file', 'serial_stage_s', 'files_per_hour_one_rank') if k in r}
        except Exception as e:      # noqa: BLE001
            out['end_to_end_cli_f32_out'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def mode_leg(ctx, era, coeffs, deltas, case, a, which, files=4, warmup=2):
    """ms per file, passes and the per-kernel HIP-event table (launches and ms per file, algorithmic GB per launch where
    bench.py defines them) of the step_03 path in another of the reference's modes: which = 'local' (p_ref_inp = None, the
    LOCAL form of the multi-pass loop kernel) or 'reinterp' (i_reinterp = 1: ta / hur and their deltas re-interpolated onto
    the current levels in every pass, ua / va once at the end)."""
    from pgw4era5_amd import _lib, step_03_apply_to_era as s3
    s = era['T'].dtype.itemsize
    nt, N, nlat, nlon = era['T'].shape
    ncol, S = nlat * nlon, len(case['plev'])
    outb = {}

    def one(i):
        stamp = case['target_dt'] + dt_mod.timedelta(hours=i)
        if which == 'local':
            return s3.process_file_device(ctx, era, coeffs, deltas, stamp, True, p_ref='local', out=outb)[1]
        return s3.process_file_device_reinterp(ctx, era, coeffs, deltas, stamp, True, out=outb)[1]
    ctx.profile(True)
    for i in range(warmup):
        one(i)
    ctx.sync()
    ctx.profile_reset()
    t0 = time.perf_counter()
    infos = [one(warmup + i) for i in range(files)]
    ctx.sync()
    el = (time.perf_counter() - t0) / files
    prof = {k: ctx.profile_get(k) for k in _lib.KERNEL_IDS}
    ctx.profile(False); ctx.profile_reset()
    for v in outb.values():
        v.free()
    passes = sum(i['n_iter'] for i in infos)
    bk = coeffs['bk']
    n_pure = 0
    while n_pure < N and (0.5 * (bk[n_pure + 1] - bk[n_pure]) + bk[n_pure]) == 0.0:
        n_pure += 1
    touched = sum(i.get('levels_touched', 0) for i in infos)
    so = 8 if s3.ref_dtype_mode(era['T'].dtype) else s          # float32 files in reference-dtype mode: float64 level arrays out
    kinfo = dict(so=so, levels_per_launch=touched / max(passes, 1), qv_done_levels=n_pure if ctx.get_option('quad') != 0 else 0,
                 passes_per_launch=(sum(i.get('passes_launched', 0) for i in infos) / prof['ps_loop_multi'][0])
                 if prof['ps_loop_multi'][0] else 1.0)
    kern = {}
    for k, (cnt, ms) in prof.items():
        if not cnt:
            continue
        # i_reinterp: the pair kernel (k_reinterp_pair: two ERA fields + two deltas onto the current levels) runs under the
        # vert_interp_delta id: 2 fields in, 2 out, 2 records x S x 2 variables, ~6 2-D fields
        if which == 'reinterp' and k == 'vert_interp_delta':
            # ta + hur inside the loop (n_iter launches per file): T (s) and RELHUM (so) in, ta_pgw, hur_pgw and e (so) out;
            # ua + va once: U, V (s) in, two fields (so) out; 2 records x S x 2 variables and ~6 2-D fields each
            n_th, n_w = sum(i['n_iter'] for i in infos), len(infos)
            b = (n_th * (N * (s + so) + 3 * N * so) + n_w * (2 * N * s + 2 * N * so)) / (n_th + n_w) * ncol + (4 * S + 6) * ncol * s
        elif k in ('adjust_ps_step', 'ps_loop_multi', 'phi_ref_hybrid') and not touched:
            b = 0                                # levels read per pass not reported by this path: no byte figure
        else:
            b = kernel_bytes(k, N, S, ncol, s, kinfo)
        kern[k] = dict(launches_per_file=round(cnt / files, 2), ms_per_file=round(ms / files, 3), avg_launch_ms=round(ms / cnt, 4),
                       algo_GB_per_launch=round(b / 1e9, 3) if b else None,
                       GBps=round(b / 1e9 / (ms / cnt / 1e3), 1) if b else None)
    n_iter = [i['n_iter'] for i in infos]
    return dict(ms_per_file=round(el * 1e3, 3), files_per_hour=round(3600.0 / el, 1), files=files,
                iterations=n_iter[0] if len(set(n_iter)) == 1 else n_iter, kernels=kern)


def pcie_pipelined(ctx, era, coeffs, deltas, case, ref, names, hp, n4, files=8, narrow=False):
    """The same host-to-host work as `pcie_inclusive`, organised like the file driver (step_03_apply_to_era.py stages):
    uploads on the 'h2d' stream, kernels on the context's stream, downloads on the 'd2h' stream, two device buffer sets
    each way, one host thread per stage - so the upload of file i+1 and the download of file i-1 run during the kernels of
    file i and PCIe carries both directions at once.  ms per file = wall time of `files` files / files (fill and drain
    included).  n4 = bytes of one input field; an output field is downloaded whole (twice n4 in reference-dtype mode:
    float64 T, QV, U, V of a float32 file), or, with narrow (settings.f32_out_dtype = 'float32'), after
    pgw_narrow_f64_f32 on the GPU (n4 again)."""
    import ctypes as C
    import queue
    import threading
    import numpy as np
    from pgw4era5_amd import step_03_apply_to_era as s3
    up, dn = ctx.side('h2d'), ctx.side('d2h')
    second = {k: (ctx.empty(v.shape, v.dtype) if k in names else v) for k, v in era.items()}   # small fields: shared, read-only
    in_sets, out_sets = [era, second], [{}, {}]
    for o in out_sets:                       # allocate the outputs outside the timed region
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=o, ref_dtype=ref)
    ctx.sync()
    nout = n4 if narrow else out_sets[0][names[0]].nbytes
    narrowed = [{k: ctx.empty(era[k].shape, np.float32) for k in names} for _ in out_sets] if narrow else None
    hout = []
    for _ in range(4):
        p = C.c_void_p()
        ctx._check(ctx.lib.pgw_host_alloc(ctx.handle, nout, C.byref(p)))
        hout.append(p)
    in_free, out_free, q_c, q_d = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
    for i in (0, 1):
        in_free.put(i); out_free.put(i)
    errs = []

    def guard(fn):
        def run():
            try:
                fn()
            except BaseException as e:       # noqa: BLE001
                errs.append(repr(e))
                q_c.put(None); q_d.put(None)
        return run

    def uploader():
        for _ in range(files):
            s = in_free.get()
            for k, p in zip(names, hp):
                up._check(up.lib.pgw_memcpy_h2d(up.handle, in_sets[s][k].ptr, p, n4))
            up.sync()
            q_c.put(s)
        q_c.put(None)

    def computer():
        while True:
            s = q_c.get()
            if s is None:
                break
            o = out_free.get()
            s3.process_file_device(ctx, in_sets[s], coeffs, deltas, case['target_dt'], True, out=out_sets[o], ref_dtype=ref)
            if narrow:
                for k in names:
                    ctx._check(ctx.lib.pgw_narrow_f64_f32(ctx.handle, out_sets[o][k].size, out_sets[o][k].ptr, narrowed[o][k].ptr, 1))
            ctx.sync()
            in_free.put(s)
            q_d.put(o)
        q_d.put(None)

    def downloader():
        while True:
            o = q_d.get()
            if o is None:
                break
            src = narrowed[o] if narrow else out_sets[o]
            for k, p in zip(names, hout):
                dn._check(dn.lib.pgw_memcpy_d2h(dn.handle, p, src[k].ptr, nout))
            dn.sync()
            out_free.put(o)
    th = [threading.Thread(target=guard(f)) for f in (uploader, computer, downloader)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = (time.perf_counter() - t0) / files
    for p in hout:
        ctx.lib.pgw_host_free(ctx.handle, p)
    for k in names:
        second[k].free()
        if narrow:
            for nb in narrowed:
                nb[k].free()
    for o in out_sets:
        for v in o.values():
            v.free()
    if errs:
        return {'error': errs[0]}
    return dict(ms_per_file=round(el * 1e3, 2), files_per_hour=round(3600.0 / el, 1), files=files,
                GB_in=round(4 * n4 / 1e9, 3), GB_out=round(4 * nout / 1e9, 3),
                note='three HIP streams (h2d, kernels, d2h), two device buffer sets each way, one host thread per stage')


def f32_storage(ctx, case, coeffs, a, np, steps=10, warmup=2):
    """The timed region again on a float32 copy of the same file - what real ERA5 files are: reference-dtype mode
    (float64 4-D outputs, the reference's roundings; settings.f32_file_mode default) and float64 arithmetic with float32
    outputs.  Per mode: ms per file, the per-kernel HIP-event table and the `roofline` object of its dominant kernel
    (algorithmic bytes of THAT storage layout; PMC traffic / rocprofv3 average from the committed summaries of
    `bench.py --storage f32 [--f32-mode fast]`)."""
    import argparse
    from pgw4era5_amd import step_03_apply_to_era as s3
    f32 = np.float32
    era32 = {k: (v.astype(f32) if isinstance(v, np.ndarray) and v.ndim >= 3 else v) for k, v in case['era'].items()}
    d32 = {k: v.astype(f32) for k, v in case['deltas'].items()}
    deltas = s3.DeltaSet(ctx, d32, case['delta_times'], case['plev'], f32)
    era = s3._upload_era(ctx, era32, f32)
    del era32, d32
    N, S, ncol = a.nlev, len(case['plev']), a.nlat * a.nlon
    bk = case['era']['bk']
    n_pure = 0
    while n_pure < N and (0.5 * (bk[n_pure + 1] - bk[n_pure]) + bk[n_pure]) == 0.0:
        n_pure += 1
    res = {}
    for mode, ref in (('reference', True), ('fast', False)):
        outb, infos = {}, []
        ctx.profile(True)
        for i in range(warmup):
            s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=outb, ref_dtype=ref)
        ctx.sync()
        ctx.profile_reset()
        walls = []
        t0 = time.perf_counter()
        for i in range(steps):
            t1 = time.perf_counter()
            _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'] + dt_mod.timedelta(hours=i), True,
                                             out=outb, ref_dtype=ref)      # returns after the file's one host round trip
            walls.append(time.perf_counter() - t1)
            infos.append(info)
        ctx.sync()
        el_mean = (time.perf_counter() - t0) / steps
        # the median file: this leg runs late in a long process, and one host hiccup of a few ms (seen: 4 ms once in ten files,
        # never in a fresh process, tools/f32ref_gap.py) would otherwise read as 0.4 ms on every file; the mean is kept beside it
        el = sorted(walls)[len(walls) // 2]
        prof = {k: ctx.profile_get(k) for k in FILE_KERNELS}
        ctx.profile(False); ctx.profile_reset()
        passes = sum(i['n_iter'] for i in infos)
        launches_multi = prof['ps_loop_multi'][0]
        kinfo = dict(levels_per_launch=sum(i['levels_touched'] for i in infos) / max(passes, 1),
                     qv_done_levels=n_pure if (not a.full_column and ctx.get_option('quad') != 0) else 0, so=8 if ref else 4,
                     passes_per_launch=(sum(i.get('passes_launched', 0) for i in infos) / launches_multi) if launches_multi else 1.0)
        kern = kernel_table(prof, N, S, ncol, 4, kinfo)
        res[mode] = dict(ms_per_file=round(el * 1e3, 3), ms_per_file_mean=round(el_mean * 1e3, 3),
                         ms_per_file_each=[round(w * 1e3, 3) for w in walls], files_per_hour=round(3600.0 / el, 1),
                         iterations=info['n_iter'], files=steps, kernels=kern)
        cand = [k for k in kern if kern[k]['GBps']]
        if cand:
            like = argparse.Namespace(storage='f32', f32_mode=mode, nlat=a.nlat, nlon=a.nlon, nlev=a.nlev)
            res[mode]['roofline'] = roofline_object(max(cand, key=lambda k: kern[k]['total_ms']), kern, like)
        for v in outb.values():
            v.free()
    # PCIe-inclusive, pipelined, for a float32 file in reference-dtype mode - the production case: 2.3 GB in, 4.55 GB out
    # (float64 T, QV, U, V like the reference writes them), and with settings.f32_out_dtype = 'float32' (2.3 GB out)
    try:
        import ctypes as C
        names = ('T', 'QV', 'U', 'V')
        n4 = era['T'].nbytes
        hp = []
        for _ in names:
            p = C.c_void_p()
            ctx._check(ctx.lib.pgw_host_alloc(ctx.handle, n4, C.byref(p)))
            C.memset(p, 0, 64)
            hp.append(p)
        for k, p in zip(names, hp):                       # the file's own values (a re-upload must not change the result)
            ctx._check(ctx.lib.pgw_memcpy_d2h(ctx.handle, p, era[k].ptr, n4))
        ctx.sync()
        res['pcie_inclusive_reference'] = pcie_pipelined(ctx, era, coeffs, deltas, case, True, names, hp, n4)
        res['pcie_inclusive_reference_f32_out'] = pcie_pipelined(ctx, era, coeffs, deltas, case, True, names, hp, n4, narrow=True)
        for p in hp:
            ctx.lib.pgw_host_free(ctx.handle, p)
    except Exception as e:      # noqa: BLE001
        res['pcie_inclusive_reference'] = {'error': '%s: %s' % (type(e).__name__, e)}
    for v in era.values():
        v.free()
    deltas.free()
    return res


def end_to_end(a, out_dtype='float64'):
    """tools/e2e_cli.py in a child process: K float32 files through the step_03 command line INCLUDING NetCDF-3 read /
    write (settings.f32_file_mode default; out_dtype 'float64': T, QV, U, V written as float64 like the reference,
    'float32': settings.f32_out_dtype = 'float32', narrowed on the GPU)."""
    import subprocess
    import tempfile
    d = a.e2e_dir or tempfile.mkdtemp(prefix='pgw_e2e_')
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'e2e_cli.py'), '--files', str(a.e2e_files), '--nlat', str(a.nlat),
           '--nlon', str(a.nlon), '--nlev', str(a.nlev), '--dir', d, '--out-dtype', out_dtype]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith('{'):
            return json.loads(ln)
    raise RuntimeError('e2e_cli.py failed (%d): %s' % (r.returncode, r.stderr[-400:]))


def microbench(ctx, era, coeffs, a, np, reps=5):
    """Signature-faithful kernels of SURVEY.md section 8(a) that the fused file path no longer
    launches, timed on the same device arrays outside the timed region: `integ_geopot`
    (functions.py:128-189, full column - the kernel BASELINE.json's metric names) and the
    hybrid-pressure kernel ("integ_pressure", step_03:64-88)."""
    from pgw4era5_amd.device import dtype_tag
    dt = era['T'].dtype
    s = dt.itemsize
    nt, N, nlat, nlon = era['T'].shape
    ncol = nlat * nlon
    tag = dtype_tag(dt)
    ctx.set_levels(coeffs['ak'], coeffs['bk'])
    # pa_hl and pa - the two write streams of the pressure kernel - from the context's placement pool when it has stock of that
    # size (settings.placement): one stretch of the card's memory takes 5.45 TB/s of writes, two take 6.75
    pa_hl = ctx.level_array((nt, N + 1, nlat, nlon), dt, cls=0)
    pa = ctx.level_array((nt, N, nlat, nlon), dt, cls=1)
    phi = ctx.empty((nt, nlat, nlon), dt)
    out = {}
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(reps + 1):
        ctx._check(ctx.lib.pgw_pressure_levels(ctx.handle, tag, nt, ncol, era['PS'].ptr, pa_hl.ptr, pa.ptr))
        ctx._check(ctx.lib.pgw_integ_geopot(ctx.handle, tag, nt, N, ncol, pa_hl.ptr, era['FIS'].ptr, era['T'].ptr,
                                            era['QV'].ptr, 30000.0, None, phi.ptr, 1))
    # interp_logp_4d (functions.py:434-580), signature-faithful: var, source_P (S levels) and targ_P in, N levels out
    S = 19
    lev = np.linspace(1, N, S).round().astype(int)
    src_p = ctx.empty((nt, S, nlat, nlon), dt)
    src_v = ctx.empty((nt, S, nlat, nlon), dt)
    plane = nlat * nlon * s
    for k, l in enumerate(lev):          # S half levels of every column as an ascending source axis
        ctx._check(ctx.lib.pgw_memcpy_d2d(ctx.handle, src_p.ptr + k * plane, pa_hl.ptr + int(l) * plane, plane))
        ctx._check(ctx.lib.pgw_memcpy_d2d(ctx.handle, src_v.ptr + k * plane, era['T'].ptr + int(l - 1) * plane, plane))
    interp_out = ctx.empty((nt, N, nlat, nlon), dt)
    for _ in range(reps + 1):
        ctx._check(ctx.lib.pgw_interp_logp_4d(ctx.handle, tag, nt, S, N, ncol, src_v.ptr, src_p.ptr, pa.ptr, 2, 0,
                                              interp_out.ptr))
    for x in (src_p, src_v, interp_out):
        x.free()
    for name, nbytes in (('integ_geopot', (3 * N + 3) * ncol * s), ('pressure', (2 * N + 2) * ncol * s),
                         ('interp_logp', (2 * S + 2 * N) * ncol * s)):
        cnt, ms = ctx.profile_get(name)
        avg = ms / cnt
        out[name] = dict(launches=cnt, avg_ms=round(avg, 4), algo_GB=round(nbytes / 1e9, 4),
                         GBps=round(nbytes / 1e9 / (avg / 1e3), 1), frac_of_peak=round(nbytes / 1e9 / (avg / 1e3) / HBM_PEAK_GBS, 4))
    ctx.profile(False)
    ctx.profile_reset()
    for x in (pa_hl, pa, phi):
        x.free()
    return out


PMC_KERNEL = {'integ_geopot': 'k_integ_geopot', 'adjust_ps_step': 'k_adjust_ps_step',
              'vert_interp_delta': 'k_vert_interp_delta', 'q_to_rh': 'k_humidity_hybrid', 'rh_to_q': 'k_humidity_hybrid',
              'finalize': 'k_finalize_ps_hus', 'pressure': 'k_pressure_levels',
              'thermo_delta': 'k_delta_pair<double, 2, true>', 'wind_delta': 'k_delta_pair<double, 2, false>',
              'phi_ref_hybrid': 'k_phi_ref_hybrid', 'quad_delta': 'k_delta_quad', 'ps_loop_multi': 'k_ps_loop_multi',
              'interp_logp': 'k_interp_logp_stream', 'reinterp_pair': 'k_reinterp_pair'}


def _profile_tag(a):
    return a.storage if a.storage == 'f64' else ('f32' if a.f32_mode == 'fast' else 'f32ref')


def _instantiation_ok(name, a):
    """The PMC passes also see the float32 instantiations of the `f32_storage` side measurement: keep the kernels of
    this run's storage types - <T, TL, ...> = <double, double> / <float, double> (reference mode) / <float, float>."""
    if '<' not in name:
        return True
    args = [x.strip() for x in name[name.index('<') + 1:].split(',')]
    t = args[0]
    tl = args[1] if len(args) > 1 and args[1] in ('double', 'float') else None
    want_t = 'double' if a.storage == 'f64' else 'float'
    want_tl = 'double' if (a.storage == 'f64' or a.f32_mode == 'reference') else 'float'
    return t == want_t and (tl is None or tl == want_tl)


def rocprof_frac(kernel, a, algo_GB):
    """The roofline fraction recomputed from the committed rocprofv3 --kernel-trace summary of this same command
    (profiles/kernel_stats_<tag>_<storage>.csv, written by profiles/summarize.py from the per-dispatch trace with the
    warm-up launches dropped): algorithmic bytes / average duration of the TIMED launches / 8 TB/s."""
    import csv
    import glob
    if (a.nlat, a.nlon, a.nlev) != (721, 1440, 137):
        return {}
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'kernel_stats_*_%s.csv' % _profile_tag(a))))
    if not files:
        return {}
    pat = PMC_KERNEL.get(kernel, '?')

    def row_of(path):
        for r in csv.DictReader(open(path)):
            if r['kernel'].startswith(pat) and _instantiation_ok(r['kernel'], a) and r.get('timed_avg_us'):
                return r
        return None
    # the latest committed summary is THE figure; the round's other summaries are listed beside it (tag -> fraction): r02k, r02m
    # and r02n are the same float64 quad kernel on three boxes of the pool (it differs by 10 % between boxes), earlier tags are
    # earlier builds of the round
    every = {}
    for f in files:
        r = row_of(f)
        tag = os.path.basename(f)[len('kernel_stats_'):].split('_')[0]
        if r is not None and tag[:3] in ('r02', 'r03'):
            every[tag] = round(algo_GB / (float(r['timed_avg_us']) / 1e6) / HBM_PEAK_GBS, 4)
    r = row_of(files[-1])
    if r is None:
        return {}
    avg = float(r['timed_avg_us'])
    return {'frac_rocprof': round(algo_GB / (avg / 1e6) / HBM_PEAK_GBS, 4), 'rocprof_timed_avg_launch_ms': round(avg / 1e3, 4),
            'rocprof_timed_launches': int(r['timed_calls']), 'rocprof_source': os.path.relpath(files[-1], ROOT),
            'frac_rocprof_by_committed_profile': every}


def pmc_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_summary_*.json, produced by profiles/summarize.py from separate FETCH_SIZE /
    WRITE_SIZE runs of this same command).  PMC counters cannot be read from inside the process,
    so the number is only reported for the configuration the passes were taken on (default
    shape, matching storage); otherwise null."""
    import glob
    if (a.nlat, a.nlon, a.nlev) != (721, 1440, 137):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_summary_*_%s.json' % _profile_tag(a))))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    pat = PMC_KERNEL.get(kernel, '?')
    if 'k_delta_pair' in pat:           # k_delta_pair<T, V, THERMO, U, TPB, STAGED>: the third argument tells the pair
        want = 'true' if kernel == 'thermo_delta' else 'false'
        match = lambda k: k.startswith('k_delta_pair<') and k[len('k_delta_pair<'):].split(', ')[2] == want
    else:
        match = lambda k: k.startswith(pat)
    vals = [v['hbm_bytes_per_launch'] for k, v in d.items() if match(k) and _instantiation_ok(k, a) and 'hbm_bytes_per_launch' in v]
    if not vals:
        return None, None
    return round(sum(vals) / len(vals) / 1e9, 4), os.path.relpath(files[-1], ROOT)


FP64_VALU_PEAK_LANE_OPS = 78.6e12 / 2          # MI355X: 78.6 TFLOP/s fp64 vector = 39.3e12 FMA lanes / s


def pmc_valu(kernel, a, avg_ms):
    """For kernels that are bound by fp64 vector issue rather than HBM (the ta+hur pair kernel: one log,
    2-4 exp and 7 IEEE divisions per level and column): VALU wave-instructions per launch from the committed
    SQ PMC pass x 64 lanes / launch time, against the fp64 FMA issue peak."""
    import glob
    if (a.nlat, a.nlon, a.nlev) != (721, 1440, 137):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_summary_*_%s.json' % _profile_tag(a))))
    if not files:
        return None
    d = json.load(open(files[-1]))
    want = 'true' if kernel == 'thermo_delta' else 'false'
    pat = PMC_KERNEL.get(kernel, '?')
    for k, v in d.items():
        ok = (k.startswith('k_delta_pair<') and k[len('k_delta_pair<'):].split(', ')[2] == want) if 'k_delta_pair' in pat \
            else k.startswith(pat)
        if ok and _instantiation_ok(k, a) and 'SQ_INSTS_VALU' in v:
            rate = v['SQ_INSTS_VALU'] * 64 / (avg_ms / 1e3)
            return {'valu_wave_insts_per_launch': round(v['SQ_INSTS_VALU']), 'lane_ops_per_s': round(rate / 1e12, 2),
                    'unit': 'T lane-ops/s', 'peak': FP64_VALU_PEAK_LANE_OPS / 1e12,
                    'frac_of_fp64_issue_peak': round(rate / FP64_VALU_PEAK_LANE_OPS, 3),
                    'valu_active_share_of_wave_time': round(v.get('valu_active_share_of_wave_time', 0), 3),
                    'simd_valu_busy': round(v['simd_valu_busy'], 3) if 'simd_valu_busy' in v else None,
                    'mean_waves_per_simd': round(v['mean_waves_per_simd'], 2) if 'mean_waves_per_simd' in v else None,
                    'source': os.path.relpath(files[-1], ROOT)}
    return None


_CPU_CASE = None          # the synthetic case, inherited by the forked workers of the file-parallel CPU leg


def _band(case, np, j0, rows):
    sl = slice(j0, j0 + rows)
    f64 = np.float64
    era = {}
    for k, v in case['era'].items():
        era[k] = np.ascontiguousarray(v[..., sl, :], dtype=f64) if (isinstance(v, np.ndarray) and v.ndim >= 3) else v
    deltas = {k: np.ascontiguousarray(v[..., sl, :], dtype=f64) for k, v in case['deltas'].items()}
    return era, deltas


def _cpu_worker(args):
    """One worker of the file-parallel leg: the oracle (serial C column loops) on its own latitude band (reference: one
    file per pool worker, parallel.py:20-27); returns (seconds, passes)."""
    import numpy as np
    from oracle import pgw_oracle as O, pgw_oracle_c as C
    j0, rows = args
    case = _CPU_CASE
    era, deltas = _band(case, np, j0, rows)
    t0 = time.perf_counter()
    out = O.pgw_for_era5_arrays(era, deltas, case['delta_times'], case['plev'], case['target_dt'], ignore_top_pressure_error=True,
                                vert_interp=C.vert_interp_delta)
    return time.perf_counter() - t0, out['n_iter']


def cpu_baseline(case, a, np):
    """The CPU oracle (a port: the reference's xarray/numba stack is not installable here) on latitude bands of the
    same synthetic file, scaled to files/hour.  One process = the reference's default `-p 1`
    (step_03_apply_to_era.py:542) in two forms: (a) `c_column_loops` - the reference's serial per-column loops
    (numba interp_1d_for_timelatlon, the np.vectorize'd replace_delta_sfc) as plain C -O2, the rest level-wise numpy like
    the reference; (b) `numpy_vectorised` - the same with those loops vectorised over columns.
    `value` = the faster of the two.  Then `procs` processes at once, each on its own band = the reference's file-parallel
    `-p N` (parallel.py:20-27), `parallel` (form a).  Both forms are kinder to the CPU than the reference's own stack
    (its np.vectorize is one Python call per column), which cannot run here."""
    global _CPU_CASE
    import multiprocessing as mp
    from oracle import pgw_oracle as O, pgw_oracle_c as C
    C.lib()
    rows = min(a.cpu_rows, a.nlat)
    j0 = max((a.nlat - rows) // 2, 0)
    era, deltas = _band(case, np, j0, rows)
    frac = rows / a.nlat
    legs = {}
    for name, vi in (('c_column_loops', C.vert_interp_delta), ('numpy_vectorised', None)):
        t0 = time.perf_counter()
        out = O.pgw_for_era5_arrays(era, deltas, case['delta_times'], case['plev'], case['target_dt'],
                                    ignore_top_pressure_error=True, vert_interp=vi)
        t = time.perf_counter() - t0
        legs[name] = {'files_per_hour': round(3600.0 / (t / frac), 3), 'seconds': round(t, 1), 'iterations': out['n_iter']}
        del out
    best = max(legs, key=lambda k: legs[k]['files_per_hour'])
    res = {'value': legs[best]['files_per_hour'], 'unit': 'files/hour', 'cores': 1, 'kind': 'port',
           'sample': '%d of %d latitude rows (%d columns) of the same file through oracle/pgw_oracle.py (numpy fp64) with the '
                     'per-column loops %s, %.1f s, %d iterations; host has %d cores'
                     % (rows, a.nlat, rows * a.nlon, 'in serial C (oracle/pgw_oracle_c.c)' if best == 'c_column_loops'
                        else 'vectorised over columns', legs[best]['seconds'], legs[best]['iterations'], os.cpu_count()),
           'one_process': legs}
    del era, deltas
    procs = a.cpu_procs or min(os.cpu_count() or 1, 16)
    if procs > 1:
        prow = max(min(a.nlat // procs, 40), 1)            # ~4 s of work per process; 16 bands of 40 rows = 640 of 721 rows
        _CPU_CASE = case
        bands = [(i * prow, prow) for i in range(procs)]
        t0 = time.perf_counter()
        with mp.get_context('fork').Pool(procs) as pool:   # fork: the workers share the case copy-on-write (no GPU state yet)
            times = pool.map(_cpu_worker, bands, chunksize=1)
        wall = time.perf_counter() - t0
        _CPU_CASE = None
        files_done = procs * prow / a.nlat
        res['parallel'] = {'value': round(files_done / wall * 3600.0, 3), 'unit': 'files/hour', 'cores': procs,
                           'sample': '%d processes at once, %d latitude rows each (%.2f of a file in all), %.1f s wall '
                                     '(slowest worker %.1f s); host has %d cores'
                                     % (procs, prow, files_done, wall, max(x[0] for x in times), os.cpu_count())}
    return res


if __name__ == '__main__':
    sys.exit(main())
