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

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant
kernel, HIP-event timed inside this run) and `cpu_baseline` (the numpy oracle, rank 0, N = 1
only, on a bounded latitude band of the same file).
"""
import argparse
import datetime as dt
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--storage', choices=['f64', 'f32'], default='f64')
    p.add_argument('--nlat', type=int, default=721)
    p.add_argument('--nlon', type=int, default=1440)
    p.add_argument('--nlev', type=int, default=137)
    p.add_argument('--cpu-rows', type=int, default=160, help='latitude rows of the CPU-baseline sample')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--full-column', action='store_true',
                   help='pass kernel reads every level (input-independent traffic) instead of stopping above p_ref')
    return p.parse_args()


def kernel_bytes(name, N, S, ncol, s, info):
    """Algorithmic bytes one launch of kernel `name` moves."""
    if name == 'integ_geopot':        # pa_hl, ta, hus, zgs in; phi_ref out          (3N+3) ncol
        return (3 * N + 3) * ncol * s
    if name == 'adjust_ps_step':      # ta, e per level read; PS,FIS (storage) + 6 fp64 state words
        lv = info.get('levels_per_launch', N * ncol)
        return 2 * lv * s + ncol * (2 * s + 6 * 8)
    if name == 'vert_interp_delta':   # 2 records x S, add_to N in, N out, ps + surface pairs
        return (2 * S + 2 * N) * ncol * s + 5 * ncol * s
    if name == 'q_to_rh':             # QV, T in, RH out, PS
        return (3 * N + 1) * ncol * s
    if name == 'rh_to_q':             # hur, ta in, e out (vapour pressure pre-pass)
        return 3 * N * ncol * s
    if name == 'finalize':            # e in, QV out, PS in/out, delta_ps
        return 2 * N * ncol * s + ncol * (2 * s + 8)
    if name == 'pressure':            # ps in, pa_hl out
        return (N + 2) * ncol * s
    return 0


def main():
    a = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        # torch first: libpgw_hip.so then binds to the HIP runtime torch loaded (pgw4era5_amd/_lib.py)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    import numpy as np
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
    from pgw4era5_amd.device import Context

    if a.full_column:
        os.environ['PGW_FULL_COLUMN'] = '1'
    dtype = np.float64 if a.storage == 'f64' else np.float32
    s = np.dtype(dtype).itemsize
    ctx = Context(local)
    t0 = time.time()
    case = synthetic.make_case(nlat=a.nlat, nlon=a.nlon, nlev=a.nlev, seed=1 + rank, dtype=dtype)
    t_gen = time.time() - t0
    ncol = a.nlat * a.nlon
    N, S = a.nlev, len(case['plev'])
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
            import torch
            torch.cuda.synchronize()

    infos = []
    for i in range(a.warmup):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, stamps[i], True, out=out)
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, stamps[a.warmup + i], True, out=out)
        infos.append(info)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {k: ctx.profile_get(k) for k in ('integ_geopot', 'adjust_ps_step', 'vert_interp_delta', 'q_to_rh',
                                            'rh_to_q', 'finalize', 'pressure', 'time_lerp', 'surface')}
    ctx.profile(False)
    if dist is not None:
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        n_iter = [i['n_iter'] for i in infos]
        passes = sum(n_iter)
        lv_per_launch = sum(i['levels_touched'] for i in infos) / max(passes, 1)
        kinfo = dict(levels_per_launch=lv_per_launch)
        kern = {}
        for k, (cnt, ms) in prof.items():
            if cnt == 0:
                continue
            avg_ms = ms / cnt
            b = kernel_bytes(k, N, S, ncol, s, kinfo)
            kern[k] = dict(launches=cnt, avg_ms=round(avg_ms, 4), total_ms=round(ms, 3),
                           algo_GB=round(b / 1e9, 4), GBps=round(b / 1e9 / (avg_ms / 1e3), 1) if b else None)
        dom = max((k for k in kern if kern[k]['GBps']), key=lambda k: kern[k]['total_ms'])
        traffic, tsrc = pmc_traffic(dom, a)
        roof = dict(bound='hbm', kernel=dom, achieved=kern[dom]['GBps'], peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(kern[dom]['GBps'] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_unit='GB/launch',
                    traffic_source=tsrc, avg_launch_ms=kern[dom]['avg_ms'],
                    algorithmic_GB_per_launch=kern[dom]['algo_GB'])
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
                       'storage': a.storage, 'files_per_rank': a.steps, 'sharding': 'one file per rank per step',
                       'iterations_per_file': n_iter[0] if len(set(n_iter)) == 1 else n_iter,
                       'pass_kernel': 'full_column' if a.full_column else 'stops_above_p_ref',
                       'mean_levels_read_per_column_per_pass': round(lv_per_launch / ncol, 2)},
            'roofline': roof,
            'kernels': kern,
            'device': ctx.device_name(),
            'setup_s': round(t_gen, 1),
        }
        if not a.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(case, a, np)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


PMC_KERNEL = {'integ_geopot': 'k_integ_geopot', 'adjust_ps_step': 'k_adjust_ps_step',
              'vert_interp_delta': 'k_vert_interp_delta', 'q_to_rh': 'k_humidity_hybrid', 'rh_to_q': 'k_humidity_hybrid',
              'finalize': 'k_finalize_ps_hus', 'pressure': 'k_pressure_levels', 'thermo_delta': 'k_thermo_delta',
              'wind_delta': 'k_wind_delta'}


def pmc_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_summary_*.json, produced by profiles/summarize.py from separate FETCH_SIZE /
    WRITE_SIZE runs of this same command).  PMC counters cannot be read from inside the process,
    so the number is only reported for the configuration the passes were taken on (default
    shape, matching storage); otherwise null."""
    import glob
    if (a.nlat, a.nlon, a.nlev) != (721, 1440, 137):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_summary_*_%s.json' % a.storage)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    vals = [v['hbm_bytes_per_launch'] for k, v in d.items()
            if k.startswith(PMC_KERNEL.get(kernel, '?')) and 'hbm_bytes_per_launch' in v]
    if not vals:
        return None, None
    return round(sum(vals) / len(vals) / 1e9, 4), os.path.relpath(files[-1], ROOT)


def cpu_baseline(case, a, np):
    """The numpy oracle (a port: the reference's xarray/numba stack is not installable here) on
    a latitude band of the same synthetic file, one host thread, scaled to files/hour."""
    from oracle import pgw_oracle as O
    rows = min(a.cpu_rows, a.nlat)
    j0 = max((a.nlat - rows) // 2, 0)
    sl = slice(j0, j0 + rows)
    f64 = np.float64
    era = {}
    for k, v in case['era'].items():
        if isinstance(v, np.ndarray) and v.ndim >= 3:
            era[k] = np.ascontiguousarray(v[..., sl, :], dtype=f64)
        else:
            era[k] = v
    deltas = {k: np.ascontiguousarray(v[..., sl, :], dtype=f64) for k, v in case['deltas'].items()}
    t0 = time.perf_counter()
    out = O.pgw_for_era5_arrays(era, deltas, case['delta_times'], case['plev'], case['target_dt'],
                                ignore_top_pressure_error=True)
    t = time.perf_counter() - t0
    frac = rows / a.nlat
    return {'value': round(3600.0 / (t / frac), 3), 'unit': 'files/hour', 'cores': 1, 'kind': 'port',
            'sample': '%d of %d latitude rows (%d columns) of the same file through oracle/pgw_oracle.py '
                      '(numpy fp64), %.1f s, %d iterations; host has %d cores'
                      % (rows, a.nlat, rows * a.nlon, t, out['n_iter'], os.cpu_count())}


if __name__ == '__main__':
    main()
