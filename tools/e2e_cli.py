#!/usr/bin/env python
"""End-to-end timing of the step_03 command line INCLUDING file I/O (not the bench.py metric):
writes K synthetic 0.25 deg L137 ERA5 files + the delta directory as NetCDF-3, runs
`python -m pgw4era5_amd.step_03_apply_to_era` over them in this process and reports seconds per
file for (a) the pipelined launcher and (b) the three stages run serially.

    python tools/e2e_cli.py [--files 3] [--nlat 721 --nlon 1440 --nlev 137] [--dir /tmp/pgw_e2e]
"""
import argparse
import datetime as dt
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--files', type=int, default=3)
    p.add_argument('--nlat', type=int, default=721)
    p.add_argument('--nlon', type=int, default=1440)
    p.add_argument('--nlev', type=int, default=137)
    p.add_argument('--dir', default='/tmp/pgw_e2e')
    p.add_argument('--out-dtype', choices=['float64', 'float32'], default='float64',
                   help="settings.f32_out_dtype: T, QV, U, V of the float32 files written as float64 (the reference) or narrowed to float32")
    p.add_argument('--ranks', type=int, default=1, help='worker processes (-p); > 1: only the pipelined total is timed')
    a = p.parse_args()
    import numpy as np
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3, settings as S
    S.i_debug = 0
    S.f32_out_dtype = a.out_dtype
    shutil.rmtree(a.dir, ignore_errors=True)
    t0 = time.time()
    case = synthetic.make_case(a.nlat, a.nlon, a.nlev, seed=1, dtype=np.float32)
    first = dt.datetime(2006, 8, 2, 0)
    for i in range(a.files):
        case['target_dt'] = first + dt.timedelta(hours=i)
        synthetic.write_case_files(case, os.path.join(a.dir, 'era'), os.path.join(a.dir, 'deltas'))
    t_write = time.time() - t0
    size = os.path.getsize(os.path.join(a.dir, 'era', S.era5_file_name_base.format(first)))
    last = first + dt.timedelta(hours=a.files - 1)
    argv = ['-i', os.path.join(a.dir, 'era'), '-o', os.path.join(a.dir, 'out'), '-d', os.path.join(a.dir, 'deltas'),
            '-f', first.strftime('%Y%m%d%H'), '-l', last.strftime('%Y%m%d%H'), '-H', '1', '-p', str(a.ranks), '-t']
    if a.ranks > 1:                       # workers are separate processes: time the whole command only
        t0 = time.time()
        n_iter = s3._cli(argv)
        t_pipe = time.time() - t0
        print(json.dumps(dict(files=a.files, ranks=a.ranks, file_GB=round(size / 1e9, 3), n_iter=n_iter,
                              total_s=round(t_pipe, 2), s_per_file=round(t_pipe / a.files, 3),
                              note='includes process start, delta upload and pinning in every rank')))
        shutil.rmtree(a.dir, ignore_errors=True)
        return
    t0 = time.time()
    s3.load_delta_set(s3.default_context(), os.path.join(a.dir, 'deltas'), np.float32)      # once per run
    t_deltas = time.time() - t0
    done = []
    stages = s3.pgw_for_era5.stages
    store = stages[-1]

    def store_logged(item):
        out = store(item)
        done.append(time.time())
        return out
    s3.pgw_for_era5.stages = stages[:-1] + (store_logged,)
    t0 = time.time()
    n_iter = s3._cli(argv)
    t_pipe = time.time() - t0
    s3.pgw_for_era5.stages = stages
    done.sort()
    half = len(done) // 2                 # second half of the run: pinned buffers allocated, pipeline full
    steady = (done[-1] - done[half - 1]) / (len(done) - half) if half >= 1 and len(done) > half else None
    # serial stages, timed individually on the first file
    kw = dict(inp_era_file_path=os.path.join(a.dir, 'era', S.era5_file_name_base.format(first)),
              out_era_file_path=os.path.join(a.dir, 'out', 'serial.nc'), delta_input_dir=os.path.join(a.dir, 'deltas'),
              era_step_dt=first, ignore_top_pressure_error=True)
    t0 = time.time(); item = s3._stage_load(**kw); t_load = time.time() - t0
    t0 = time.time(); item = s3._stage_download(s3._stage_compute(s3._stage_upload(item))); t_comp = time.time() - t0
    t0 = time.time(); s3._stage_store(item); t_store = time.time() - t0
    print(json.dumps(dict(files=a.files, file_GB=round(size / 1e9, 3), n_iter=n_iter,
                          setup_write_s=round(t_write, 1), delta_load_s=round(t_deltas, 2),
                          pipelined_s_per_file=round(t_pipe / a.files, 3),
                          steady_state_s_per_file=None if steady is None else round(steady, 3),
                          io_raw=os.environ.get('PGW_IO_RAW', '1') != '0',
                          serial_stage_s=dict(read=round(t_load, 3), upload_compute_download=round(t_comp, 3), write=round(t_store, 3)),
                          files_per_hour_one_rank=round(3600.0 / (steady or t_pipe / a.files), 1))))
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == '__main__':
    main()
