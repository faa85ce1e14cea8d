#!/usr/bin/env python
"""Condense rocprofv3 output directories (under gpurun_out/) into the small tracked summaries
in profiles/.

    python profiles/summarize.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir>]]
    python profiles/summarize.py trace <round-tag> <trace_dir> <warmup W> <steps K>

`trace` (round 2 on): kernel_stats_<tag>.csv from the per-dispatch kernel trace of
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py --steps K --warmup W
        --no-cpu-baseline --overlap-streams 0 --no-extras
with the warm-up launches dropped: per kernel the dispatches are ordered by start time; a kernel launched c times per
step (calls = (W + K) * c) loses its first W * c dispatches, any other kernel (the signature micro-benchmarks: one
warm-up launch + reps) its first one.  Columns: all launches (calls, avg_us, min_us, max_us) as rocprofv3 --stats gives
them, and timed_calls, timed_avg_us, timed_median_us, timed_min_us, timed_max_us of the timed launches - bench.py's
`roofline.frac_rocprof` divides the algorithmic bytes by timed_avg_us.

stats_dir: output of  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py ...
pmc dirs : output of  rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
                      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
(separate passes, as MI355X_MICROARCH.md "rocprofv3 PMC slots" requires).

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B: both counters are in KiB and on
gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read
(MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for streaming stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.split('(')[0]
    return name.replace('pgw::', '')


def trace_stats():
    import statistics
    tag, d, W, K = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    here = os.path.dirname(os.path.abspath(__file__))
    f = glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[short(r['Kernel_Name'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    with open(os.path.join(here, 'kernel_stats_%s.csv' % tag), 'w') as o:
        o.write('kernel,calls,avg_us,min_us,max_us,total_ms,timed_calls,timed_avg_us,timed_median_us,timed_min_us,timed_max_us\n')
        rows = []
        for k, v in per.items():
            v.sort()
            dur = [(e - b) / 1e3 for b, e in v]
            n = len(dur)
            drop = n * W // (W + K) if (n % (W + K) == 0 and n >= W + K) else (1 if n > 1 else 0)
            t = dur[drop:]
            rows.append((sum(dur), '"%s",%d,%.1f,%.1f,%.1f,%.3f,%d,%.1f,%.1f,%.1f,%.1f\n' % (
                k, n, sum(dur) / n, min(dur), max(dur), sum(dur) / 1e3, len(t), sum(t) / len(t), statistics.median(t), min(t), max(t))))
        for _, line in sorted(rows, reverse=True):
            o.write(line)
            print(line, end='')


def clean_rows(path):
    """Rows of a rocprofv3 counter table with the dispatches of bench.py's side measurements dropped for the kernels of the
    timed region: the PCIe-pipelined leg of `extras` runs copies (blit kernels) on two more streams, and the counters of a
    file-path kernel that overlaps them include their instructions and bytes (r03a: 889 M instead of 593 M VALU
    wave-instructions, 17.1 instead of 4.5 GB written per k_delta_quad launch).  `extras` starts with the step_02
    regridding: a kernel that has dispatches before the first k_zonal_mean_rows / k_regrid dispatch is summarised from
    those alone; kernels that only the side measurements launch keep all their dispatches."""
    rows = list(csv.DictReader(open(path)))
    cut = None
    for r in rows:
        if short(r['Kernel_Name']).startswith(('k_zonal_mean_rows', 'k_regrid')):
            d = int(r['Dispatch_Id'])
            cut = d if cut is None else min(cut, d)
    if cut is None:
        return rows
    early = {short(r['Kernel_Name']) for r in rows if int(r['Dispatch_Id']) < cut}
    return [r for r in rows if int(r['Dispatch_Id']) < cut or short(r['Kernel_Name']) not in early]


def main():
    if sys.argv[1] == 'trace':
        return trace_stats()
    tag, stats_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    f = glob.glob(os.path.join(stats_dir, '**', '*_kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(here, 'kernel_stats_%s.csv' % tag), 'w') as o:
        o.write('kernel,calls,avg_us,min_us,max_us,total_ms,percent\n')
        for r in rows:
            o.write('%s,%s,%.1f,%.1f,%.1f,%.3f,%s\n' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3,
                                                      float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3,
                                                      float(r['TotalDurationNs']) / 1e6, r['Percentage']))
            out.setdefault(short(r['Name']), {})['avg_us'] = float(r['AverageNs']) / 1e3
    if len(sys.argv) >= 5:
        for kind, d in (('FETCH_SIZE', sys.argv[3]), ('WRITE_SIZE', sys.argv[4])):
            f = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)[0]
            agg = collections.defaultdict(list)
            for r in clean_rows(f):
                if r['Counter_Name'] == kind:
                    agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
            for k, v in agg.items():
                out.setdefault(k, {})[kind + '_KiB'] = sum(v) / len(v)
        for k, v in out.items():
            if 'FETCH_SIZE_KiB' in v and 'WRITE_SIZE_KiB' in v:
                v['hbm_bytes_per_launch'] = (2 * v['FETCH_SIZE_KiB'] + v['WRITE_SIZE_KiB']) * 1024
    if len(sys.argv) >= 6:
        # SQ pass (rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY ...):
        # wave-instructions per launch; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles (x4 = cycles)
        f = glob.glob(os.path.join(sys.argv[5], '**', '*_counter_collection.csv'), recursive=True)[0]
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in clean_rows(f):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            m = {c: sum(x) / len(x) for c, x in v.items()}
            d = out.setdefault(k, {})
            for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
                if c in m:
                    d[c] = m[c]
            if m.get('GRBM_GUI_ACTIVE') and 'SQ_WAVE_CYCLES' in m:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = the launch's duration in shader-clock cycles; 1024 SIMDs
                simd_cycles = m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
                d['shader_cycles'] = m['GRBM_GUI_ACTIVE'] / 8.0
                d['mean_waves_per_simd'] = 4.0 * m['SQ_WAVE_CYCLES'] / simd_cycles
                if 'SQ_ACTIVE_INST_VALU' in m:
                    d['simd_valu_busy'] = 4.0 * m['SQ_ACTIVE_INST_VALU'] / simd_cycles
                    if m.get('SQ_INSTS_VALU'):
                        d['cycles_per_valu_inst'] = 4.0 * m['SQ_ACTIVE_INST_VALU'] / m['SQ_INSTS_VALU']
            if 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES'] > 0:
                for c, name in (('SQ_ACTIVE_INST_VALU', 'valu_active_share_of_wave_time'),
                                ('SQ_WAIT_INST_ANY', 'issue_stall_share_of_wave_time'),
                                ('SQ_WAIT_ANY', 'waitcnt_share_of_wave_time')):
                    if c in m:
                        d[name] = m[c] / m['SQ_WAVE_CYCLES']
    with open(os.path.join(here, 'pmc_summary_%s.json' % tag), 'w') as o:
        json.dump(out, o, indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print(k, v)


if __name__ == '__main__':
    main()
