#!/usr/bin/env python
"""The figures of a bench.py line one looks at first.  Usage: python tools/print_bench.py gpurun_out/bench.json"""
import json
import sys

j = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = j['roofline']
print('files/hour %s  ms/file %s  n_gpus %s  %s' % (j['value'], j['ms_per_step'], j['n_gpus'], j['config'].get('storage')))
print('dominant kernel %s: %.4f ms  frac %.4f (rocprof %s)  bare pattern on the same arrays %s  -> frac of it %s'
      % (r['kernel'], r['avg_launch_ms'], r['frac'], r.get('frac_rocprof'), (r.get('bare_pattern_same_arrays') or {}).get('GBps'),
         r.get('frac_of_bare_pattern')))
for k, v in (j.get('kernels') or {}).items():
    print('  %-16s %3d x %.4f ms  %s GB  %s GB/s' % (k, v['launches'], v['avg_ms'], v['algo_GB'], v['GBps']))
d = j.get('device_state') or {}
print('device_state: cap %s W  timed region %s' % (d.get('power_cap_w'), d.get('timed_region')))
print('              whole run %s' % (d.get('whole_run'),))
f = j.get('roofline_f32ref')
if f:
    print('float32 reference mode: %.4f ms quad, frac %.4f, %.3f ms per file' % (f['avg_launch_ms'], f['frac'], f['ms_per_file']))
ex = j.get('extras') or {}
for k in ('local_p_ref', 'i_reinterp'):
    if k in ex and isinstance(ex[k], dict):
        print('%s: %s ms per file' % (k, ex[k].get('ms_per_file')))
c = j.get('cpu_baseline')
if c:
    print('cpu_baseline: %s %s on %s cores (%s)' % (c.get('value'), c.get('unit'), c.get('cores'), c.get('kind')))
