#!/usr/bin/env python
"""A/B timing helper: bench.py's timed region with the loop cut to ONE pass (threshold infinite), for library builds
whose arithmetic is deliberately perturbed in an experiment (PGW_LIB) - only the per-kernel times are meaningful."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pgw4era5_amd import settings as S
S.thresh_phi_ref_max_error = 1e300
import bench
sys.exit(bench.main(['--no-cpu-baseline', '--no-extras', '--overlap-streams', '0', '--steps', '10', '--warmup', '2'] + sys.argv[1:]))
