"""pgw4era5_amd - MI355X-native compute path for PGW4ERA5's step_03 / step_02 hot path.

Python host code that mirrors the reference's `functions.py` / `step_03_apply_to_era.py`
call surface and drives hand-written HIP kernels (gfx950) through the C-ABI declared in
`include/pgw_hip.h`.  Importing the package does not load the HIP library; the first
compute call does, and fails loudly if it is missing (there is no CPU fallback).
"""
__version__ = '0.1.0'
