"""Run-time settings with the names the reference's settings.py exports (settings.py:15-150).

Same module-global style as the reference (scripts import the names), same defaults, so a
user's edited settings.py carries over value by value.
"""
i_debug = 2                                            # settings.py:15

file_name_bases = {'SCEN-HIST': '{}_delta.nc', 'HIST': '{}_historical.nc'}   # :20-23
era5_file_name_base = 'cas{:%Y%m%d%H}0000.nc'          # :26

# dimension names, ERA5 file (:30-35), GCM delta files (:38-42), GCM ocean grid (:45-47)
TIME_ERA, LON_ERA, LAT_ERA, LEV_ERA, HLEV_ERA, SOIL_HLEV_ERA = 'time', 'lon', 'lat', 'level', 'level1', 'soil1'
TIME_GCM, LON_GCM, LAT_GCM, PLEV_GCM, LEV_GCM = 'time', 'lon', 'lat', 'plev', 'lev'
TIME_GCM_OCEAN, LON_GCM_OCEAN, LAT_GCM_OCEAN = 'time', 'longitude', 'latitude'

# CMOR name -> variable name in the ERA5 files (:57-104)
var_name_map = dict(
    ta='T', ua='U', va='V', hur='RELHUM', zg='PHI',
    tas=None, hurs=None, tos=None,
    ps='PS', hus='QV', zgs='FIS', ts='T_SKIN', st='T_SO', sftlf='FR_LAND', sic='FR_SEA_ICE',
)

# step_02 regridding (:120-129).  The xESMF branch is not part of this build.
i_use_xesmf_regridding = 0
nan_interp_kernel_radius = 1000000
nan_interp_sharpness = 4

# surface-pressure adjustment (:140-150)
p_ref_inp = 30000
adj_factor = 0.95
thresh_phi_ref_max_error = 0.15
max_n_iter = 20
i_reinterp = 0

# ---- not in the reference: arithmetic on float32 ERA5 files ----------------------------------
# 'reference': what the reference computes on float32 files - numpy's promotion puts float32 roundings into it
#     (phi_hl stored in FIS' dtype, functions.py:141,149; float32 tav / e_sat of the ERA state; float32 delta_ps and
#     ps_pgw, step_03:182-193) and writes T, QV, U, V as float64 (`era + delta`, step_03:170-173).  The kernels
#     reproduce those roundings (pgw_file_args.ref_dtype = 1), so iteration count and PS follow the reference.
# 'fast': float64 arithmetic on the stored float32 values, float32 outputs (half the output bytes); PS then differs from
#     the reference by a few 1e-7 relative and, when the last pass ends within ~0.03 m2/s2 of the threshold, by one pass.
f32_file_mode = 'reference'

# Output dtype of T, QV, U, V on float32 files in 'reference' mode.  'float64': what the reference writes (`era + delta`
# promotes; the output file is twice the input).  'float32': the same float64 fields, narrowed on the GPU on the way out -
# half the download and half the file written (file I/O is what bounds the end-to-end rate); PS and the pass count are the
# reference's either way.
f32_out_dtype = 'float64'

# Placement of the level arrays (T, QV, U, V in and out, the vapour-pressure workspace) in the card's memory.  'spread': the
# context draws them so that half lie in one stretch of physical memory and half in another (device.SpreadPool; the quad
# kernel runs 12 % faster than with all of them in one stretch, which is what consecutive hipMallocs give; DESIGN.md section
# 4).  'plain': plain allocations.  `PGW_PLACEMENT` overrides.  No influence on any result.
placement = 'spread'
