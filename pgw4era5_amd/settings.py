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
