"""
step_02: regrid GCM climate deltas (and HIST climatologies) to the ERA5 grid on MI355X.

Command line of the reference's step_02_preproc_deltas.py (:27-87): positional
{smoothing,regridding}, -i, -o, -e, -v.  For every variable both `{var}_historical.nc` and
`{var}_delta.nc` are processed (:116-119).  `regridding` runs the bilinear lat-then-lon kernel
(functions.regrid_lat_lon); `smoothing` runs the annual-cycle filter for daily deltas
(functions.filter_data, reference functions.py:603-740); tos / siconc on the ocean grid go through the
NaN-ignoring Gaussian-kernel interpolation (functions.nan_ignoring_interp, reference functions.py:900-1060).
"""
import argparse
import os
from pathlib import Path

from . import ncio
from .functions import filter_data, interp_wrapper
from .settings import file_name_bases, i_use_xesmf_regridding, nan_interp_kernel_radius, nan_interp_sharpness

DEFAULT_VARS = 'ta,hur,ua,va,zg,hurs,tas,ps,tos,ts,siconc'


def main(argv=None):
    p = argparse.ArgumentParser(description='PGW for ERA5: regrid GCM deltas to the ERA5 grid (MI355X).')
    p.add_argument('processing_step', type=str, choices=['smoothing', 'regridding'])
    p.add_argument('-i', '--input_dir', type=str, help='directory with {var}_delta.nc and {var}_historical.nc')
    p.add_argument('-o', '--output_dir', type=str, help='directory for the regridded files')
    p.add_argument('-e', '--era5_file_path', type=str, default=None, help='example ERA5 file (target grid)')
    p.add_argument('-v', '--var_names', type=str, default=DEFAULT_VARS, help='comma separated variable names')
    args = p.parse_args(argv)
    print(args)
    if args.input_dir is None:
        raise ValueError('Input directory (-i) is required.')
    if args.output_dir is None:
        raise ValueError('Output directory (-o) is required.')
    if args.processing_step == 'regridding' and args.era5_file_path is None:
        raise ValueError('era5_file_path is required for regridding step.')
    Path(args.output_dir).mkdir(exist_ok=True, parents=True)
    var_names = args.var_names.split(',')
    print('Run {} for variable names {}.'.format(args.processing_step, var_names))
    smoothing = args.processing_step == 'smoothing'
    ds_era5 = None if smoothing else ncio.open_dataset(args.era5_file_path, decode_times=False)
    done = []
    for var_name in var_names:
        print(var_name)
        if smoothing:                                        # step_02_preproc_deltas.py:129-132
            for clim_period in ['HIST', 'SCEN-HIST']:
                fname = file_name_bases[clim_period].format(var_name)
                inp, out = os.path.join(args.input_dir, fname), os.path.join(args.output_dir, fname)
                filter_data(inp, var_name, out)
                done.append(out)
            continue
        for clim_period in ['HIST', 'SCEN-HIST']:
            fname = file_name_bases[clim_period].format(var_name)
            inp, out = os.path.join(args.input_dir, fname), os.path.join(args.output_dir, fname)
            if not os.path.exists(inp):
                raise ValueError('Files for variable ' + var_name + ' are missing')
            ds_gcm = ncio.open_dataset(inp)
            ds_out = interp_wrapper(ds_gcm, ds_era5, var_name, i_use_xesmf=i_use_xesmf_regridding,
                                    nan_interp_kernel_radius=nan_interp_kernel_radius,
                                    nan_interp_sharpness=nan_interp_sharpness)
            ncio.to_netcdf(ds_out, out)
            done.append(out)
    return done


if __name__ == '__main__':
    main()
