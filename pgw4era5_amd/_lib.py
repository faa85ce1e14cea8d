"""
ctypes binding of libpgw_hip.so (C-ABI: include/pgw_hip.h).

This is the whole FFI layer: one `argtypes/restype` line per exported function, a loader
that fails loudly, and the status-code -> exception mapping that reproduces the reference's
error behaviour (ValueError with the reference's message).

There is no CPU fallback.  If the library is missing or cannot be loaded, `load()` raises
ImportError naming the build command; every compute entry point goes through `load()`.

HIP-runtime note: when the process also uses torch (multi-GPU launcher: RCCL barrier),
import torch BEFORE the first call here, so that libpgw_hip.so binds to the libamdhip64.so.7
torch already loaded instead of a second copy from /opt/rocm.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PGW_LIB') or os.path.join(_HERE, 'libpgw_hip.so')   # PGW_LIB: A/B builds only

PGW_F32, PGW_F64 = 0, 1
EXTRAP = {'off': 0, 'linear': 1, 'constant': 2, 'nan': 3}
KERNEL_IDS = dict(pressure=0, q_to_rh=1, rh_to_q=2, integ_geopot=3, interp_logp=4, time_lerp=5,
                  vert_interp_delta=6, adjust_ps_step=7, regrid=8, surface=9, finalize=10,
                  thermo_delta=11, wind_delta=12, phi_ref_hybrid=13, quad_delta=14, byteswap=15, harmonic=16, gauss_interp=17, ps_loop_multi=18)

# enum pgw_option (include/pgw_hip.h)
OPTIONS = dict(quad=0, full_column=1, force_vec1=2, multipass=3, loop_guess=4, force_off64=5, test_fail=6)

PGW_OK = 0
PGW_ERR_HIP = 1
PGW_ERR_ARG = 2
PGW_ERR_PREF_AT_TOP = 14
PGW_ERR_PS_HIST_ABOVE_TOP = 15
PGW_ERR_NOT_CONVERGED = 17
PGW_ERR_REDUCE = 20

_vp, _i, _ll, _d, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_double, C.c_size_t
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vpp = C.POINTER(C.c_void_p)          # array of device pointers

# name -> (restype, argtypes); must list every symbol declared in include/pgw_hip.h
SIGNATURES = {
    'pgw_device_count': (_i, [_ip]),
    'pgw_device_pci_bus_id': (_i, [_i, C.c_char_p, _i]),
    'pgw_ctx_create': (_i, [_i, C.POINTER(_vp)]),
    'pgw_ctx_destroy': (_i, [_vp]),
    'pgw_set_option': (_i, [_vp, _i, _i]),
    'pgw_band_abort': (_i, [_vp, _i, _i]),
    'pgw_get_option': (_i, [_vp, _i, _ip]),
    'pgw_last_error': (C.c_char_p, [_vp]),
    'pgw_error_column': (_ll, [_vp]),
    'pgw_version': (C.c_char_p, []),
    'pgw_device_name': (_i, [_vp, C.c_char_p, _sz]),
    'pgw_malloc': (_i, [_vp, _sz, C.POINTER(_vp)]),
    'pgw_free': (_i, [_vp, _vp]),
    'pgw_host_alloc': (_i, [_vp, _sz, C.POINTER(_vp)]),
    'pgw_host_free': (_i, [_vp, _vp]),
    'pgw_memcpy_h2d': (_i, [_vp, _vp, _vp, _sz]),
    'pgw_memcpy_d2h': (_i, [_vp, _vp, _vp, _sz]),
    'pgw_memcpy_d2d': (_i, [_vp, _vp, _vp, _sz]),
    'pgw_memset': (_i, [_vp, _vp, _i, _sz]),
    'pgw_sync': (_i, [_vp]),
    'pgw_mem_info': (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    'pgw_profile_enable': (_i, [_vp, _i]),
    'pgw_profile_reset': (_i, [_vp]),
    'pgw_profile_get': (_i, [_vp, _i, C.POINTER(_ll), _dp]),
    'pgw_timer_start': (_i, [_vp]),
    'pgw_timer_stop': (_i, [_vp, _dp]),
    'pgw_set_levels': (_i, [_vp, _i, _dp, _dp, _dp, _dp]),
    'pgw_get_full_level_coeffs': (_i, [_vp, _dp, _dp]),
    'pgw_pressure_levels': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp]),
    'pgw_specific_to_relative_humidity': (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp]),
    'pgw_relative_to_specific_humidity': (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp]),
    'pgw_humidity_leaf': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp]),
    'pgw_specific_to_relative_humidity_hybrid': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp]),
    'pgw_relative_to_specific_humidity_hybrid': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp]),
    'pgw_integ_geopot': (_i, [_vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp, _d, _vp, _vp, _i]),
    'pgw_interp_logp_4d': (_i, [_vp, _i, _i, _i, _i, _ll, _vp, _vp, _vp, _i, _i, _vp]),
    'pgw_time_lerp': (_i, [_vp, _i, _ll, _vp, _vp, _d, _d, _vp]),
    'pgw_vert_interp_delta': (_i, [_vp, _i, _i, _i, _i, _ll, _dp, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _i, _vp, _vp]),
    'pgw_reinterp_field': (_i, [_vp, _i, _i, _i, _ll, _dp, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'pgw_reinterp_pair': (_i, [_vp, _i, _i, _i, _ll, _dp, _vpp, _vpp, _d, _d, _vpp, _vpp, _vp, _vp, _vpp, _vp, _vp, _i, _vpp]),
    'pgw_reinterp_pass': (_i, [_vp, _i, _i, _i, _ll, _dp, _vpp, _vpp, _d, _d, _vpp, _vpp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _d, _d, _i, _vp, _vp, _vp, _dp]),
    'pgw_replace_delta_sfc': (_i, [_vp, _i, _i, _i, _ll, _dp, _vp, _vp, _vp, _vp, _vp]),
    'pgw_integrate_tos': (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp, _vp]),
    'pgw_adjust_ps_step': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _i, _dp]),
    'pgw_update_ps': (_i, [_vp, _i, _ll, _vp, _vp, _vp, _vp]),
    'pgw_phi_ref_hybrid': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp, _d, _vp, _vp]),
    'pgw_adjust_ps_loop': (_i, [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i,
                                _vp, _vp, _ip, _dp]),
    'pgw_last_levels_touched': (C.c_ulonglong, [_vp]),
    'pgw_regrid_bilinear': (_i, [_vp, _i, _ll, _i, _i, _i, _i, _vp, _ip, _ip, _dp, _dp, _ip,
                                 _ip, _ip, _dp, _dp, _ip, _i, _i, _vp]),
    'pgw_surface_update': (_i, [_vp, _i, _i, _ll, _i, _dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
}



class FileArgs(C.Structure):
    """`pgw_file_args` of include/pgw_hip.h (whole-file entry pgw_step03_file)."""
    _fields_ = (
        [(n, C.c_int) for n in ('dtype', 'ntime', 'nlev', 'nplev', 'nsoil', 'ignore_top', 'max_n_iter', 'local_p_ref',
                                'ref_dtype', '_pad0')] +
        [('ncol', C.c_longlong)] +
        [(n, C.c_void_p) for n in ('PS', 'FIS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_LAND', 'FR_SEA_ICE')] +
        [('soil_depth', _dp), ('plev', _dp)] +
        [(n, C.c_void_p) for n in ('ta_b', 'ta_a', 'hur_b', 'hur_a', 'ua_b', 'ua_a', 'va_b', 'va_a', 'zg_b', 'zg_a', 'zg3_b', 'zg3_a',
                                   'tas_b', 'tas_a', 'hurs_b', 'hurs_a', 'pshist_b', 'pshist_a',
                                   'siconc_b', 'siconc_a', 'ts_b', 'ts_a', 'tos_b', 'tos_a', 'ts_clim')] +
        [(n, C.c_double) for n in ('x_hi', 'x_new', 'p_ref', 'adj_factor', 'thresh')] +
        [(n, C.c_void_p) for n in ('PS_out', 'T_out', 'QV_out', 'U_out', 'V_out', 'hur_pgw_out',
                                   'T_SKIN_out', 'T_SO_out', 'FR_SEA_ICE_out')] +
        [('n_iter', C.c_int), ('passes_launched', C.c_int), ('levels_touched', C.c_ulonglong),
         ('max_err_hist', C.c_double * 32), ('per_var_time', C.c_int), ('i_reinterp', C.c_int)] +
        [(n, C.c_double) for n in ('zg_x_hi', 'zg_x_new', 'siconc_x_hi', 'siconc_x_new', 'ts_x_hi', 'ts_x_new',
                                   'tos_x_hi', 'tos_x_new')])


SIGNATURES['pgw_step03_file'] = (_i, [_vp, C.POINTER(FileArgs)])
REDUCE_MAX_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)      # pgw_reduce_max_fn
SIGNATURES['pgw_set_reduce_hook'] = (_i, [_vp, REDUCE_MAX_FN, _vp])
SIGNATURES['pgw_test_log'] = (_i, [_vp, _ll, _vp, _vp])
SIGNATURES['pgw_test_log_table'] = (_i, [_vp, _ll, _vp, _vp])
SIGNATURES['pgw_test_exp'] = (_i, [_vp, _ll, _vp, _vp, _vp])
SIGNATURES['pgw_test_shared_div'] = (_i, [_vp, _ll, _vp, _vp, _vp])
SIGNATURES['pgw_test_rh_f32'] = (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
SIGNATURES['pgw_byteswap'] = (_i, [_vp, _i, _ll, _vp, _vp])
SIGNATURES['pgw_placement_probe'] = (_i, [_vp, _i, C.POINTER(C.c_void_p), _i, C.POINTER(C.c_void_p), _ll, _ll, _i, C.POINTER(C.c_double)])
SIGNATURES['pgw_ws_adopt'] = (_i, [_vp, _i, _vp, C.c_size_t])
SIGNATURES['pgw_narrow_f64_f32'] = (_i, [_vp, _ll, _vp, _vp, _i])
SIGNATURES['pgw_harmonic_smooth'] = (_i, [_vp, _i, _i, _ll, _dp, _dp, _vp, _vp])
SIGNATURES['pgw_gauss_interp'] = (_i, [_vp, _ll, _vp, _vp, _i, _i, _d, _d, _d, _vp, _ll, _vp, _vp, _vp, _i, _d, _d, _vp])
SIGNATURES['pgw_planar_metres'] = (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp])

_lib = None


class PGWHipError(RuntimeError):
    """HIP runtime failure or bad argument reported by libpgw_hip.so."""


def load():
    """Load libpgw_hip.so and bind every symbol.  Raises ImportError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libpgw_hip.so is not built (%s). Build it with '
            '`make -C pgw4era5_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`. '
            'pgw4era5_amd has no CPU fallback.' % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError('cannot load %s: %s (is the ROCm runtime libamdhip64.so.7 on the '
                          'library path?)' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(lib, name)
        except AttributeError:
            raise ImportError('libpgw_hip.so does not export %s - rebuild it' % name)
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, rc):
    """Map a pgw_status to the exception the reference raises at that point."""
    if rc == PGW_OK:
        return
    lib = load()
    msg = lib.pgw_last_error(ctx_handle)
    msg = msg.decode() if msg else 'error %d' % rc
    col = lib.pgw_error_column(ctx_handle) if ctx_handle else -1
    if rc == PGW_ERR_HIP:
        raise PGWHipError(msg)
    if rc == PGW_ERR_PREF_AT_TOP:
        raise KeyError(0)                       # tav.sel(level=0), functions.py:176
    if rc == PGW_ERR_PS_HIST_ABOVE_TOP:
        e = ValueError()                        # bare ValueError(), functions.py:360-361
        e.column = col
        e.detail = msg
        raise e
    e = ValueError(msg)                         # all other data errors are ValueError
    e.status = rc
    e.column = col
    raise e
