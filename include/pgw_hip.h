/*
 * pgw_hip.h  --  C-ABI of libpgw_hip.so: the MI355X (gfx950) compute path for the
 * PGW4ERA5 step_03 / step_02 hot path.
 *
 * Every entry point replaces one piece of the reference's Python interface; the
 * citation after each declaration is the reference file:line it stands in for
 * (paths relative to the reference repository root).  The reference has no FFI of
 * its own (it is pure Python); the binding a maintainer adds is the ctypes stub shown
 * in INTEGRATION.md (pgw4era5_amd/_lib.py is that stub in full).
 *
 * Conventions
 *  - plain C: opaque context, raw pointers, sizes; no C++/torch types.
 *  - all `field` pointers are DEVICE pointers (from pgw_malloc or any hipMalloc'd
 *    buffer, e.g. a torch tensor's data_ptr()); small coefficient vectors (ak, bk,
 *    plev, coordinate tables) are HOST pointers, copied by the call.
 *  - `dtype` is the STORAGE type of field arrays (PGW_F32 / PGW_F64); arithmetic is
 *    always IEEE fp64, no fast-math.
 *  - 4-D fields are C-order (time, level, lat, lon); `ncol` = nlat*nlon; pressure
 *    ascends with the level index (functions.py:500-503 of the reference).
 *  - calls on one context are stream-ordered on the context's HIP stream and are not
 *    re-entrant; functions that report data errors synchronise before returning.
 *  - return value: 0 = PGW_OK, otherwise a pgw_status code; pgw_last_error() gives
 *    the text, pgw_error_column() the first offending column (or -1).
 */
#ifndef PGW_HIP_H
#define PGW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgw_ctx pgw_ctx;

enum pgw_status {
    PGW_OK = 0,
    PGW_ERR_HIP = 1,                 /* HIP runtime failure (text in pgw_last_error)            */
    PGW_ERR_ARG = 2,                 /* bad argument / shape (functions.py:407-412,457-465)     */
    PGW_ERR_SRC_NOT_ASCENDING = 10,  /* 'Source pressure values must be ascending!'  :500-501   */
    PGW_ERR_TARG_NOT_ASCENDING = 11, /* 'Target pressure values must be ascending!'  :502-503   */
    PGW_ERR_EXTRAP_OFF = 12,         /* 'Extrapolation deactivated but data out of bounds.' :564 */
    PGW_ERR_PREF_BELOW_SURFACE = 13, /* 'p_ref locally lies below the surface...'    :162-165   */
    PGW_ERR_PREF_AT_TOP = 14,        /* k* = 0: tav.sel(level=0) KeyError            :176       */
    PGW_ERR_PS_HIST_ABOVE_TOP = 15,  /* replace_delta_sfc ValueError()               :360-363   */
    PGW_ERR_TOP_PRESSURE = 16,       /* 'ERA5 top pressure is lower than ...'        :417-425   */
    PGW_ERR_NOT_CONVERGED = 17,      /* 'Pressure adjustment did not converge' step_03:313-319  */
    PGW_ERR_GRID_EXTENT = 18,        /* regrid: target exceeds source    functions.py:845-888   */
    PGW_ERR_NO_P_REF = 19,           /* 'No reference pressure level above ...'  step_03:245-251 */
    PGW_ERR_REDUCE = 20              /* the caller's reduce hook (pgw_set_reduce_hook) returned non-zero               */
};

enum pgw_dtype { PGW_F32 = 0, PGW_F64 = 1 };

/* interp_logp_4d `extrapolate` argument (functions.py:434-446) */
enum pgw_extrap { PGW_EXTRAP_OFF = 0, PGW_EXTRAP_LINEAR = 1, PGW_EXTRAP_CONSTANT = 2, PGW_EXTRAP_NAN = 3 };

/* kernel ids for the per-context launch profiler (pgw_profile_*) */
enum pgw_kernel_id {
    PGW_K_PRESSURE = 0, PGW_K_Q_TO_RH = 1, PGW_K_RH_TO_Q = 2, PGW_K_INTEG_GEOPOT = 3,
    PGW_K_INTERP_LOGP = 4, PGW_K_TIME_LERP = 5, PGW_K_VERT_INTERP_DELTA = 6,
    PGW_K_ADJUST_PS_STEP = 7, PGW_K_REGRID = 8, PGW_K_SURFACE = 9, PGW_K_FINALIZE = 10,
    PGW_K_THERMO_DELTA = 11, PGW_K_WIND_DELTA = 12, PGW_K_PHI_REF_HYBRID = 13, PGW_K_QUAD_DELTA = 14,
    PGW_K_BYTESWAP = 15, PGW_K_HARMONIC = 16, PGW_K_GAUSS_INTERP = 17, PGW_K_PS_LOOP_MULTI = 18,
    PGW_K_COUNT = 19
};

/* per-context options (pgw_set_option).  Defaults come from the environment variables named below, which are read
 * ONCE, in pgw_ctx_create; nothing on the launch path calls getenv. */
enum pgw_option {
    PGW_OPT_QUAD = 0,         /* 1 (default): ta+hur and ua+va deltas in one kernel; 0: the two pair kernels   [PGW_QUAD]        */
    PGW_OPT_FULL_COLUMN = 1,  /* 1: loop passes read every level (input-independent traffic); default 0: a wave stops once
                                 all its columns are above p_ref                                                 [PGW_FULL_COLUMN] */
    PGW_OPT_FORCE_VEC1 = 2,   /* 1: one column per thread everywhere (test knob for the scalar-column code path)  [PGW_FORCE_VEC1]  */
    PGW_OPT_MULTIPASS = 3,    /* 1 (default): several passes of the surface-pressure loop per launch with the column held
                                 on chip; 0: one launch per pass                                                 [PGW_MULTIPASS]   */
    PGW_OPT_LOOP_GUESS = 4,   /* passes the next file's first multi-pass launch runs (1..8); updated by every file to its own pass
                                 count (consecutive ERA5 files of a run need the same number); initial value 6                  */
    PGW_OPT_FORCE_OFF64 = 5,  /* 1: the 64-bit byte-offset instantiations of the kernels that address arrays below 4 GiB with
                                 32-bit offsets (k_delta_quad, k_reinterp_pair) - test knob: the path a 0.125 deg L137 file takes    */
    PGW_OPT_TEST_FAIL = 6,    /* test knob, default 0: pgw_step03_file fails on purpose - 1: the loop's workspace (ws_get), 2: before the
                                 first loop launch, 3: before a continuation launch - so that the latitude-band protocol can be
                                 tested where one band stops on its own                                                           */
    PGW_OPT_COUNT = 7
};

/* ---------------------------------------------------------------- context ------------ */
/* One context = one HIP device + one stream; replaces the implicit per-process state of a
 * reference worker (parallel.py:18-32: one process per file, no shared state). */
int pgw_device_count(int *n);
/* PCI address ("0000:c1:00.0", NUL-terminated; len >= 13) of HIP device `device`: the key under /sys/bus/pci/devices/ whose
 * `numa_node` a rank binds its host threads and pinned buffers to (pgw4era5_amd/parallel.py bind_rank_to_numa).  The
 * reference's workers are unplaced `multiprocessing.Pool` processes (parallel.py:18-32). */
int pgw_device_pci_bus_id(int device, char *buf, int len);
int pgw_ctx_create(int device, pgw_ctx **out);
int pgw_ctx_destroy(pgw_ctx *ctx);
int pgw_set_option(pgw_ctx *ctx, int option, int value);
int pgw_get_option(pgw_ctx *ctx, int option, int *value);
/* One ERA5 file split into latitude bands over several contexts / ranks: the columns of
 * a file are independent except for the loop's stopping test, the maximum of |phi error| over ALL columns
 * (step_03_apply_to_era.py:189, 308).  With a hook set, pgw_step03_file hands the per-pass figures of its band to
 * `fn(vals, n, user)`, which must replace every element by its MAXIMUM over the ranks holding the file's other bands
 * (an 8..200-byte all-reduce per loop launch: RCCL / gloo in the caller, see pgw4era5_amd/parallel.py) and return 0.
 * Every rank then takes the same stopping decision, and an error status of one band is returned by all.  n is the same
 * on all ranks at every call.  fn = NULL removes the hook.  Needs the multi-pass loop (PGW_OPT_MULTIPASS = 1,
 * PGW_OPT_FULL_COLUMN = 0; fixed or local p_ref). */
typedef int (*pgw_reduce_max_fn)(double *vals, int n, void *user);
int pgw_set_reduce_hook(pgw_ctx *ctx, pgw_reduce_max_fn fn, void *user);
/* A band that cannot even call pgw_step03_file for the file all bands are about to process (its host-side set-up raised:
 * a failed upload, an out-of-memory) calls this instead: one reduce of the length the others' first loop launch makes,
 * carrying `code`, so that their pgw_step03_file returns `code` instead of waiting for this band until the collective
 * backend's timeout.  (A failure INSIDE pgw_step03_file does this by itself, whenever it happens.) */
int pgw_band_abort(pgw_ctx *ctx, int code, int max_n_iter);
const char *pgw_last_error(pgw_ctx *ctx);
long long pgw_error_column(pgw_ctx *ctx);
const char *pgw_version(void);
int pgw_device_name(pgw_ctx *ctx, char *buf, size_t len);

/* ---------------------------------------------------------------- memory / stream ---- */
int pgw_malloc(pgw_ctx *ctx, size_t bytes, void **dptr);
int pgw_free(pgw_ctx *ctx, void *dptr);
int pgw_host_alloc(pgw_ctx *ctx, size_t bytes, void **hptr);      /* pinned host memory */
int pgw_host_free(pgw_ctx *ctx, void *hptr);
int pgw_memcpy_h2d(pgw_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async */
int pgw_memcpy_d2h(pgw_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async */
int pgw_memcpy_d2d(pgw_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async */
int pgw_memset(pgw_ctx *ctx, void *dst, int value, size_t bytes);             /* async */
int pgw_sync(pgw_ctx *ctx);
int pgw_mem_info(pgw_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* HIP-event launch profiler: when enabled every kernel launched by this library is
 * bracketed by two events on the context stream; pgw_profile_get synchronises and returns
 * launch count and summed device time of kernel `kid` since the last reset. */
int pgw_profile_enable(pgw_ctx *ctx, int on);
int pgw_profile_reset(pgw_ctx *ctx);
int pgw_profile_get(pgw_ctx *ctx, int kid, long long *launches, double *total_ms);
int pgw_timer_start(pgw_ctx *ctx);                 /* event on the context stream */
int pgw_timer_stop(pgw_ctx *ctx, double *ms);      /* records, synchronises, returns elapsed */

/* ---------------------------------------------------------------- vertical grid ------ */
/* Hybrid coefficients of the ERA5 file: ak, bk on half levels (nlev+1 values), akm/bkm on
 * full levels (nlev values) or NULL -> 0.5*diff + lower, as step_03_apply_to_era.py:68-85.
 * Host pointers; kept in the context until the next call. */
int pgw_set_levels(pgw_ctx *ctx, int nlev, const double *ak, const double *bk,
                   const double *akm, const double *bkm);
int pgw_get_full_level_coeffs(pgw_ctx *ctx, double *akm_out, double *bkm_out);

/* a1 "integ_pressure": pa_hl = ak + ps*bk ; pa = akm + ps*bkm
 * (step_03_apply_to_era.py:64-66, 87-88, 196-199).  Either output may be NULL.
 * ps (ntime, ncol) ; pa_hl (ntime, nlev+1, ncol) ; pa (ntime, nlev, ncol). */
int pgw_pressure_levels(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                        const void *ps, void *pa_hl, void *pa);

/* ---------------------------------------------------------------- humidity ----------- */
/* a2 specific_to_relative_humidity(hus, pa, ta)   functions.py:107-116 (with :58-64,:74-105)
 * a3 relative_to_specific_humidity(hur, pa, ta)   functions.py:118-125 (with :66-72)
 * flat elementwise over n elements. */
int pgw_specific_to_relative_humidity(pgw_ctx *ctx, int dtype, long long n,
                                      const void *hus, const void *pa, const void *ta, void *hur);
int pgw_relative_to_specific_humidity(pgw_ctx *ctx, int dtype, long long n,
                                      const void *hur, const void *pa, const void *ta, void *hus);
/* the leaf helpers of the same block, for callers that use them directly (flat, n elements; b is ignored by 2..4):
 *   which = 0  specific_humidity_to_vapor_pressure(hus = a, pa = b)             functions.py:58-64
 *           1  vapor_pressure_to_specific_humidity(vapp = a, pa = b)            :66-72
 *           2  saturation_vapor_pressure_water_or_ice(pa, ta = a, water=True)   :74-89
 *           3  ... water=False (over ice)
 *           4  saturation_vapor_pressure_water_and_ice(pa, ta = a)              :91-105 */
int pgw_humidity_leaf(pgw_ctx *ctx, int dtype, int which, long long n, const void *a, const void *b, void *out);
/* same, with pa = akm + ps*bkm rebuilt in registers instead of read (saves the 4-D pa array
 * step_03:87-94 / :196-197,262-266 materialise).  fields (ntime, nlev, ncol), ps (ntime, ncol) */
int pgw_specific_to_relative_humidity_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                                             const void *hus, const void *ps, const void *ta, void *hur);
int pgw_relative_to_specific_humidity_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                                             const void *hur, const void *ps, const void *ta, void *hus);

/* ---------------------------------------------------------------- integ_geopot ------- */
/* a4 integ_geopot(pa_hl, zgs, ta, hus, level1, p_ref)     functions.py:128-189
 * pa_hl (ntime, nlev+1, ncol); zgs (ntime, ncol); ta, hus (ntime, nlev, ncol);
 * p_ref: scalar `p_ref`, or per-column field `p_ref_field` (ntime, ncol) when non-NULL;
 * phi_ref (ntime, ncol) out.  nlev = number of full levels (level1 labels are 1..nlev+1).
 * full_column != 0 reads every level of every column (signature-faithful traffic,
 * (3*nlev+3)*ncol elements, exact for any pressure ordering); 0 is the caller's assertion
 * that pressure ascends strictly with the level index, and lets a wave stop once all of its
 * columns have passed p_ref (identical results under that assertion). */
int pgw_integ_geopot(pgw_ctx *ctx, int dtype, int ntime, int nlev, long long ncol,
                     const void *pa_hl, const void *zgs, const void *ta, const void *hus,
                     double p_ref, const void *p_ref_field, void *phi_ref, int full_column);

/* ---------------------------------------------------------------- interp_logp_4d ----- */
/* a6 interp_logp_4d(var, source_P, targ_P, extrapolate)   functions.py:434-477 and the
 * column kernels :479-580.  var, source_P (ntime, nsrc, ncol); targ_P, out (ntime, ntarg, ncol).
 * logp_in == 0: pressures in, logs taken inside (:470-471) = interp_logp_4d;
 * logp_in != 0: the arrays already hold ln p = interp_1d_for_timelatlon (:479-508) and, with
 * ntime = ncol = 1, interp_extrap_1d (:511-580). */
int pgw_interp_logp_4d(pgw_ctx *ctx, int dtype, int ntime, int nsrc, int ntarg, long long ncol,
                       const void *var, const void *source_P, const void *targ_P,
                       int extrapolate, int logp_in, void *out);

/* ---------------------------------------------------------------- deltas ------------- */
/* a7 the arithmetic of load_delta's time interpolation (functions.py:288-292 -> scipy
 * interp1d linear): out = (v_after - v_before)/x_hi * x_new + v_before, flat over n. */
int pgw_time_lerp(pgw_ctx *ctx, int dtype, long long n, const void *v_before, const void *v_after,
                  double x_hi, double x_new, void *out);

/* a8 vert_interp_delta(delta, target_P, delta_sfc, ps_hist, ignore_top)  functions.py:369-431
 * (+ replace_delta_sfc :343-366, + the time lerp of load_delta, + `era + delta` of
 * step_03_apply_to_era.py:170-173) fused per column.
 *  plev (nplev, host): the delta file's plev coordinate in FILE order; it is reversed like :383-384.
 *  delta_b / delta_a (nplev, ncol) per time: the two bracketing records (delta_a NULL, or
 *    x_hi == 0 -> delta_b is used as is, functions.py:282-283); same for the optional
 *    surface pairs dsfc_*, pshist_* (ntime, ncol) (NULL -> no surface insertion).
 *  target pressure: targ_P (ntime, nlev_t, ncol) if non-NULL, else akm + ps*bkm from
 *    pgw_set_levels with ps (ntime, ncol).
 *  add_to (ntime, nlev_t, ncol) or NULL: out = add_to + delta_interp.
 *  ignore_top != 0 skips the model-top check (:417-425). */
int pgw_vert_interp_delta(pgw_ctx *ctx, int dtype, int ntime, int nplev, int nlev_t, long long ncol,
                          const double *plev,
                          const void *delta_b, const void *delta_a, double x_hi, double x_new,
                          const void *dsfc_b, const void *dsfc_a,
                          const void *pshist_b, const void *pshist_a,
                          const void *targ_P, const void *ps,
                          int ignore_top, const void *add_to, void *out);

/* f3  settings.i_reinterp = 1, one variable of one loop pass (step_03_apply_to_era.py:202-216; ua / va after the loop,
 * :330-343):  out = interp_logp_4d(era_field, pa_era, pa_pgw, extrapolate='constant') + load_delta_interp(var, pa_pgw)
 * as ONE kernel: pa_era = akm + ps_era*bkm (source axis of the first term) and pa_pgw = akm + ps_pgw*bkm (target of
 * both) are rebuilt in registers from the two surface-pressure fields (ntime, ncol) and pgw_set_levels, the delta
 * arguments are those of pgw_vert_interp_delta.  era_field, out: (ntime, nlev, ncol). */
int pgw_reinterp_field(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                       const void *delta_b, const void *delta_a, double x_hi, double x_new, const void *dsfc_b,
                       const void *dsfc_a, const void *pshist_b, const void *pshist_a, const void *era_field,
                       const void *ps_era, const void *ps_pgw, int ignore_top, void *out);

/* f3  the same for TWO variables that share both pressure axes - ta + hur (one ps_hist; step_03_apply_to_era.py:202-216)
 * or ua + va (:330-343; dsfc_b = NULL) - in one kernel: one target logarithm per level, one source logarithm per level of
 * the ERA column and one bracket search per axis serve both.  delta_b / delta_a / dsfc_b / dsfc_a / era_field / out are
 * arrays of two device pointers (dsfc_b, dsfc_a, delta_a may be NULL as in pgw_reinterp_field); results are the bits of
 * two pgw_reinterp_field calls. */
int pgw_reinterp_pair(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                      const void *const *delta_b, const void *const *delta_a, double x_hi, double x_new,
                      const void *const *dsfc_b, const void *const *dsfc_a, const void *pshist_b, const void *pshist_a,
                      const void *const *era_field, const void *ps_era, const void *ps_pgw, int ignore_top, void *const *out);

/* f3  one pass of the surface-pressure loop with settings.i_reinterp = 1 (step_03_apply_to_era.py:192-216, 262-308) in one
 * call and one host synchronisation: delta_ps += adj_ps, ps_pgw = PS + delta_ps (:192-193); ta_pgw / hur_pgw =
 * interp_logp_4d(T / RELHUM of the ERA state, pa_era, pa_pgw, 'constant') + load_delta_interp(ta / hur, pa_pgw) (:202-216,
 * pgw_reinterp_pair, which also leaves e = hur_pgw / 100 * e_sat(ta_pgw) in a workspace); then the pass of pgw_adjust_ps_step
 * on them: adj_ps and the global max |phi error| (:262-308).  Same bits as pgw_update_ps + pgw_reinterp_pair +
 * pgw_adjust_ps_step(apply_adj = 0).  ps_pgw (ntime, ncol), ta_pgw, hur_pgw (ntime, nlev, ncol) are outputs. */
int pgw_reinterp_pass(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                      const void *const *delta_b, const void *const *delta_a, double x_hi, double x_new,
                      const void *const *dsfc_b, const void *const *dsfc_a, const void *pshist_b, const void *pshist_a,
                      const void *T_era, const void *RELHUM_era, const void *PS, const void *FIS, const double *phi_ref_era,
                      const double *dphi_clim, double *delta_ps, double *adj_ps, double p_ref, double adj_factor,
                      int ignore_top, void *ps_pgw, void *ta_pgw, void *hur_pgw, double *max_abs_err);

/* replace_delta_sfc(source_P, ps_hist, delta, delta_sfc)  functions.py:343-366, on many columns:
 * plev_asc (nplev, host, ascending); delta (ntime, nplev, ncol) in ascending order;
 * delta_sfc, ps_hist (ntime, ncol); outputs out_P, out_delta (ntime, nplev, ncol). */
int pgw_replace_delta_sfc(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol,
                          const double *plev_asc, const void *delta, const void *delta_sfc,
                          const void *ps_hist, void *out_P, void *out_delta);

/* ---------------------------------------------------------------- ps fixed-point loop - */
/* a5 one pass of the loop body step_03_apply_to_era.py:192-308 for fixed p_ref, fused:
 * delta_ps += adj_ps ; ps = PS + delta_ps ; pa, pa_hl ; hus = rh_to_q(hur_pgw, pa, ta_pgw) ;
 * phi_ref_pgw = integ_geopot(...) ; err = (phi_ref_pgw - phi_ref_era) - dphi_clim ;
 * adj_ps = -adj_factor*ps/(Rd*ta_pgw[lowest])*err ; max|err| (NaN skipped).
 *  ta_pgw, hur_pgw (ntime, nlev, ncol) storage dtype; PS, FIS (ntime, ncol) storage dtype;
 *  phi_ref_era, dphi_clim, delta_ps, adj_ps (ntime, ncol) DOUBLE (loop state is fp64);
 *  p_ref_field (double, (ntime,ncol)) or NULL -> scalar p_ref.
 *  max_abs_err (host out). */
int pgw_adjust_ps_step(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                       const void *ta_pgw, const void *hur_pgw, const void *PS, const void *FIS,
                       const double *phi_ref_era, const double *dphi_clim,
                       double *delta_ps, double *adj_ps,
                       double p_ref, const double *p_ref_field, double adj_factor,
                       int apply_adj, double *max_abs_err);
/* apply_adj != 0: the pass itself does delta_ps += adj_ps (step_03:192); 0: the caller already did
 * (pgw_update_ps), as needed when the model-level pressures of the pass are required beforehand
 * (i_reinterp = 1, step_03:202-216). */

/* step_03:192-193: delta_ps += adj_ps ; ps_pgw = PS + delta_ps.  PS, ps_pgw storage dtype (ntime,ncol);
 * delta_ps, adj_ps double. */
int pgw_update_ps(pgw_ctx *ctx, int dtype, long long n, const void *PS, double *delta_ps,
                  const double *adj_ps, void *ps_pgw);

/* phi_ref of a hybrid-level state (T, QV, PS, FIS) at scalar or per-column p_ref, fp64 output:
 * integ_geopot(ak + PS*bk, FIS, T, QV, ., p_ref) without the 4-D pressure array
 * (step_03:280-287 with :64-66). */
int pgw_phi_ref_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *T, const void *QV,
                       const void *PS, const void *FIS, double p_ref, const double *p_ref_field,
                       double *phi_ref);

/* a5 the whole loop step_03_apply_to_era.py:182-319 with its control flow: passes until
 * max|err| <= thresh; error when the pass counter exceeds max_n_iter (so at most
 * max_n_iter-1 passes succeed); the adj_ps of the last pass is NOT applied.
 *  T, QV: the ERA fields for phi_ref_era (computed once; it is constant for fixed p_ref);
 *  dzg_pref (ntime, ncol) storage dtype: zg delta [m] at plev == p_ref (multiplied by g here);
 *  outputs: ps_pgw (ntime, ncol), hus_pgw (ntime, nlev, ncol) storage dtype (either may be NULL),
 *  n_iter, max_err_hist[max_n_iter] (host, may be NULL). */
int pgw_adjust_ps_loop(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                       const void *PS, const void *FIS, const void *T, const void *QV,
                       const void *ta_pgw, const void *hur_pgw, const void *dzg_pref,
                       double p_ref, double adj_factor, double thresh, int max_n_iter,
                       void *ps_pgw, void *hus_pgw, int *n_iter, double *max_err_hist);

/* sum over columns and passes of the full levels the last pgw_adjust_ps_loop actually read
 * (the pass kernel stops a wave above p_ref); used for the bytes-moved accounting. */
unsigned long long pgw_last_levels_touched(pgw_ctx *ctx);

/* ---------------------------------------------------------------- whole file ---------- */
/* The per-file compute path of pgw_for_era5 (reference step_03_apply_to_era.py:62-346 with
 * i_reinterp = 0 and a fixed p_ref) as ONE call on device-resident arrays, using fused kernels:
 *   ta+hur: RELHUM of the ERA state (:91-94), both delta interpolations incl. surface insertion
 *           (functions.py:306-431), the add (:170-173) and e = hur_pgw/100*e_sat(ta_pgw)
 *           (functions.py:123) in one pass over T, QV;
 *   ua+va : both delta interpolations + add in one pass over U, V;
 *   phi_ref of the ERA state from (T, QV, PS, FIS) without a 4-D pressure array (:280-287);
 *   the fixed-point loop (:182-319) and the final PS / QV (:193, :262-266, :369-371);
 *   the surface riders (:103-146).
 * All field pointers are device pointers in storage `dtype`; delta records are the two
 * bracketing time records (`*_a` may be NULL / x_hi == 0 for an exact hit, functions.py:282-283).
 * Results are identical to calling the function-level entry points in the reference's order. */
typedef struct pgw_file_args {
    /* shapes */
    int dtype, ntime, nlev, nplev, nsoil, ignore_top, max_n_iter;
    int local_p_ref;      /* != 0: p_ref_inp = None, reference pressure chosen per column and pass (step_03:219-253) */
    int ref_dtype;        /* != 0 (dtype must be PGW_F32): reference-dtype mode - the float32 roundings numpy's promotion
                             puts into the reference on float32 ERA5 files are reproduced (phi_hl stored float32 per level,
                             functions.py:141,149; float32 tav of the ERA state :144; float32 delta_ps / ps_pgw,
                             step_03:182-193; float32 e_sat chain of RELHUM, functions.py:74-105) and, like the reference's
                             `era + delta` (step_03:170-173), the 4-D outputs T_out, QV_out, U_out, V_out (and hur_pgw_out)
                             are FLOAT64 arrays; PS_out and the surface riders stay float32.
                             0: float64 arithmetic on the stored values, outputs in `dtype`. */
    int _pad0;
    long long ncol;
    /* ERA5 file (device) + small host tables */
    const void *PS, *FIS, *T, *QV, *U, *V;                  /* (ntime,ncol) / (ntime,nlev,ncol)   */
    const void *T_SKIN, *T_SO, *FR_LAND, *FR_SEA_ICE;       /* surface riders, NULL to skip them  */
    const double *soil_depth;                               /* host, nsoil                        */
    const double *plev;                                     /* host, nplev, file order            */
    /* delta records (device) */
    const void *ta_b, *ta_a, *hur_b, *hur_a, *ua_b, *ua_a, *va_b, *va_a;   /* (ntime,nplev,ncol) */
    const void *zg_b, *zg_a;                                /* (ntime,ncol): zg delta at plev == p_ref */
    const void *zg3_b, *zg3_a;                              /* (ntime,nplev,ncol): full zg records (local_p_ref) */
    const void *tas_b, *tas_a, *hurs_b, *hurs_a, *pshist_b, *pshist_a;     /* (ntime,ncol)       */
    const void *siconc_b, *siconc_a, *ts_b, *ts_a, *tos_b, *tos_a, *ts_clim;
    double x_hi, x_new;                                     /* time-lerp abscissae (pgw_time_lerp) */
    /* loop controls (settings.py:140-150 of the reference) */
    double p_ref, adj_factor, thresh;
    /* outputs (device); T_out..V_out (ntime,nlev,ncol); hur_pgw_out optional (may be NULL) */
    void *PS_out, *T_out, *QV_out, *U_out, *V_out, *hur_pgw_out;
    void *T_SKIN_out, *T_SO_out, *FR_SEA_ICE_out;
    /* results */
    int n_iter;
    int passes_launched;  /* passes the loop kernels executed, including passes speculated beyond convergence (multi-pass launches) */
    unsigned long long levels_touched;
    double max_err_hist[32];
    /* The reference loads every delta file on its own (load_delta, functions.py:195-303): files may have different time
     * axes - monthly tos / siconc beside daily 3-D deltas - so a variable's bracketing records and abscissae are its own.
     * per_var_time != 0: the records of zg, siconc, ts and tos are interpolated with the pairs below (x_hi == 0: the
     * instant is a record of that file, `_a` ignored); x_hi / x_new above remain the pair of ta, hur, ua, va, tas, hurs and
     * ps_hist, which the quad kernel interpolates together (a member of that group on another axis is handed over
     * already interpolated by the caller, `_a` == `_b`).  per_var_time == 0: one pair for all (x_hi / x_new). */
    int per_var_time;
    /* != 0: settings.i_reinterp = 1 (step_03_apply_to_era.py:202-216, 330-343) - in every pass of the loop the ERA ta / hur
     * fields and their deltas are interpolated onto the CURRENT model-level pressures, ua / va once after convergence;
     * fixed or local p_ref, every storage mode incl. ref_dtype.  One kernel sequence and one host read-back per pass
     * (the multi-pass loop kernel needs iterate-independent T_pgw / e); no latitude-band sharding. */
    int i_reinterp;
    double zg_x_hi, zg_x_new, siconc_x_hi, siconc_x_new, ts_x_hi, ts_x_new, tos_x_hi, tos_x_new;
} pgw_file_args;

int pgw_step03_file(pgw_ctx *ctx, pgw_file_args *args);

/* ---------------------------------------------------------------- step_02 regridding - */
/* a10 regrid_lat_lon, xarray branch (functions.py:774-789, 817-893): separable linear
 * interpolation, latitude first then longitude, on tables the host derives from the
 * coordinates (pole rows = zonal mean, periodic +-360 copies, scipy interp1d index rule).
 *  src (nfield, nlat_s, nlon_s); out (nfield, nlat_t, nlon_t);
 *  lat_lo/lat_hi (nlat_t): source row of the lower/upper neighbour in the EXTENDED ascending
 *    latitude axis: -1 = south-pole row, nlat_s = north-pole row, else physical row index
 *    (already un-flipped); lat_dx = x_new - x_lo, lat_Dx = x_hi - x_lo; lat_oob != 0 -> NaN.
 *  lon_lo/lon_hi (nlon_t): physical source column (periodic copies folded); lon_dx, lon_Dx, lon_oob.
 *  pole rows use the NaN-skipping zonal mean of physical row `south_row` / `north_row`. */
int pgw_regrid_bilinear(pgw_ctx *ctx, int dtype, long long nfield, int nlat_s, int nlon_s,
                        int nlat_t, int nlon_t, const void *src,
                        const int *lat_lo, const int *lat_hi, const double *lat_dx, const double *lat_Dx,
                        const int *lat_oob,
                        const int *lon_lo, const int *lon_hi, const double *lon_dx, const double *lon_Dx,
                        const int *lon_oob,
                        int south_row, int north_row, void *out);

/* ---------------------------------------------------------------- surface riders ----- */
/* a9 step_03_apply_to_era.py:103-146 + integrate_tos functions.py:1145-1186, 2-D fields (n = ntime*ncol)
 *  sic_out = clip(sic + dsic/100, 0, 1)
 *  dts_comb = integrate_tos(dtos, dts, land, sic_out) ; tskin_out = tskin + dts_comb
 *  tso_out[s] = tso[s] + clim + exp(-soil_depth[s]/2.8)*(dts_comb - clim)   (nsoil levels) */
int pgw_surface_update(pgw_ctx *ctx, int dtype, int ntime, long long ncol, int nsoil,
                       const double *soil_depth,
                       const void *sic, const void *dsic, const void *dtos, const void *dts,
                       const void *land, const void *ts_clim, const void *tskin, const void *tso,
                       void *sic_out, void *dts_comb_out, void *tskin_out, void *tso_out);

/* Spectral smoothing of a daily annual cycle, step_02 `smoothing`: filter_data / harmonic_ac_analysis,
 * functions.py:603-740.  in / out: (ntime, inner) C-order, inner = product of the non-time dimensions; every
 * column's series is replaced by mean + its first three harmonics (Storch & Zwiers 12.19-12.23), all NaN if the
 * series holds a NaN (:694-696).  cos_tab / sin_tab: host tables [3][ntime] of cos / sin(2*pi*i/ntime * t), t = 1..ntime,
 * i = 1..3, evaluated by the caller as the reference does (:716, 727).  ntime < 8 -> PGW_ERR_ARG with the reference's
 * message (:734-737); ntime <= 1365. */
int pgw_harmonic_smooth(pgw_ctx *ctx, int dtype, int ntime, long long inner, const double *cos_tab,
                        const double *sin_tab, const void *in, void *out);

/* NaN-ignoring Gaussian-kernel interpolation of a point cloud onto target points, step_02 for tos / siconc:
 * nan_ignoring_interp, functions.py:900-1060 (pyvista PolyData.interpolate = VTK vtkPointInterpolator + vtkGaussianKernel,
 * radius footprint, null value NaN).  All pointers are DEVICE pointers, coordinates planar metres (functions.py:958-1023;
 * host side: pgw4era5_amd/geodesy.py).
 *  tx, ty (ntarg): target points.  sx, sy (nsrc), sval (nsrc, nfield): source points SORTED by cell of a uniform grid of
 *  square cells (edge `cell` >= radius, origin (x0, y0), ncx x ncy cells, cell id = ix * ncy + iy), `cell_start`
 *  (ncx*ncy + 1 ints): first point of each cell.  sval may hold NaN: that point is skipped for that field (month).
 *  out (nfield, ntarg): sum_i w_i v_i / sum_i w_i, w_i = exp(-(sharpness/radius)^2 d_i^2) over the points with
 *  d_i <= radius; the value of a coincident point (d^2 < 256 eps); NaN where no point lies within the radius. */
int pgw_gauss_interp(pgw_ctx *ctx, long long ntarg, const double *tx, const double *ty, int ncx, int ncy,
                     double x0, double y0, double cell, const int *cell_start, long long nsrc, const double *sx,
                     const double *sy, const double *sval, int nfield, double radius, double sharpness, double *out);

/* The planar "metre" coordinates of that interpolation (functions.py:958-975, 1010-1023: per point three
 * pyproj.Geod(ellps="WGS84").inv lengths - the meridian arc from the equator, the geodesic between (0, lat) and
 * (lon, lat), and the one between (0, lat) and (180, lat)) for n points: lat, lon [deg], lon in (-180, 180]; outputs
 * lat_m = sign(lat) * arc, lon_m = sign(lon) * geodesic, lon_off = over-the-pole length [m].  Device pointers.  Vincenty's
 * series arranged as in pgw4era5_amd/geodesy.py (bisection on the departure azimuth; no iteration that can fail). */
int pgw_planar_metres(pgw_ctx *ctx, long long n, const double *lat, const double *lon, double *lat_m, double *lon_m,
                      double *lon_off);

/* Placement of the level arrays in HBM (no counterpart in the reference: numpy arrays live wherever malloc put them).
 * On an MI355X the rate a column kernel gets depends on WHERE hipMalloc put its arrays relative to each other: an array
 * written while an array of the same stretch of physical memory is read runs at 5.0-5.4 TB/s, arrays of different stretches
 * at 5.7-6.2.  pgw_placement_probe measures it for a given set: the column kernels' access pattern
 * with no arithmetic over n_src (0..4) read streams and n_dst (0..4) write streams of rows x ncol float64 each, `reps`
 * timed launches after one warm-up, *gbps = bytes moved / time.  The dst arrays are OVERWRITTEN.  Synchronous.
 * pgw_ws_adopt hands the library a pgw_malloc'ed buffer as its workspace `slot` (0 = the vapour-pressure field of the
 * file path, one ERA5 level field of float64) so that the host layer can choose that array's place too; the library owns
 * and frees it from then on; (NULL, 0) releases the slot. */
int pgw_placement_probe(pgw_ctx *ctx, int n_src, const void *const *src, int n_dst, void *const *dst, long long rows,
                        long long ncol, int reps, double *gbps);
int pgw_ws_adopt(pgw_ctx *ctx, int slot, void *dptr, size_t bytes);

/* Byte-order conversion on the device: dst[i] = byte-reversed src[i] for n elements of 4 or 8 bytes (in place
 * allowed).  NetCDF classic files are big-endian (the reference reads / writes them through xarray,
 * step_03_apply_to_era.py:60, 378, which converts on the host); with this entry the raw file bytes are uploaded
 * and converted at HBM speed, and results are converted before the download, so no host pass touches the fields. */
int pgw_byteswap(pgw_ctx *ctx, int elem_bytes, long long n, const void *src, void *dst);

/* dst[i] = (float)src[i], optionally written byte-reversed (big_endian != 0: the NetCDF classic byte order), n elements,
 * device pointers.  On float32 ERA5 files the reference's `era + delta` promotes T, QV, U, V to float64 and writes them
 * so (step_03_apply_to_era.py:170-173, 369-378: the output file is twice the input).  With settings.f32_out_dtype =
 * 'float32' the file driver computes exactly the same float64 fields and narrows them here on the way out - half the
 * download and half the file; values = the reference's, rounded once to float32.  No reference counterpart. */
int pgw_narrow_f64_f32(pgw_ctx *ctx, long long n, const double *src, void *dst, int big_endian);

/* diagnostic: out[i] = ln(in[i]) with the device logarithm every kernel uses (pgw_device.h
 * pgw_log: fdlibm log kernel for positive normal finite x, ocml log otherwise); device fp64 arrays */
int pgw_test_log(pgw_ctx *ctx, long long n, const double *in, double *out);
/* same for the table-driven logarithm of the hybrid-level loops (pgw_device.h pgw_log_tab: 128-entry table, r = fma(z, 1/c, -1),
 * degree-7 polynomial; generated by tools/gen_log_table.py) */
int pgw_test_log_table(pgw_ctx *ctx, long long n, const double *in, double *out);

/* diagnostic: out[i] = pgw_exp(in[i]) (pgw_device.h: the device library's exp arithmetic written with explicit FMAs, the
 * exponential of every e_sat evaluation), ref[i] = exp(in[i]) of the device library; device fp64 arrays.  Tests require
 * out == ref bit for bit and <= 1 ulp from numpy. */
int pgw_test_exp(pgw_ctx *ctx, long long n, const double *in, double *out, double *ref);

/* diagnostic: RELHUM of a float32 ERA state as numpy's promotion evaluates functions.py:58-116 on float32 files
 * (reference-dtype mode, step_03_apply_to_era.py:91-94): out = the form the quad kernel uses (the one phase a temperature
 * needs, scale-free divisions, expf without range selects), lit = the expression as written (both phases, IEEE divisions,
 * library expf); es / es_lit = the float32 e_sat alone.  hus, ta float32, pa float64; device arrays.  Tests require
 * out == lit and es == es_lit bit for bit over the physical range and the same special values outside it. */
int pgw_test_rh_f32(pgw_ctx *ctx, long long n, const float *hus, const double *pa, const float *ta,
                    double *out, double *lit, float *es, float *es_lit);

/* diagnostic: out[i] = num[i] / den[i] through the shared-divisor path (pgw_device.h SharedDivisor: reciprocal once,
 * three instructions per quotient) that the regridding and delta kernels use where many numerators share a divisor;
 * device fp64 arrays.  Tests require the IEEE quotient bit for bit. */
int pgw_test_shared_div(pgw_ctx *ctx, long long n, const double *num, const double *den, double *out);

/* integrate_tos(tos_field, ts_field, land_frac, ice_frac)  functions.py:1145-1186, flat over n */
int pgw_integrate_tos(pgw_ctx *ctx, int dtype, long long n, const void *tos, const void *ts,
                      const void *land, const void *ice, void *out);

#ifdef __cplusplus
}
#endif
#endif /* PGW_HIP_H */
