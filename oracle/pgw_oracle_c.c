/*
 * CPU oracle, C part  --  TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Plain serial C restatement of the reference's per-column loops: the functions the
 * reference runs as serial numba (`@njit`) or through `np.vectorize`, i.e. one call per
 * (time, lat, lon) column on strided column views `[t, :, j, i]`.  The numpy oracle
 * (oracle/pgw_oracle.py) vectorises these over columns; this file keeps the reference's
 * loop order and memory access pattern, so that
 *   - the vectorised oracle has an independent column-by-column check, and
 *   - bench.py's `cpu_baseline` can time the loops as the reference executes them
 *     ("per-column loop ... compiled single-thread to stand in for
 *     serial numba").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file (oracle/_build/libpgw_oracle_c.so, `make -C oracle`).
 *
 * Parity pin: `pgwc_interp_extrap_1d`, `pgwc_interp_1d_for_timelatlon` and
 * `pgwc_replace_delta_sfc` are checked against tests/golden/ref_leaf_vectors.npz, which
 * holds outputs of the reference's own functions (oracle/make_golden.py).  The
 * `np.vectorize` wrapper of vert_interp_delta is xarray-bound in the reference and is
 * checked against the numpy oracle only ("parity unpinned against a reference run").
 *
 * Citations are file:line of /root/reference.  Arithmetic is double; compiled with
 * -ffp-contract=off so that every operation rounds like numpy's.
 */
#include <math.h>
#include <stddef.h>

enum { PGWC_OFF = 0, PGWC_LINEAR = 1, PGWC_CONSTANT = 2, PGWC_NAN = 3 };
enum { PGWC_OK = 0, PGWC_ERR_EXTRAP_OFF = 1, PGWC_ERR_SRC_ORDER = 2, PGWC_ERR_TARG_ORDER = 3,
       PGWC_ERR_PS_ABOVE_TOP = 4 };

/* functions.py:511-580  interp_extrap_1d(src_x, src_y, targ_x, extrapolate).
 * sx/sy/tx/ty are element strides of the column views (numba works on the strided views
 * `[t, :, j, i]`, functions.py:496-498). */
int pgwc_interp_extrap_1d(const double *src_x, ptrdiff_t sx, const double *src_y, ptrdiff_t sy, int ns,
                          const double *targ_x, ptrdiff_t tx, int nt, double *targ_y, ptrdiff_t ty, int mode)
{
    for (int ti = 0; ti < nt; ti++) {                              /* :524 */
        const double x = targ_x[ti * tx];
        int i1 = -1, i2 = -1, require_extrap = 0;                  /* :525-527 */
        for (int si = 0; si < ns; si++) {                          /* :528 */
            const double s = src_x[si * sx];
            if (si == 0 && s > x) {                                /* :530-538  below the first source point */
                if (mode == PGWC_LINEAR) { i1 = 0; i2 = 1; }
                else if (mode == PGWC_CONSTANT) { i1 = 0; i2 = 0; }
                require_extrap = 1;
                break;
            } else if (s == x) {                                   /* :540-543  exact hit */
                i1 = si; i2 = si;
                break;
            } else if (s > x) {                                    /* :545-548  bracket found */
                i1 = si - 1; i2 = si;
                break;
            }
        }
        if (i1 == -1 && !require_extrap) {                         /* :554-561  above the last source point */
            if (mode == PGWC_LINEAR) { i1 = ns - 2; i2 = ns - 1; }
            else if (mode == PGWC_CONSTANT) { i1 = ns - 1; i2 = ns - 1; }
            require_extrap = 1;
        }
        if (require_extrap && mode == PGWC_OFF)                    /* :564-566 */
            return PGWC_ERR_EXTRAP_OFF;
        if (require_extrap && mode == PGWC_NAN) {                  /* :569-570 */
            targ_y[ti * ty] = NAN;
        } else if (i1 == i2) {                                     /* :572-573 */
            targ_y[ti * ty] = src_y[i1 * sy];
        } else {                                                   /* :575-578 */
            const double y1 = src_y[i1 * sy], y2 = src_y[i2 * sy];
            const double x1 = src_x[i1 * sx], x2 = src_x[i2 * sx];
            targ_y[ti * ty] = y1 + (x - x1) * (y2 - y1) / (x2 - x1);
        }
    }
    return PGWC_OK;
}

/* functions.py:479-508  interp_1d_for_timelatlon: serial triple loop over (time, lat, lon), one
 * interp_extrap_1d per column; arrays are C-order (time, level, lat, lon), inputs already ln p.
 * *bad receives the flat column index (t*nlat*nlon + j*nlon + i) of the first failure. */
int pgwc_interp_1d_for_timelatlon(const double *orig, const double *src_p, const double *targ_p, double *out,
                                  int ntime, int nsrc, int ntarg, int nlat, int nlon, int mode, long long *bad)
{
    const ptrdiff_t plane = (ptrdiff_t)nlat * nlon;
    for (int t = 0; t < ntime; t++)                                /* :490 */
        for (int j = 0; j < nlat; j++)                             /* :491 */
            for (int i = 0; i < nlon; i++) {                       /* :492 */
                const ptrdiff_t col = (ptrdiff_t)j * nlon + i;
                const double *sp = src_p + (ptrdiff_t)t * nsrc * plane + col;
                const double *sy = orig + (ptrdiff_t)t * nsrc * plane + col;
                const double *tp = targ_p + (ptrdiff_t)t * ntarg * plane + col;
                double *o = out + (ptrdiff_t)t * ntarg * plane + col;
                int rc = PGWC_OK;
                if (sp[(ptrdiff_t)(nsrc - 1) * plane] < sp[0]) rc = PGWC_ERR_SRC_ORDER;         /* :500-501 */
                else if (tp[(ptrdiff_t)(ntarg - 1) * plane] < tp[0]) rc = PGWC_ERR_TARG_ORDER;  /* :502-503 */
                else rc = pgwc_interp_extrap_1d(sp, plane, sy, plane, nsrc, tp, plane, ntarg, o, plane, mode);
                if (rc != PGWC_OK) {
                    if (bad) *bad = (long long)t * plane + col;
                    return rc;
                }
            }
    return PGWC_OK;
}

/* functions.py:343-366  replace_delta_sfc(source_P, ps_hist, delta, delta_sfc) on one column;
 * source_P ascending (:383-384 reversed it).  st = element stride of the in/out column views. */
int pgwc_replace_delta_sfc(const double *source_P, ptrdiff_t ps_stride, int ns, double ps_hist,
                           const double *delta, ptrdiff_t st, double delta_sfc,
                           double *out_P, double *out_D, ptrdiff_t ost)
{
    double pmax = source_P[0], pmin = source_P[0];
    int nan_seen = 0;
    for (int k = 0; k < ns; k++) {                                 /* copies, :352-353; np.max / np.min propagate NaN */
        const double p = source_P[k * ps_stride];
        out_P[k * ost] = p;
        out_D[k * ost] = delta[k * st];
        if (p != p) nan_seen = 1;
        if (p > pmax) pmax = p;
        if (p < pmin) pmin = p;
    }
    if (nan_seen) pmax = pmin = NAN;
    if (ps_hist > pmax) {                                          /* :356-359 */
        out_P[(ptrdiff_t)(ns - 1) * ost] = ps_hist;
        out_D[(ptrdiff_t)(ns - 1) * ost] = delta_sfc;
    } else if (ps_hist < pmin) {                                   /* :360-361 */
        return PGWC_ERR_PS_ABOVE_TOP;
    } else {                                                       /* :362-365 */
        int sfc = -1;
        for (int k = 0; k < ns; k++)
            if (ps_hist > source_P[k * ps_stride]) sfc = k;        /* np.max(np.argwhere(ps_hist > source_P)) */
        if (sfc < 0) return PGWC_ERR_PS_ABOVE_TOP;                 /* np.max of an empty array raises ValueError too */
        for (int k = sfc; k < ns; k++) out_D[k * ost] = delta_sfc;
        out_P[(ptrdiff_t)sfc * ost] = ps_hist;
    }
    return PGWC_OK;
}

/* functions.py:395-405  the np.vectorize call of vert_interp_delta: replace_delta_sfc once per
 * (time, lat, lon) column of delta (time, S, lat, lon) with the 1-D ascending plev axis; writes
 * the 4-D source_P and the modified delta. */
int pgwc_replace_delta_sfc_columns(const double *plev_asc, int ns, const double *ps_hist, const double *delta,
                                   const double *delta_sfc, double *out_P, double *out_D,
                                   int ntime, int nlat, int nlon, long long *bad)
{
    const ptrdiff_t plane = (ptrdiff_t)nlat * nlon;
    for (int t = 0; t < ntime; t++)
        for (ptrdiff_t c = 0; c < plane; c++) {
            const ptrdiff_t off = (ptrdiff_t)t * ns * plane + c;
            int rc = pgwc_replace_delta_sfc(plev_asc, 1, ns, ps_hist[(ptrdiff_t)t * plane + c], delta + off, plane,
                                            delta_sfc[(ptrdiff_t)t * plane + c], out_P + off, out_D + off, plane);
            if (rc != PGWC_OK) {
                if (bad) *bad = (long long)t * plane + c;
                return rc;
            }
        }
    return PGWC_OK;
}
