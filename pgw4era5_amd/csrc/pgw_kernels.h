// pgw_kernels.h -- hand-written gfx950 kernels for the PGW4ERA5 step_03 / step_02 hot path.
// All arithmetic is IEEE fp64 (no fast-math); T is the storage type of field arrays.
// Citations are reference file:line (functions.py unless prefixed).
#pragma once
#include "pgw_device.h"

namespace pgw {

// Signature-faithful streaming kernels read and write each element once: streaming (non-temporal) forms (pgw_device.h)
#define SIG_LOADV loadv_nt
#define SIG_LD(p) __builtin_nontemporal_load(p)
#define SIG_ST(v, p) __builtin_nontemporal_store((v), (p))

// Vertical-grid tables of a context, in device memory; every access is wave-uniform
// (index = level loop counter) so the compiler emits scalar (s_load) reads.
struct Levels {
    const double *ak;    // nlev+1
    const double *bk;    // nlev+1
    const double *akm;   // nlev
    const double *bkm;   // nlev
    int nlev;
    double ps_mono_min;  // columns with ps >= this have strictly ascending half-level pressure
};

// The tables are staged in LDS by every kernel that walks levels: a coefficient read through the
// kernel-argument pointer compiles to a VECTOR global load (the compiler cannot prove the table is
// not aliased by the kernel's stores), and its `s_waitcnt vmcnt` then also waits for every
// prefetched row and earlier store of the wave (vmcnt retires in order) - one such load per level
// serialises the whole software pipeline.  ds_read uses lgkmcnt and leaves vmcnt traffic in flight.
constexpr int MAX_NLEV = 256;
struct LevTab {            // LDS image: ak[0..N] | bk[0..N] | akm[0..N-1] | bkm[0..N-1] | table of pgw_log_tab
    const double *ak, *bk, *akm, *bkm, *logtab;
};
template <bool HALF, bool FULL>
__device__ __forceinline__ LevTab stage_levels(const Levels &lv, double *lds, int nthreads) {
    const int N = lv.nlev;
    LevTab t;
    t.ak = lds; t.bk = lds + (MAX_NLEV + 1); t.akm = lds + 2 * (MAX_NLEV + 1); t.bkm = t.akm + MAX_NLEV;
    t.logtab = t.bkm + MAX_NLEV;
    stage_log_table(lds + 2 * (MAX_NLEV + 1) + 2 * MAX_NLEV, nthreads);
    double *w = lds;
    for (int i = threadIdx.x; i <= N; i += nthreads) {
        if (HALF) { w[i] = lv.ak[i]; w[(MAX_NLEV + 1) + i] = lv.bk[i]; }
        if (FULL && i < N) { w[2 * (MAX_NLEV + 1) + i] = lv.akm[i]; w[2 * (MAX_NLEV + 1) + MAX_NLEV + i] = lv.bkm[i]; }
    }
    __syncthreads();
    return t;
}
constexpr int LEVTAB_DOUBLES = 2 * (MAX_NLEV + 1) + 2 * MAX_NLEV + 2 * LOG_TABLE_N;

// flat column group -> (time, column) and base offsets
struct ColIdx {
    long long t, c;
};
__device__ __forceinline__ ColIdx col_index(long long g, int V, long long ncol) {
    long long flat = g * V;
    ColIdx r;
    r.t = flat / ncol;
    r.c = flat - r.t * ncol;
    return r;
}

// =====================================================================================
// a1  pressure on half / full levels          step_03_apply_to_era.py:64-66,87-88,196-199
// write-bound: (2N+1) rows out per column, 1 element in.
// =====================================================================================
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_pressure_levels(Levels lv, int ntime, long long ncol,
                                                           const T *__restrict__ ps,
                                                           T *__restrict__ pa_hl, T *__restrict__ pa) {
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<true, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    if (g >= ngroups) return;
    ColIdx ix = col_index(g, V, ncol);
    double p[V];
    loadv<T, V>(ps + ix.t * ncol + ix.c, p);
    const int N = lv.nlev;
    if (pa_hl) {
        T *o = pa_hl + ix.t * (N + 1) * ncol + ix.c;
#pragma unroll 4
        for (int k = 0; k <= N; ++k) {
            double a = lt.ak[k], b = lt.bk[k], r[V];
#pragma unroll
            for (int v = 0; v < V; ++v) r[v] = a + p[v] * b;
            storev<T, V>(o + (long long)k * ncol, r);
        }
    }
    if (pa) {
        T *o = pa + ix.t * N * ncol + ix.c;
#pragma unroll 4
        for (int l = 0; l < N; ++l) {
            double a = lt.akm[l], b = lt.bkm[l], r[V];
#pragma unroll
            for (int v = 0; v < V; ++v) r[v] = a + p[v] * b;
            storev<T, V>(o + (long long)l * ncol, r);
        }
    }
}

// =====================================================================================
// a2 / a3  humidity conversions, flat elementwise             functions.py:107-125
// MODE 0: q -> RH ; MODE 1: RH -> q ; MODE 2: RH -> e (vapour pressure, :123)
// =====================================================================================
template <typename T, int V, int MODE>
__global__ __launch_bounds__(BLOCK) void k_humidity_flat(long long n, const T *__restrict__ x,
                                                         const T *__restrict__ pa,
                                                         const T *__restrict__ ta, T *__restrict__ out) {
    long long stride = (long long)gridDim.x * BLOCK;
    for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g * V < n; g += stride) {
        double a[V], p[V], t[V], r[V];
        loadv<T, V>(x + g * V, a);
        loadv<T, V>(pa + g * V, p);
        loadv<T, V>(ta + g * V, t);
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = (MODE == 0) ? q_to_rh(a[v], p[v], t[v]) : rh_to_q(a[v], p[v], t[v]);
        storev<T, V>(out + g * V, r);
    }
}

// the leaf helpers (functions.py:58-105), literal forms with IEEE divisions
template <typename T, int WHICH>
__global__ __launch_bounds__(BLOCK) void k_humidity_leaf(long long n, const T *__restrict__ a, const T *__restrict__ b,
                                                         T *__restrict__ out) {
    long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const double x = (double)a[i];
        double r;
        if (WHICH == 0) { const double p = (double)b[i]; r = x * p / (CON_MW_MD + 0.378 * x); }
        else if (WHICH == 1) { const double p = (double)b[i]; r = CON_MW_MD * x / (p - (1 - CON_MW_MD) * x); }
        else if (WHICH == 2) r = 611.21 * pgw_exp(17.502 * (x - 273.16) / (x - 32.19));
        else if (WHICH == 3) r = 611.21 * pgw_exp(22.587 * (x - 273.16) / (x - (-0.7)));
        else r = esat_mixed(x);
        out[i] = (T)r;
    }
}

// RELHUM of a float32 ERA state in reference-dtype mode (step_03:91-94 on a float32 file): float32 QV, T, PS in, the float64
// field numpy's promotion produces out (q_to_rh_f32: float64 pressure, float32 e_sat chain)
__global__ __launch_bounds__(BLOCK) void k_relhum_ref(Levels lv, int ntime, long long ncol, const float *__restrict__ hus,
                                                      const float *__restrict__ ps, const float *__restrict__ ta,
                                                      double *__restrict__ out) {
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<false, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= (long long)ntime * ncol) return;
    const long long t = g / ncol, c = g - t * ncol;
    const double p = (double)ps[g];
    const int N = lv.nlev;
    const long long base = t * N * ncol + c;
#pragma unroll 4
    for (int l = 0; l < N; ++l) {
        const long long o = base + (long long)l * ncol;
        out[o] = q_to_rh_f32(hus[o], lt.akm[l] + p * lt.bkm[l], ta[o]);
    }
}

// same with pa = akm + ps*bkm rebuilt in registers (no 4-D pressure array)
template <typename T, int V, int MODE>
__global__ __launch_bounds__(BLOCK) void k_humidity_hybrid(Levels lv, int ntime, long long ncol,
                                                           const T *__restrict__ x, const T *__restrict__ ps,
                                                           const T *__restrict__ ta, T *__restrict__ out) {
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<false, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    if (g >= ngroups) return;
    ColIdx ix = col_index(g, V, ncol);
    double p[V];
    loadv<T, V>(ps + ix.t * ncol + ix.c, p);
    const int N = lv.nlev;
    long long base = ix.t * N * ncol + ix.c;
#pragma unroll 2
    for (int l = 0; l < N; ++l) {
        double a[V], t[V], r[V];
        loadv<T, V>(x + base + (long long)l * ncol, a);
        loadv<T, V>(ta + base + (long long)l * ncol, t);
        double am = lt.akm[l], bm = lt.bkm[l];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double pa = am + p[v] * bm;
            r[v] = (MODE == 0) ? q_to_rh(a[v], pa, t[v]) : (MODE == 1) ? rh_to_q(a[v], pa, t[v]) : rh_to_e(a[v], t[v]);
        }
        storev<T, V>(out + base + (long long)l * ncol, r);
    }
}

// =====================================================================================
// a4  integ_geopot                                             functions.py:128-189
// One thread = V columns, bottom-up scan; the half level k* with the smallest non-negative
// p_hl - p_ref (ties -> lowest k, like nanargmin) is tracked while scanning, so non-monotone
// columns give the reference's answer too.
// =====================================================================================
struct GeoAcc {            // per-column running state of the upward scan
    double phi;            // phi at the lower half level of the current layer
    double p_lo, lnp_lo;   // pressure / ln p at that half level
    double dmin;           // smallest non-negative p_hl - p_ref so far
    double phi_s, rtv_s, lnp_s;   // phi, CON_RD*tv and ln p captured at the candidate k*
    int kstar;             // -1 = none yet
};

// Reference-dtype mode (REF, float32 ERA5 files): `phi_hl` is created from the float32
// `zgs` (functions.py:141), so every assignment phi_hl[l] = ... (:149-152) rounds to float32; `tav` and `CON_RD * tav`
// are float32 products for the float32 ERA state (:144, :150).  All of these are single IEEE operations, reproduced
// exactly: cvt f64->f32->f64 for the store, v_mul_f32 / v_add_f32 for the products (-ffp-contract=off).
template <bool REF>
__device__ __forceinline__ double phi_store(double x) { return REF ? (double)(float)x : x; }
// CON_RD * (ta * (1 + 0.61 * hus)) for float32 ta, hus as numpy evaluates it: python floats become float32 scalars
__device__ __forceinline__ double rd_tv_f32(double t, double q) {
    const float tv = (float)t * (1.0f + 0.61f * (float)q);       // :144
    return (double)(287.05f * tv);                               // :150  CON_RD * tav.sel(...)
}

// `logtab`: LDS table of pgw_log_tab (the hybrid-level kernels), or nullptr: fdlibm kernel (pgw_log)
__device__ __forceinline__ void geo_init(GeoAcc &a, double zgs, double p_bottom, const double *logtab = nullptr) {
    a.phi = zgs;
    a.p_lo = fix_p(p_bottom);
    a.lnp_lo = logtab ? pgw_log_tab(a.p_lo, logtab) : pgw_log(a.p_lo);
    a.dmin = __builtin_inf();
    a.kstar = -1;
    a.phi_s = a.rtv_s = a.lnp_s = 0.0;
}
// process layer l (between half levels l and l+1); rtv = CON_RD * tv of the layer; p_top = pa_hl[l] raw
template <bool REF = false, bool TAB = false>
__device__ __forceinline__ void geo_layer(GeoAcc &a, int l, double rtv, double p_top, double p_ref, const double *logtab = nullptr) {
    double d = a.p_lo - p_ref;                        // candidate k = l+1         :160-161
    if (d >= 0 && d <= a.dmin) {
        a.dmin = d; a.kstar = l + 1; a.phi_s = a.phi; a.rtv_s = rtv; a.lnp_s = a.lnp_lo;
    }
    double p_hi = fix_p(p_top);                       // :135
    double lnp_hi = TAB ? pgw_log_tab(p_hi, logtab) : pgw_log_f3(p_hi);      // the log of the level loops
    a.phi = phi_store<REF>(a.phi + rtv * (a.lnp_lo - lnp_hi));   // :149-152, dlnpa :136-138
    a.p_lo = p_hi; a.lnp_lo = lnp_hi;
}
// returns phi_ref; reports errors
__device__ __forceinline__ double geo_finish(GeoAcc &a, double p_ref, DevStatus *st, long long col) {
    double d = a.p_lo - p_ref;                        // candidate k = 0
    if (d >= 0 && d <= a.dmin) a.kstar = 0;
    if (a.kstar < 0) { report(st, 13 /*PGW_ERR_PREF_BELOW_SURFACE*/, col); return __builtin_nan(""); }
    if (a.kstar == 0) { report(st, 14 /*PGW_ERR_PREF_AT_TOP*/, col); return __builtin_nan(""); }
    return a.phi_s - a.rtv_s * (pgw_log(p_ref) - a.lnp_s);    // :174-179
}

// TO = type of the phi_ref output (T for the signature-faithful call, double for the loop state)
template <typename T, int V, int U, typename TO>
__global__ __launch_bounds__(BLOCK) void k_integ_geopot(int nlev, int ntime, long long ncol,
                                                        const T *__restrict__ pa_hl, const T *__restrict__ zgs,
                                                        const T *__restrict__ ta, const T *__restrict__ hus,
                                                        double p_ref_s, const T *__restrict__ p_ref_f,
                                                        TO *__restrict__ phi_ref, int full_column,
                                                        DevStatus *st) {
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    if (g >= ngroups) return;
    ColIdx ix = col_index(g, V, ncol);
    const int N = nlev;
    const T *ph = pa_hl + ix.t * (N + 1) * ncol + ix.c;
    const T *pt = ta + ix.t * N * ncol + ix.c;
    const T *pq = hus + ix.t * N * ncol + ix.c;
    long long c2 = ix.t * ncol + ix.c;
    double z[V], pb[V], pref[V];
    loadv<T, V>(zgs + c2, z);
    loadv<T, V>(ph + (long long)N * ncol, pb);
    if (p_ref_f) loadv<T, V>(p_ref_f + c2, pref);
    else {
#pragma unroll
        for (int v = 0; v < V; ++v) pref[v] = p_ref_s;
    }
    GeoAcc acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) geo_init(acc[v], z[v], pb[v]);
    int l = N - 1;
    // chunks of U levels: issue all 3*U row loads, then the dependent scan
    for (; l >= U - 1; l -= U) {
        double p[U][V], t[U][V], q[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            SIG_LOADV<T, V>(ph + (long long)(l - u) * ncol, p[u]);
            SIG_LOADV<T, V>(pt + (long long)(l - u) * ncol, t[u]);
            SIG_LOADV<T, V>(pq + (long long)(l - u) * ncol, q[u]);
        }
        bool above = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                double tv = t[u][v] * (1 + 0.61 * q[u][v]);          // :144
                geo_layer(acc[v], l - u, CON_RD * tv, p[u][v], pref[v]);
            }
        }
        if (!full_column) {
            // a column is finished once its pressure has dropped below p_ref and it has been
            // strictly ascending so far (then no higher half level can satisfy p >= p_ref)
#pragma unroll
            for (int v = 0; v < V; ++v) above = above && (acc[v].p_lo < pref[v]) && (acc[v].kstar >= 0);
            if (__all(above)) { l -= U; goto done; }
        }
    }
    for (; l >= 0; --l) {
        double p[V], t[V], q[V];
        SIG_LOADV<T, V>(ph + (long long)l * ncol, p);
        SIG_LOADV<T, V>(pt + (long long)l * ncol, t);
        SIG_LOADV<T, V>(pq + (long long)l * ncol, q);
#pragma unroll
        for (int v = 0; v < V; ++v) geo_layer(acc[v], l, CON_RD * (t[v] * (1 + 0.61 * q[v])), p[v], pref[v]);
    }
done:
    double r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = geo_finish(acc[v], pref[v], st, c2 + v);
    storev<TO, V>(phi_ref + c2, r);
}

// =====================================================================================
// a5  fused pass of the surface-pressure loop         step_03_apply_to_era.py:192-308
// The vapour pressure e = hur_pgw/100 * e_sat(ta_pgw) (functions.py:123) does not depend on
// the iterate, so it is precomputed once per file (k_humidity_hybrid MODE 2) and each pass
// does q = 0.622 e / (pa - 0.378 e) (:66-72) - the same values the reference recomputes.
// State (delta_ps, adj_ps) and the constant phi_ref_era / dphi_clim are fp64.
// =====================================================================================
// Upward scan of V hybrid-pressure columns from the surface to p_ref: shared by the loop pass
// (second field = vapour pressure e, q = e_to_q(e, pa)) and by phi_ref of the ERA state (second
// field = QV itself).  Levels are processed in chunks of U with the next chunk's 2*U row loads
// already in flight (software pipeline; ~2*U KiB per wave outstanding).
// TL = storage type of the two level arrays.  REF (reference-dtype mode, float32 files): phi rounded to float32 per
// level; for the ERA state (SECOND_IS_Q, float32 T and QV) tav and CON_RD*tav are float32 products.
template <typename TL, int V, int U, bool SECOND_IS_Q, bool REF>
__device__ __forceinline__ void scan_columns(const Levels &lv, const LevTab &lt, long long ncol, const TL *__restrict__ pt,
                                             const TL *__restrict__ pe, const double (&ps)[V], const double (&z)[V],
                                             const double (&pref)[V], int full_column, DevStatus *st, long long c2,
                                             double (&phi_ref)[V], double (&tlow)[V], int &touched) {
    const int N = lv.nlev;
    // the ERA-state scan (SECOND_IS_Q) reads its rows once per file; the loop passes re-read theirs (keep those cacheable)
#define SCAN_LOADV(p, o) do { if constexpr (SECOND_IS_Q) loadv_nt<TL, V>(p, o); else loadv<TL, V>(p, o); } while (0)
    GeoAcc acc[V];
    bool mono[V];
    {
        double akN = lt.ak[N], bkN = lt.bk[N];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            geo_init(acc[v], z[v], akN + ps[v] * bkN, lt.logtab);   // step_03:198
            mono[v] = ps[v] >= lv.ps_mono_min;                  // false for NaN
        }
    }
    double tn[U][V], en[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int lu = (N - 1 - u) > 0 ? (N - 1 - u) : 0;
        SCAN_LOADV(pt + (long long)lu * ncol, tn[u]);
        SCAN_LOADV(pe + (long long)lu * ncol, en[u]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) tlow[v] = tn[0][v];             // ta at the lowest full level, :303
    for (int l = N - 1; l >= 0; l -= U) {
        double t[U][V], e[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < V; ++v) { t[u][v] = tn[u][v]; e[u][v] = en[u][v]; }
        if (l - U >= 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int lu = (l - U - u) > 0 ? (l - U - u) : 0;
                SCAN_LOADV(pt + (long long)lu * ncol, tn[u]);
                SCAN_LOADV(pe + (long long)lu * ncol, en[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int lc = l - u;
            if (lc >= 0) {
                double am = lt.akm[lc], bm = lt.bkm[lc], a = lt.ak[lc], b = lt.bk[lc];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    double rtv;
                    if (SECOND_IS_Q) {
                        if (REF) rtv = rd_tv_f32(t[u][v], e[u][v]);                    // float32 T, QV: float32 products
                        else rtv = CON_RD * (t[u][v] * (1 + 0.61 * e[u][v]));          // functions.py:144
                    } else {
                        // e_to_q (:196, :262-266) with the quotient formed like SharedDivisor's: reciprocal, two Newton steps,
                        // product, one residual correction - the compiler's IEEE sequence without v_div_scale / v_div_fixup
                        // (3 of its 11 instructions), the same bits for every divisor that needs no scaling (here: a pressure
                        // minus a fraction of a vapour pressure, 1e-4 .. 1.1e5 Pa)
                        const double pm = am + ps[v] * bm;
                        double q = SharedDivisor(pm - (1 - CON_MW_MD) * e[u][v]).divide(CON_MW_MD * e[u][v]);
                        rtv = CON_RD * (t[u][v] * (1 + 0.61 * q));
                    }
                    geo_layer<REF, true>(acc[v], lc, rtv, a + ps[v] * b, pref[v], lt.logtab);
                }
                touched += V;
            }
        }
        if (!full_column) {
            bool above = true;
#pragma unroll
            for (int v = 0; v < V; ++v) above = above && mono[v] && (acc[v].p_lo < pref[v]) && (acc[v].kstar >= 0);
            if (__all(above)) break;
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) phi_ref[v] = geo_finish(acc[v], pref[v], st, c2 + v);
}

// step_03:192-193.  REF: delta_ps is a float32 array updated in place (`delta_ps += adj_ps` casts the float64 sum back,
// step_03:182-192) and ps_pgw = PS + delta_ps is a float32 sum; the fp64 state array then holds float32 values.
template <bool REF>
__device__ __forceinline__ double next_delta_ps(double dps, double adj) { return REF ? (double)(float)(dps + adj) : dps + adj; }
template <bool REF>
__device__ __forceinline__ double ps_of(double ps0, double dps) { return REF ? (double)((float)ps0 + (float)dps) : ps0 + dps; }

template <typename T, typename TL, int V, int U, bool REF>
__global__ __launch_bounds__(BLOCK) void k_adjust_ps_step(Levels lv, int ntime, long long ncol,
                                                          const TL *__restrict__ ta, const TL *__restrict__ evap,
                                                          const T *__restrict__ PS, const T *__restrict__ FIS,
                                                          const double *__restrict__ phi_ref_era,
                                                          const double *__restrict__ dphi_clim,
                                                          double *__restrict__ delta_ps, double *__restrict__ adj_ps,
                                                          double p_ref_s, const double *__restrict__ p_ref_f,
                                                          double adj_factor, int full_column, int apply_adj,
                                                          DevStatus *st, DevStatus *clear) {
    __shared__ double s_max[BLOCK / 64];
    __shared__ unsigned int s_valid[BLOCK / 64];
    __shared__ double s_lev[LEVTAB_DOUBLES];
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) {       // status block of the NEXT pass
        DevStatus z;
        z.code = 0; z.nan_seen = 0; z.col = ~0ull; z.max_bits = 0; z.valid = 0;
        z.min_targ_bits = ~0ull; z.min_src_bits = ~0ull; z.levels_touched = 0;
        *clear = z;
    }
    LevTab lt = stage_levels<true, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    double amax = -1.0;       // max |err| of this thread's valid columns (-1 = none)
    int touched = 0;          // full levels read by this thread (x V columns)
    if (g < ngroups) {
        ColIdx ix = col_index(g, V, ncol);
        const int N = lv.nlev;
        long long c2 = ix.t * ncol + ix.c;
        double ps0[V], z[V], dps[V], adj[V], ps[V], pref[V], tlow[V], phi_ref[V];
        loadv<T, V>(PS + c2, ps0);
        loadv<T, V>(FIS + c2, z);
        loadv<double, V>(delta_ps + c2, dps);
        loadv<double, V>(adj_ps + c2, adj);
        if (p_ref_f) loadv<double, V>(p_ref_f + c2, pref);
        else {
#pragma unroll
            for (int v = 0; v < V; ++v) pref[v] = p_ref_s;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (apply_adj) dps[v] = next_delta_ps<REF>(dps[v], adj[v]);   // step_03:192 (already applied by k_local_p_ref otherwise)
            ps[v] = ps_of<REF>(ps0[v], dps[v]);                           // :193
        }
        if (apply_adj) storev<double, V>(delta_ps + c2, dps);
        scan_columns<TL, V, U, false, REF>(lv, lt, ncol, ta + ix.t * N * ncol + ix.c, evap + ix.t * N * ncol + ix.c, ps, z, pref,
                                           full_column, st, c2, phi_ref, tlow, touched);
        double nadj[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double err = (phi_ref[v] - phi_ref_era[c2 + v]) - dphi_clim[c2 + v];      // :289,298
            // :301-304.  REF: `-adj_factor * ps_pgw` is a float32 product (python float x float32 array)
            const double fps = REF ? (double)((float)(-adj_factor) * (float)ps[v]) : -adj_factor * ps[v];
            nadj[v] = fps / (CON_RD * tlow[v]) * err;
            double ae = fabs(err);
            if (ae == ae) amax = fmax(amax, ae);                                      // :308 skipna
        }
        storev<double, V>(adj_ps + c2, nadj);
    }
    // block max (exact: max is order independent) -> one atomic per block
    double wm = wave_max(amax);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) touched += __shfl_xor(touched, off, 64);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_max[w] = wm; s_valid[w] = (unsigned int)touched; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_max[0];
        unsigned long long tch = s_valid[0];
#pragma unroll
        for (int i = 1; i < BLOCK / 64; ++i) { m = fmax(m, s_max[i]); tch += s_valid[i]; }
        if (m >= 0.0) {
            atomicMax(&st->max_bits, dbits(m));
            atomicAdd(&st->valid, 1ull);
        }
        atomicAdd(&st->levels_touched, tch);
    }
}

// phi_ref of the ERA state from (T, QV, PS, FIS) with the hybrid pressure rebuilt in registers
// (step_03:280-287 with pa_hl_era of :64-66): no 4-D pressure array, and like the pass kernel it
// only reads the levels below p_ref.
template <typename T, int V, int U, bool REF>
__global__ __launch_bounds__(BLOCK) void k_phi_ref_hybrid(Levels lv, int ntime, long long ncol,
                                                          const T *__restrict__ ta, const T *__restrict__ hus,
                                                          const T *__restrict__ PS, const T *__restrict__ FIS,
                                                          double p_ref_s, const double *__restrict__ p_ref_f,
                                                          double *__restrict__ phi_out, int full_column, DevStatus *st) {
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<true, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    if (g >= ngroups) return;
    ColIdx ix = col_index(g, V, ncol);
    const int N = lv.nlev;
    long long c2 = ix.t * ncol + ix.c;
    double ps[V], z[V], pref[V], tlow[V], phi_ref[V];
    int touched = 0;
    loadv<T, V>(PS + c2, ps);
    loadv<T, V>(FIS + c2, z);
    if (p_ref_f) loadv<double, V>(p_ref_f + c2, pref);
    else {
#pragma unroll
        for (int v = 0; v < V; ++v) pref[v] = p_ref_s;
    }
    scan_columns<T, V, U, true, REF>(lv, lt, ncol, ta + ix.t * N * ncol + ix.c, hus + ix.t * N * ncol + ix.c, ps, z, pref,
                                     full_column, st, c2, phi_ref, tlow, touched);
    storev<double, V>(phi_out + c2, phi_ref);
}

// final outputs of the loop: ps_pgw = PS + delta_ps (step_03:193,369), hus_pgw from e (:262-266,370)
template <typename T, typename TL, int V, bool REF>
__global__ __launch_bounds__(BLOCK) void k_finalize_ps_hus(Levels lv, int ntime, long long ncol,
                                                           const T *__restrict__ PS, const double *__restrict__ delta_ps,
                                                           const TL *__restrict__ evap, T *__restrict__ ps_out,
                                                           TL *__restrict__ hus_out, int l_start) {
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<false, true>(lv, s_lev, BLOCK);
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    if (g >= ngroups) return;
    ColIdx ix = col_index(g, V, ncol);
    long long c2 = ix.t * ncol + ix.c;
    double ps0[V], dps[V], ps[V];
    loadv<T, V>(PS + c2, ps0);
    loadv<double, V>(delta_ps + c2, dps);
#pragma unroll
    for (int v = 0; v < V; ++v) ps[v] = ps_of<REF>(ps0[v], dps[v]);
    if (ps_out) storev<T, V>(ps_out + c2, ps);
    if (hus_out) {
        const int N = lv.nlev;
        long long base = ix.t * N * ncol + ix.c;
        // e is read for the last time and QV is not read again by this library: streaming loads / stores
#pragma unroll 4
        for (int l = l_start; l < N; ++l) {            // levels < l_start were written by k_delta_quad
            double e[V], r[V];
            loadv_nt<TL, V>(evap + base + (long long)l * ncol, e);
            double am = lt.akm[l], bm = lt.bkm[l];
#pragma unroll
            for (int v = 0; v < V; ++v) r[v] = e_to_q(e[v], am + ps[v] * bm);
            storev_nt<TL, V>(hus_out + base + (long long)l * ncol, r);
        }
    }
}

// =====================================================================================
// a6  interp_logp_4d, signature-faithful                       functions.py:434-580
// Each target level is located by the reference's "first s with src[s] == x or src[s] > x" rule.  The scan
// resumes from the previous hit while targets ascend (all earlier sources are < the previous target <= x, so
// they cannot match) and restarts from 0 otherwise -> identical selection, O(N+S).
// =====================================================================================
// No field is staged (only the 2 KB table of the logarithm).  A column keeps a window of its source profile in registers - levels j-2, j-1, j (the scan
// position) and j+1, with the loads of level j+2 in flight - and moves it forward when a target passes level j, so
// every source element is read exactly once, one scan step ahead of its use, and the kernel runs at full occupancy
// (a [level][thread] LDS tile of the source columns caps the CU at 8 waves at S = 19: 0.91 vs 0.54 ms).
// A non-ascending / NaN target restarts the scan (re-reads the column from level 0; rare).
#ifndef PGW_INTERP_MINB
#define PGW_INTERP_MINB 1
#endif
#define PGW_LOGT(x) pgw_log_tab((x), s_logt)
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK, PGW_INTERP_MINB) void k_interp_logp_stream(int ntime, int S, int N, long long ncol,
                                                              const T *__restrict__ var, const T *__restrict__ srcP,
                                                              const T *__restrict__ trgP, T *__restrict__ out,
                                                              int logp_in, DevStatus *st) {
    // np.log of both pressure fields (:470-471) through the table-driven logarithm of the level loops (pgw_log_tab: ~27
    // instead of ~40 instructions, same <= 1 ulp; S + N = 156 logarithms per column made this kernel issue-bound: 0.86 busy).
    // Source and target logarithms come from the one implementation, so the exact-hit rule (:540) is unaffected.
    __shared__ double s_logt[2 * LOG_TABLE_N];
    stage_log_table(s_logt, BLOCK);
    __syncthreads();
    long long flat = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (flat >= (long long)ntime * ncol) return;
    long long t = flat / ncol, c = flat - t * ncol;
    const T *pv = var + t * S * ncol + c;
    const T *pp = srcP + t * S * ncol + c;
    const T *pt = trgP + t * N * ncol + c;
    T *po = out + t * N * ncol + c;
    const double s_first = logp_in ? (double)pp[0] : PGW_LOGT((double)pp[0]);                         // :470
    {
        double s_last = (double)pp[(long long)(S - 1) * ncol];
        if (!logp_in) s_last = PGW_LOGT(s_last);
        if (s_last < s_first) { report(st, 10, flat); }              // :500-501
        double x_first = (double)pt[0], x_last = (double)pt[(long long)(N - 1) * ncol];
        if (!logp_in) { x_first = PGW_LOGT(x_first); x_last = PGW_LOGT(x_last); }
        if (x_last < x_first) { report(st, 11, flat); }              // :502-503
    }
    // source window
    int j;
    double xmm = 0, ymm = 0, xm = 0, ym = 0, xj, yj, xn, yn, rx, ry;
    auto reset = [&]() {
        j = 0;
        xj = s_first; yj = (double)pv[0];
        xn = (double)pp[ncol]; yn = (double)pv[ncol];                // S >= 2
        if (!logp_in) xn = PGW_LOGT(xn);
        const long long o = (long long)(2 < S ? 2 : S - 1) * ncol;
        rx = (double)SIG_LD(pp + o); ry = (double)SIG_LD(pv + o);
    };
    reset();
    double xprev = -__builtin_inf();
    constexpr int U = 4;            // chunks of 4 target levels: the next chunk's loads are in flight while this one is done
    double nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nx[u] = (double)SIG_LD(pt + (long long)(u < N ? u : N - 1) * ncol);
    for (int l0 = 0; l0 < N; l0 += U) {
        double cx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cx[u] = nx[u];
        if (l0 + U < N) {
#pragma unroll
            for (int u = 0; u < U; ++u) nx[u] = (double)SIG_LD(pt + (long long)((l0 + U + u) < N ? (l0 + U + u) : N - 1) * ncol);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u;
            if (l < N) {
                double x = cx[u];
                if (!logp_in) x = PGW_LOGT(x);                            // :471
                if (__builtin_expect(!(x >= xprev), 0)) {                // restart (descending or NaN target)
                    x = no_speculate(x);
                    reset();
                }
                while (j < S && !(xj == x || xj > x)) {                  // first s with src[s] == x or src[s] > x
                    xmm = xm; ymm = ym; xm = xj; ym = yj; xj = xn; yj = yn;
                    xn = logp_in ? rx : PGW_LOGT(rx); yn = ry;
                    ++j;
                    const long long o = (long long)(j + 2 < S ? j + 2 : S - 1) * ncol;
                    rx = (double)SIG_LD(pp + o); ry = (double)SIG_LD(pv + o);
                }
                bool extrap = false;
                double x1, y1, x2, y2;
                bool same;                                               // i1 == i2: take y1
                if (j >= S) {                                            // above range            :554-561
                    extrap = true;
                    same = (MODE != 1);
                    x1 = (MODE == 1) ? xmm : xm; y1 = (MODE == 1) ? ymm : ym; x2 = xm; y2 = ym;
                } else if (xj == x) {                                    // exact                  :540-543
                    same = true; x1 = x2 = xj; y1 = y2 = yj;
                } else if (j == 0) {                                     // below range            :530-538
                    extrap = true;
                    same = (MODE != 1);
                    x1 = xj; y1 = yj; x2 = xn; y2 = yn;
                } else {                                                 // bracket                :545-548
                    same = false; x1 = xm; y1 = ym; x2 = xj; y2 = yj;
                }
                double y;
                if (extrap && MODE == 3) y = __builtin_nan("");          // :569-570
                else if (same) y = y1;                                   // :572-573
                else y = y1 + (x - x1) * (y2 - y1) / (x2 - x1);          // :575-578
                if (MODE == 0 && extrap) report(st, 12, flat);           // :564-566
                SIG_ST((T)y, po + (long long)l * ncol);
                xprev = (x == x) ? x : __builtin_inf();                  // after a NaN target restart
            }
        }
    }
}

#undef PGW_LOGT
// =====================================================================================
// a7  time lerp of load_delta                                   functions.py:288-292
// =====================================================================================
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_time_lerp(long long n, const T *__restrict__ vb, const T *__restrict__ va,
                                                     double x_hi, double x_new, T *__restrict__ out) {
    long long stride = (long long)gridDim.x * BLOCK;
    for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g * V < n; g += stride) {
        double b[V], a[V], r[V];
        loadv<T, V>(vb + g * V, b);
        loadv<T, V>(va + g * V, a);
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = (a[v] - b[v]) / x_hi * x_new + b[v];
        storev<T, V>(out + g * V, r);
    }
}

// =====================================================================================
// a8  vert_interp_delta fused per column                        functions.py:343-431
// (+ time lerp :288-292, + era + delta step_03:170-173).  The source axis is the 1-D plev
// table (uniform, passed by value) except for the one level replace_delta_sfc moves to
// ps_hist, so nothing is staged: ln(plev) is a scalar table, delta values are gathered from
// the two bracketing records as the scan advances (each source level is read ~once).
// =====================================================================================
constexpr int MAX_PLEV = 64;
struct PlevTable {
    double p[MAX_PLEV];      // ascending-index order (file order reversed, :383-384)
    double lnp[MAX_PLEV];
    double pmax, pmin;
    int n;
};

template <typename T>
struct DeltaSrc {
    const T *b, *a;          // bracketing records (a may be null)
    double x_hi, x_new;
    // REF (reference-dtype mode): scipy's interp1d._call_linear takes y_hi - y_lo in the dtype of the file (float32 deltas:
    // one float32 subtraction), slope and result in float64 (functions.py:288-292 through xarray .interp)
    template <bool REF = false>
    __device__ __forceinline__ double get(long long off) const {
        const T rb = b[off];
        if (!a) return (double)rb;                                // :282-283
        const T ra = a[off];
        const double diff = REF ? (double)(T)(ra - rb) : (double)ra - (double)rb;
        return diff / x_hi * x_new + (double)rb;                  // :288-292 (scipy interp1d linear)
    }
    // get() addressed by byte offset (ld_off)
    template <typename O>
    __device__ __forceinline__ double get_off(O byte_off) const {
        const T rb = ld_off(b, byte_off);
        if (!a) return (double)rb;
        const T ra = ld_off(a, byte_off);
        return ((double)ra - (double)rb) / x_hi * x_new + (double)rb;
    }
    // same, addressed by byte offset (ld_off); the division by the kernel-wide x_hi goes through a reciprocal the
    // caller computed once (SharedDivisor: same quotient bits)
    // LERP is the compile-time form of `a != nullptr` (all records of a file share the instant, so it is one
    // property of the launch): no per-source null test - those tests were uniform 64-bit masks kept in spilled SGPRs
    template <bool LERP, bool REF, typename O>
    __device__ __forceinline__ double get_at(O byte_off, const SharedDivisor &by_x_hi) const {
        const T rb = ld_off(b, byte_off);
        if (!LERP) return (double)rb;
        const T ra = ld_off(a, byte_off);
        const double diff = REF ? (double)(T)(ra - rb) : (double)ra - (double)rb;
        return by_x_hi.divide(diff) * x_new + (double)rb;
    }
};

template <typename T, bool HAS_SFC>
__global__ __launch_bounds__(BLOCK) void k_vert_interp_delta(PlevTable pt, Levels lv, int ntime, int N, long long ncol,
                                                             DeltaSrc<T> dsrc, DeltaSrc<T> sfc, DeltaSrc<T> psh,
                                                             const T *__restrict__ trgP, const T *__restrict__ ps,
                                                             int check_top, const T *__restrict__ add_to,
                                                             T *__restrict__ out, DevStatus *st) {
    __shared__ double s_mint[BLOCK / 64], s_mins[BLOCK / 64];
    __shared__ int s_nan[BLOCK / 64];
    // plev / ln(plev) staged in LDS: lanes index them with their own (divergent) scan position
    __shared__ double s_p[MAX_PLEV], s_lnp[MAX_PLEV];
    __shared__ double s_lev[LEVTAB_DOUBLES];
    // the delta kernels take every logarithm from pgw_log_tab - like k_log_table, which makes ln(plev): the exact-hit test
    // `src_x == targ_x` (functions.py:540) compares values of ONE implementation
    LevTab lt;
    lt.akm = lt.bkm = nullptr;
    lt.logtab = s_lev + 2 * (MAX_NLEV + 1) + 2 * MAX_NLEV;
    if (!trgP) lt = stage_levels<false, true>(lv, s_lev, BLOCK);
    else stage_log_table(s_lev + 2 * (MAX_NLEV + 1) + 2 * MAX_NLEV, BLOCK);
    const int S = pt.n;
    if (threadIdx.x < MAX_PLEV) {
        s_p[threadIdx.x] = pt.p[threadIdx.x];
        s_lnp[threadIdx.x] = pt.lnp[threadIdx.x];
    }
    __syncthreads();
    long long flat = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double min_t = __builtin_inf(), min_s = __builtin_inf();
    int nanflag = 0;
    if (flat < (long long)ntime * ncol) {
        long long t = flat / ncol, c = flat - t * ncol;
        long long c2 = flat;
        long long dbase = t * S * ncol + c;      // delta records are (ntime, S, ncol), file order
        int ksfc = -1;                           // level moved to ps_hist
        bool fill_below = false;
        double d_sfc = 0.0, lnps = 0.0, pshv = 0.0;
        bool bad = false;
        if (HAS_SFC) {
            pshv = psh.get(c2);
            d_sfc = sfc.get(c2);
            if (pshv > pt.pmax) {                                  // :356-359
                ksfc = S - 1;
            } else if (pshv < pt.pmin) {                           // :360-361
                bad = true;
            } else {                                               // :362-365
                for (int i = 0; i < S; ++i) if (pshv > s_p[i]) ksfc = i;
                if (ksfc < 0) bad = true;                          // np.max of empty argwhere
                fill_below = true;
            }
            if (bad) { report(st, 15, flat); ksfc = -1; }
            lnps = pgw_log_tab(pshv, lt.logtab);
        }
        auto srcx = [&](int i) -> double { return (HAS_SFC && i == ksfc) ? lnps : s_lnp[i]; };
        auto srcy = [&](int i) -> double {
            if (HAS_SFC && ksfc >= 0 && (i == ksfc || (fill_below && i > ksfc))) return d_sfc;
            return dsrc.get(dbase + (long long)(S - 1 - i) * ncol);
        };
        if (check_top) {
            // np.min(source_P) over this column (:417)
            for (int i = 0; i < S; ++i) {
                double p = (HAS_SFC && i == ksfc) ? pshv : s_p[i];
                if (p != p) nanflag |= 2; else min_s = fmin(min_s, p);
            }
        }
        double psv = 0.0;
        const T *ptg = nullptr;
        if (trgP) ptg = trgP + t * N * ncol + c; else psv = (double)ps[c2];
        long long obase = t * N * ncol + c;
        int j = 0;
        double xprev = -__builtin_inf();
        int ci = -2;                 // cached bracket index: values y[ci], y[ci+1]
        double y_lo = 0.0, y_hi = 0.0;
        // chunks of 4 target levels: the next chunk's target pressures and addends are in flight while this one is
        // interpolated (one memory latency per chunk instead of per level; each element is touched once: streaming forms)
        constexpr int U = 4;
        double np_[U], na_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long o = (long long)(u < N ? u : N - 1) * ncol;
            np_[u] = ptg ? (double)SIG_LD(ptg + o) : 0.0;
            na_[u] = add_to ? (double)SIG_LD(add_to + obase + o) : 0.0;
        }
        for (int l0 = 0; l0 < N; l0 += U) {
        double cp_[U], ca_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { cp_[u] = np_[u]; ca_[u] = na_[u]; }
        if (l0 + U < N) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long o = (long long)((l0 + U + u) < N ? (l0 + U + u) : N - 1) * ncol;
                if (ptg) np_[u] = (double)SIG_LD(ptg + o);
                if (add_to) na_[u] = (double)SIG_LD(add_to + obase + o);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u;
            if (l >= N) break;
            double p = ptg ? cp_[u] : (lt.akm[l] + psv * lt.bkm[l]);
            if (check_top) { if (p != p) nanflag |= 1; else min_t = fmin(min_t, p); }
            double x = pgw_log_tab(p, lt.logtab);
            if (!(x >= xprev)) j = 0;
            while (j < S) {
                double xs = srcx(j);
                if (xs == x || xs > x) break;
                ++j;
            }
            double y;
            if (j >= S) {                                   // above range, constant :558-560
                y = srcy(S - 1);
            } else {
                double xs = srcx(j);
                if (xs == x) y = srcy(j);                   // exact :540-543
                else if (j == 0) y = srcy(0);               // below range, constant :534-536
                else {
                    if (ci != j - 1) { y_lo = srcy(j - 1); y_hi = srcy(j); ci = j - 1; }
                    double x1 = srcx(j - 1);
                    y = y_lo + (x - x1) * (y_hi - y_lo) / (xs - x1);    // :575-578
                }
            }
            if (add_to) y = ca_[u] + y;                                        // step_03:170-173
            SIG_ST((T)y, out + obase + (long long)l * ncol);
            xprev = (x == x) ? x : __builtin_inf();
        }
        }
    }
    if (check_top) {
        double wt = wave_min(min_t), ws = wave_min(min_s);
        int wn = nanflag;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wn |= __shfl_xor(wn, off, 64);
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_mint[w] = wt; s_mins[w] = ws; s_nan[w] = wn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mt = s_mint[0], ms = s_mins[0];
            int nn = s_nan[0];
            for (int i = 1; i < BLOCK / 64; ++i) { mt = fmin(mt, s_mint[i]); ms = fmin(ms, s_mins[i]); nn |= s_nan[i]; }
            // pressures are compared as ordered bit patterns; negative values (unphysical) map to 0
            if (mt < __builtin_inf()) atomicMin(&st->min_targ_bits, mt > 0 ? dbits(mt) : 0ull);
            if (ms < __builtin_inf()) atomicMin(&st->min_src_bits, ms > 0 ? dbits(ms) : 0ull);
            if (nn) atomicOr(&st->nan_seen, nn);
        }
    }
}

// settings.i_reinterp = 1 (step_03_apply_to_era.py:202-216, 330-343), one variable, one pass, one kernel:
//   out = interp_logp_4d(era_field, pa_era, pa_pgw, 'constant')  +  load_delta_interp(var, pa_pgw)
// with both hybrid pressure fields rebuilt in registers from the two surface pressures (pa_era = akm + ps_era bkm is the
// SOURCE axis of the first term, pa_pgw = akm + ps_pgw bkm the TARGET of both), so neither 4-D pressure array nor the
// re-interpolated ERA field is materialised: reads the ERA field once, writes the result once (composed from the
// function-level entries the same work is two launches and five 4-D passes).  The ERA column is followed with a
// three-value window (level j-1, j and the raw value of j+1 in flight): the two pressure sets differ by the loop's
// surface-pressure increment, so the bracket of target level l lies next to source level l.  Selection rule and lerp
// arithmetic are interp_extrap_1d's (functions.py:527-578, 'constant'); all logarithms from pgw_log_tab.
template <typename T, bool HAS_SFC>
__global__ __launch_bounds__(BLOCK) void k_reinterp_field(PlevTable pt, Levels lv, int ntime, long long ncol,
                                                          DeltaSrc<T> dsrc, DeltaSrc<T> sfc, DeltaSrc<T> psh,
                                                          const T *__restrict__ era_field, const T *__restrict__ ps_era,
                                                          const T *__restrict__ ps_pgw, int check_top,
                                                          T *__restrict__ out, DevStatus *st) {
    __shared__ double s_mint[BLOCK / 64], s_mins[BLOCK / 64];
    __shared__ int s_nan[BLOCK / 64];
    __shared__ double s_p[MAX_PLEV], s_lnp[MAX_PLEV];
    __shared__ double s_lev[LEVTAB_DOUBLES];
    LevTab lt = stage_levels<false, true>(lv, s_lev, BLOCK);
    const int S = pt.n, N = lv.nlev;
    if (threadIdx.x < MAX_PLEV) {
        s_p[threadIdx.x] = pt.p[threadIdx.x];
        s_lnp[threadIdx.x] = pt.lnp[threadIdx.x];
    }
    __syncthreads();
    long long flat = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double min_t = __builtin_inf(), min_s = __builtin_inf();
    int nanflag = 0;
    if (flat < (long long)ntime * ncol) {
        long long t = flat / ncol, c = flat - t * ncol;
        long long c2 = flat;
        long long dbase = t * S * ncol + c;      // delta records are (ntime, S, ncol), file order
        int ksfc = -1;                           // level moved to ps_hist
        bool fill_below = false;
        double d_sfc = 0.0, lnps = 0.0, pshv = 0.0;
        bool bad = false;
        if (HAS_SFC) {
            pshv = psh.get(c2);
            d_sfc = sfc.get(c2);
            if (pshv > pt.pmax) {                                  // functions.py:356-359
                ksfc = S - 1;
            } else if (pshv < pt.pmin) {                           // :360-361
                bad = true;
            } else {                                               // :362-365
                for (int i = 0; i < S; ++i) if (pshv > s_p[i]) ksfc = i;
                if (ksfc < 0) bad = true;
                fill_below = true;
            }
            if (bad) { report(st, 15, flat); ksfc = -1; }
            lnps = pgw_log_tab(pshv, lt.logtab);
        }
        auto srcx = [&](int i) -> double { return (HAS_SFC && i == ksfc) ? lnps : s_lnp[i]; };
        auto srcy = [&](int i) -> double {
            if (HAS_SFC && ksfc >= 0 && (i == ksfc || (fill_below && i > ksfc))) return d_sfc;
            return dsrc.get(dbase + (long long)(S - 1 - i) * ncol);
        };
        if (check_top) {
            for (int i = 0; i < S; ++i) {                          // np.min(source_P) over this column (:417)
                double p = (HAS_SFC && i == ksfc) ? pshv : s_p[i];
                if (p != p) nanflag |= 2; else min_s = fmin(min_s, p);
            }
        }
        const double pse = (double)ps_era[c2], psv = (double)ps_pgw[c2];
        const long long obase = t * (long long)N * ncol + c;
        const T *pf = era_field + obase;
        // ---- window over the ERA column: (wxm, wym) level wj - 1, (wxj, wyj) level wj, and the raw values of levels
        // wj + 1 .. wj + 4 in flight (q0..q3: the window moves one level per target level, four loads stay outstanding)
        int wj;
        double wxm = 0, wym = 0, wxj, wyj, q0, q1, q2, q3;
        auto wload = [&](int lev) -> double { return (double)SIG_LD(pf + (long long)(lev < N ? lev : N - 1) * ncol); };
        auto wreset = [&]() {
            wj = 0;
            wxj = pgw_log_tab(lt.akm[0] + pse * lt.bkm[0], lt.logtab);
            wyj = wload(0);
            q0 = wload(1); q1 = wload(2); q2 = wload(3); q3 = wload(4);
        };
        wreset();
        int j = 0;
        double xprev = -__builtin_inf();
        int ci = -2;                 // cached bracket index of the delta: values y[ci], y[ci+1]
        double y_lo = 0.0, y_hi = 0.0;
        for (int l = 0; l < N; ++l) {
            const double p = lt.akm[l] + psv * lt.bkm[l];                                   // step_03:196-197
            if (check_top) { if (p != p) nanflag |= 1; else min_t = fmin(min_t, p); }
            const double x = pgw_log_tab(p, lt.logtab);
            if (__builtin_expect(!(x >= xprev), 0)) { j = 0; wreset(); }                    // descending / NaN target: both scans restart
            // -- the ERA field at this pressure (interp_extrap_1d, 'constant')
            while (wj < N && !(wxj == x || wxj > x)) {
                wxm = wxj; wym = wyj;
                ++wj;
                if (wj < N) {
                    wxj = pgw_log_tab(lt.akm[wj] + pse * lt.bkm[wj], lt.logtab);
                    wyj = q0; q0 = q1; q1 = q2; q2 = q3;
                    q3 = wload(wj + 4);
                }
            }
            double e;
            if (wj >= N) e = wym;                                   // beyond the last source level: its value   :558-560
            else if (wxj == x || wj == 0) e = wyj;                  // exact :540-543 / before the first: its value :534-536
            else e = wym + (x - wxm) * (wyj - wym) / (wxj - wxm);   // :575-578
            // -- the climate delta at this pressure (as k_vert_interp_delta)
            while (j < S) {
                double xs = srcx(j);
                if (xs == x || xs > x) break;
                ++j;
            }
            double y;
            if (j >= S) {
                y = srcy(S - 1);
            } else {
                double xs = srcx(j);
                if (xs == x) y = srcy(j);
                else if (j == 0) y = srcy(0);
                else {
                    if (ci != j - 1) { y_lo = srcy(j - 1); y_hi = srcy(j); ci = j - 1; }
                    double x1 = srcx(j - 1);
                    y = y_lo + (x - x1) * (y_hi - y_lo) / (xs - x1);
                }
            }
            SIG_ST((T)(e + y), out + obase + (long long)l * ncol);                          // vars_era + deltas  :216
            xprev = (x == x) ? x : __builtin_inf();
        }
    }
    if (check_top) {
        double wt = wave_min(min_t), ws = wave_min(min_s);
        int wn = nanflag;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wn |= __shfl_xor(wn, off, 64);
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_mint[w] = wt; s_mins[w] = ws; s_nan[w] = wn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mt = s_mint[0], ms = s_mins[0];
            int nn = s_nan[0];
            for (int i = 1; i < BLOCK / 64; ++i) { mt = fmin(mt, s_mint[i]); ms = fmin(ms, s_mins[i]); nn |= s_nan[i]; }
            if (mt < __builtin_inf()) atomicMin(&st->min_targ_bits, mt > 0 ? dbits(mt) : 0ull);
            if (ms < __builtin_inf()) atomicMin(&st->min_src_bits, ms > 0 ? dbits(ms) : 0ull);
            if (nn) atomicOr(&st->nan_seen, nn);
        }
    }
}

// settings.i_reinterp = 1, TWO variables that share both pressure axes in one kernel (ta + hur, whose surface insertion
// uses the same ps_hist, or ua + va): per level ONE target logarithm, one source logarithm per window step, one bracket
// search on each axis and one reciprocal per interval serve both variables (k_reinterp_field spends them per variable),
// and the delta records of a bracket are cached like k_delta_quad's (a step to the next source level fetches one level,
// not two; the constant ranges above / below the delta file's levels fetch nothing).  Quotients go through SharedDivisor
// (the compiler's own division steps: same bits as k_reinterp_field's `/` for the finite, normal-range operands here).
//
// The ERA columns are streamed with a STATIC schedule - at target level l the row of level l + 8 is requested and the
// row of level l + 4 (requested four levels earlier) is put into a per-thread LDS ring of 8 levels - and the window that
// follows the source axis reads its values from the ring by level index.  k_reinterp_field moves the window through
// registers inside its data-dependent `while` loop (value rotation q0 <- q1 <- ...): every register copy has to wait for
// the load it copies, so each level exposes a full memory latency (2 TB/s).  Here the loads are consumed by the ring
// writes in issue order with four rows in flight, and the window's `while` loop only touches LDS.  A window position the
// ring does not hold (the two surface pressures more than 3-4 levels apart, or a restart) reads global memory directly.
// T: storage type of the delta records and the surface pressures; TE0 / TE1: of the two ERA fields; TO: of the outputs.
// All equal except in reference-dtype mode on float32 files (T = float): the ERA temperature is the file's float32, RELHUM
// of the ERA state is float64 (functions.py:58-116 with a float64 pressure), and `era + delta` is float64 (step_03:209-216).
template <typename T, typename TE0 = T, typename TE1 = T, typename TO = T>
struct ReinterpPair {
    DeltaSrc<T> d[2], sfc[2];
    const TE0 *era0;
    const TE1 *era1;
    TO *out[2];
    TO *evap;              // EVAP: the vapour pressure of (out[1], out[0]) = (hur_pgw, ta_pgw), what the loop pass reads
};
constexpr int RING = 8, RING_LEAD = 4;

#ifndef RP_MINW_EVAP
#define RP_MINW_EVAP 3     // with the e_sat chain 128 VGPRs spill 116 bytes into the level loop (scratch reloads are vector loads: they drain the row prefetch)
#endif
#ifndef RP_MINW
#define RP_MINW 4
#endif
#ifndef RP_SPARE_ALL
#define RP_SPARE_ALL 0
#endif
// O: byte-offset type of ld_off / st_off (32-bit when every array is smaller than 4 GiB)
// EVAP (ta + hur inside the loop): also writes e = hur_pgw / 100 * e_sat(ta_pgw) (functions.py:123) of the STORED values,
// the iterate-independent half of relative_to_specific_humidity that k_adjust_ps_step reads - the bits of a separate
// k_humidity_hybrid<.., 2> pass over the two outputs, without reading them back
// REF (reference-dtype mode): the roundings numpy / numba put into the reference on float32 files - the record difference of
// the time interpolation in float32 (DeltaSrc::get<REF>), numba's `src_y[i2] - src_y[i1]` of interp_extrap_1d in the dtype
// of the source (float32 for an ERA field of the file, float64 for RELHUM; for the deltas float32 only when the instant is
// a record, functions.py:282-283, 575-578), everything else float64.
template <typename T, bool HAS_SFC, typename O, bool EVAP = false, typename TE0 = T, typename TE1 = T, typename TO = T, bool REF = false>
__global__ __launch_bounds__(BLOCK, (EVAP ? RP_MINW_EVAP : RP_MINW)) void k_reinterp_pair(PlevTable pt, Levels lv, int ntime, long long ncol,
                                                            ReinterpPair<T, TE0, TE1, TO> rv,
                                                            DeltaSrc<T> psh, const T *__restrict__ ps_era,
                                                            const T *__restrict__ ps_pgw, int check_top, DevStatus *st) {
    extern __shared__ double lds_rp[];               // akm[N] | bkm[N]
    __shared__ double s_mint[BLOCK / 64], s_mins[BLOCK / 64];
    __shared__ int s_nan[BLOCK / 64];
    __shared__ double s_p[MAX_PLEV], s_lnp[MAX_PLEV];
    __shared__ double s_logt[2 * LOG_TABLE_N];
    // slot RING: see era_at.  Not for float64 fields without `e` (RP_SPARE): that instantiation runs 4 blocks per CU, and the
    // ninth slot (2 x 2 KB) would cost it one of them.
    constexpr bool SPARE = RP_SPARE_ALL || EVAP || (sizeof(TE0) + sizeof(TE1) < 16);
    __shared__ TE0 s_ring0[RING + (SPARE ? 1 : 0)][BLOCK];
    __shared__ TE1 s_ring1[RING + (SPARE ? 1 : 0)][BLOCK];
    const int S = pt.n, N = lv.nlev;
    double *s_akm = lds_rp, *s_bkm = lds_rp + N;
    stage_log_table(s_logt, BLOCK);
    for (int i = threadIdx.x; i < N; i += BLOCK) { s_akm[i] = lv.akm[i]; s_bkm[i] = lv.bkm[i]; }
    if (threadIdx.x < MAX_PLEV) {
        s_p[threadIdx.x] = pt.p[threadIdx.x];
        s_lnp[threadIdx.x] = pt.lnp[threadIdx.x];
    }
    __syncthreads();
    long long flat = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double min_t = __builtin_inf(), min_s = __builtin_inf();
    int nanflag = 0;
    if (flat < (long long)ntime * ncol) {
        long long t = flat / ncol, c = flat - t * ncol;
        const O row = (O)((unsigned long long)ncol * sizeof(T));
        const O dbase = (O)((unsigned long long)(t * S * ncol + c) * sizeof(T));      // delta records are (ntime, S, ncol), file order
        int ksfc = -1;                                 // level moved to ps_hist
        bool fill_below = false;
        double d_sfc0 = 0.0, d_sfc1 = 0.0, lnps = 0.0, pshv = 0.0;
        if (HAS_SFC) {
            bool bad = false;
            pshv = psh.template get<REF>(flat);
            d_sfc0 = rv.sfc[0].template get<REF>(flat);
            d_sfc1 = rv.sfc[1].template get<REF>(flat);
            if (pshv > pt.pmax) {                                  // functions.py:356-359
                ksfc = S - 1;
            } else if (pshv < pt.pmin) {                           // :360-361
                bad = true;
            } else {                                               // :362-365
                for (int i = 0; i < S; ++i) if (pshv > s_p[i]) ksfc = i;
                if (ksfc < 0) bad = true;
                fill_below = true;
            }
            if (bad) { report(st, 15, flat); ksfc = -1; }
            lnps = pgw_log_tab(pshv, s_logt);
        }
        auto srcx = [&](int i) -> double { return (HAS_SFC && i == ksfc) ? lnps : s_lnp[i]; };
        auto is_sfc = [&](int i) -> bool { return HAS_SFC && ksfc >= 0 && (i == ksfc || (fill_below && i > ksfc)); };
        if (check_top) {
            for (int i = 0; i < S; ++i) {                          // np.min(source_P) over this column (:417)
                double p = (HAS_SFC && i == ksfc) ? pshv : s_p[i];
                if (p != p) nanflag |= 2; else min_s = fmin(min_s, p);
            }
        }
        // delta records of the bracket (ci, ci + 1) of both variables; all loads of a change before the first use
        int ci = -2;
        double a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
        const bool lerp = rv.d[0].a != nullptr;          // one instant for every record of the launch
        const double x_new = rv.d[0].x_new;
        const SharedDivisor by_x_hi(lerp ? rv.d[0].x_hi : 1.0);
        auto tl = [&](T rb, T ra) -> double {             // DeltaSrc::get (functions.py:282-292)
            if (!lerp) return (double)rb;
            const double diff = REF ? (double)(T)(ra - rb) : (double)ra - (double)rb;
            return by_x_hi.divide(diff) * x_new + (double)rb;
        };
        // value differences of the column interpolations (functions.py:575-578) in the dtype numba sees them in
        auto ydiff = [&](double hi, double lo) -> double { return (REF && !lerp) ? (double)((float)hi - (float)lo) : hi - lo; };
        auto ediff0 = [](double hi, double lo) -> double { return (REF && sizeof(TE0) == 4) ? (double)((float)hi - (float)lo) : hi - lo; };
        auto ediff1 = [](double hi, double lo) -> double { return (REF && sizeof(TE1) == 4) ? (double)((float)hi - (float)lo) : hi - lo; };
        auto fetch = [&](int i1) {
            if (ci == i1) return;
            const int ih = (i1 + 1 < S) ? i1 + 1 : i1;
            const O oh = dbase + (O)(S - 1 - ih) * row, ol = dbase + (O)(S - 1 - i1) * row;
            const bool seq = (ci + 1 == i1);
            const bool need_h = !is_sfc(ih), need_l = !seq && !is_sfc(i1);
            double h0 = 0, h1 = 0, l0 = 0, l1 = 0;
            // raw records first, then the time interpolation (the quotient by the launch-wide x_hi through one reciprocal)
            T hb0 = 0, hb1 = 0, ha0 = 0, ha1 = 0, lb0 = 0, lb1 = 0, la0 = 0, la1 = 0;
            if (need_h) { hb0 = ld_off(rv.d[0].b, oh); hb1 = ld_off(rv.d[1].b, oh); if (lerp) { ha0 = ld_off(rv.d[0].a, oh); ha1 = ld_off(rv.d[1].a, oh); } }
            if (need_l) { lb0 = ld_off(rv.d[0].b, ol); lb1 = ld_off(rv.d[1].b, ol); if (lerp) { la0 = ld_off(rv.d[0].a, ol); la1 = ld_off(rv.d[1].a, ol); } }
            if (need_h) { h0 = tl(hb0, ha0); h1 = tl(hb1, ha1); }
            if (need_l) { l0 = tl(lb0, la0); l1 = tl(lb1, la1); }
            if (seq) { a_lo = a_hi; b_lo = b_hi; }
            else { a_lo = need_l ? l0 : d_sfc0; b_lo = need_l ? l1 : d_sfc1; }
            a_hi = need_h ? h0 : d_sfc0;
            b_hi = need_h ? h1 : d_sfc1;
            ci = i1;
        };
        const double pse = (double)ps_era[flat], psv = (double)ps_pgw[flat];
        // byte offsets of the level arrays in units of the first ERA field's element; the other field and the outputs scale
        // them by the ratio of the element sizes (1 unless the types differ: reference-dtype mode)
        const O erow = (O)((unsigned long long)ncol * sizeof(TE0));
        const O obase = (O)((unsigned long long)(t * (long long)N * ncol + c) * sizeof(TE0));
        static_assert(sizeof(TE1) % sizeof(TE0) == 0 && sizeof(TO) % sizeof(TE0) == 0, "element sizes");
        auto off1 = [](O o) -> O { return o * (O)(sizeof(TE1) / sizeof(TE0)); };
        auto offo = [](O o) -> O { return o * (O)(sizeof(TO) / sizeof(TE0)); };
        const TE0 *pf0 = rv.era0;
        const TE1 *pf1 = rv.era1;
        auto lev_off = [&](int lev) -> O { return obase + (O)(lev < N ? lev : N - 1) * erow; };
        // ---- the ring: at target level l it holds the ERA levels [l - 3, l + 5)
        const int tid = threadIdx.x;
        int ring_lo = 0;                                               // l - 3 (may be negative)
        // A level the ring does not hold (the two surface pressures more than 3-4 levels apart, or a restart) is fetched from
        // global memory INTO the thread's spare ring slot and read from there like any other: the window's values then never
        // depend on a vector-memory load outside this branch.  (Returned in registers, the compiler had to put
        // `s_waitcnt vmcnt(0)` at the head of the window loop - where the two paths merge - which drained the eight prefetched
        // rows and every store in flight at almost every level: 0.52 of the wave time was spent in s_waitcnt.)
        auto era_at = [&](int lev, double &u, double &v) {
            if constexpr (SPARE) {
                int slot = lev & (RING - 1);
                if (__builtin_expect(!((unsigned)(lev - ring_lo) < (unsigned)RING), 0)) {
                    const O o = lev_off(lev);
                    s_ring0[RING][tid] = ld_off(pf0, o);
                    s_ring1[RING][tid] = ld_off(pf1, off1(o));
                    slot = RING;
                }
                u = (double)s_ring0[slot][tid]; v = (double)s_ring1[slot][tid];
            } else {
                if ((unsigned)(lev - ring_lo) < (unsigned)RING) { u = (double)s_ring0[lev & (RING - 1)][tid]; v = (double)s_ring1[lev & (RING - 1)][tid]; }
                else { const O o = lev_off(lev); u = (double)ld_off(pf0, o); v = (double)ld_off(pf1, off1(o)); }
            }
        };
        TE0 na[RING_LEAD];                                              // rows in flight: levels l + 4 .. l + 7 at the top of level l
        TE1 nb[RING_LEAD];
        {
            TE0 ia[RING_LEAD];
            TE1 ib[RING_LEAD];
#pragma unroll
            for (int u = 0; u < RING_LEAD; ++u) { ia[u] = ld_off_nt(pf0, lev_off(u)); ib[u] = ld_off_nt(pf1, off1(lev_off(u))); }
#pragma unroll
            for (int u = 0; u < RING_LEAD; ++u) { na[u] = ld_off_nt(pf0, lev_off(RING_LEAD + u)); nb[u] = ld_off_nt(pf1, off1(lev_off(RING_LEAD + u))); }
#pragma unroll
            for (int u = 0; u < RING_LEAD; ++u) { s_ring0[u][tid] = ia[u]; s_ring1[u][tid] = ib[u]; }
        }
        // ---- window over the source axis: level wj - 1 (wxm, um, vm) and level wj (wxj, uj, vj)
        int wj;
        double wxm = 0, wxj, um = 0, uj, vm = 0, vj;
        auto wreset = [&]() {
            wj = 0;
            wxj = pgw_log_tab(s_akm[0] + pse * s_bkm[0], s_logt);
            era_at(0, uj, vj);
        };
        int j = 0;
        double xprev = -__builtin_inf();
        int wdiv = -1, ddiv = -2;                       // window position by_W / delta bracket by_D belong to
        SharedDivisor by_W(1.0, 1.0), by_D(1.0, 1.0);
        double dx1 = 0.0;
        for (int l0 = 0; l0 < N; l0 += RING_LEAD) {
#pragma unroll
            for (int u = 0; u < RING_LEAD; ++u) {
                const int l = l0 + u;
                if (l < N) {
                    // row l + 4 into the ring (it replaces level l - 4), row l + 8 requested
                    s_ring0[(l + RING_LEAD) & (RING - 1)][tid] = na[u];
                    s_ring1[(l + RING_LEAD) & (RING - 1)][tid] = nb[u];
                    ring_lo = l + RING_LEAD + 1 - RING;
                    { const O o = lev_off(l + 2 * RING_LEAD); na[u] = ld_off_nt(pf0, o); nb[u] = ld_off_nt(pf1, off1(o)); }
                    if (l == 0) wreset();
                    const double p = s_akm[l] + psv * s_bkm[l];                                 // step_03:196-197
                    if (check_top) { if (p != p) nanflag |= 1; else min_t = fmin(min_t, p); }
                    const double x = pgw_log_tab(p, s_logt);
                    if (__builtin_expect(!(x >= xprev), 0)) { j = 0; wreset(); wdiv = -1; }     // descending / NaN target: both scans restart
                    // -- the ERA fields at this pressure (interp_extrap_1d, 'constant')
                    while (wj < N && !(wxj == x || wxj > x)) {
                        wxm = wxj; um = uj; vm = vj;
                        ++wj;
                        if (wj < N) {
                            wxj = pgw_log_tab(s_akm[wj] + pse * s_bkm[wj], s_logt);
                            era_at(wj, uj, vj);
                        }
                    }
                    double e0, e1;
                    if (wj >= N) { e0 = um; e1 = vm; }                      // beyond the last source level: its value   :558-560
                    else if (wxj == x || wj == 0) { e0 = uj; e1 = vj; }     // exact :540-543 / before the first: its value :534-536
                    else {                                                  // :575-578
                        if (wdiv != wj) { by_W = SharedDivisor(wxj - wxm); wdiv = wj; }
                        const double dx = x - wxm;
                        e0 = um + by_W.divide(dx * ediff0(uj, um));
                        e1 = vm + by_W.divide(dx * ediff1(vj, vm));
                    }
                    // -- the climate deltas at this pressure (as k_vert_interp_delta)
                    while (j < S) {
                        double xs = srcx(j);
                        if (xs == x || xs > x) break;
                        ++j;
                    }
                    int i1, i2;
                    if (j >= S) { i1 = i2 = S - 1; }
                    else {
                        double xs = srcx(j);
                        if (xs == x) { i1 = i2 = j; }
                        else if (j == 0) { i1 = i2 = 0; }
                        else { i1 = j - 1; i2 = j; }
                    }
                    fetch(i1);
                    double y0 = a_lo, y1 = b_lo;
                    if (i1 != i2) {
                        if (ddiv != i1) { dx1 = srcx(i1); by_D = SharedDivisor(srcx(i2) - dx1); ddiv = i1; }   // once per delta bracket
                        const double x1 = dx1;
                        const double dx = x - x1;
                        y0 = a_lo + by_D.divide(dx * ydiff(a_hi, a_lo));
                        y1 = b_lo + by_D.divide(dx * ydiff(b_hi, b_lo));
                    }
                    const O o = offo(obase + (O)l * erow);
                    const TO r0 = (TO)(e0 + y0), r1 = (TO)(e1 + y1);                             // vars_era + deltas  :216
                    st_off_nt(rv.out[0], o, r0);
                    st_off_nt(rv.out[1], o, r1);
                    if (EVAP) st_off(rv.evap, o, (TO)rh_to_e((double)r1, (double)r0));
                    xprev = x;           // a NaN target leaves NaN here: the next level's `!(x >= xprev)` restarts the scans, as +inf did
                }
            }
        }
    }
    if (check_top) {
        double wt = wave_min(min_t), ws = wave_min(min_s);
        int wn = nanflag;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wn |= __shfl_xor(wn, off, 64);
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_mint[w] = wt; s_mins[w] = ws; s_nan[w] = wn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mt = s_mint[0], ms = s_mins[0];
            int nn = s_nan[0];
            for (int i = 1; i < BLOCK / 64; ++i) { mt = fmin(mt, s_mint[i]); ms = fmin(ms, s_mins[i]); nn |= s_nan[i]; }
            if (mt < __builtin_inf()) atomicMin(&st->min_targ_bits, mt > 0 ? dbits(mt) : 0ull);
            if (ms < __builtin_inf()) atomicMin(&st->min_src_bits, ms > 0 ? dbits(ms) : 0ull);
            if (nn) atomicOr(&st->nan_seen, nn);
        }
    }
}

// =====================================================================================
// Fused per-file delta kernels (production path of pgw_step03_file).
//
// k_delta_pair<THERMO=true>:  ta + hur   (step_03:91-94 RELHUM of the ERA state, functions.py:306-431
//     for both variables incl. replace_delta_sfc, step_03:170-173 add, and the iterate-independent
//     vapour pressure e = hur_pgw/100 * e_sat(ta_pgw) of functions.py:123) in ONE pass:
//     reads T, QV (+ S-level delta records), writes T_pgw and e_pgw.  RELHUM / hur_pgw are never
//     materialised (the reference deletes RELHUM before writing, step_03:373).
// k_delta_pair<THERMO=false>: ua + va    (no surface insertion): reads U, V, writes U_pgw, V_pgw.
//
// Both variables of a pair see the same target ln p and (because tas/hurs share ps_hist) the same
// modified source axis, so one log and one bracket search per level serve both.
// V adjacent columns per thread (16 B per lane); level loads are software-pipelined 2 deep.
// =====================================================================================
struct ColScan {
    int ksfc;          // source level moved to ps_hist (-1: none)
    double lnps;       // ln(ps_hist)
    int j;             // current "first source index with sx >= x"
    double xprev;
};

template <typename T>
struct PairSrc {       // the two variables of a pair
    DeltaSrc<T> a, b;
};

// Source values are gathered from global memory when a column's bracket changes and cached in registers
// (staging the S source values of every column in LDS, 304 B per column at S = 19, capped the CU at 7 waves and
// measured 25 % slower: 2.44 vs 1.85 ms).
// THERMO: 4 waves/SIMD (VGPR <= 128) with 2-level chunks measured 6 % faster than 3 waves with 4-level
// chunks (fp64-VALU/latency bound); the wind pair is HBM bound and prefers the deeper prefetch.
template <typename T, int V, bool THERMO, int U, int TPB>
__global__ __launch_bounds__(TPB, THERMO ? 4 : 1) void k_delta_pair(PlevTable pt, Levels lv, int ntime, long long ncol,
                                                    const T *__restrict__ fa, const T *__restrict__ fb,
                                                    const T *__restrict__ PS,
                                                    PairSrc<T> d3, PairSrc<T> dsfc, DeltaSrc<T> psh,
                                                    int check_top, T *__restrict__ out_a, T *__restrict__ out_b,
                                                    T *__restrict__ out_hur, DevStatus *st) {
    extern __shared__ double lds_pair[];            // akm[N] | bkm[N]
    __shared__ double s_mint[TPB / 64], s_mins[TPB / 64];
    __shared__ int s_nan[TPB / 64];
    __shared__ double s_lnp[MAX_PLEV];
    __shared__ double s_logt[2 * LOG_TABLE_N];
    const int S = pt.n;
    double *s_akm = lds_pair, *s_bkm = s_akm + lv.nlev;
    stage_log_table(s_logt, TPB);
    for (int i = threadIdx.x; i < MAX_PLEV; i += TPB) s_lnp[i] = pt.lnp[i];
    for (int i = threadIdx.x; i < lv.nlev; i += TPB) {
        s_akm[i] = lv.akm[i];
        s_bkm[i] = lv.bkm[i];
    }
    __syncthreads();
    long long g = (long long)blockIdx.x * TPB + threadIdx.x;
    long long ngroups = (long long)ntime * ncol / V;
    double min_t = __builtin_inf(), min_s = __builtin_inf();
    int nanflag = 0;
    if (g < ngroups) {
        ColIdx ix = col_index(g, V, ncol);
        const int N = lv.nlev;
        long long c2 = ix.t * ncol + ix.c;
        long long dbase = ix.t * S * ncol + ix.c;     // delta records (ntime, S, ncol), file order
        long long base = ix.t * N * ncol + ix.c;
        double ps[V];
        loadv<T, V>(PS + c2, ps);
        ColScan sc[V];
        bool fillv[V];
        double sfav[V], sfbv[V];
        int ci[V];                                    // cached bracket (levels ci, ci+1)
        double ca_lo[V], ca_hi[V], cb_lo[V], cb_hi[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            sc[v].ksfc = -1; sc[v].lnps = 0.0; sc[v].j = 0; sc[v].xprev = -__builtin_inf();
            ci[v] = -2; ca_lo[v] = ca_hi[v] = cb_lo[v] = cb_hi[v] = 0.0;
            bool fill = false;
            double sfa = 0.0, sfb = 0.0, pshv = 0.0;
            if (THERMO) {
                pshv = psh.get(c2 + v);
                sfa = dsfc.a.get(c2 + v);
                sfb = dsfc.b.get(c2 + v);
                bool bad = false;
                if (pshv > pt.pmax) sc[v].ksfc = S - 1;                        // functions.py:356-359
                else if (pshv < pt.pmin) bad = true;                          // :360-361
                else {                                                        // :362-365
                    for (int i = 0; i < S; ++i) if (pshv > pt.p[i]) sc[v].ksfc = i;     // uniform index: scalar loads
                    if (sc[v].ksfc < 0) bad = true;
                    fill = true;
                }
                if (bad) { report(st, 15, c2 + v); sc[v].ksfc = -1; fill = false; }
                sc[v].lnps = pgw_log_tab(pshv, s_logt);
            }
            fillv[v] = fill; sfav[v] = sfa; sfbv[v] = sfb;
            if (check_top) {                                                  // np.min(source_P), :417
                for (int i = 0; i < S; ++i) {
                    double p = (THERMO && i == sc[v].ksfc) ? pshv : pt.p[i];
                    if (p != p) nanflag |= 2; else min_s = fmin(min_s, p);
                }
            }
        }
        // values of source levels (i, i+1) of column v, from the register cache or global memory (ascending order i <-> file index S-1-i)
        auto fetch = [&](int v, int i1, int i2, double &a1, double &b1, double &a2, double &b2) {
            auto one = [&](int i, double &a, double &b) {
                bool sfc = THERMO && sc[v].ksfc >= 0 && (i == sc[v].ksfc || (fillv[v] && i > sc[v].ksfc));
                long long o = dbase + v + (long long)(S - 1 - i) * ncol;
                a = sfc ? sfav[v] : d3.a.get(o);
                b = sfc ? sfbv[v] : d3.b.get(o);
            };
            if (ci[v] != i1) {
                if (ci[v] + 1 == i1) { ca_lo[v] = ca_hi[v]; cb_lo[v] = cb_hi[v]; }
                else one(i1, ca_lo[v], cb_lo[v]);
                int ih = (i1 + 1 < S) ? i1 + 1 : i1;
                one(ih, ca_hi[v], cb_hi[v]);
                ci[v] = i1;
            }
            a1 = ca_lo[v]; b1 = cb_lo[v];
            if (i2 != i1) { a2 = ca_hi[v]; b2 = cb_hi[v]; } else { a2 = a1; b2 = b1; }
        };
        auto srcx = [&](int v, int i) -> double {          // unconditional LDS read + select, as in k_delta_quad
            double x = s_lnp[i]; asm("" : "+v"(x)); return (THERMO && i == sc[v].ksfc) ? sc[v].lnps : x; };
        // software pipeline: chunks of U levels; the next chunk's 2*U row loads are in flight while
        // the current chunk is processed
        double na[U][V], nb[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int lu = u < N ? u : N - 1;
            loadv<T, V>(fa + base + (long long)lu * ncol, na[u]);
            loadv<T, V>(fb + base + (long long)lu * ncol, nb[u]);
        }
        for (int l0 = 0; l0 < N; l0 += U) {
            double ca[U][V], cb[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) { ca[u][v] = na[u][v]; cb[u][v] = nb[u][v]; }
            if (l0 + U < N) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int lu = (l0 + U + u) < N ? (l0 + U + u) : N - 1;
                    loadv<T, V>(fa + base + (long long)lu * ncol, na[u]);
                    loadv<T, V>(fb + base + (long long)lu * ncol, nb[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = l0 + u;
                if (l < N) {
                    double ra[V], rb[V], rh[V];
                    double am = s_akm[l], bm = s_bkm[l];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        double pa = am + ps[v] * bm;                                   // step_03:87-88
                        if (check_top) { if (pa != pa) nanflag |= 1; else min_t = fmin(min_t, pa); }
                        double x = pgw_log_tab(pa, s_logt);                           // functions.py:471
                        ColScan &c = sc[v];
                        if (!(x >= c.xprev)) c.j = 0;
                        while (c.j < S) {
                            double xs = srcx(v, c.j);
                            if (xs == x || xs > x) break;
                            ++c.j;
                        }
                        c.xprev = (x == x) ? x : __builtin_inf();
                        // i1 == i2: single value; else bracket (i1, i2 = i1 + 1)
                        int i1, i2;
                        if (c.j >= S) { i1 = i2 = S - 1; }                            // above range, constant :558-560
                        else {
                            double xs = srcx(v, c.j);
                            if (xs == x) { i1 = i2 = c.j; }                           // exact                 :540-543
                            else if (c.j == 0) { i1 = i2 = 0; }                       // below range, constant :534-536
                            else { i1 = c.j - 1; i2 = c.j; }                          // bracket               :545-548
                        }
                        double a1, b1, a2 = 0.0, b2 = 0.0;
                        fetch(v, i1, i2, a1, b1, a2, b2);
                        double da = a1, db = b1;
                        if (i1 != i2) {                                               // :575-578
                            double x1 = srcx(v, i1), x2 = srcx(v, i2);
                            const double dx = x - x1;
                            const SharedDivisor by_Dx(x2 - x1);                       // x1 < x < x2: finite, positive
                            da = a1 + by_Dx.divide(dx * (a2 - a1));
                            db = b1 + by_Dx.divide(dx * (b2 - b1));
                        }
                        if (THERMO) {
                            double rh_era = q_to_rh(cb[u][v], pa, ca[u][v]);           // step_03:91-94
                            double ta_pgw = ca[u][v] + da;                             // step_03:170-173
                            double hur_pgw = rh_era + db;
                            ra[v] = ta_pgw;
                            rb[v] = rh_to_e(hur_pgw, ta_pgw);                          // functions.py:123
                            rh[v] = hur_pgw;
                        } else {
                            ra[v] = ca[u][v] + da;
                            rb[v] = cb[u][v] + db;
                        }
                    }
                    storev<T, V>(out_a + base + (long long)l * ncol, ra);
                    storev<T, V>(out_b + base + (long long)l * ncol, rb);
                    if (THERMO && out_hur) storev<T, V>(out_hur + base + (long long)l * ncol, rh);
                }
            }
        }
    }
    if (check_top) {
        double wt = wave_min(min_t), ws = wave_min(min_s);
        int wn = nanflag;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wn |= __shfl_xor(wn, off, 64);
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_mint[w] = wt; s_mins[w] = ws; s_nan[w] = wn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mt = s_mint[0], ms = s_mins[0];
            int nn = s_nan[0];
            for (int i = 1; i < TPB / 64; ++i) { mt = fmin(mt, s_mint[i]); ms = fmin(ms, s_mins[i]); nn |= s_nan[i]; }
            if (mt < __builtin_inf()) atomicMin(&st->min_targ_bits, mt > 0 ? dbits(mt) : 0ull);
            if (ms < __builtin_inf()) atomicMin(&st->min_src_bits, ms > 0 ? dbits(ms) : 0ull);
            if (nn) atomicOr(&st->nan_seen, nn);
        }
    }
}

// -------------------------------------------------------------------------------------
// k_delta_quad: ta + hur AND ua + va in one pass (production kernel of pgw_step03_file).
// The ta+hur pair is bound by fp64 VALU issue and leaves ~60 % of the HBM bandwidth idle; the
// ua+va pair is HBM bound and leaves the VALU mostly idle.  In one kernel the two overlap
// wave by wave (the memory pipe streams U, V while other waves compute e_sat), one ln(p) per
// level serves all four variables, and the level-loop overhead is paid once.
// Two source axes are scanned per column: the surface-modified one (ta, hur) and the plain
// plev axis (ua, va); both use the reference's "first s with sx == x or sx > x" rule.
// One column per thread, source values gathered on bracket change and cached in registers.
// -------------------------------------------------------------------------------------
#ifndef QUAD_MINW
#define QUAD_MINW 3
#endif
#ifndef QUAD_MINW_F32
#define QUAD_MINW_F32 4
#endif
// TO = storage type of the 4-D outputs.  REF (reference-dtype mode; T = float, TO = double): what numpy's promotion
// computes on float32 files - RELHUM of the ERA state through the float32 e_sat chain
// (q_to_rh_f32), float32 record differences in the time interpolation, era (float32) + delta (float64) = float64 outputs.
// streaming forms for what this kernel touches once (ERA fields in; U, V and the pure-level QV out); T_pgw and e are re-read by the loop
#define QLD ld_off_nt
#define QST st_off_nt
#define QST2 st_off
template <typename T, typename TO, int U, int TPB, typename O, bool LERP, bool REF>
__global__ __launch_bounds__(TPB, (sizeof(T) == 4 ? QUAD_MINW_F32 : QUAD_MINW)) void k_delta_quad(PlevTable pt, Levels lv, int ntime, long long ncol,
                                                       const T *__restrict__ fT, const T *__restrict__ fQ,
                                                       const T *__restrict__ fU, const T *__restrict__ fV,
                                                       const T *__restrict__ PS,
                                                       PairSrc<T> dth, PairSrc<T> dsfc, DeltaSrc<T> psh, PairSrc<T> dw,
                                                       int check_top, TO *__restrict__ oT, TO *__restrict__ oE,
                                                       TO *__restrict__ oHur, TO *__restrict__ oU, TO *__restrict__ oV,
                                                       TO *__restrict__ oQ, int n_pure, int n_pure_lv, DevStatus *st) {
    // n_pure > 0: the first n_pure full levels are pure-pressure levels (bkm == 0): their pressure does not depend
    // on the surface pressure, so the final QV = e_to_q(e, akm) (step_03:262-266,370) is written here already
    // (instead of e, which only the levels below p_ref and k_finalize_ps_hus need) and the finalize kernel skips them.
    // n_pure_lv: number of leading pure-pressure levels (n_pure is 0 when the QV shortcut is off); their pressure akm[l]
    // is the same in every column, so ln(akm[l]) is taken once per block instead of once per column and level
    extern __shared__ double lds_quad[];            // akm[N] | bkm[N] | ln(akm)[N] (first n_pure_lv entries)
    __shared__ double s_mint[TPB / 64], s_mins[TPB / 64];
    __shared__ int s_nan[TPB / 64];
    __shared__ double s_lnp[MAX_PLEV];
    __shared__ double s_logt[2 * LOG_TABLE_N];      // table of pgw_log_tab: every logarithm of the delta kernels (see k_vert_interp_delta)
    const int S = pt.n;
    double *s_akm = lds_quad, *s_bkm = lds_quad + lv.nlev, *s_lnpa = lds_quad + 2 * lv.nlev;
    stage_log_table(s_logt, TPB);
    for (int i = threadIdx.x; i < MAX_PLEV; i += TPB) s_lnp[i] = pt.lnp[i];
    for (int i = threadIdx.x; i < lv.nlev; i += TPB) { s_akm[i] = lv.akm[i]; s_bkm[i] = lv.bkm[i]; }
    __syncthreads();
    for (int i = threadIdx.x; i < n_pure_lv; i += TPB) s_lnpa[i] = pgw_log_tab(s_akm[i], s_logt);
    __syncthreads();
    long long flat = (long long)blockIdx.x * TPB + threadIdx.x;
    double min_t = __builtin_inf(), min_s = __builtin_inf();
    int nanflag = 0;
    // every delta record of a file is interpolated to the same instant: ONE (x_hi, x_new) pair and one reciprocal
    // instead of the seven copies in the argument structs (28 SGPRs; the kernel was spilling scalars to VGPR lanes)
    const double x_hi = dth.a.x_hi, x_new = dth.a.x_new;
    const SharedDivisor by_x_hi(x_hi);                      // unused (a == nullptr) when the instant is a record
    const DeltaSrc<T> sTa{dth.a.b, dth.a.a, x_hi, x_new}, sHur{dth.b.b, dth.b.a, x_hi, x_new};
    const DeltaSrc<T> sUa{dw.a.b, dw.a.a, x_hi, x_new}, sVa{dw.b.b, dw.b.a, x_hi, x_new};
    if (flat < (long long)ntime * ncol) {
        const int N = lv.nlev;
        long long t = flat / ncol, c = flat - t * ncol;
        // byte offsets (type O, see ld_off): delta records (ntime, S, ncol), fields (ntime, N, ncol)
        const O row = (O)((unsigned long long)ncol * sizeof(T));
        const O dbase = (O)((unsigned long long)(t * S * ncol + c) * sizeof(T));
        const O base = (O)((unsigned long long)(t * N * ncol + c) * sizeof(T));
        const O orow = (O)((unsigned long long)ncol * sizeof(TO));          // outputs: their own element size
        const O obase = (O)((unsigned long long)(t * N * ncol + c) * sizeof(TO));
        const double ps = (double)PS[flat];
        const bool ps_finite = __builtin_fabs(ps) <= 1.7976931348623157e308;   // false for NaN, +-inf
        // ---- surface insertion for ta / hur (replace_delta_sfc, functions.py:343-366)
        int ksfc = -1;
        bool fill = false;
        double pshv = psh.template get<REF>(flat), sfa = dsfc.a.template get<REF>(flat), sfb = dsfc.b.template get<REF>(flat);
        {
            bool bad = false;
            if (pshv > pt.pmax) ksfc = S - 1;                                  // :356-359
            else if (pshv < pt.pmin) bad = true;                              // :360-361
            else {                                                            // :362-365
                for (int i = 0; i < S; ++i) if (pshv > pt.p[i]) ksfc = i;
                if (ksfc < 0) bad = true;
                fill = true;
            }
            if (bad) { report(st, 15, flat); ksfc = -1; fill = false; }
        }
        const double lnps = pgw_log_tab(pshv, s_logt);
        if (check_top) {                                                      // np.min(source_P), :417
            for (int i = 0; i < S; ++i) {
                double p = (i == ksfc) ? pshv : pt.p[i];
                if (p != p) nanflag |= 2; else min_s = fmin(min_s, p);
            }
        }
        // modified axis: unconditional LDS read + select (the compiler otherwise wraps the read in a divergent branch;
        // the empty asm keeps the load where it is; 8 of 82 exec-mask regions of the loop gone, 2.28 -> 2.25 ms)
        auto sx1 = [&](int i) -> double { double v = s_lnp[i]; asm("" : "+v"(v)); return (i == ksfc) ? lnps : v; };
        auto is_sfc = [&](int i) -> bool { return ksfc >= 0 && (i == ksfc || (fill && i > ksfc)); };
        // register caches: source levels (ci, ci+1) of each pair
        int ci1 = -2, ci2 = -2;
        double a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;       // ta, hur
        double c_lo = 0, c_hi = 0, d_lo = 0, d_hi = 0;       // ua, va
        // A bracket change gathers the records of the new source level(s) from global memory.  All loads of one change are
        // issued back to back BEFORE the first use (raw values into registers, then the time interpolation): one exposed
        // memory latency per change instead of one per value (the first version waited after every record pair - 2 to 4
        // serial round trips per change, the largest part of the kernel's 0.51 s_waitcnt share).
        auto tl = [&](const DeltaSrc<T> &sv, T rb, T ra) -> double {          // load_delta's time interpolation of one value
            if (!LERP) return (double)rb;
            const double diff = REF ? (double)(T)(ra - rb) : (double)ra - (double)rb;
            return by_x_hi.divide(diff) * sv.x_new + (double)rb;
        };
        auto off_of = [&](int i) -> O { return dbase + (O)(S - 1 - i) * row; };
        auto fetch1 = [&](int i1) {
            if (ci1 == i1) return;
            const int ih = (i1 + 1 < S) ? i1 + 1 : i1;
            const O oh = off_of(ih);
            const bool seq = (ci1 + 1 == i1);
            T hb0 = ld_off(sTa.b, oh), hb1 = ld_off(sHur.b, oh), ha0 = 0, ha1 = 0, lb0 = 0, lb1 = 0, la0 = 0, la1 = 0;
            if (LERP) { ha0 = ld_off(sTa.a, oh); ha1 = ld_off(sHur.a, oh); }
            if (!seq) {
                const O o = off_of(i1);
                lb0 = ld_off(sTa.b, o); lb1 = ld_off(sHur.b, o);
                if (LERP) { la0 = ld_off(sTa.a, o); la1 = ld_off(sHur.a, o); }
            }
            if (seq) { a_lo = a_hi; b_lo = b_hi; }
            else { a_lo = is_sfc(i1) ? sfa : tl(sTa, lb0, la0); b_lo = is_sfc(i1) ? sfb : tl(sHur, lb1, la1); }
            a_hi = is_sfc(ih) ? sfa : tl(sTa, hb0, ha0);
            b_hi = is_sfc(ih) ? sfb : tl(sHur, hb1, ha1);
            ci1 = i1;
        };
        // plain-axis bracket (ci2, ci2 + 1): its lower abscissa and the reciprocal of its ln-pressure interval, taken when the
        // bracket changes instead of at every level (2 LDS reads, v_rcp_f64 and 4 FMAs per level less: 1.60 -> 1.55 ms same
        // box).  float32 storage only: the float64 instantiation sits at its 3-wave register budget (164 of 168 VGPRs; with
        // the six more live registers it spills) and its time is its bytes.
        double px1 = 0.0;
        SharedDivisor by_Dp(1.0, 1.0);
        constexpr bool HOIST_DP = (sizeof(T) == 4);
        auto plain_bracket = [&](int i1, int ih) { if (HOIST_DP) { px1 = s_lnp[i1]; by_Dp = SharedDivisor(s_lnp[ih] - px1); } };
        auto fetch2 = [&](int i1) {
            if (ci2 == i1) return;
            const int ih = (i1 + 1 < S) ? i1 + 1 : i1;
            const O oh = off_of(ih);
            const bool seq = (ci2 + 1 == i1);
            plain_bracket(i1, ih);
            T hb0 = ld_off(sUa.b, oh), hb1 = ld_off(sVa.b, oh), ha0 = 0, ha1 = 0, lb0 = 0, lb1 = 0, la0 = 0, la1 = 0;
            if (LERP) { ha0 = ld_off(sUa.a, oh); ha1 = ld_off(sVa.a, oh); }
            if (!seq) {
                const O o = off_of(i1);
                lb0 = ld_off(sUa.b, o); lb1 = ld_off(sVa.b, o);
                if (LERP) { la0 = ld_off(sUa.a, o); la1 = ld_off(sVa.a, o); }
            }
            if (seq) { c_lo = c_hi; d_lo = d_hi; }
            else { c_lo = tl(sUa, lb0, la0); d_lo = tl(sVa, lb1, la1); }
            c_hi = tl(sUa, hb0, ha0);
            d_hi = tl(sVa, hb1, ha1);
            ci2 = i1;
        };
        // both axes step to the same bracket (they coincide above the surface insertion): the records of all four
        // variables in one batch
        auto fetch12 = [&](int i1) {
            if (!(ci1 + 1 == i1 && ci2 + 1 == i1)) { fetch2(i1); fetch1(i1); return; }
            const int ih = (i1 + 1 < S) ? i1 + 1 : i1;
            const O oh = off_of(ih);
            T b0 = ld_off(sTa.b, oh), b1 = ld_off(sHur.b, oh), b2 = ld_off(sUa.b, oh), b3 = ld_off(sVa.b, oh);
            T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            if (LERP) { a0 = ld_off(sTa.a, oh); a1 = ld_off(sHur.a, oh); a2 = ld_off(sUa.a, oh); a3 = ld_off(sVa.a, oh); }
            plain_bracket(i1, ih);
            a_lo = a_hi; b_lo = b_hi; c_lo = c_hi; d_lo = d_hi;
            a_hi = is_sfc(ih) ? sfa : tl(sTa, b0, a0);
            b_hi = is_sfc(ih) ? sfb : tl(sHur, b1, a1);
            c_hi = tl(sUa, b2, a2);
            d_hi = tl(sVa, b3, a3);
            ci1 = ci2 = i1;
        };
        // y_hi - y_lo of the column interpolation (functions.py:575-578): numba takes it in the delta's dtype - float64
        // after a time interpolation, the file's float32 when the instant is a record (REF && !LERP)
        auto ydiff = [](double hi, double lo) -> double {
            return (REF && !LERP) ? (double)((float)hi - (float)lo) : hi - lo; };
        int j1 = 0, j2 = 0;
        double xprev = -__builtin_inf();
        // ---- level loop, chunks of U levels with the next chunk's 4*U rows in flight
        T nT[U], nQ[U], nU[U], nV[U];                 // prefetched rows stay in the storage type until they are used
#pragma unroll
        for (int u = 0; u < U; ++u) {
            O o = base + (O)(u < N ? u : N - 1) * row;
            nT[u] = QLD(fT, o); nQ[u] = QLD(fQ, o); nU[u] = QLD(fU, o); nV[u] = QLD(fV, o);
        }
        for (int l0 = 0; l0 < N; l0 += U) {
            T cT[U], cQ[U], cU[U], cV[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cT[u] = nT[u]; cQ[u] = nQ[u]; cU[u] = nU[u]; cV[u] = nV[u]; }
            if (l0 + U < N) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    O o = base + (O)((l0 + U + u) < N ? (l0 + U + u) : N - 1) * row;
                    nT[u] = QLD(fT, o); nQ[u] = QLD(fQ, o); nU[u] = QLD(fU, o); nV[u] = QLD(fV, o);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = l0 + u;
                if (l < N) {
                    double pa = s_akm[l] + ps * s_bkm[l];                          // step_03:87-88
                    if (check_top) { if (pa != pa) nanflag |= 1; else min_t = fmin(min_t, pa); }
                    double x;                                                      // functions.py:471
                    if (l < n_pure_lv) x = ps_finite ? s_lnpa[l] : pa;             // pa == akm[l]; NaN for a non-finite ps
                    else x = pgw_log_tab(pa, s_logt);
                    if (!(x >= xprev)) { j1 = 0; j2 = 0; }
                    while (j2 < S) { double xs = s_lnp[j2]; if (xs == x || xs > x) break; ++j2; }
                    xprev = x;           // a NaN target leaves NaN here: the next level's `!(x >= xprev)` restarts the scans, as +inf did
                    // ua, va on the plain plev axis
                    double dc, dd;
                    int p1, p2;                                                    // its bracket
                    if (j2 >= S) { p1 = p2 = S - 1; }                              // above range, constant :558-560
                    else {
                        double xs = s_lnp[j2];
                        if (xs == x) { p1 = p2 = j2; }                             // exact                 :540-543
                        else if (j2 == 0) { p1 = p2 = 0; }                         // below range, constant :534-536
                        else { p1 = j2 - 1; p2 = j2; }                             // bracket               :545-548
                    }
                    const bool same_axis = (ksfc < 0 || j2 < ksfc);       // ta / hur stand at the same bracket (see below)
                    if (same_axis) fetch12(p1); else fetch2(p1);
                    dc = c_lo; dd = d_lo;
                    double dxp = 0.0;
                    if (p1 != p2) {                                                // :575-578
                        if (HOIST_DP) dxp = x - px1;
                        else {
                            double x1 = s_lnp[p1], x2 = s_lnp[p2];
                            dxp = x - x1;
                            by_Dp = SharedDivisor(x2 - x1);                        // x1 < x < x2: finite, positive
                        }
                        dc = c_lo + by_Dp.divide(dxp * ydiff(c_hi, c_lo));
                        dd = d_lo + by_Dp.divide(dxp * ydiff(d_hi, d_lo));
                    }
                    // ta, hur on the axis modified by the surface insertion (level ksfc moved to ps_hist, :362-365).
                    // Source levels below index ksfc are untouched, so while the plain scan stands at j2 < ksfc the
                    // modified scan stands there too (same values, same rule): same bracket, same x - x1, same divisor.
                    double da, db;
                    if (same_axis) {
                        da = a_lo; db = b_lo;
                        if (p1 != p2) {
                            da = a_lo + by_Dp.divide(dxp * ydiff(a_hi, a_lo));
                            db = b_lo + by_Dp.divide(dxp * ydiff(b_hi, b_lo));
                        }
                    } else {
                        // j1 may lag behind (it only moves here); targets ascend, so resuming from it finds the same index
                        while (j1 < S) { double xs = sx1(j1); if (xs == x || xs > x) break; ++j1; }
                        int i1, i2;
                        if (j1 >= S) { i1 = i2 = S - 1; }
                        else {
                            double xs = sx1(j1);
                            if (xs == x) { i1 = i2 = j1; }
                            else if (j1 == 0) { i1 = i2 = 0; }
                            else { i1 = j1 - 1; i2 = j1; }
                        }
                        fetch1(i1);
                        da = a_lo; db = b_lo;
                        if (i1 != i2) {
                            double x1 = sx1(i1), x2 = sx1(i2);
                            const double dx = x - x1;
                            const SharedDivisor by_Dx(x2 - x1);
                            da = a_lo + by_Dx.divide(dx * ydiff(a_hi, a_lo));
                            db = b_lo + by_Dx.divide(dx * ydiff(b_hi, b_lo));
                        }
                    }
                    const O o = obase + (O)l * orow;
                    QST(oU, o, (TO)((double)cU[u] + dc));                                  // step_03:170-173
                    QST(oV, o, (TO)((double)cV[u] + dd));
                    double rh_era;                                                 // step_03:91-94
                    if (REF) rh_era = q_to_rh_f32((float)cQ[u], pa, (float)cT[u]);
                    else rh_era = q_to_rh((double)cQ[u], pa, (double)cT[u]);
                    double ta_pgw = (double)cT[u] + da;
                    double hur_pgw = rh_era + db;
                    QST2(oT, o, (TO)ta_pgw);
                    double e_pgw = rh_to_e(hur_pgw, ta_pgw);                       // functions.py:123
                    if (l < n_pure) QST(oQ, o, (TO)e_to_q_ns(e_pgw, pa));          // pa == akm[l] for every finite ps
                    else QST2(oE, o, (TO)e_pgw);
                    if (oHur) st_off(oHur, o, (TO)hur_pgw);
                }
            }
        }
    }
    if (check_top) {
        double wt = wave_min(min_t), ws = wave_min(min_s);
        int wn = nanflag;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wn |= __shfl_xor(wn, off, 64);
        int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_mint[w] = wt; s_mins[w] = ws; s_nan[w] = wn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double mt = s_mint[0], ms = s_mins[0];
            int nn = s_nan[0];
            for (int i = 1; i < TPB / 64; ++i) { mt = fmin(mt, s_mint[i]); ms = fmin(ms, s_mins[i]); nn |= s_nan[i]; }
            if (mt < __builtin_inf()) atomicMin(&st->min_targ_bits, mt > 0 ? dbits(mt) : 0ull);
            if (ms < __builtin_inf()) atomicMin(&st->min_src_bits, ms > 0 ? dbits(ms) : 0ull);
            if (nn) atomicOr(&st->nan_seen, nn);
        }
    }
}

// =====================================================================================
// a10  bilinear regridding (lat then lon)                      functions.py:817-893
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_zonal_mean_rows(long long nfield, int nlat_s, int nlon_s,
                                                           const T *__restrict__ src, int south_row, int north_row,
                                                           double *__restrict__ pole /* [nfield][2] */) {
    // one wave per (field, pole); NaN-skipping mean like xarray .mean(dim=lon) (:836,:841).
    // Sequential pairwise-free summation order differs from numpy's pairwise sum by O(eps).
    long long wv = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wv >= nfield * 2) return;
    long long f = wv >> 1;
    int which = (int)(wv & 1);
    int row = which ? north_row : south_row;
    if (row < 0) return;
    const T *r = src + (f * nlat_s + row) * nlon_s;
    double sum = 0.0;
    long long cnt = 0;
    for (int i = lane; i < nlon_s; i += 64) {
        double v = (double)r[i];
        if (v == v) { sum += v; cnt += 1; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    if (lane == 0) pole[f * 2 + which] = cnt > 0 ? sum / (double)cnt : __builtin_nan("");
}

struct RegridTables {
    const int *lat_lo, *lat_hi, *lat_oob;
    const double *lat_dx, *lat_Dx;
    const int *lon_lo, *lon_hi, *lon_oob;
    const double *lon_dx, *lon_Dx;
};

// A block owns 256 * W consecutive target longitudes of one target latitude row (W = 2 where the row length allows:
// each thread then stores two adjacent values per plane as one 16-B store - the kernel is write-dominated, 1.89 of its
// 2.03 GB, and 8-B-per-lane stores reached 3.4 TB/s where the 16-B kernels of this library reach 5.5-6) and walks the
// fields (time x plev planes).  All its points interpolate between the same two source rows, and 512 target longitudes
// of a 0.25 deg grid fall on ~140 consecutive source columns, so per plane the block first does the LATITUDE pass once
// per needed source column (coalesced loads of the two source rows; the values the reference's first interp1d produces
// at those columns, :859) into LDS, and each thread then takes its neighbours from LDS for the LONGITUDE pass (:892) -
// instead of four uncoalesced 8-byte gathers per output, which kept the texture-address path, not HBM, busy
// (2.4 TB/s).  Arithmetic per output value is unchanged: slope*(x_new-x_lo)+y_lo as scipy interp1d, twice.
// Source columns are addressed relative to the first valid lane's lower neighbour, modulo nlon_s (periodic
// wrap); a block whose points span more than REGRID_SPAN source columns (coarse or unsorted targets) gathers
// directly like the first version.  FU planes per step; the next step's source loads are issued before this
// step's stores.
constexpr int REGRID_SPAN = BLOCK;       // one source column per thread

template <typename T, int FU, int W>
__global__ __launch_bounds__(BLOCK) void k_regrid(long long nfield, int nlat_s, int nlon_s, int nlat_t, int nlon_t,
                                                  unsigned int bx, unsigned int gz, const T *__restrict__ src, RegridTables tb,
                                                  const double *__restrict__ pole, T *__restrict__ out) {
    __shared__ double s_y[FU][REGRID_SPAN];
    __shared__ int s_first, s_span;
    // 1-D grid of bx * nlat_t * gz blocks.  Blocks b and b + 8 run on the same XCD (MI355X_MICROARCH.md, workgroup
    // dispatch), each XCD with its own L2: with gz a multiple of 8, XCD group b % 8 takes the z-slices g, g + 8, ... so
    // a source plane is fetched into ONE L2 instead of all eight, and the blocks resident on an XCD together are
    // neighbours in (x, j) of the same planes.  Speed only; any placement is correct.
    int bxi, j, bz;
    {
        const unsigned int L = blockIdx.x, per_slice = bx * (unsigned int)nlat_t;
        unsigned int w = L, g = 0;
        if (gz % 8 == 0) { g = L & 7u; w = L >> 3; }
        const unsigned int sub = w / per_slice, rem = w - sub * per_slice;
        bz = (gz % 8 == 0) ? (int)(sub * 8u + g) : (int)sub;
        j = (int)(rem / bx);                                    // target lat
        bxi = (int)(rem - (unsigned int)j * bx);
    }
    const int i0 = (bxi * BLOCK + threadIdx.x) * W;             // first target lon of this thread (W | nlon_t when W > 1)
    const bool active = i0 < nlon_t;
    const int jl = tb.lat_lo[j], jh = tb.lat_hi[j];
    const double ldx = tb.lat_dx[j], lDx = tb.lat_Dx[j];
    const bool lat_oob = tb.lat_oob[j] != 0;
    int il[W], ih[W];
    double odx[W], oDx[W];
    bool oob[W], valid[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int ii = active ? i0 + w : nlon_t - 1;
        il[w] = tb.lon_lo[ii]; ih[w] = tb.lon_hi[ii];
        odx[w] = tb.lon_dx[ii]; oDx[w] = tb.lon_Dx[ii];
        oob[w] = lat_oob || tb.lon_oob[ii];
        valid[w] = active && !tb.lon_oob[ii];
    }
    const long long plane_s = (long long)nlat_s * nlon_s, plane_t = (long long)nlat_t * nlon_t;
    const bool lo_pole = jl < 0 || jl >= nlat_s, hi_pole = jh < 0 || jh >= nlat_s;
    const long long row_lo = (long long)(lo_pole ? 0 : jl) * nlon_s, row_hi = (long long)(hi_pole ? 0 : jh) * nlon_s;
    const int lo_which = jl < 0 ? 0 : 1, hi_which = jh < 0 ? 0 : 1;
    // the grid spacings divide every plane's differences: reciprocals once per thread (SharedDivisor)
    const SharedDivisor by_lDx(lDx);
    struct DivW { SharedDivisor d[W]; };
    const DivW by_o = [&]() { if constexpr (W == 1) return DivW{{SharedDivisor(oDx[0])}}; else return DivW{{SharedDivisor(oDx[0]), SharedDivisor(oDx[1])}}; }();
    const SharedDivisor (&by_oDx)[W] = by_o.d;
    // source-column window of this block: starts at the lower neighbour of the first valid target
    if (threadIdx.x == 0) { s_first = BLOCK * W; s_span = 0; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w) if (valid[w]) atomicMin(&s_first, (int)threadIdx.x * W + w);
    __syncthreads();
    const int first = s_first;                            // BLOCK * W if the block has no valid point
    int c0 = 0;
    if (first < BLOCK * W) c0 = tb.lon_lo[bxi * BLOCK * W + first];
    int rl[W], rh[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        rl[w] = il[w] - c0; rh[w] = ih[w] - c0;
        if (rl[w] < 0) rl[w] += nlon_s;
        if (rh[w] < 0) rh[w] += nlon_s;
        if (valid[w]) atomicMax(&s_span, (rl[w] > rh[w] ? rl[w] : rh[w]) + 1);
    }
    __syncthreads();
    const int span = s_span;
    const bool staged = !lat_oob && span <= REGRID_SPAN;
    // fields of this z-slice
    long long per = (nfield + gz - 1) / gz;
    long long f0 = (long long)bz * per, f1 = f0 + per < nfield ? f0 + per : nfield;
    T *po = out + (long long)j * nlon_t + (active ? i0 : 0);
    auto store = [&](long long f, const double (&r)[W]) {
        storev_nt<T, W>(po + f * plane_t, r);             // W * sizeof(T) aligned: W | nlon_t and W | i0; streaming: the output
                                                          // must not push the source planes out of L2
    };
    if (staged) {
        // this thread's source column of the window (threads >= span idle in the latitude pass)
        const bool loader = (int)threadIdx.x < span;
        int col = c0 + (int)threadIdx.x;
        if (col >= nlon_s) col -= nlon_s;
        if (!loader) col = 0;
        int li[W], hi[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { li[w] = valid[w] ? rl[w] : 0; hi[w] = valid[w] ? rh[w] : 0; }
        double v_lo[FU], v_hi[FU];
        auto fetch = [&](long long f) {                    // the two source rows of FU planes at this thread's column
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                long long ff = (f + u < f1) ? f + u : f1 - 1;
                const T *sp = src + ff * plane_s;
                v_lo[u] = lo_pole ? pole[ff * 2 + lo_which] : (double)sp[row_lo + col];
                v_hi[u] = hi_pole ? pole[ff * 2 + hi_which] : (double)sp[row_hi + col];
            }
        };
        if (loader && f0 < f1) fetch(f0);
        for (long long f = f0; f < f1; f += FU) {
            if (loader) {
#pragma unroll
                for (int u = 0; u < FU; ++u) s_y[u][threadIdx.x] = by_lDx.divide(v_hi[u] - v_lo[u]) * ldx + v_lo[u];   // lat pass :859
            }
            __syncthreads();
            if (loader && f + FU < f1) fetch(f + FU);      // next planes' loads fly during this group's stores
            double ya[FU][W], yb[FU][W];
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int w = 0; w < W; ++w) { ya[u][w] = s_y[u][li[w]]; yb[u][w] = s_y[u][hi[w]]; }
            __syncthreads();                               // s_y is rewritten by the next group of planes
            if (active) {
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    if (f + u < f1) {
                        double r[W];
#pragma unroll
                        for (int w = 0; w < W; ++w)
                            r[w] = oob[w] ? __builtin_nan("") : by_oDx[w].divide(yb[u][w] - ya[u][w]) * odx[w] + ya[u][w];   // lon pass :892
                        store(f + u, r);
                    }
                }
            }
        }
        return;
    }
    // window wider than the tile (target coarser than the source, unsorted target longitudes) or latitude out of
    // bounds: four direct gathers per point, like the first version of this kernel
    for (long long f = f0; f < f1; f += FU) {
        double ya[FU][W], yb[FU][W];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            long long ff = (f + u < f1) ? f + u : f1 - 1;
            const T *sp = src + ff * plane_s;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                double a_lo = 0, b_lo = 0, a_hi = 0, b_hi = 0;
                if (!oob[w]) {
                    a_lo = lo_pole ? pole[ff * 2 + lo_which] : (double)sp[row_lo + il[w]];
                    b_lo = lo_pole ? pole[ff * 2 + lo_which] : (double)sp[row_lo + ih[w]];
                    a_hi = hi_pole ? pole[ff * 2 + hi_which] : (double)sp[row_hi + il[w]];
                    b_hi = hi_pole ? pole[ff * 2 + hi_which] : (double)sp[row_hi + ih[w]];
                }
                ya[u][w] = by_lDx.divide(a_hi - a_lo) * ldx + a_lo;                      // lat pass at the lower lon  :859
                yb[u][w] = by_lDx.divide(b_hi - b_lo) * ldx + b_lo;                      // lat pass at the upper lon
            }
        }
        if (active) {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                if (f + u < f1) {
                    double r[W];
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        r[w] = oob[w] ? __builtin_nan("") : by_oDx[w].divide(yb[u][w] - ya[u][w]) * odx[w] + ya[u][w];   // lon pass  :892
                    store(f + u, r);
                }
            }
        }
    }
}

// =====================================================================================
// a9  surface riders               step_03_apply_to_era.py:103-146, functions.py:1145-1186
// =====================================================================================
constexpr int MAX_SOIL = 16;
struct SoilTable { double w[MAX_SOIL]; int n; };     // w = exp(-depth/2.8), computed on the host

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_surface_update(int ntime, long long ncol, SoilTable soil,
                                                          const T *__restrict__ sic, const T *__restrict__ dsic,
                                                          const T *__restrict__ dtos, const T *__restrict__ dts,
                                                          const T *__restrict__ land, const T *__restrict__ clim,
                                                          const T *__restrict__ tskin, const T *__restrict__ tso,
                                                          T *__restrict__ sic_out, T *__restrict__ comb_out,
                                                          T *__restrict__ tskin_out, T *__restrict__ tso_out) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long n = (long long)ntime * ncol;
    if (i >= n) return;
    long long t = i / ncol, c = i - t * ncol;
    double ice = (double)sic[i] + (double)dsic[i] / 100;             // step_03:105
    ice = fmin(fmax(ice, 0.0), 1.0);                                  // :106-107 (np.clip keeps NaN)
    if ((double)sic[i] != (double)sic[i] || (double)dsic[i] != (double)dsic[i]) ice = __builtin_nan("");
    if (sic_out) sic_out[i] = (T)ice;
    double tos = (double)dtos[i], ts = (double)dts[i];
    // land fraction / sea ice of time step 0 (step_03:121-122 .isel(time=0))
    double lf = (double)land[c];
    double ice0 = ice;
    if (t != 0) {
        double i0 = (double)sic[c] + (double)dsic[c] / 100;
        i0 = fmin(fmax(i0, 0.0), 1.0);
        if ((double)sic[c] != (double)sic[c] || (double)dsic[c] != (double)dsic[c]) i0 = __builtin_nan("");
        ice0 = i0;
    }
    double comb = ts;                                                 // functions.py:1180-1181
    if (ice0 == ice0 && tos == tos) {                                 // :1173
        double fr = fmin(fmax(ice0 + lf, 0.0), 1.0);                  // :1183
        if (lf != lf) fr = __builtin_nan("");                         // np.clip keeps NaN
        comb = fr * ts + (1 - fr) * tos;                              // :1184
    }
    if (comb_out) comb_out[i] = (T)comb;
    if (tskin_out) tskin_out[i] = (T)((double)tskin[i] + comb);       // step_03:124
    if (tso_out) {
        double cl = (double)clim[c];                                  // annual mean ts delta :134-136
#pragma unroll
        for (int s = 0; s < MAX_SOIL; ++s) {
            if (s < soil.n) {
                long long o = (t * soil.n + s) * ncol + c;
                tso_out[o] = (T)((double)tso[o] + (cl + soil.w[s] * (comb - cl)));   // :139-144
            }
        }
    }
}

// replace_delta_sfc (functions.py:343-366) on many ascending-pressure columns: source_P is the
// broadcast 1-D table, delta (ntime, S, ncol) ascending order; outputs the modified columns.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_replace_delta_sfc(PlevTable pt, int ntime, long long ncol,
                                                             const T *__restrict__ delta, const T *__restrict__ dsfc,
                                                             const T *__restrict__ pshist, T *__restrict__ outP,
                                                             T *__restrict__ outD, DevStatus *st) {
    __shared__ double s_p[MAX_PLEV];
    const int S = pt.n;
    if (threadIdx.x < MAX_PLEV) s_p[threadIdx.x] = pt.p[threadIdx.x];
    __syncthreads();
    long long flat = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (flat >= (long long)ntime * ncol) return;
    long long t = flat / ncol, c = flat - t * ncol;
    long long base = t * S * ncol + c;
    double ps = (double)pshist[flat], ds = (double)dsfc[flat];
    int k = -1;
    bool fill = false;
    if (ps > pt.pmax) k = S - 1;                                   // :356-359
    else if (ps < pt.pmin) { report(st, 15, flat); }               // :360-361
    else {                                                         // :362-365
        for (int i = 0; i < S; ++i) if (ps > s_p[i]) k = i;
        if (k < 0) report(st, 15, flat);
        fill = true;
    }
    for (int i = 0; i < S; ++i) {
        double P = s_p[i], D = (double)delta[base + (long long)i * ncol];
        if (k >= 0) {
            if (i == k) { P = ps; D = ds; }
            else if (fill && i > k) D = ds;
        }
        outP[base + (long long)i * ncol] = (T)P;
        outD[base + (long long)i * ncol] = (T)D;
    }
}

// integrate_tos (functions.py:1145-1186), flat over n
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_integrate_tos(long long n, const T *__restrict__ tos, const T *__restrict__ ts,
                                                         const T *__restrict__ land, const T *__restrict__ ice,
                                                         T *__restrict__ out) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double o = (double)tos[i], s = (double)ts[i], ic = (double)ice[i];
    double r = s;                                                   // :1180-1181
    if (ic == ic && o == o) {                                       // :1173
        double fr = fmin(fmax(ic + (double)land[i], 0.0), 1.0);     // :1183 (np.clip propagates NaN)
        if ((double)land[i] != (double)land[i]) fr = __builtin_nan("");
        r = fr * s + (1 - fr) * o;                                  // :1184
    }
    out[i] = (T)r;
}

// p_ref_inp = None (step_03:219-253 with determine_p_ref, functions.py:583-598): per column the first
// plev (file order) with 0.95*ps_era > p and 0.95*ps_pgw > p, never below the previous pass's choice;
// also applies delta_ps += adj_ps (step_03:192) and gathers g * zg-delta at the chosen level (:292-295).
template <typename T, bool REF>
__global__ __launch_bounds__(BLOCK) void k_local_p_ref(PlevTable pt /* p[] in FILE order here */, double akN, double bkN,
                                                       long long n, const T *__restrict__ PS,
                                                       double *__restrict__ delta_ps, const double *__restrict__ adj_ps,
                                                       DeltaSrc<T> zg /* (nplev, ncol) records */, long long ncol, int first_pass,
                                                       double *__restrict__ p_ref, int *__restrict__ p_idx,
                                                       double *__restrict__ dphi, DevStatus *st) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double dps = next_delta_ps<REF>(delta_ps[i], adj_ps[i]);        // step_03:192
    delta_ps[i] = dps;
    double ps0 = (double)PS[i];
    double p_min_era = (akN + ps0 * bkN) * 0.95;                    // :227-228
    double p_min_pgw = (akN + ps_of<REF>(ps0, dps) * bkN) * 0.95;   // :229-230
    double p = __builtin_nan("");
    int k = -1;
    for (int j = 0; j < pt.n; ++j) {
        if ((p_min_era > pt.p[j]) && (p_min_pgw > pt.p[j])) { p = pt.p[j]; k = j; break; }   // functions.py:593-596
    }
    if (k >= 0 && !first_pass) {
        double last = p_ref[i];
        if (last < p) { p = last; k = p_idx[i]; }                    // min(p, p_ref_last)  :598
    }
    if (k < 0) { report(st, 19, i); p_ref[i] = p; p_idx[i] = -1; dphi[i] = p; return; }   // step_03:245-251
    p_ref[i] = p;
    p_idx[i] = k;
    long long t = i / ncol, c = i - t * ncol;
    dphi[i] = zg.template get<REF>((t * pt.n + k) * ncol + c) * CON_G;            // step_03:292-295
}

// step_03:192-193.  apply_adj = 0: delta_ps already carries this pass's increment (k_local_p_ref applied it), only
// ps_pgw = PS + delta_ps is formed.  REF: the float32 sums of the reference on float32 files (next_delta_ps, ps_of).
template <typename T, bool REF = false>
__global__ __launch_bounds__(BLOCK) void k_update_ps(long long n, const T *__restrict__ PS, double *__restrict__ delta_ps,
                                                     const double *__restrict__ adj_ps, T *__restrict__ ps_out, int apply_adj = 1) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double d = delta_ps[i];
    if (apply_adj) { d = next_delta_ps<REF>(d, adj_ps[i]); delta_ps[i] = d; }
    ps_out[i] = (T)ps_of<REF>((double)PS[i], d);
}

// g * time-interpolated zg delta at p_ref -> fp64 loop constant (step_03:292-295)
// REF: a record that needed no time interpolation stays float32 (functions.py:282-283) and `* CON_G` is a float32 product
template <typename T, bool REF>
__global__ __launch_bounds__(BLOCK) void k_dphi_clim(long long n, DeltaSrc<T> z, double g, double *__restrict__ out,
                                                     double *__restrict__ zero_a, double *__restrict__ zero_b) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        if (REF && !z.a) out[i] = (double)((float)z.b[i] * (float)g);
        else out[i] = z.template get<REF>(i) * g;
        if (zero_a) { zero_a[i] = 0.0; zero_b[i] = 0.0; }          // delta_ps, adj_ps of the loop start (step_03:182-184)
    }
}

// -------------------------------------------------------------------------------------
// Several passes of the loop in ONE launch (fixed p_ref).  A column's trajectory delta_ps(k) depends on that column
// alone - only the stopping test max|err| <= thresh is global (step_03:189, 308) - so a block can run `npass` passes
// on its columns back to back and leave, per pass, the block maxima (status block k) and the delta_ps the pass used
// (dps_hist[k]); the host then applies the reference's control flow to the recorded maxima and finalises from the
// first pass that met the threshold.  Passes beyond it were speculated and are discarded.
// What this buys is NOT memory traffic (the pass is bound by fp64 issue and latency, not by HBM: the float32-storage
// build reads half the bytes in the same time) but the fixed cost of a launch (block dispatch, table staging, tail:
// 0.03-0.045 ms of a 0.22 ms pass) and one host round trip per pass.  Each pass re-reads its rows (L2 / Infinity Cache / HBM).
// `first` != 0 (first launch of a file): phi_ref of the ERA state (step_03:280-287) and g * dzg (:292-295) are computed
// here and stored for continuation launches; delta_ps = adj_ps = 0 (:182-184).
// -------------------------------------------------------------------------------------
constexpr int MULTI_MAX_PASS = 8;
#ifndef MULTI_MINW
#define MULTI_MINW 4      // 128 VGPRs (12 bytes of scratch per lane): 1.28 -> 1.21 ms against 3 waves / 136 VGPRs; 5 waves spill 132 bytes and lose
#endif
// LOCAL: settings.p_ref_inp = None (step_03:219-253) - the reference level is chosen per column and pass (the first delta
// level, in file order, that lies above 0.95 of the lowest half-level pressure of both states, functions.py:583-598; never
// lower than in the pass before, :598), phi_ref of the ERA state is taken at that level (:280-287: recomputed here only
// when the level changed, it depends on nothing else) and g * dzg at it (:292-295).  `zg` then holds the full
// (nplev, ncol) records, `pt.p` the plev coordinate in file order, `p_ref_col` / `p_idx_col` the per-column memory that a
// continuation launch resumes from.  One column per lane (V = 1).
struct LocalPRef { PlevTable pt; double akN, bkN; double *p_ref_col; int *p_idx_col; };

template <typename T, typename TL, int V, int U, bool REF, bool LOCAL>
__global__ __launch_bounds__(BLOCK, MULTI_MINW) void k_ps_loop_multi(Levels lv, int ntime, long long ncol,
                                                         const T *__restrict__ Tera, const T *__restrict__ QVera,
                                                         const TL *__restrict__ ta, const TL *__restrict__ evap,
                                                         const T *__restrict__ PS, const T *__restrict__ FIS,
                                                         DeltaSrc<T> zg, double *__restrict__ phi_ref_era,
                                                         double *__restrict__ dphi_clim,
                                                         double *__restrict__ delta_ps, double *__restrict__ adj_ps,
                                                         double *__restrict__ dps_hist /* [npass][ntime*ncol] */,
                                                         double p_ref_s, double adj_factor, int first, int npass,
                                                         DevStatus *st0 /* errors of the ERA-state scan */,
                                                         DevStatus *st /* [npass] */, LocalPRef loc) {
    static_assert(!LOCAL || V == 1, "local p_ref: one column per lane");
    __shared__ unsigned long long s_max[MULTI_MAX_PASS];          // per pass: max |err| as ordered bits, #valid, levels read
    __shared__ unsigned int s_valid[MULTI_MAX_PASS];
    __shared__ unsigned long long s_touched[MULTI_MAX_PASS];
    __shared__ double s_lev[LEVTAB_DOUBLES];
    if (threadIdx.x < MULTI_MAX_PASS) { s_max[threadIdx.x] = 0ull; s_valid[threadIdx.x] = 0u; s_touched[threadIdx.x] = 0ull; }
    LevTab lt = stage_levels<true, true>(lv, s_lev, BLOCK);        // ends with a barrier
    const long long n2 = (long long)ntime * ncol;
    long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long ngroups = n2 / V;
    // The per-pass wave reductions below (wave_max, the __shfl_xor sum) need all 64 lanes: a lane beyond the last group
    // (1440 x 721 columns leave 32 lanes of the last wave idle) walks the last group again - no stores, nothing contributed -
    // instead of sitting out in a divergent branch, where the shuffles would read its registers undefined (a zero from
    // there made an all-NaN partial wave "valid" with |err| = 0).
    const bool live = g < ngroups;
    if (!live) g = ngroups - 1;
    {
        ColIdx ix = col_index(g, V, ncol);
        const int N = lv.nlev;
        long long c2 = ix.t * ncol + ix.c;
        const long long lbase = ix.t * N * ncol + ix.c;
        double ps0[V], z[V], dps[V], adj[V], pref[V], tlow[V], phi_ref[V], phi_era[V], dphi[V];
        int idx = -2;                                              // LOCAL: index of the column's current level (-2: none yet)
        loadv<T, V>(PS + c2, ps0);
        loadv<T, V>(FIS + c2, z);
#pragma unroll
        for (int v = 0; v < V; ++v) pref[v] = p_ref_s;
        if (first) {
            if (!LOCAL) {
                int touched0 = 0;
                scan_columns<T, V, U, true, REF>(lv, lt, ncol, Tera + lbase, QVera + lbase, ps0, z, pref, 0, st0, c2, phi_era, tlow, touched0);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (REF && !zg.a) dphi[v] = (double)((float)zg.b[c2 + v] * (float)CON_G);      // see k_dphi_clim
                    else dphi[v] = zg.template get<REF>(c2 + v) * CON_G;                             // step_03:292-295
                }
                if (live) {
                    storev<double, V>(phi_ref_era + c2, phi_era);
                    storev<double, V>(dphi_clim + c2, dphi);
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) { dps[v] = 0.0; adj[v] = 0.0; }                              // :182-184
        } else {
            loadv<double, V>(phi_ref_era + c2, phi_era);
            loadv<double, V>(dphi_clim + c2, dphi);
            loadv<double, V>(delta_ps + c2, dps);
            loadv<double, V>(adj_ps + c2, adj);
            if (LOCAL) { pref[0] = loc.p_ref_col[c2]; idx = loc.p_idx_col[c2]; }
        }
        for (int k = 0; k < npass; ++k) {
            double ps[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                dps[v] = next_delta_ps<REF>(dps[v], adj[v]);              // step_03:192
                ps[v] = ps_of<REF>(ps0[v], dps[v]);                       // :193
            }
            if (live) storev<double, V>(dps_hist + (long long)k * n2 + c2, dps);
            if (LOCAL) {
                const double p_min_era = (loc.akN + ps0[0] * loc.bkN) * 0.95;                  // :227-228
                const double p_min_pgw = (loc.akN + ps[0] * loc.bkN) * 0.95;                   // :229-230
                double p = __builtin_nan("");
                int j = -1;
                for (int i = 0; i < loc.pt.n; ++i)
                    if ((p_min_era > loc.pt.p[i]) && (p_min_pgw > loc.pt.p[i])) { p = loc.pt.p[i]; j = i; break; }   // functions.py:593-596
                if (j >= 0 && idx >= 0 && pref[0] < p) { p = pref[0]; j = idx; }               // min(p, p_ref_last)  :598
                if (j < 0) { report(st + k, 19, c2); p = __builtin_nan(""); }                   // step_03:245-251
                if (j != idx) {                                                                 // (first pass: idx == -2)
                    pref[0] = p;
                    idx = j;
                    if (j >= 0) {
                        int touched0 = 0;
                        scan_columns<T, V, U, true, REF>(lv, lt, ncol, Tera + lbase, QVera + lbase, ps0, z, pref, 0, st + k, c2, phi_era,
                                                         tlow, touched0);                       // :280-287
                        dphi[0] = zg.template get<REF>((ix.t * loc.pt.n + j) * ncol + ix.c) * CON_G;          // :292-295
                    } else {
                        phi_era[0] = p; dphi[0] = p;
                    }
                }
            }
            int touched = 0;
            scan_columns<TL, V, U, false, REF>(lv, lt, ncol, ta + lbase, evap + lbase, ps, z, pref, 0, st + k, c2, phi_ref, tlow,
                                               touched);
            double amax = -1.0;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                double err = (phi_ref[v] - phi_era[v]) - dphi[v];                         // :289,298
                const double fps = REF ? (double)((float)(-adj_factor) * (float)ps[v]) : -adj_factor * ps[v];
                adj[v] = fps / (CON_RD * tlow[v]) * err;                                  // :301-304
                double ae = fabs(err);
                if (ae == ae) amax = fmax(amax, ae);                                      // :308 skipna
            }
            if (!live) { amax = -1.0; touched = 0; }
            double wm = wave_max(amax);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) touched += __shfl_xor(touched, off, 64);
            if ((threadIdx.x & 63) == 0) {
                if (wm >= 0.0) { atomicMax(&s_max[k], dbits(wm)); atomicAdd(&s_valid[k], 1u); }
                atomicAdd(&s_touched[k], (unsigned long long)touched);
            }
        }
        if (live) {
            storev<double, V>(delta_ps + c2, dps);        // state after the last pass: a continuation launch resumes here
            storev<double, V>(adj_ps + c2, adj);
        }
        if (LOCAL && live) {
            loc.p_ref_col[c2] = pref[0]; loc.p_idx_col[c2] = idx;
            phi_ref_era[c2] = phi_era[0]; dphi_clim[c2] = dphi[0];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < npass) {
        const int k = threadIdx.x;
        if (s_valid[k]) {
            atomicMax(&st[k].max_bits, s_max[k]);
            atomicAdd(&st[k].valid, 1ull);
        }
        atomicAdd(&st[k].levels_touched, s_touched[k]);
    }
}

// surface riders with the time lerp of the three 2-D deltas fused in (step_03:103-146)
// REF: the sea-ice fraction the blend sees is the float32 value stored back into the file's array (step_03:105-107), and
// `ice + land`, `1 - frac` are float32 operations on the file's float32 fractions (functions.py:1183-1184)
template <typename T, bool REF>
__global__ __launch_bounds__(BLOCK) void k_surface_update_lerp(int ntime, long long ncol, SoilTable soil,
                                                               const T *__restrict__ sic, DeltaSrc<T> dsic, DeltaSrc<T> dtos,
                                                               DeltaSrc<T> dts, const T *__restrict__ land,
                                                               const T *__restrict__ clim, const T *__restrict__ tskin,
                                                               const T *__restrict__ tso, T *__restrict__ sic_out,
                                                               T *__restrict__ tskin_out, T *__restrict__ tso_out) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long n = (long long)ntime * ncol;
    if (i >= n) return;
    long long t = i / ncol, c = i - t * ncol;
    // REF and the instant IS a delta record (no time interpolation: the delta stays the file's float32, functions.py:282-283):
    // numpy then takes `delta / 100`, the sum and the blend below in float32
    const bool f32_delta = REF && !dsic.a;
    auto ice_of = [&](long long k) -> double {
        double s0 = (double)sic[k], d0 = dsic.template get<REF>(k);
        double v = s0 + d0 / 100;                                     // step_03:105
        if (f32_delta) v = (double)((float)s0 + (float)d0 / 100.0f);
        v = fmin(fmax(v, 0.0), 1.0);                                  // :106-107
        if (s0 != s0 || d0 != d0) v = __builtin_nan("");              // np.clip keeps NaN
        return REF ? (double)(T)v : v;
    };
    double ice = ice_of(i);
    sic_out[i] = (T)ice;
    double ice0 = (t == 0) ? ice : ice_of(c);                         // .isel(time=0), :121-122
    double tos = dtos.template get<REF>(i), ts = dts.template get<REF>(i);
    double comb = ts;                                                 // functions.py:1180-1181
    if (ice0 == ice0 && tos == tos) {                                 // :1173
        double fr, omf;
        if (REF) {
            float f = fminf(fmaxf((float)ice0 + (float)land[c], 0.0f), 1.0f);
            if ((float)land[c] != (float)land[c]) f = __builtin_nanf("");
            fr = (double)f; omf = (double)(1.0f - f);
        } else {
            fr = fmin(fmax(ice0 + (double)land[c], 0.0), 1.0);        // :1183
            if ((double)land[c] != (double)land[c]) fr = __builtin_nan("");
            omf = 1 - fr;
        }
        comb = fr * ts + omf * tos;                                   // :1184
        if (f32_delta) comb = (double)((float)fr * (float)ts + (float)omf * (float)tos);   // float32 products and sum
    }
    tskin_out[i] = (T)((double)tskin[i] + comb);                      // step_03:124
    if (tso_out) {
        double cl = (double)clim[c];                                  // :134-136
#pragma unroll
        for (int s = 0; s < MAX_SOIL; ++s) {
            if (s < soil.n) {
                long long o = (t * soil.n + s) * ncol + c;
                tso_out[o] = (T)((double)tso[o] + (cl + soil.w[s] * (comb - cl)));   // :139-144
            }
        }
    }
}

// =====================================================================================
// step_02 `smoothing`: filter_data / harmonic_ac_analysis          functions.py:603-740
// in / out (ntime, inner): one thread per column (level, y, x flattened), lanes along x, so each time step of a
// wave is one coalesced row segment.  Pass 1 streams the series once (NaN test, sum, three cos / sin projections),
// pass 2 writes mean + the three harmonics.  cos / sin tables [3][ntime] come from the host (numpy, evaluated as
// the reference does at :716, 727) and are staged in LDS; every lane reads the same entry (broadcast).
// =====================================================================================
template <typename T, int U>
__global__ __launch_bounds__(BLOCK) void k_harmonic_smooth(int ntime, long long inner, const double *__restrict__ tabs,
                                                           const T *__restrict__ in, T *__restrict__ out) {
    extern __shared__ double s_tab[];                 // cos[3][ntime], sin[3][ntime]
    for (int i = threadIdx.x; i < 6 * ntime; i += BLOCK) s_tab[i] = tabs[i];
    __syncthreads();
    const long long c = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= inner) return;
    const double *c1 = s_tab, *c2 = s_tab + ntime, *c3 = s_tab + 2 * ntime;
    const double *s1 = s_tab + 3 * ntime, *s2 = s_tab + 4 * ntime, *s3 = s_tab + 5 * ntime;
    const T *pi = in + c;
    double sum = 0, a1 = 0, a2 = 0, a3 = 0, b1 = 0, b2 = 0, b3 = 0;
    bool nan = false;
    for (int t0 = 0; t0 < ntime; t0 += U) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (double)SIG_LD(pi + (long long)(t0 + u < ntime ? t0 + u : ntime - 1) * inner);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < ntime) {
                nan |= (x[u] != x[u]);                                   // :694
                sum += x[u];                                             // :699
                a1 += x[u] * c1[t]; b1 += x[u] * s1[t];                  // :728-730
                a2 += x[u] * c2[t]; b2 += x[u] * s2[t];
                a3 += x[u] * c3[t]; b3 += x[u] * s3[t];
            }
        }
    }
    const double lt = (double)ntime;
    const double mean = sum / lt, f = 2. / lt;
    a1 = f * a1; b1 = f * b1; a2 = f * a2; b2 = f * b2; a3 = f * a3; b3 = f * b3;
    T *po = out + c;
    for (int t = 0; t < ntime; ++t) {
        double h1 = a1 * c1[t] + b1 * s1[t];                             // :733
        double h2 = a2 * c2[t] + b2 * s2[t];
        double h3 = a3 * c3[t] + b3 * s3[t];
        double r = ((h1 + h2) + h3) + mean;                              // :739
        SIG_ST((T)(nan ? __builtin_nan("") : r), po + (long long)t * inner);     // :695-696
    }
}

// =====================================================================================
// step_02 NaN-ignoring interpolation of ocean-grid deltas (tos, siconc)      functions.py:900-1060
// The reference hands two point clouds in planar "metre" coordinates (functions.py:958-1023; host: geodesy.py) to
// pyvista's PolyData.interpolate(points, null_value=nan, radius=R, sharpness=s) (:1038-1048) = VTK's vtkPointInterpolator
// with a vtkGaussianKernel on a RADIUS footprint.  Per target point x (VTK 9.2 vtkGaussianKernel::ComputeWeights,
// vtkPointInterpolator):
//     neighbours  = source points with |x - p|^2 <= R^2 ;  none -> null value (NaN)
//     |x - p|^2 < 256 eps for some p  ->  the value of (the first such) p
//     else  sum_i w_i v_i / sum_i w_i ,  w_i = exp(-(s/R)^2 |x - p_i|^2)
// (VTK divides the weights by their sum first and then forms sum (w_i / W) v_i; here the quotient of the two sums is
// taken once - equal up to rounding; parity with VTK is unpinned anyway: VTK / pyvista / pyproj are not installable.)
// Source points are binned on the host into square cells of edge R (sorted by cell, `cell_start` = first point of each
// cell), so a target looks at its 3 x 3 block of cells.  One thread per target point; all `nm` months of a variable in
// ONE pass (the geometry - the expensive part, one exp per pair - is shared; the reference repeats it 12 times).  A
// month's NaN values are skipped for that month (the reference removes them from that month's cloud, :944-948).
// =====================================================================================
constexpr int GAUSS_MAX_FIELDS = 16;
// One target per thread; the SOURCE points are staged through LDS by the block: the targets of a block (consecutive
// indices: neighbours on the ERA5 grid) need the 3 x 3 cell neighbourhoods of a small box of cells, and for one cell row ix the
// cells iy_lo .. iy_hi are one contiguous run of the cell-sorted source arrays.  The block copies that run in chunks of
// GAUSS_CHUNK points (coordinates and the values of all months: coalesced) into LDS and every thread walks the chunk with
// broadcast reads; a point outside a thread's own neighbourhood fails the radius test (cells are at least one radius wide).
// Per thread the accepted points arrive in the order (ix, iy, p) of a direct walk over its 3 x 3 cells, so the sums are the
// bits of that walk.  The first form of this kernel did walk the cells per thread from global memory: one exposed memory
// latency per source point, 115 ms for 12 months of tos on the 0.25 deg grid at 1.4 waves per SIMD.
// A target with NaN coordinates is inactive.  The host hands the targets over in tiles of 16 x 16 grid points (edge tiles
// filled with such targets): a block's box of cells stays small, and so does the set of points a wave accepts for some of
// its lanes only (a wave runs the weight arithmetic of a point when ANY lane has it within the radius).
constexpr int GAUSS_CHUNK = 256;
template <int NM>
__global__ __launch_bounds__(BLOCK) void k_gauss_interp(long long ntarg, const double *__restrict__ tx, const double *__restrict__ ty,
                                                        int ncx, int ncy, double x0, double y0, double inv_h,
                                                        const int *__restrict__ cell_start, const double *__restrict__ sx,
                                                        const double *__restrict__ sy, const double *__restrict__ sval /* [nsrc][nm] */,
                                                        int nm, double r2, double f2, double *__restrict__ out /* [nm][ntarg] */) {
    __shared__ double s_x[GAUSS_CHUNK], s_y[GAUSS_CHUNK];
    __shared__ double s_v[GAUSS_CHUNK * NM];
    __shared__ int s_box[4];                                    // cx_min, cx_max, cy_min, cy_max of the block's active targets
    __shared__ int s_nan[GAUSS_CHUNK];                          // the point has a NaN month (rare: the common path tests nothing)
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double x = __builtin_nan(""), y = __builtin_nan("");
    if (i < ntarg) { x = tx[i]; y = ty[i]; }
    const bool active = (x == x) && (y == y);
    if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = -0x7fffffff; s_box[2] = 0x7fffffff; s_box[3] = -0x7fffffff; }
    __syncthreads();
    if (active) {
        const int cx = (int)floor((x - x0) * inv_h), cy = (int)floor((y - y0) * inv_h);
        atomicMin(&s_box[0], cx); atomicMax(&s_box[1], cx); atomicMin(&s_box[2], cy); atomicMax(&s_box[3], cy);
    }
    __syncthreads();
    double sw[NM], swv[NM], hit[NM];
    bool has_hit[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) { sw[m] = 0.0; swv[m] = 0.0; hit[m] = 0.0; has_hit[m] = false; }
    const double tol = 256.0 * 2.220446049250313e-16;           // vtkMathUtilities::FuzzyCompare(d2, 0.0, eps * 256)
    if (s_box[1] >= s_box[0]) {                                  // at least one active target (uniform over the block)
        int ix_lo = s_box[0] - 1, ix_hi = s_box[1] + 1, iy_lo = s_box[2] - 1, iy_hi = s_box[3] + 1;
        if (ix_lo < 0) ix_lo = 0;
        if (iy_lo < 0) iy_lo = 0;
        if (ix_hi > ncx - 1) ix_hi = ncx - 1;
        if (iy_hi > ncy - 1) iy_hi = ncy - 1;
        for (int ix = ix_lo; ix <= ix_hi && iy_lo <= iy_hi; ++ix) {
            const int p_lo = cell_start[ix * ncy + iy_lo], p_hi = cell_start[ix * ncy + iy_hi + 1];
            for (int p0 = p_lo; p0 < p_hi; p0 += GAUSS_CHUNK) {
                const int cnt = (p_hi - p0 < GAUSS_CHUNK) ? (p_hi - p0) : GAUSS_CHUNK;
                __syncthreads();                                 // the previous chunk has been read by everyone
                for (int q = threadIdx.x; q < cnt; q += BLOCK) { s_x[q] = sx[p0 + q]; s_y[q] = sy[p0 + q]; s_nan[q] = 0; }
                __syncthreads();
                for (int q = threadIdx.x; q < cnt * nm; q += BLOCK) {
                    const int pt = q / nm, m = q - pt * nm;
                    const double v = sval[(long long)p0 * nm + q];
                    s_v[pt * NM + m] = v;
                    if (v != v) s_nan[pt] = 1;
                }
                __syncthreads();
                if (active) {
                    for (int q = 0; q < cnt; ++q) {
                        const double dx = x - s_x[q], dy = y - s_y[q];
                        const double d2 = dx * dx + dy * dy;     // the third coordinate is 0 on both sides (:986-988, 1028-1030)
                        if (!(d2 <= r2)) continue;               // vtkStaticPointLocator::FindPointsWithinRadius
                        const double w = pgw_exp(-f2 * d2);
                        const bool exact = d2 < tol;
                        if (!s_nan[q] && !exact) {               // every month valid, not a coincident point: nothing to test
#pragma unroll
                            for (int m = 0; m < NM; ++m) {
                                if (m < nm) { const double vm = s_v[q * NM + m]; sw[m] += w; swv[m] += w * vm; }
                            }
                            continue;
                        }
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            if (m < nm) {
                                const double vm = s_v[q * NM + m];
                                if (vm == vm) {                  // a NaN month of this point: skipped
                                    sw[m] += w; swv[m] += w * vm;
                                    if (exact && !has_hit[m]) { has_hit[m] = true; hit[m] = vm; }
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (i < ntarg) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m < nm) {
                double r = __builtin_nan("");                    // null_value (:1041)
                if (has_hit[m]) r = hit[m];
                else if (sw[m] > 0.0) r = swv[m] / sw[m];
                out[(long long)m * ntarg + i] = r;
            }
        }
    }
}

// float64 -> float32 (round to nearest even, the conversion numpy's astype does), two elements per lane, optionally byte-reversed
// -------------------------------------------------------------------------------------
// Planar "metre" coordinates of the ocean-grid interpolation (functions.py:958-975, 1010-1023: three pyproj Geod.inv
// calls per point) on the GPU: the arithmetic of pgw4era5_amd/geodesy.py (Vincenty's series arranged so that nothing
// iterates to failure - meridian arc and over-the-pole length from the direct series with azimuth 0, the geodesic between
// two points of one parallel by bisection on the departure azimuth), one point per thread.  Same formulas and iteration
// count as the host form; sin / cos / atan2 are the device library's, so results agree to the last bits (~1e-9 m), not
// bit for bit.
// -------------------------------------------------------------------------------------
namespace geo {
constexpr double A = 6378137.0, F = 1.0 / 298.257223563, B = A * (1.0 - F), PI = 3.14159265358979323846;
struct Series { double a, b, c; };
__device__ __forceinline__ Series series(double cos2_alpha) {
    const double u2 = cos2_alpha * (A * A - B * B) / (B * B);
    Series s;
    s.a = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)));
    s.b = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)));
    s.c = F / 16 * cos2_alpha * (4 + F * (4 - 3 * cos2_alpha));
    return s;
}
__device__ __forceinline__ double arc(double sigma, double cos_2sm, double a, double b) {
    const double sin_s = sin(sigma), cos_s = cos(sigma);
    const double dsig = b * sin_s * (cos_2sm + b / 4 * (cos_s * (-1 + 2 * cos_2sm * cos_2sm) -
                                                        b / 6 * cos_2sm * (-3 + 4 * sin_s * sin_s) * (-3 + 4 * cos_2sm * cos_2sm)));
    return B * a * (sigma - dsig);
}
__device__ __forceinline__ double reduced_latitude(double lat_deg) { return atan((1.0 - F) * tan(lat_deg * (PI / 180.0))); }
__device__ __forceinline__ double meridian_arc(double lat_deg) {
    double U = fabs(reduced_latitude(lat_deg));
    if (fabs(lat_deg) >= 90.0) U = 0.5 * PI;
    const Series s = series(1.0);
    return arc(U, cos(U), s.a, s.b);
}
// leave reduced latitude U (>= 0) with azimuth a1 in [0, pi/2], travel until latitude U is reached again
__device__ __forceinline__ void symmetric(double U, double a1, double &L, double &len) {
    const double sinU = sin(U), cosU = cos(U), sin_a1 = sin(a1), cos_a1 = cos(a1);
    const double s1 = atan2(sinU, cosU * cos_a1);
    const double sigma = PI - 2.0 * s1;
    const double sin_alpha = cosU * sin_a1;
    const double cos2_alpha = 1.0 - sin_alpha * sin_alpha;
    const Series s = series(cos2_alpha);
    const double sin_s = sin(sigma), cos_s = cos(sigma);
    double omega = atan2(sin_s * sin_a1, cosU * cos_s - sinU * sin_s * cos_a1);
    if (omega < 0) omega += 2 * PI;
    const double cos_2sm = -1.0;
    L = omega - (1 - s.c) * F * sin_alpha * (sigma + s.c * sin_s * (cos_2sm + s.c * cos_s * (-1 + 2 * cos_2sm * cos_2sm)));
    len = arc(sigma, cos_2sm, s.a, s.b);
}
}  // namespace geo

// lat, lon [deg] (lon folded to (-180, 180] by the caller) -> lat_m, lon_m, lon_off; quarter = meridian_arc(90)
__global__ __launch_bounds__(BLOCK) void k_planar_metres(long long n, const double *__restrict__ lat, const double *__restrict__ lon,
                                                        double *__restrict__ lat_m, double *__restrict__ lon_m,
                                                        double *__restrict__ lon_off) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double la = lat[i], lo = lon[i];
    const double quarter = geo::meridian_arc(90.0);
    const double arc_m = geo::meridian_arc(la);
    const double over_pole = 2.0 * (quarter - arc_m);
    const double sgn_la = (la > 0) - (la < 0), sgn_lo = (lo > 0) - (lo < 0);
    const double dl = fabs(lo), U = fabs(geo::reduced_latitude(la)), Ls = dl * (geo::PI / 180.0);
    double a_lo = 0.0, a_hi = 0.5 * geo::PI, L, len;
    for (int it = 0; it < 70; ++it) {
        const double mid = 0.5 * (a_lo + a_hi);
        geo::symmetric(U, mid, L, len);
        if (L > Ls) a_lo = mid; else a_hi = mid;
    }
    geo::symmetric(U, 0.5 * (a_lo + a_hi), L, len);
    double s = len;
    if (U < 1e-15 && Ls <= (1.0 - geo::F) * geo::PI) s = geo::A * Ls;      // the equator itself
    if (dl >= 180.0) s = over_pole;
    if (fabs(la) >= 90.0) s = 0.0;
    if (dl == 0.0) s = 0.0;
    if (la != la || lo != lo) s = __builtin_nan("");
    lat_m[i] = arc_m * sgn_la;
    lon_m[i] = s * sgn_lo;
    lon_off[i] = over_pole;
}

template <bool SWAP>
__global__ __launch_bounds__(BLOCK) void k_narrow_f64_f32(long long n2, long long n, const double *__restrict__ src, unsigned int *__restrict__ dst) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src) + i);
        u2 o;
        o.x = __float_as_uint((float)v.x); o.y = __float_as_uint((float)v.y);
        if (SWAP) { o.x = __builtin_bswap32(o.x); o.y = __builtin_bswap32(o.y); }
        __builtin_nontemporal_store(o, reinterpret_cast<u2 *>(dst) + i);
    }
    for (long long e = 2 * n2 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        unsigned int o = __float_as_uint((float)src[e]);
        dst[e] = SWAP ? __builtin_bswap32(o) : o;
    }
}

// Placement probe (pgw_placement_probe): the access pattern of the column kernels with no arithmetic - one thread per column,
// blocks of 128 columns, `ns` read streams and `nd` write streams of (row, column) float64 arrays, two rows per step with the
// next step's rows requested one step ahead, streaming loads and stores.  What it measures: the rate this set of arrays gets
// WHERE hipMalloc put them - on an MI355X an array that is written while another one of the same stretch of physical memory
// is read gets 5.0-5.4 TB/s, arrays of different stretches 5.7-6.2 (tools/micro/pair_matrix.hip).
struct ProbeStreams { const double *src[4]; double *dst[4]; int ns, nd; };
__global__ __launch_bounds__(128) void k_placement_probe(long long rows, long long ncol, ProbeStreams s) {
    const long long c = (long long)blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    auto row_sum = [&](long long r) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < s.ns) a += __builtin_nontemporal_load(s.src[i] + r * ncol + c);
        return a;
    };
    auto row_put = [&](long long r, double v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < s.nd) __builtin_nontemporal_store(v, s.dst[i] + r * ncol + c);
    };
    double a = row_sum(0), b = rows > 1 ? row_sum(1) : 0.0;
    long long r = 0;
    for (; r + 1 < rows; r += 2) {
        double a2 = 0.0, b2 = 0.0;
        if (r + 2 < rows) a2 = row_sum(r + 2);
        if (r + 3 < rows) b2 = row_sum(r + 3);
        row_put(r, a); row_put(r + 1, b);
        a = a2; b = b2;
    }
    if (r < rows) row_put(r, a);
}

// Byte-order conversion of a field (NetCDF classic data are big-endian): every 4- or 8-byte element of `src` is
// written byte-reversed to `dst` (in place allowed), 16 B per lane, grid-stride.  HBM-bound: 2 x n x W bytes.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_byteswap(long long n16, long long n, const uint4 *src, uint4 *dst) {   // may alias
    const long long stride = (long long)gridDim.x * blockDim.x;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src) + i);   // read once, written once
        uint4 v = make_uint4(t.x, t.y, t.z, t.w);
        if (W == 4) {
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
            v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
        } else {
            unsigned int a = __builtin_bswap32(v.y), b = __builtin_bswap32(v.x);
            unsigned int c = __builtin_bswap32(v.w), d = __builtin_bswap32(v.z);
            v.x = a; v.y = b; v.z = c; v.w = d;
        }
        { u4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(dst) + i); }
    }
    // tail elements (n not a multiple of 16 / W) and the unaligned case (n16 == 0): element by element
    const long long done = n16 * (16 / W);
    for (long long e = done + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        if (W == 4) {
            ((unsigned int *)dst)[e] = __builtin_bswap32(((const unsigned int *)src)[e]);
        } else {
            ((unsigned long long *)dst)[e] = __builtin_bswap64(((const unsigned long long *)src)[e]);
        }
    }
}

// pgw_log / pgw_log_tab over an array (diagnostic entry pgw_test_log; tests compare them with numpy's log)
__global__ void k_test_log(long long n, const double *__restrict__ in, double *__restrict__ out, int table) {
    __shared__ double s_tab[2 * LOG_TABLE_N];
    stage_log_table(s_tab, blockDim.x);
    __syncthreads();
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table ? pgw_log_tab(in[i], s_tab) : pgw_log(in[i]);
}

// pgw_exp and the device library's exp over an array (diagnostic entry pgw_test_exp): out[i] = pgw_exp(in[i]),
// ref[i] = exp(in[i])
__global__ void k_test_exp(long long n, const double *__restrict__ in, double *__restrict__ out, double *__restrict__ ref) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = pgw_exp(in[i]); ref[i] = exp(in[i]); }
}

// RELHUM of a float32 ERA state in reference-dtype mode (diagnostic entry pgw_test_rh_f32): the fast form the quad kernel
// uses and the literal expression, side by side
__global__ void k_test_rh_f32(long long n, const float *__restrict__ hus, const double *__restrict__ pa,
                              const float *__restrict__ ta, double *__restrict__ out, double *__restrict__ lit,
                              float *__restrict__ es, float *__restrict__ es_lit) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = q_to_rh_f32(hus[i], pa[i], ta[i]);
        lit[i] = q_to_rh_f32_literal(hus[i], pa[i], ta[i]);
        es[i] = esat_mixed_f32(ta[i]);
        es_lit[i] = esat_mixed_f32_literal(ta[i]);
    }
}

// SharedDivisor over arrays (diagnostic entry pgw_test_shared_div; tests compare it with IEEE division)
__global__ void k_test_shared_div(long long n, const double *__restrict__ num, const double *__restrict__ den,
                                  double *__restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (den[i] == 100.0) ? div_by_100(num[i]) : SharedDivisor(den[i]).divide(num[i]);
}

// ln() of a small table with the device log (so table entries and per-column logs come from
// the same implementation)
__global__ void k_log_table(int n, const double *in, double *out) {
    __shared__ double s_tab[2 * LOG_TABLE_N];
    stage_log_table(s_tab, blockDim.x);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pgw_log_tab(in[i], s_tab);
}

}  // namespace pgw
