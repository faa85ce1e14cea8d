// pgw_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Column kernels: one thread owns V adjacent (lat,lon) columns (V*sizeof(T) = 16 B where the
// grid allows it), consecutive lanes own consecutive groups, so every level access of a wave is
// one fully coalesced 1 KiB row segment of the (time, lev, lat, lon) C-order array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pgw_log_table.h"

namespace pgw {

constexpr double CON_RD = 287.05;      // constants.py:3-7 of the reference
constexpr double CON_G = 9.80665;
constexpr double CON_MW_MD = 0.622;

constexpr int BLOCK = 256;             // 4 waves of 64

// device-resident status block of a context (zeroed before a launch that can report)
struct DevStatus {
    int code;                          // first pgw_status reported by a kernel
    int nan_seen;                      // bit 0: NaN in target min, bit 1: NaN in source min
    unsigned long long col;            // min offending column (init ~0ull)
    unsigned long long max_bits;       // max |err| as ordered bits of a non-negative double
    unsigned long long valid;          // number of non-NaN |err| contributions (>0 flag)
    unsigned long long min_targ_bits;  // min target pressure (bits of positive double)
    unsigned long long min_src_bits;   // min source pressure
    unsigned long long levels_touched; // sum over columns of levels read (early-exit kernels)
};

template <typename T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

template <typename T, int V>
__device__ __forceinline__ void loadv(const T *__restrict__ p, double (&out)[V]) {
    Pack<T, V> t = *reinterpret_cast<const Pack<T, V> *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = (double)t.v[i];
}

template <typename T, int V>
__device__ __forceinline__ void storev(T *__restrict__ p, const double (&in)[V]) {
    Pack<T, V> t;
#pragma unroll
    for (int i = 0; i < V; ++i) t.v[i] = (T)in[i];
    *reinterpret_cast<Pack<T, V> *>(p) = t;
}

// streaming (non-temporal) forms of loadv / storev, see ld_off_nt below
template <typename T, int V> struct VecOf { typedef T type __attribute__((ext_vector_type(V))); };
template <typename T, int V>
__device__ __forceinline__ void loadv_nt(const T *__restrict__ p, double (&out)[V]) {
    if constexpr (V == 1) out[0] = (double)__builtin_nontemporal_load(p);
    else {
        typename VecOf<T, V>::type t = __builtin_nontemporal_load(reinterpret_cast<const typename VecOf<T, V>::type *>(p));
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = (double)t[i];
    }
}
template <typename T, int V>
__device__ __forceinline__ void storev_nt(T *__restrict__ p, const double (&in)[V]) {
    if constexpr (V == 1) __builtin_nontemporal_store((T)in[0], p);
    else {
        typename VecOf<T, V>::type t;
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = (T)in[i];
        __builtin_nontemporal_store(t, reinterpret_cast<typename VecOf<T, V>::type *>(p));
    }
}

// ---- addressing with 32-bit byte offsets ---------------------------------------------------
// `base + (long long)index` makes the compiler carry a 64-bit address per array and lane (two VGPRs and a 64-bit
// add per access).  When a whole array is smaller than 4 GiB the element can be addressed as uniform base (SGPR
// pair) + 32-bit byte offset in one VGPR, which is the form global_load / global_store take natively (saddr +
// voffset): one offset register serves every array of the same shape.  O is the offset type: unsigned int for
// that form, unsigned long long for arrays of 4 GiB and more.
using boff32 = unsigned int;          // byte offset types for ld_off / st_off
using boff64 = unsigned long long;
template <typename T, typename O>
__device__ __forceinline__ T ld_off(const T *__restrict__ base, O byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T, typename O>
__device__ __forceinline__ void st_off(T *__restrict__ base, O byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// Streaming forms (the `nt` bit of global_load / global_store): data that is touched once per launch - the ERA fields in,
// the PGW fields out - should not displace what IS re-read (delta records, the loop's rows) from L2 / Infinity Cache, and a
// mixed read + write stream runs ~10 % faster with them (tools/micro/chunk_pattern.hip: 5.0 -> 5.6 TB/s at 128 threads).
template <typename T, typename O>
__device__ __forceinline__ T ld_off_nt(const T *__restrict__ base, O byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off));
}
template <typename T, typename O>
__device__ __forceinline__ void st_off_nt(T *__restrict__ base, O byte_off, T v) {
    __builtin_nontemporal_store(v, reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off));
}

// first reported code wins (kernels of a stream run in order, so an earlier kernel's error outlives later ones);
// the column is the smallest one that reported THAT code
__device__ __forceinline__ void report(DevStatus *st, int code, long long col) {
    int prev = atomicCAS(&st->code, 0, code);
    if (prev == 0 || prev == code) atomicMin(&st->col, (unsigned long long)col);
}

// ---- division by a value that many numerators share ---------------------------------------
// The compiler expands an IEEE fp64 division into v_div_scale x2, v_rcp, two Newton steps on the reciprocal,
// q = n*r, one residual correction (v_div_fmas) and v_div_fixup: 11 instructions.  When many numerators are
// divided by the same d, the reciprocal part (v_rcp + 4 FMAs) is computed once and each quotient costs the last
// three instructions.  The arithmetic is the compiler's own sequence, so quotients are bit-identical whenever
// v_div_scale would not rescale (finite operands whose exponents are far from the denormal / overflow range -
// true for every divisor this is used with: grid spacings and ln-pressure intervals); NaN numerators propagate,
// a -0 numerator gives +0 (v_div_fixup would restore the sign; the value is the same).
// Not for divisors that can be 0, infinite or denormal-scale.
struct SharedDivisor {
    double d, r;
    __device__ __forceinline__ explicit SharedDivisor(double den) : d(den) {
        double y = __builtin_amdgcn_rcp(den);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
        r = y;
    }
    // divisor known at compile time: r must be the double nearest to 1/den (then the quotient below is the
    // correctly rounded one - Markstein's theorem - and therefore the same bits as any IEEE division)
    __device__ __forceinline__ constexpr SharedDivisor(double den, double rcp_nearest) : d(den), r(rcp_nearest) {}
    __device__ __forceinline__ double divide(double n) const {
        double q = n * r;
        return __builtin_fma(__builtin_fma(-d, q, n), r, q);
    }
};

// n / d by the same steps for a divisor used once: the compiler's IEEE fp64 division without v_div_scale (x2) and
// v_div_fixup - 8 instead of 11 instructions, the correctly rounded quotient (the same bits) whenever neither operand
// needs scaling and d is finite and non-zero; used where the divisor is a physical quantity of known range
// (a temperature offset, a pressure minus a vapour-pressure fraction, 0.622 + 0.378 q, e_sat).  NaN operands give
// NaN like the division; d = 0 or inf gives NaN where the division gives inf / 0.
__device__ __forceinline__ double div_ns(double n, double d) { return SharedDivisor(d).divide(n); }

// ---- natural logarithm ------------------------------------------------------------------
// Every kernel on the path takes one ln(p) per level and column, and with fp64 vector math at
// half rate the generic ocml log (~65 instructions, double-double internals, denormal /
// negative / inf handling) made the column kernels VALU-bound.  pgw_log is the classic
// fdlibm __ieee754_log kernel (argument reduction to [sqrt(1/2), sqrt(2)), s = f/(2+f),
// degree-14 even polynomial; documented error < 1 ulp) for positive normal finite x, with the
// quotient formed by v_rcp_f64 + two Newton steps + one residual correction; everything else
// (0, negative, denormal, inf, NaN) goes to the ocml log.  One implementation serves table
// entries and per-column values, so `src_x == targ_x` comparisons (functions.py:540) are
// consistent.
// keeps a rarely taken block a real (wave-uniform) branch: volatile asm cannot be speculated, so
// the compiler does not if-convert the block into "compute both sides and select"
__device__ __forceinline__ double no_speculate(double x) {
    asm volatile("" : "+v"(x));
    return x;
}

// one Horner step a*b + c as the three-address v_fma_f64, c (a coefficient that stays live) in its own register
__device__ __forceinline__ double fma3(double a, double b, double c) {
#ifdef PGW_NO_FMA3
    return __builtin_fma(a, b, c);
#else
    double d;
#ifdef PGW_FMA3_SGPR
    // A/B knob: the coefficient in a scalar register pair (the one constant-bus operand a VOP3 instruction may take) - no
    // vector register is held for it, s_mov_b32 literals on the scalar unit materialise it.  Every file-path kernel then
    // compiles without scratch (k_delta_quad<float, double, REF>: 127 VGPRs / 0 B instead of 128 / 180 B; k_reinterp_pair
    // with e: 139 / 0 instead of 168 / 116) and runs in the same time (round 3, same box: 2.13 / 2.14, 1.61 / 1.60,
    // 1.594 / 1.596 ms) - the spilled values are constants of cold paths.  Not the default.
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
#else
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
#endif
    return d;
#endif
}
// F3: Horner steps as three-address FMAs (fma3) - for kernels whose register budget is not the binding one
template <bool F3, bool CHECK = true>
__device__ __forceinline__ double pgw_log_impl(double x) {
    if (CHECK && __builtin_expect(!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308), 0))
        return log(no_speculate(x));
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int k = __builtin_amdgcn_frexp_exp(x);                 // x = m * 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 0.70710678118654752440) { m = m * 2.0; k -= 1; }
    double f = m - 1.0;                                    // exact
    double d = 2.0 + f;                                    // in [1.70, 2.42)
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double s = f * r;
    s = __builtin_fma(__builtin_fma(-d, s, f), r, s);      // s = f / (2 + f), correctly rounded up to ~0.5 ulp
    double dk = (double)k;
    double z = s * s;
    double w = z * z;
    double t1, t2;
    if (F3) {
        t1 = w * fma3(w, fma3(w, Lg6, Lg4), Lg2);
        t2 = z * fma3(w, fma3(w, fma3(w, Lg7, Lg5), Lg3), Lg1);
    } else {
        t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
        t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    }
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f);
}
__device__ __forceinline__ double pgw_log(double x) { return pgw_log_impl<false>(x); }
__device__ __forceinline__ double pgw_log_f3(double x) { return pgw_log_impl<true>(x); }    // same bits

// ---- table-driven natural logarithm for the level loops ------------------------------------
// The level loops of the surface-pressure iteration are bound by fp64 issue (94 % of the SIMD cycles), and the
// fdlibm kernel above is ~40 of their ~90 instructions per level.  Tang-style table method (the construction of
// glibc's / ARM optimized-routines' log): x = 2^k z, z in [0.6875, 1.375) cut into 128 intervals by the top mantissa
// bits; per interval a double invc ~ 1/centre and logc = -log(invc), the pair chosen (tools/gen_log_table.py) such that
// logc is within 2.2e-19 of the exact value; r = fma(z, invc, -1) (|r| < 0.0040, one rounding), and
//     log x = (k ln2_hi + logc) + r + [r^2 (A0 + A1 r + ... + A5 r^5) + k ln2_lo]     (truncation r^8/8 < 1e-20)
// with the leading sum split exactly into hi + lo.  ~25 instructions and one 16-B LDS read.  The two intervals next to
// z = 1 with k = 0 (results near zero, where the table's absolute accuracy is not a relative one) and everything
// outside the positive normal range go to pgw_log.  tests/...::test_device_log_accuracy: <= 1 ulp of numpy's log.
// The table lives in LDS (stage_log_table): lanes index it with their own interval.
__device__ const double LOG_TABLE_DEV[2 * LOG_TABLE_N] = {
#define PGW_LOG_TABLE_BODY
#include "pgw_log_table.h"
#undef PGW_LOG_TABLE_BODY
};
__device__ __forceinline__ void stage_log_table(double *lds, int nthreads) {      // caller synchronises afterwards
    for (int i = threadIdx.x; i < 2 * LOG_TABLE_N; i += nthreads) lds[i] = LOG_TABLE_DEV[i];
}
__device__ __forceinline__ double pgw_log_tab(double x, const double *tab) {
    const unsigned int hi = (unsigned int)__double2hiint(x);
    const unsigned int thi = hi - 0x3FE60000u;                      // (bits(x) - OFF) >> 32; OFF's low word is 0: no borrow
    // positive normal finite x  <=>  0x00100000 <= hi < 0x7FF00000
    const int i = (int)((thi >> 13) & 127u);
    const int k = (int)thi >> 20;
    // one divergent region for both exceptions: not a positive normal finite number (library log), or within two table
    // intervals of 1, where r = z/c - 1 cancels (fdlibm kernel; pgw_log_impl sends non-normal arguments to the library itself)
    if (__builtin_expect(!(hi - 0x00100000u < 0x7FE00000u) || (k == 0 && (i == 79 || i == 80)), 0))
        return pgw_log_impl<true, true>(no_speculate(x));
    const double z = __hiloint2double((int)(hi - (thi & 0xFFF00000u)), __double2loint(x));
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, 0x1.62e42fefa3800p-1, logc);       // k ln2_hi is exact (11 trailing zero bits)
    const double h = w + r;
    const double lo = __builtin_fma(kd, 0x1.ef35793c76730p-45, (w - h) + r);
    const double r2 = r * r;
    double p = fma3(r, 0x1.2492492492492p-3, -0x1.5555555555555p-3);       //  1/7, -1/6
    p = fma3(r, p, 0x1.999999999999ap-3);                                   //  1/5
    p = fma3(r, p, -0x1.0p-2);                                              // -1/4
    p = fma3(r, p, 0x1.5555555555555p-2);                                   //  1/3
    p = fma3(r, p, -0x1.0p-1);                                              // -1/2
    return __builtin_fma(r2, p, lo) + h;
}

// ---- exponential --------------------------------------------------------------------------
// The same arithmetic as the device library's exp(double) - n = rint(x log2 e), r = x - n ln2 (two FMAs),
// degree-11 polynomial in r (Horner), ldexp, the library's two range selects - written out with explicit FMAs.
// Same operations and constants, hence the same bits as exp() (tests/...::test_device_exp_is_library_exp); the point
// is code generation: for most of the inlined library instances in the delta kernels the compiler forms each Horner
// step as v_mov_b64 (copy of the coefficient) + v_fmac, two instructions, where one three-address v_fma_f64 does.
template <bool RANGE>
__device__ __forceinline__ double pgw_exp_impl(double x) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-0x1.62e42fefa39efp-1, n, x);
    r = __builtin_fma(-0x1.abc9e3b39803fp-56, n, r);
    double p = fma3(0x1.ade156a5dcb37p-26, r, 0x1.28af3fca7ab0cp-22);
    p = fma3(r, p, 0x1.71dee623fde64p-19);
    p = fma3(r, p, 0x1.a01997c89e6b0p-16);
    p = fma3(r, p, 0x1.a01a014761f6ep-13);
    p = fma3(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma3(r, p, 0x1.1111111122322p-7);
    p = fma3(r, p, 0x1.55555555502a1p-5);
    p = fma3(r, p, 0x1.5555555555511p-3);
    p = fma3(r, p, 0x1.000000000000bp-1);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    double v = __builtin_ldexp(p, (int)n);
    if (RANGE) {
        v = (1024.0 < x) ? __builtin_inf() : v;      // NaN compares false: the NaN from the arithmetic is kept
        v = (x < -1075.0) ? 0.0 : v;
    }
    return v;
}
__device__ __forceinline__ double pgw_exp(double x) { return pgw_exp_impl<true>(x); }
// without the library's two range selects: the same bits for every finite x (v_ldexp_f64 saturates to 0 / inf by itself,
// the conversion of n saturates too) and for NaN; only x = +-inf gives NaN instead of inf / 0.  For arguments of known range.
__device__ __forceinline__ double pgw_exp_finite(double x) { return pgw_exp_impl<false>(x); }

// ---- humidity thermodynamics (functions.py:58-125), operation order as written there ----
__device__ __forceinline__ double esat_water(double ta) {   // :74-89 water
    return 611.21 * pgw_exp(div_ns(17.502 * (ta - 273.16), ta - 32.19));
}
__device__ __forceinline__ double esat_ice(double ta) {     // :74-89 ice (a4 = -0.7)
    return 611.21 * pgw_exp(22.587 * (ta - 273.16) / (ta - (-0.7)));
}
// :91-105.  alpha = 1 (T>=T0), 0 (T<=Ti), ((T-Ti)/(T0-Ti))^2 in between, NaN for NaN T.
// alpha*e_w + (1-alpha)*e_i is evaluated in full only in the mixed range; for alpha in {0,1}
// the dropped term is exactly 0*finite = 0 for every finite T (e_w, e_i are finite for all
// T > 32.19 K), so the value is unchanged.
// One unconditional division + exp evaluates the phase every temperature needs (water for
// T >= T0, ice otherwise; the coefficients are selected, not the results), and only lanes in the
// mixed range take a real branch (no_speculate) for the second one.
// the temperatures the one-exponential value does not serve: mixed phase, unphysically cold, NaN
__device__ __forceinline__ double esat_special(double ta, double e1) {
    const double T0 = 273.16, Ti = 250.16;
    if (ta < T0 && ta > Ti) {                                    // mixed phase
        double ew = esat_water(ta);
        // (ta - Ti) / (T0 - Ti): T0 - Ti = 273.16 - 250.16 = 0x1.7000000000008p+4 in double (not 23); with the double nearest to
        // its reciprocal the two-step quotient is the correctly rounded one (Markstein), i.e. the bits of the division
        static_assert(273.16 - 250.16 == 0x1.7000000000008p+4, "T0 - Ti");
        double r = SharedDivisor(0x1.7000000000008p+4, 0x1.642c8590b215cp-5).divide(ta - Ti);
        double alpha = r * r;                                    // np.power(x, 2.) == x*x
        return alpha * ew + (1 - alpha) * e1;
    }
    if (ta <= 40.0) {                                            // unphysical cold: the literal expression, 0 * e_w may be NaN / inf
        const double ew = 611.21 * pgw_exp(17.502 * (ta - 273.16) / (ta - 32.19));     // IEEE division: the divisor is 0 at 32.19 K
        const double ei = 611.21 * pgw_exp(22.587 * (ta - 273.16) / (ta - (-0.7)));    // and at -0.7 K: exp(-inf) = 0
        return 0.0 * ew + 1.0 * ei;
    }
    return __builtin_nan("");                                    // NaN temperature
}
__device__ __forceinline__ double esat_mixed(double ta) {
    const double T0 = 273.16, Ti = 250.16;
    const bool warm = (ta >= T0);
    const double a3 = warm ? 17.502 : 22.587;
    const double a4 = warm ? 32.19 : -0.7;
    // e_w if warm else e_i (NaN for NaN ta); |argument| < 130 for every T the value is used for: no range selects
    double e1 = 611.21 * pgw_exp_finite(div_ns(a3 * (ta - T0), ta - a4));
    // ONE divergent region per call (every branch costs scalar instructions and issue slots the few resident waves of the
    // delta kernels cannot hide): e1 is the answer for T >= T0 and for 40 K < T <= Ti
    if (__builtin_expect(!(warm || (ta <= Ti && ta > 40.0)), 0)) e1 = esat_special(no_speculate(ta), e1);
    return e1;
}
__device__ __forceinline__ double q_to_e(double hus, double pa) {          // :58-64
    return div_ns(hus * pa, CON_MW_MD + 0.378 * hus);
}
__device__ __forceinline__ double e_to_q(double vapp, double pa) {         // :66-72
    return CON_MW_MD * vapp / (pa - (1 - CON_MW_MD) * vapp);
}
// the same with the scale-free quotient, for pressures of the model's levels (pa - 0.378 e is a pressure of known range)
__device__ __forceinline__ double e_to_q_ns(double vapp, double pa) {
    return div_ns(CON_MW_MD * vapp, pa - (1 - CON_MW_MD) * vapp);
}
__device__ __forceinline__ double q_to_rh(double hus, double pa, double ta) {   // :107-116
    return div_ns(q_to_e(hus, pa), esat_mixed(ta)) * 100;
}
// ---- reference-dtype mode: RELHUM of a float32 ERA state as numpy's promotion evaluates functions.py:58-116 ----
// e_sat of a float32 temperature is float32 throughout (alpha = full_like(ta): :95-98; a1*np.exp(a3*(ta-T0)/(ta-a4)):
// :88 with the python floats taken as float32 scalars); every operation below is one IEEE float32 operation except
// expf (<= 1 ulp here; numpy's SIMD float32 exp differs from it by a few float32 ulp - the reference's own last bits).
__device__ __forceinline__ float esat_x_f32(float ta, float a3, float a4) {
    return 611.21f * expf(a3 * (ta - 273.16f) / (ta - a4));
}
// the expression as written (:91-105): both phases, IEEE divisions, the library's expf
__device__ __forceinline__ float esat_mixed_f32_literal(float ta) {
    const float T0 = 273.16f, Ti = 250.16f;
    float alpha = __builtin_nanf("");
    if (ta >= T0) alpha = 1.0f;
    if (ta <= Ti) alpha = 0.0f;
    if (ta < T0 && ta > Ti) { const float r = (ta - Ti) / 23.0f; alpha = r * r; }   // (T0 - Ti) = 23.000000000000028 -> 23.0f
    return alpha * esat_x_f32(ta, 17.502f, 32.19f) + (1.0f - alpha) * esat_x_f32(ta, 22.587f, -0.7f);
}
// n / d in float32 by the compiler's own IEEE sequence without v_div_scale (x2) and v_div_fixup: v_rcp_f32, one Newton
// step, q = n r, two residual corrections - 8 instead of 11 instructions and the same bits whenever neither operand
// needs scaling (div_ns above is the float64 form).  Used for a3 (T - T0) / (T - a4) with 60 K < T < 10^4 K.
__device__ __forceinline__ float div_ns_f32(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
// the device library's expf arithmetic - x log2(e) as a float32 pair, n = rint, v_exp_f32 of the remainder, ldexp -
// without its two range selects (callers keep |x| < 87): the same bits as expf() there
// (tests/...::test_reference_mode_esat_f32_fast_path_is_the_literal_expression)
__device__ __forceinline__ float pgw_expf_core(float x) {
    const float log2e = 0x1.715476p+0f;                        // 0x3fb8aa3b
    const float ph = x * log2e;
    float pl = __builtin_fmaf(x, log2e, -ph);
    pl = __builtin_fmaf(x, __uint_as_float(0x32a5705fu), pl);   // + x * (log2(e) - float(log2(e)))
    const float n = __builtin_rintf(ph);
    const float a = (ph - n) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)n);
}
__device__ __forceinline__ float no_speculate_f(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
// alpha e_w + (1 - alpha) e_i for alpha in {0, 1} is exactly the phase the temperature needs: the dropped term is
// 0 * finite = +0 and x + 0 = x in float32 as in float64 (esat_mixed above).  One division and one exponential for
// T >= T0 (water) and for 60 K < T <= Ti (ice); mixed phase, colder than 60 K (e_i approaches the float32 denormals,
// e_w overflows below 32.19 K), hotter than 10^4 K and NaN take the literal expression behind ONE divergent region.
__device__ __forceinline__ float esat_mixed_f32(float ta) {
#ifdef PGW_ESAT_F32_LITERAL                                      // A/B knob: round 2's form
    return esat_mixed_f32_literal(ta);
#endif
    const float T0 = 273.16f, Ti = 250.16f;
    const bool warm = (ta >= T0);
    const float a3 = warm ? 17.502f : 22.587f;
    const float a4 = warm ? 32.19f : -0.7f;
    float e1 = 611.21f * pgw_expf_core(div_ns_f32(a3 * (ta - T0), ta - a4));
    if (__builtin_expect(!((warm && ta < 1.0e4f) || (ta <= Ti && ta > 60.0f)), 0)) e1 = esat_mixed_f32_literal(no_speculate_f(ta));
    return e1;
}
// hus float32, pa float64 (ak/bk are float64): hus*pa is float64, the denominator CON_MW_MD + 0.378*hus float32 (:63).
// The two float64 quotients through div_ns like q_to_e / q_to_rh above (divisors of known range: 0.622 + 0.378 q, e_sat).
__device__ __forceinline__ double q_to_rh_f32_literal(float hus, double pa, float ta);
__device__ __forceinline__ double q_to_rh_f32(float hus, double pa, float ta) {
#ifdef PGW_ESAT_F32_LITERAL
    return q_to_rh_f32_literal(hus, pa, ta);
#endif
    const double vapp = div_ns((double)hus * pa, (double)(0.622f + 0.378f * hus));
    return div_ns(vapp, (double)esat_mixed_f32(ta)) * 100;
}
__device__ __forceinline__ double q_to_rh_f32_literal(float hus, double pa, float ta) {
    const double vapp = (double)hus * pa / (double)(0.622f + 0.378f * hus);
    return (vapp / (double)esat_mixed_f32_literal(ta)) * 100;
}

__device__ __forceinline__ double div_by_100(double x) { return SharedDivisor(100.0, 0.01).divide(x); }
__device__ __forceinline__ double rh_to_e(double hur, double ta) {         // :123
    return div_by_100(hur) * esat_mixed(ta);
}
__device__ __forceinline__ double rh_to_q(double hur, double pa, double ta) {   // :118-125
    return e_to_q(rh_to_e(hur, ta), pa);
}

// functions.py:135  pa_hl.where(pa_hl > 0, 0.0001): NaN > 0 is False -> 1e-4
__device__ __forceinline__ double fix_p(double p) { return (p > 0) ? p : 0.0001; }

// ordered-bits view of a non-negative (or any positive) double for atomicMax/atomicMin
__device__ __forceinline__ unsigned long long dbits(double x) { return (unsigned long long)__double_as_longlong(x); }

__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
    return x;
}
__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmin(x, __shfl_xor(x, off, 64));
    return x;
}

}  // namespace pgw
