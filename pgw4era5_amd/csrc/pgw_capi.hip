// pgw_capi.hip -- C-ABI (include/pgw_hip.h) over the gfx950 kernels in pgw_kernels.h.
// Built as libpgw_hip.so with hipcc --offload-arch=gfx950 (pgw4era5_amd/csrc/Makefile).
#include "../../include/pgw_hip.h"
#include "pgw_kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

using namespace pgw;

struct ProfRec { int kid; hipEvent_t e0, e1; };

struct pgw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    long long err_col = -1;
    DevStatus *d_status = nullptr;     // device; [1] = alternate block of the loop passes (cleared by the pass before);
                                       // [2 .. 2 + MULTI_MAX_PASS) = per-pass blocks of the multi-pass loop kernel
    DevStatus *h_status = nullptr;     // pinned host mirror ([0]) + [1 .. 1 + MULTI_MAX_PASS]: read-back of the multi-pass launch,
                                       // [2 + MULTI_MAX_PASS ...): cleared template for the per-pass blocks
    int last_passes_launched = 0;
    pgw_reduce_max_fn reduce_fn = nullptr;      // latitude-band sharding of one file: MAX over the ranks (pgw_set_reduce_hook)
    void *reduce_user = nullptr;
    // protocol state of the band reduces of the file in progress (pgw_step03_file): how many have been made, how long the
    // next one is (passes of the next loop launch), and whether the error being returned is one every band has seen
    int band_reduces = 0, band_next_np = 0;
    bool band_agreed = false;
    // options (pgw_set_option; defaults from the environment, read ONCE in pgw_ctx_create)
    int opt[PGW_OPT_COUNT];
    // vertical grid
    int nlev = 0;
    double *d_levels = nullptr;        // ak | bk | akm | bkm
    std::vector<double> h_akm, h_bkm;
    double h_akN = 0.0, h_bkN = 0.0;   // ak[nlev], bk[nlev] (surface half level)
    double ps_mono_min = 0.0;
    int n_pure = 0;                    // leading full levels with bkm == 0 (pure-pressure levels)
    // plev table cache for vert_interp_delta
    std::vector<double> plev_key;
    PlevTable plev_tab;
    double *d_small = nullptr;         // small scratch (tables), 64 KiB
    // named workspaces grown on demand
    void *ws[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ws_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // profiler
    bool prof_on = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> event_pool;      // recycled profiling events (create/destroy per launch is slow)
    long long prof_count[PGW_K_COUNT];
    double prof_ms[PGW_K_COUNT];
    hipEvent_t t0 = nullptr, t1 = nullptr;
    unsigned long long last_levels_touched = 0;
};

static const size_t SMALL_BYTES = 64 * 1024;

static int fail(pgw_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, PGW_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                       \
    } while (0)

#define NEED(ctx, cond, msg)                                                                       \
    do {                                                                                           \
        if (!(cond)) return fail(ctx, PGW_ERR_ARG, "%s: %s", __func__, msg);                       \
    } while (0)

static const char *status_text(int code) {
    switch (code) {
        case PGW_ERR_SRC_NOT_ASCENDING: return "Source pressure values must be ascending!";
        case PGW_ERR_TARG_NOT_ASCENDING: return "Target pressure values must be ascending!";
        case PGW_ERR_EXTRAP_OFF: return "Extrapolation deactivated but data out of bounds.";
        case PGW_ERR_PREF_BELOW_SURFACE:
            return "p_ref locally lies below the surface. Please set a lower reference pressue (p_ref_inp) in settings.py";
        case PGW_ERR_PREF_AT_TOP: return "p_ref is matched by the top half level (level 0 does not exist)";
        case PGW_ERR_PS_HIST_ABOVE_TOP: return "historical surface pressure is not below the top climate-delta pressure level";
        case PGW_ERR_TOP_PRESSURE:
            return "ERA5 top pressure is lower than climate delta top pressure. If you are certain that you do not need "
                   "the data beyond to upper-most pressure level of the climate delta, you can set the flag "
                   "--ignore_top_pressure_error and re-run the script.";
        case PGW_ERR_NOT_CONVERGED: return "ERROR! Pressure adjustment did not converge";
        case PGW_ERR_GRID_EXTENT:      // functions.py:845-856 / 877-888 ("North or South" / "East or West" is chosen by the host)
            return "ERA5 dataset extends further than GCM dataset!. Perhaps consider using ERA5 on a subdomain only if "
                   "global coverage is not required?";
        case PGW_ERR_NO_P_REF:
            return "No reference pressure level above the required local minimum pressure level could not be found "
                   "everywhere. This is likely the case because your geopotential data set does not reach up high enough "
                   "(e.g. only to 500 hPa instead of e.g. 300 hPa?)";
        default: return "error";
    }
}

// ------------------------------------------------------------------ launch helpers
struct Prof {
    pgw_ctx *c; int kid; hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEvent_t take() {
        hipEvent_t e = nullptr;
        if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;      // launch goes unprofiled (see ~Prof)
        return e;
    }
    hipStream_t st;
    Prof(pgw_ctx *c_, int kid_, hipStream_t st_ = nullptr) : c(c_), kid(kid_), st(st_ ? st_ : c_->stream) {
        if (c->prof_on) {
            e0 = take(); e1 = take();
            if (e0 && e1) hipEventRecord(e0, st);
            else { if (e0) c->event_pool.push_back(e0); if (e1) c->event_pool.push_back(e1); e0 = e1 = nullptr; }
        }
    }
    ~Prof() {
        if (e0 && e1) {
            hipEventRecord(e1, st);
            c->prof_pending.push_back({kid, e0, e1});
        }
    }
};

static int prof_resolve(pgw_ctx *ctx) {
    if (ctx->prof_pending.empty()) return PGW_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof_pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        ctx->prof_count[r.kid] += 1;
        ctx->prof_ms[r.kid] += ms;
        ctx->event_pool.push_back(r.e0); ctx->event_pool.push_back(r.e1);
    }
    ctx->prof_pending.clear();
    return PGW_OK;
}

static inline unsigned int nblocks(long long n, int per) { return (unsigned int)((n + per - 1) / per); }

static inline bool aligned16(const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

// columns per thread: 16 B per lane when shape and alignment allow, else 1
// `max_v` caps the width for kernels whose register footprint makes the widest form slower.
static int pick_vec(pgw_ctx *ctx, int dtype, long long ncol, std::initializer_list<const void *> ptrs, int max_v = 4) {
    int v = (dtype == PGW_F64) ? 2 : 4;
    if (v > max_v) v = max_v;
    if (ctx->opt[PGW_OPT_FORCE_VEC1]) return 1;          // test knob: scalar columns per thread
    for (const void *p : ptrs) if (!aligned16(p)) return 1;
    while (v > 1 && ncol % v != 0) v >>= 1;
    return v;
}

static int status_reset(pgw_ctx *ctx) {
    DevStatus z;
    memset(&z, 0, sizeof(z));
    z.col = ~0ull;
    z.min_targ_bits = ~0ull;
    z.min_src_bits = ~0ull;
    *ctx->h_status = z;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_status, ctx->h_status, sizeof(DevStatus), hipMemcpyHostToDevice, ctx->stream));
    return PGW_OK;
}
static int status_fetch(pgw_ctx *ctx, const DevStatus *from = nullptr) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status, from ? from : ctx->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGW_OK;
}
static int status_check(pgw_ctx *ctx, const DevStatus *from = nullptr) {
    int rc = status_fetch(ctx, from);
    if (rc) return rc;
    if (ctx->h_status->code != 0) {
        ctx->err_col = (long long)ctx->h_status->col;
        ctx->err = status_text(ctx->h_status->code);
        return ctx->h_status->code;
    }
    return PGW_OK;
}

static int ws_get(pgw_ctx *ctx, int slot, size_t bytes, void **out) {
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->ws[slot])); }
        ctx->ws[slot] = nullptr; ctx->ws_bytes[slot] = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws[slot], bytes));
        ctx->ws_bytes[slot] = bytes;
    }
    *out = ctx->ws[slot];
    return PGW_OK;
}

static Levels levels_of(pgw_ctx *ctx) {
    Levels lv;
    int n = ctx->nlev;
    lv.ak = ctx->d_levels;
    lv.bk = ctx->d_levels + (n + 1);
    lv.akm = ctx->d_levels + 2 * (n + 1);
    lv.bkm = ctx->d_levels + 2 * (n + 1) + n;
    lv.nlev = n;
    lv.ps_mono_min = ctx->ps_mono_min;
    return lv;
}

// ------------------------------------------------------------------ context
extern "C" const char *pgw_version(void) { return "pgw_hip 0.1.0 (gfx950)"; }

extern "C" int pgw_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return PGW_ERR_HIP; }
    *n = c;
    return PGW_OK;
}

extern "C" int pgw_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 13) return PGW_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); buf[0] = 0; return PGW_ERR_ARG; }
    hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) { (void)hipGetLastError(); buf[0] = 0; return PGW_ERR_HIP; }   // not left behind for a later hipGetLastError()
    return PGW_OK;
}

static int env_flag(const char *name, int dflt) {
    const char *e = getenv(name);
    if (!e || !e[0]) return dflt;
    return (e[0] == '0') ? 0 : 1;
}

extern "C" int pgw_ctx_create(int device, pgw_ctx **out) {
    if (!out) return PGW_ERR_ARG;
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return PGW_ERR_HIP;
    if (device < 0 || device >= cnt) return PGW_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return PGW_ERR_HIP;
    pgw_ctx *c = new pgw_ctx();
    c->device = device;
    memset(c->prof_count, 0, sizeof(c->prof_count));
    memset(c->prof_ms, 0, sizeof(c->prof_ms));
    // the environment is read here and nowhere else (no getenv on the launch path)
    c->opt[PGW_OPT_QUAD] = env_flag("PGW_QUAD", 1);
    c->opt[PGW_OPT_FULL_COLUMN] = env_flag("PGW_FULL_COLUMN", 0);
    c->opt[PGW_OPT_FORCE_VEC1] = env_flag("PGW_FORCE_VEC1", 0);
    c->opt[PGW_OPT_MULTIPASS] = env_flag("PGW_MULTIPASS", 1);
    c->opt[PGW_OPT_LOOP_GUESS] = 6;
    c->opt[PGW_OPT_FORCE_OFF64] = 0;
    c->opt[PGW_OPT_TEST_FAIL] = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&c->d_status, (2 + MULTI_MAX_PASS) * sizeof(DevStatus)) != hipSuccess ||
        hipHostMalloc(&c->h_status, (2 + 2 * MULTI_MAX_PASS) * sizeof(DevStatus)) != hipSuccess ||
        hipMalloc(&c->d_small, SMALL_BYTES) != hipSuccess ||
        hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
        pgw_ctx_destroy(c);            // releases whatever was created before the failure
        return PGW_ERR_HIP;
    }
    *out = c;
    return PGW_OK;
}

extern "C" int pgw_ctx_destroy(pgw_ctx *ctx) {
    if (!ctx) return PGW_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto &r : ctx->prof_pending) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (auto &e : ctx->event_pool) hipEventDestroy(e);
    for (int i = 0; i < 8; ++i) if (ctx->ws[i]) hipFree(ctx->ws[i]);
    if (ctx->d_levels) hipFree(ctx->d_levels);
    if (ctx->d_small) hipFree(ctx->d_small);
    if (ctx->d_status) hipFree(ctx->d_status);
    if (ctx->h_status) hipHostFree(ctx->h_status);
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PGW_OK;
}

extern "C" int pgw_set_option(pgw_ctx *ctx, int option, int value) {
    if (!ctx) return PGW_ERR_ARG;
    if (option < 0 || option >= PGW_OPT_COUNT) return fail(ctx, PGW_ERR_ARG, "pgw_set_option: unknown option %d", option);
    ctx->opt[option] = value;
    return PGW_OK;
}
extern "C" int pgw_get_option(pgw_ctx *ctx, int option, int *value) {
    if (!ctx || !value) return PGW_ERR_ARG;
    if (option < 0 || option >= PGW_OPT_COUNT) return fail(ctx, PGW_ERR_ARG, "pgw_get_option: unknown option %d", option);
    *value = ctx->opt[option];
    return PGW_OK;
}

extern "C" const char *pgw_last_error(pgw_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }
extern "C" long long pgw_error_column(pgw_ctx *ctx) { return ctx ? ctx->err_col : -1; }

extern "C" int pgw_device_name(pgw_ctx *ctx, char *buf, size_t len) {
    hipDeviceProp_t p;
    HIPCHK(ctx, hipGetDeviceProperties(&p, ctx->device));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return PGW_OK;
}

// ------------------------------------------------------------------ memory
extern "C" int pgw_malloc(pgw_ctx *ctx, size_t bytes, void **dptr) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        // an allocation that does not fit is an answer, not a fault of the context: clear HIP's sticky last error so that the
        // hipGetLastError() after the next kernel launch does not report it (the placement draw probes how much fits)
        (void)hipGetLastError();
        *dptr = nullptr;
        return fail(ctx, PGW_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    return PGW_OK;
}
extern "C" int pgw_free(pgw_ctx *ctx, void *dptr) {
    if (!dptr) return PGW_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(dptr));
    return PGW_OK;
}
extern "C" int pgw_host_alloc(pgw_ctx *ctx, size_t bytes, void **hptr) {
    HIPCHK(ctx, hipSetDevice(ctx->device));                  // callable from reader / writer threads
    HIPCHK(ctx, hipHostMalloc(hptr, bytes ? bytes : 16));
    return PGW_OK;
}
extern "C" int pgw_host_free(pgw_ctx *ctx, void *hptr) {
    if (hptr) HIPCHK(ctx, hipHostFree(hptr));
    return PGW_OK;
}
extern "C" int pgw_memcpy_h2d(pgw_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_memcpy_d2h(pgw_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_memcpy_d2d(pgw_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_memset(pgw_ctx *ctx, void *dst, int value, size_t bytes) {
    if (bytes) HIPCHK(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_sync(pgw_ctx *ctx) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_mem_info(pgw_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return PGW_OK;
}

// ------------------------------------------------------------------ profiler / timer
extern "C" int pgw_profile_enable(pgw_ctx *ctx, int on) {
    int rc = prof_resolve(ctx);
    ctx->prof_on = on != 0;
    return rc;
}
extern "C" int pgw_profile_reset(pgw_ctx *ctx) {
    int rc = prof_resolve(ctx);
    memset(ctx->prof_count, 0, sizeof(ctx->prof_count));
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
    return rc;
}
extern "C" int pgw_profile_get(pgw_ctx *ctx, int kid, long long *launches, double *total_ms) {
    NEED(ctx, kid >= 0 && kid < PGW_K_COUNT, "bad kernel id");
    int rc = prof_resolve(ctx);
    if (launches) *launches = ctx->prof_count[kid];
    if (total_ms) *total_ms = ctx->prof_ms[kid];
    return rc;
}
extern "C" int pgw_timer_start(pgw_ctx *ctx) {
    HIPCHK(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return PGW_OK;
}
extern "C" int pgw_timer_stop(pgw_ctx *ctx, double *ms) {
    HIPCHK(ctx, hipEventRecord(ctx->t1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->t1));
    float f = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&f, ctx->t0, ctx->t1));
    if (ms) *ms = f;
    return PGW_OK;
}

// ------------------------------------------------------------------ vertical grid
extern "C" int pgw_set_levels(pgw_ctx *ctx, int nlev, const double *ak, const double *bk,
                              const double *akm, const double *bkm) {
    NEED(ctx, nlev >= 1 && nlev <= MAX_NLEV && ak && bk, "nlev must be in [1, 256]");
    NEED(ctx, (akm == nullptr) == (bkm == nullptr), "akm and bkm must both be given or both be NULL");
    std::vector<double> h((size_t)4 * nlev + 2);
    double *pak = h.data(), *pbk = pak + nlev + 1, *pakm = pbk + nlev + 1, *pbkm = pakm + nlev;
    memcpy(pak, ak, sizeof(double) * (nlev + 1));
    memcpy(pbk, bk, sizeof(double) * (nlev + 1));
    for (int l = 0; l < nlev; ++l) {
        if (akm) { pakm[l] = akm[l]; pbkm[l] = bkm[l]; }
        else {   // step_03_apply_to_era.py:74-85: 0.5*diff(label='lower') + lower
            pakm[l] = 0.5 * (ak[l + 1] - ak[l]) + ak[l];
            pbkm[l] = 0.5 * (bk[l + 1] - bk[l]) + bk[l];
        }
    }
    // smallest ps for which ak + ps*bk is strictly ascending over all layers:
    // d(ak) + ps*d(bk) > 0.  Layers with d(bk) <= 0 need d(ak) + ps*d(bk) > 0 for all ps of
    // interest; if d(bk) < 0 or (d(bk) == 0 and d(ak) <= 0) monotonicity is never assumed.
    double pmin = 0.0;
    for (int l = 0; l < nlev; ++l) {
        double da = ak[l + 1] - ak[l], db = bk[l + 1] - bk[l];
        if (db > 0) { double need = -da / db; if (need >= pmin) pmin = nextafter(need, INFINITY); }
        else if (db == 0 && da > 0) { /* fine for every ps */ }
        else { pmin = INFINITY; break; }
    }
    if (ctx->nlev != nlev || !ctx->d_levels) {
        if (ctx->d_levels) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->d_levels)); ctx->d_levels = nullptr; }
        HIPCHK(ctx, hipMalloc(&ctx->d_levels, sizeof(double) * h.size()));
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_levels, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // h is a stack-lifetime buffer
    ctx->nlev = nlev;
    ctx->h_akN = ak[nlev]; ctx->h_bkN = bk[nlev];
    ctx->ps_mono_min = pmin;
    ctx->n_pure = 0;
    while (ctx->n_pure < nlev && pbkm[ctx->n_pure] == 0.0) ctx->n_pure += 1;
    ctx->h_akm.assign(pakm, pakm + nlev);
    ctx->h_bkm.assign(pbkm, pbkm + nlev);
    return PGW_OK;
}

extern "C" int pgw_get_full_level_coeffs(pgw_ctx *ctx, double *akm_out, double *bkm_out) {
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    memcpy(akm_out, ctx->h_akm.data(), sizeof(double) * ctx->nlev);
    memcpy(bkm_out, ctx->h_bkm.data(), sizeof(double) * ctx->nlev);
    return PGW_OK;
}

// every compute entry binds the context's device first: a process may hold contexts on several devices
#define CHECK_COMMON(ctx, dtype, ntime, ncol)                                        \
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64"); \
    NEED(ctx, ntime >= 1 && ncol >= 1, "ntime and ncol must be positive");           \
    HIPCHK(ctx, hipSetDevice(ctx->device));

// dispatch on storage dtype and vector width
#define DISPATCH_TV(dtype, vec, ...)                                        \
    do {                                                                    \
        if (dtype == PGW_F64) {                                             \
            typedef double T;                                               \
            if (vec == 2) { constexpr int V = 2; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } \
        } else {                                                            \
            typedef float T;                                                \
            if (vec == 4) { constexpr int V = 4; __VA_ARGS__; }             \
            else if (vec == 2) { constexpr int V = 2; __VA_ARGS__; }        \
            else { constexpr int V = 1; __VA_ARGS__; }                      \
        }                                                                   \
    } while (0)

// storage type T of the ERA5 fields, TL of the PGW level arrays (ta_pgw, e, QV out), REF = reference-dtype mode
// (float32 files only: T = float, TL = double)
#define DISPATCH_TLV(dtype, ref, vec, ...)                                   \
    do {                                                                     \
        if (dtype == PGW_F64) {                                              \
            typedef double T; typedef double TL; constexpr bool REF = false; \
            if (vec == 2) { constexpr int V = 2; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } \
        } else if (ref) {                                                    \
            typedef float T; typedef double TL; constexpr bool REF = true;   \
            if (vec >= 2) { constexpr int V = 2; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } \
        } else {                                                             \
            typedef float T; typedef float TL; constexpr bool REF = false;   \
            if (vec == 4) { constexpr int V = 4; __VA_ARGS__; }              \
            else if (vec == 2) { constexpr int V = 2; __VA_ARGS__; }         \
            else { constexpr int V = 1; __VA_ARGS__; }                       \
        }                                                                    \
    } while (0)

#define DISPATCH_T(dtype, ...)                                   \
    do {                                                         \
        if (dtype == PGW_F64) { typedef double T; __VA_ARGS__; } \
        else { typedef float T; __VA_ARGS__; }                   \
    } while (0)

extern "C" int pgw_pressure_levels(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                                   const void *ps, void *pa_hl, void *pa) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, ps && (pa_hl || pa), "null pointer");
    int vec = pick_vec(ctx, dtype, ncol, {ps, pa_hl, pa});
    Levels lv = levels_of(ctx);
    {
        Prof pr(ctx, PGW_K_PRESSURE);
        DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_pressure_levels<T, V>), dim3(nblocks((long long)ntime * ncol / V, BLOCK)),
                                                    dim3(BLOCK), 0, ctx->stream, lv, ntime, ncol, (const T *)ps, (T *)pa_hl, (T *)pa));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

// ------------------------------------------------------------------ humidity
template <int MODE>
static int humidity_flat(pgw_ctx *ctx, int kid, int dtype, long long n, const void *x, const void *pa,
                         const void *ta, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, n >= 1 && x && pa && ta && out, "bad argument");
    int vec = pick_vec(ctx, dtype, n, {x, pa, ta, out});
    long long groups = n / vec;
    unsigned int nb = nblocks(groups, BLOCK);
    if (nb > 256 * 16) nb = 256 * 16;
    {
        Prof pr(ctx, kid);
        DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_humidity_flat<T, V, MODE>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n,
                                                    (const T *)x, (const T *)pa, (const T *)ta, (T *)out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}
extern "C" int pgw_specific_to_relative_humidity(pgw_ctx *ctx, int dtype, long long n, const void *hus,
                                                 const void *pa, const void *ta, void *hur) {
    return humidity_flat<0>(ctx, PGW_K_Q_TO_RH, dtype, n, hus, pa, ta, hur);
}
extern "C" int pgw_relative_to_specific_humidity(pgw_ctx *ctx, int dtype, long long n, const void *hur,
                                                 const void *pa, const void *ta, void *hus) {
    return humidity_flat<1>(ctx, PGW_K_RH_TO_Q, dtype, n, hur, pa, ta, hus);
}

extern "C" int pgw_humidity_leaf(pgw_ctx *ctx, int dtype, int which, long long n, const void *a, const void *b, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, which >= 0 && which <= 4, "which must be 0..4");
    NEED(ctx, n >= 1 && a && out && (which >= 2 || b), "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unsigned int nb = nblocks(n, BLOCK);
    if (nb > 256 * 16) nb = 256 * 16;
#define LEAF(W) hipLaunchKernelGGL((k_humidity_leaf<T, W>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n, (const T *)a, (const T *)b, (T *)out)
    DISPATCH_T(dtype, {
        switch (which) {
            case 0: LEAF(0); break;
            case 1: LEAF(1); break;
            case 2: LEAF(2); break;
            case 3: LEAF(3); break;
            default: LEAF(4); break;
        }
    });
#undef LEAF
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

template <int MODE>
static int humidity_hybrid(pgw_ctx *ctx, int kid, int dtype, int ntime, long long ncol, const void *x,
                           const void *ps, const void *ta, void *out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, x && ps && ta && out, "null pointer");
    int vec = pick_vec(ctx, dtype, ncol, {x, ps, ta, out});
    Levels lv = levels_of(ctx);
    {
        Prof pr(ctx, kid);
        DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_humidity_hybrid<T, V, MODE>), dim3(nblocks((long long)ntime * ncol / V, BLOCK)),
                                                    dim3(BLOCK), 0, ctx->stream, lv, ntime, ncol, (const T *)x, (const T *)ps,
                                                    (const T *)ta, (T *)out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}
extern "C" int pgw_specific_to_relative_humidity_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                                                        const void *hus, const void *ps, const void *ta, void *hur) {
    return humidity_hybrid<0>(ctx, PGW_K_Q_TO_RH, dtype, ntime, ncol, hus, ps, ta, hur);
}
extern "C" int pgw_relative_to_specific_humidity_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol,
                                                        const void *hur, const void *ps, const void *ta, void *hus) {
    return humidity_hybrid<1>(ctx, PGW_K_RH_TO_Q, dtype, ntime, ncol, hur, ps, ta, hus);
}

// ------------------------------------------------------------------ integ_geopot
static int launch_integ_geopot(pgw_ctx *ctx, int dtype, int nlev, int ntime, long long ncol, const void *pa_hl,
                               const void *zgs, const void *ta, const void *hus, double p_ref,
                               const void *p_ref_field, void *phi_ref, int full_column, bool out_f64 = false) {
    int vec = pick_vec(ctx, dtype, ncol, {pa_hl, zgs, ta, hus, p_ref_field, phi_ref});
    Prof pr(ctx, PGW_K_INTEG_GEOPOT);
#define LAUNCH_GEO(UU, TO_)                                                                                         \
    DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_integ_geopot<T, V, UU, TO_>), dim3(nblocks((long long)ntime * ncol / V, BLOCK)), \
                                                dim3(BLOCK), 0, ctx->stream, nlev, ntime, ncol, (const T *)pa_hl,     \
                                                (const T *)zgs, (const T *)ta, (const T *)hus, p_ref,                  \
                                                (const T *)p_ref_field, (TO_ *)phi_ref, full_column, ctx->d_status))
    if (out_f64) { LAUNCH_GEO(4, double); }      // levels per chunk: 2 / 4 / 8 measured the same
    else { LAUNCH_GEO(4, T); }
#undef LAUNCH_GEO
    return PGW_OK;
}

extern "C" int pgw_integ_geopot(pgw_ctx *ctx, int dtype, int ntime, int nlev, long long ncol, const void *pa_hl,
                                const void *zgs, const void *ta, const void *hus, double p_ref,
                                const void *p_ref_field, void *phi_ref, int full_column) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nlev >= 1, "nlev must be positive");
    NEED(ctx, pa_hl && zgs && ta && hus && phi_ref, "null pointer");
    int rc = status_reset(ctx);
    if (rc) return rc;
    launch_integ_geopot(ctx, dtype, nlev, ntime, ncol, pa_hl, zgs, ta, hus, p_ref, p_ref_field, phi_ref, full_column);
    HIPCHK(ctx, hipGetLastError());
    return status_check(ctx);
}

// ------------------------------------------------------------------ interp_logp_4d
template <typename T, int MODE>
static int launch_interp_mode(pgw_ctx *ctx, int ntime, int S, int N, long long ncol, const T *var, const T *sp,
                              const T *tp, T *out, int logp_in) {
    long long total = (long long)ntime * ncol;
    Prof pr(ctx, PGW_K_INTERP_LOGP);
    hipLaunchKernelGGL((k_interp_logp_stream<T, MODE>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream,
                       ntime, S, N, ncol, var, sp, tp, out, logp_in, ctx->d_status);
    return PGW_OK;
}

extern "C" int pgw_interp_logp_4d(pgw_ctx *ctx, int dtype, int ntime, int nsrc, int ntarg, long long ncol,
                                  const void *var, const void *source_P, const void *targ_P, int extrapolate,
                                  int logp_in, void *out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nsrc >= 2 && ntarg >= 1, "need at least 2 source levels and 1 target level");
    NEED(ctx, var && source_P && targ_P && out, "null pointer");
    if (extrapolate < 0 || extrapolate > 3) return fail(ctx, PGW_ERR_ARG, "Invalid input value for \"extrapolate\"");
    int rc = status_reset(ctx);
    if (rc) return rc;
    DISPATCH_T(dtype, {
        const T *v = (const T *)var; const T *sp = (const T *)source_P; const T *tp = (const T *)targ_P; T *o = (T *)out;
        switch (extrapolate) {
            case 0: rc = launch_interp_mode<T, 0>(ctx, ntime, nsrc, ntarg, ncol, v, sp, tp, o, logp_in); break;
            case 1: rc = launch_interp_mode<T, 1>(ctx, ntime, nsrc, ntarg, ncol, v, sp, tp, o, logp_in); break;
            case 2: rc = launch_interp_mode<T, 2>(ctx, ntime, nsrc, ntarg, ncol, v, sp, tp, o, logp_in); break;
            default: rc = launch_interp_mode<T, 3>(ctx, ntime, nsrc, ntarg, ncol, v, sp, tp, o, logp_in); break;
        }
    });
    if (rc) return rc;
    HIPCHK(ctx, hipGetLastError());
    return status_check(ctx);
}

// ------------------------------------------------------------------ time lerp
extern "C" int pgw_time_lerp(pgw_ctx *ctx, int dtype, long long n, const void *v_before, const void *v_after,
                             double x_hi, double x_new, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, n >= 1 && v_before && v_after && out, "bad argument");
    int vec = pick_vec(ctx, dtype, n, {v_before, v_after, out});
    unsigned int nb = nblocks(n / vec, BLOCK);
    if (nb > 256 * 16) nb = 256 * 16;
    {
        Prof pr(ctx, PGW_K_TIME_LERP);
        DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_time_lerp<T, V>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n,
                                                    (const T *)v_before, (const T *)v_after, x_hi, x_new, (T *)out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

// ------------------------------------------------------------------ vert_interp_delta
static int plev_table(pgw_ctx *ctx, int nplev, const double *plev) {
    if ((int)ctx->plev_key.size() == nplev && memcmp(ctx->plev_key.data(), plev, sizeof(double) * nplev) == 0)
        return PGW_OK;
    PlevTable &t = ctx->plev_tab;
    memset(&t, 0, sizeof(t));
    t.n = nplev;
    t.pmax = -INFINITY; t.pmin = INFINITY;
    bool anynan = false;
    for (int i = 0; i < nplev; ++i) {
        t.p[i] = plev[nplev - 1 - i];                       // functions.py:383-384 reversal
        if (t.p[i] != t.p[i]) anynan = true;
        if (t.p[i] > t.pmax) t.pmax = t.p[i];
        if (t.p[i] < t.pmin) t.pmin = t.p[i];
    }
    if (anynan) return fail(ctx, PGW_ERR_ARG, "vert_interp_delta: NaN in plev");
    // ln(plev) with the device log so that table and per-column logs are from one implementation
    double *d_in = ctx->d_small, *d_out = ctx->d_small + MAX_PLEV;
    HIPCHK(ctx, hipMemcpyAsync(d_in, t.p, sizeof(double) * nplev, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_log_table, dim3(1), dim3(64), 0, ctx->stream, nplev, d_in, d_out);
    HIPCHK(ctx, hipMemcpyAsync(t.lnp, d_out, sizeof(double) * nplev, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->plev_key.assign(plev, plev + nplev);
    return PGW_OK;
}

extern "C" int pgw_vert_interp_delta(pgw_ctx *ctx, int dtype, int ntime, int nplev, int nlev_t, long long ncol,
                                     const double *plev, const void *delta_b, const void *delta_a, double x_hi,
                                     double x_new, const void *dsfc_b, const void *dsfc_a, const void *pshist_b,
                                     const void *pshist_a, const void *targ_P, const void *ps, int ignore_top,
                                     const void *add_to, void *out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nplev >= 2 && nplev <= MAX_PLEV, "nplev must be in [2, 64]");
    NEED(ctx, plev && delta_b && out, "null pointer");
    NEED(ctx, targ_P || ps, "need targ_P or ps");
    NEED(ctx, (dsfc_b == nullptr) == (pshist_b == nullptr), "delta_sfc and ps_hist must be given together");
    if (!targ_P) {
        NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
        NEED(ctx, nlev_t == ctx->nlev, "nlev_t must equal the context's nlev when targ_P is NULL");
    }
    NEED(ctx, nlev_t >= 1, "nlev_t must be positive");
    if (x_hi == 0.0) { delta_a = nullptr; dsfc_a = nullptr; pshist_a = nullptr; }
    int rc = plev_table(ctx, nplev, plev);
    if (rc) return rc;
    rc = status_reset(ctx);
    if (rc) return rc;
    Levels lv = levels_of(ctx);
    if (targ_P) { lv.akm = lv.bkm = nullptr; }
    long long total = (long long)ntime * ncol;
    {
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        DISPATCH_T(dtype, {
            DeltaSrc<T> d{(const T *)delta_b, (const T *)delta_a, x_hi, x_new};
            DeltaSrc<T> s{(const T *)dsfc_b, (const T *)dsfc_a, x_hi, x_new};
            DeltaSrc<T> p{(const T *)pshist_b, (const T *)pshist_a, x_hi, x_new};
            if (dsfc_b)
                hipLaunchKernelGGL((k_vert_interp_delta<T, true>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream,
                                   ctx->plev_tab, lv, ntime, nlev_t, ncol, d, s, p, (const T *)targ_P, (const T *)ps,
                                   ignore_top ? 0 : 1, (const T *)add_to, (T *)out, ctx->d_status);
            else
                hipLaunchKernelGGL((k_vert_interp_delta<T, false>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream,
                                   ctx->plev_tab, lv, ntime, nlev_t, ncol, d, s, p, (const T *)targ_P, (const T *)ps,
                                   ignore_top ? 0 : 1, (const T *)add_to, (T *)out, ctx->d_status);
        });
    }
    HIPCHK(ctx, hipGetLastError());
    rc = status_check(ctx);
    if (rc) return rc;
    if (!ignore_top) {
        // functions.py:417-425: np.min(target_P) < np.min(source_P); NaN in either -> comparison False
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
    }
    return PGW_OK;
}

extern "C" int pgw_reinterp_field(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                                  const void *delta_b, const void *delta_a, double x_hi, double x_new, const void *dsfc_b,
                                  const void *dsfc_a, const void *pshist_b, const void *pshist_a, const void *era_field,
                                  const void *ps_era, const void *ps_pgw, int ignore_top, void *out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nplev >= 2 && nplev <= MAX_PLEV, "nplev must be in [2, 64]");
    NEED(ctx, plev && delta_b && era_field && ps_era && ps_pgw && out, "null pointer");
    NEED(ctx, (dsfc_b == nullptr) == (pshist_b == nullptr), "delta_sfc and ps_hist must be given together");
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    if (x_hi == 0.0) { delta_a = nullptr; dsfc_a = nullptr; pshist_a = nullptr; }
    int rc = plev_table(ctx, nplev, plev);
    if (rc) return rc;
    if ((rc = status_reset(ctx))) return rc;
    Levels lv = levels_of(ctx);
    const long long total = (long long)ntime * ncol;
    {
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        DISPATCH_T(dtype, {
            DeltaSrc<T> d{(const T *)delta_b, (const T *)delta_a, x_hi, x_new};
            DeltaSrc<T> s{(const T *)dsfc_b, (const T *)dsfc_a, x_hi, x_new};
            DeltaSrc<T> p{(const T *)pshist_b, (const T *)pshist_a, x_hi, x_new};
            if (dsfc_b)
                hipLaunchKernelGGL((k_reinterp_field<T, true>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, ctx->plev_tab,
                                   lv, ntime, ncol, d, s, p, (const T *)era_field, (const T *)ps_era, (const T *)ps_pgw,
                                   ignore_top ? 0 : 1, (T *)out, ctx->d_status);
            else
                hipLaunchKernelGGL((k_reinterp_field<T, false>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, ctx->plev_tab,
                                   lv, ntime, ncol, d, s, p, (const T *)era_field, (const T *)ps_era, (const T *)ps_pgw,
                                   ignore_top ? 0 : 1, (T *)out, ctx->d_status);
        });
    }
    HIPCHK(ctx, hipGetLastError());
    if ((rc = status_check(ctx))) return rc;
    if (!ignore_top) {                                     // functions.py:417-425
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
    }
    return PGW_OK;
}

template <typename T, bool SFC, typename O, bool EVAP, typename TE0, typename TE1, typename TO, bool REF>
static void launch_reinterp_pair_o(pgw_ctx *ctx, const Levels &lv, int ntime, long long ncol, const ReinterpPair<T, TE0, TE1, TO> &rv,
                                   const DeltaSrc<T> &p, const T *ps_era, const T *ps_pgw, int check_top) {
    hipLaunchKernelGGL((k_reinterp_pair<T, SFC, O, EVAP, TE0, TE1, TO, REF>), dim3(nblocks((long long)ntime * ncol, BLOCK)), dim3(BLOCK),
                       2 * lv.nlev * sizeof(double), ctx->stream, ctx->plev_tab, lv, ntime, ncol, rv, p, ps_era, ps_pgw,
                       check_top, ctx->d_status);
}
template <typename T, typename TE0 = T, typename TE1 = T, typename TO = T, bool REF = false>
static void launch_reinterp_pair(pgw_ctx *ctx, const Levels &lv, int ntime, int nplev, long long ncol,
                                 const ReinterpPair<T, TE0, TE1, TO> &rv,
                                 const DeltaSrc<T> &p, const T *ps_era, const T *ps_pgw, bool sfc, int check_top) {
    // 32-bit byte offsets when every array (fields: nlev levels, delta records: nplev levels) is smaller than 4 GiB
    const unsigned long long big = (unsigned long long)ntime * (lv.nlev > nplev ? lv.nlev : nplev) * ncol * sizeof(TO);
    const bool o32 = big < (1ull << 32) && !ctx->opt[PGW_OPT_FORCE_OFF64];
    if (rv.evap) {                                       // the loop's ta + hur pair (always with the surface insertion)
        if (o32) launch_reinterp_pair_o<T, true, boff32, true, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
        else launch_reinterp_pair_o<T, true, boff64, true, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
    } else if constexpr (!REF) {
        if (o32) {
            if (sfc) launch_reinterp_pair_o<T, true, boff32, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
            else launch_reinterp_pair_o<T, false, boff32, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
        } else {
            if (sfc) launch_reinterp_pair_o<T, true, boff64, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
            else launch_reinterp_pair_o<T, false, boff64, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
        }
    } else {                                             // reference-dtype mode without EVAP: the ua + va pair after the loop
        if (o32) launch_reinterp_pair_o<T, false, boff32, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
        else launch_reinterp_pair_o<T, false, boff64, false, TE0, TE1, TO, REF>(ctx, lv, ntime, ncol, rv, p, ps_era, ps_pgw, check_top);
    }
}

extern "C" int pgw_reinterp_pair(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                                 const void *const *delta_b, const void *const *delta_a, double x_hi, double x_new,
                                 const void *const *dsfc_b, const void *const *dsfc_a, const void *pshist_b,
                                 const void *pshist_a, const void *const *era_field, const void *ps_era, const void *ps_pgw,
                                 int ignore_top, void *const *out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nplev >= 2 && nplev <= MAX_PLEV, "nplev must be in [2, 64]");
    NEED(ctx, plev && delta_b && era_field && ps_era && ps_pgw && out, "null pointer");
    NEED(ctx, delta_b[0] && delta_b[1] && era_field[0] && era_field[1] && out[0] && out[1], "null pointer");
    NEED(ctx, (dsfc_b == nullptr) == (pshist_b == nullptr), "delta_sfc and ps_hist must be given together");
    NEED(ctx, !dsfc_b || (dsfc_b[0] && dsfc_b[1]), "null pointer");
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    const bool lerp = (x_hi != 0.0);
    NEED(ctx, !lerp || (delta_a && delta_a[0] && delta_a[1] && (!dsfc_b || (dsfc_a && dsfc_a[0] && dsfc_a[1] && pshist_a))),
         "the record after the instant is missing");
    int rc = plev_table(ctx, nplev, plev);
    if (rc) return rc;
    if ((rc = status_reset(ctx))) return rc;
    Levels lv = levels_of(ctx);
    {
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        DISPATCH_T(dtype, {
            ReinterpPair<T> rv;
            for (int v = 0; v < 2; ++v) {
                rv.d[v] = DeltaSrc<T>{(const T *)delta_b[v], lerp ? (const T *)delta_a[v] : nullptr, x_hi, x_new};
                rv.sfc[v] = DeltaSrc<T>{dsfc_b ? (const T *)dsfc_b[v] : nullptr, (dsfc_b && lerp) ? (const T *)dsfc_a[v] : nullptr, x_hi, x_new};
                rv.out[v] = (T *)out[v];
            }
            rv.era0 = (const T *)era_field[0]; rv.era1 = (const T *)era_field[1];
            rv.evap = nullptr;
            DeltaSrc<T> p{(const T *)pshist_b, lerp ? (const T *)pshist_a : nullptr, x_hi, x_new};
            launch_reinterp_pair<T>(ctx, lv, ntime, nplev, ncol, rv, p, (const T *)ps_era, (const T *)ps_pgw, dsfc_b != nullptr,
                                    ignore_top ? 0 : 1);
        });
    }
    HIPCHK(ctx, hipGetLastError());
    if ((rc = status_check(ctx))) return rc;
    if (!ignore_top) {                                     // functions.py:417-425
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
    }
    return PGW_OK;
}

extern "C" int pgw_replace_delta_sfc(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol,
                                     const double *plev_asc, const void *delta, const void *delta_sfc,
                                     const void *ps_hist, void *out_P, void *out_delta) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nplev >= 1 && nplev <= MAX_PLEV, "nplev must be in [1, 64]");
    NEED(ctx, plev_asc && delta && delta_sfc && ps_hist && out_P && out_delta, "null pointer");
    PlevTable t;
    memset(&t, 0, sizeof(t));
    t.n = nplev; t.pmax = -INFINITY; t.pmin = INFINITY;
    for (int i = 0; i < nplev; ++i) {
        t.p[i] = plev_asc[i];
        if (t.p[i] > t.pmax) t.pmax = t.p[i];
        if (t.p[i] < t.pmin) t.pmin = t.p[i];
    }
    int rc = status_reset(ctx);
    if (rc) return rc;
    long long total = (long long)ntime * ncol;
    {
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_replace_delta_sfc<T>), dim3(nblocks(total, BLOCK)), dim3(BLOCK), 0, ctx->stream, t,
                                             ntime, ncol, (const T *)delta, (const T *)delta_sfc, (const T *)ps_hist,
                                             (T *)out_P, (T *)out_delta, ctx->d_status));
    }
    HIPCHK(ctx, hipGetLastError());
    return status_check(ctx);
}

// ------------------------------------------------------------------ ps fixed-point loop
#ifndef PAIR_U
#define PAIR_U 4
#endif
#ifndef STEP_U
#define STEP_U 2      // measured 3 % faster than 4 for the pass kernel (finer stop above p_ref, fewer VGPRs)
#endif

static int launch_step(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *ta, const void *evap,
                       const void *PS, const void *FIS, const double *phi_ref_era, const double *dphi_clim,
                       double *delta_ps, double *adj_ps, double p_ref, const double *p_ref_field,
                       double adj_factor, int full_column, int apply_adj = 1, DevStatus *st = nullptr,
                       DevStatus *clear = nullptr, bool ref = false) {
    if (!st) st = ctx->d_status;
    int vec = pick_vec(ctx, dtype, ncol, {ta, evap, PS, FIS, phi_ref_era, dphi_clim, delta_ps, adj_ps, p_ref_field}, 2);
    // fp64 state arrays are read with V doubles per lane: 16*V/2 B alignment follows from ncol % V == 0
    Levels lv = levels_of(ctx);
    Prof pr(ctx, PGW_K_ADJUST_PS_STEP);
    DISPATCH_TLV(dtype, ref, vec, hipLaunchKernelGGL((k_adjust_ps_step<T, TL, V, STEP_U, REF>), dim3(nblocks((long long)ntime * ncol / V, BLOCK)),
                                                     dim3(BLOCK), 0, ctx->stream, lv, ntime, ncol, (const TL *)ta, (const TL *)evap,
                                                     (const T *)PS, (const T *)FIS, phi_ref_era, dphi_clim, delta_ps, adj_ps,
                                                     p_ref, p_ref_field, adj_factor, full_column, apply_adj, st, clear));
    return PGW_OK;
}

static int launch_phi_ref_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *ta, const void *hus,
                                 const void *PS, const void *FIS, double p_ref, double *phi_out, int full_column,
                                 const double *p_ref_field = nullptr, bool ref = false) {
    int vec = pick_vec(ctx, dtype, ncol, {ta, hus, PS, FIS, phi_out}, 2);
    Levels lv = levels_of(ctx);
    Prof pr(ctx, PGW_K_PHI_REF_HYBRID);
    DISPATCH_TLV(dtype, ref, vec, hipLaunchKernelGGL((k_phi_ref_hybrid<T, V, STEP_U, REF>), dim3(nblocks((long long)ntime * ncol / V, BLOCK)),
                                                     dim3(BLOCK), 0, ctx->stream, lv, ntime, ncol, (const T *)ta, (const T *)hus,
                                                     (const T *)PS, (const T *)FIS, p_ref, p_ref_field, phi_out,
                                                     full_column, ctx->d_status));
    return PGW_OK;
}

static double max_err_of(pgw_ctx *ctx) {
    DevStatus *h = ctx->h_status;
    if (h->valid == 0) return NAN;              // xarray .max() of an all-NaN field
    double m;
    memcpy(&m, &h->max_bits, 8);
    return m;
}

#define QUAD_TPB 128      // 64 / 256 threads measured the same (2.16 / 2.17 / 2.17 ms same box)
#ifndef QUAD_U
#define QUAD_U 2          // levels per software-pipeline chunk of k_delta_quad
#endif

extern "C" int pgw_adjust_ps_step(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *ta_pgw,
                                  const void *hur_pgw, const void *PS, const void *FIS, const double *phi_ref_era,
                                  const double *dphi_clim, double *delta_ps, double *adj_ps, double p_ref,
                                  const double *p_ref_field, double adj_factor, int apply_adj, double *max_abs_err) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, ta_pgw && hur_pgw && PS && FIS && phi_ref_era && dphi_clim && delta_ps && adj_ps, "null pointer");
    size_t es = dtype == PGW_F64 ? 8 : 4;
    void *evap = nullptr;
    int rc = ws_get(ctx, 0, (size_t)ntime * ctx->nlev * ncol * es, &evap);
    if (rc) return rc;
    rc = humidity_hybrid<2>(ctx, PGW_K_RH_TO_Q, dtype, ntime, ncol, hur_pgw, PS, ta_pgw, evap);
    if (rc) return rc;
    rc = status_reset(ctx);
    if (rc) return rc;
    launch_step(ctx, dtype, ntime, ncol, ta_pgw, evap, PS, FIS, phi_ref_era, dphi_clim, delta_ps, adj_ps, p_ref,
                p_ref_field, adj_factor, ctx->opt[PGW_OPT_FULL_COLUMN], apply_adj ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    rc = status_check(ctx);
    if (max_abs_err) *max_abs_err = max_err_of(ctx);
    return rc;
}

extern "C" int pgw_reinterp_pass(pgw_ctx *ctx, int dtype, int ntime, int nplev, long long ncol, const double *plev,
                                 const void *const *delta_b, const void *const *delta_a, double x_hi, double x_new,
                                 const void *const *dsfc_b, const void *const *dsfc_a, const void *pshist_b,
                                 const void *pshist_a, const void *T_era, const void *RELHUM_era, const void *PS,
                                 const void *FIS, const double *phi_ref_era, const double *dphi_clim, double *delta_ps,
                                 double *adj_ps, double p_ref, double adj_factor, int ignore_top, void *ps_pgw,
                                 void *ta_pgw, void *hur_pgw, double *max_abs_err) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nplev >= 2 && nplev <= MAX_PLEV, "nplev must be in [2, 64]");
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, plev && delta_b && dsfc_b && pshist_b && T_era && RELHUM_era && PS && FIS && phi_ref_era && dphi_clim && delta_ps &&
         adj_ps && ps_pgw && ta_pgw && hur_pgw, "null pointer");
    NEED(ctx, delta_b[0] && delta_b[1] && dsfc_b[0] && dsfc_b[1], "null pointer");
    const bool lerp = (x_hi != 0.0);
    NEED(ctx, !lerp || (delta_a && delta_a[0] && delta_a[1] && dsfc_a && dsfc_a[0] && dsfc_a[1] && pshist_a),
         "the record after the instant is missing");
    const size_t es = dtype == PGW_F64 ? 8 : 4;
    void *evap = nullptr;
    int rc = ws_get(ctx, 0, (size_t)ntime * ctx->nlev * ncol * es, &evap);
    if (rc) return rc;
    if ((rc = plev_table(ctx, nplev, plev))) return rc;
    if ((rc = status_reset(ctx))) return rc;
    Levels lv = levels_of(ctx);
    const long long n2 = (long long)ntime * ncol;
    {   // step_03:192-193
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_update_ps<T>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, n2,
                                             (const T *)PS, delta_ps, adj_ps, (T *)ps_pgw));
    }
    {   // :202-216 for ta and hur, + e of functions.py:123
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        DISPATCH_T(dtype, {
            ReinterpPair<T> rv;
            for (int v = 0; v < 2; ++v) {
                rv.d[v] = DeltaSrc<T>{(const T *)delta_b[v], lerp ? (const T *)delta_a[v] : nullptr, x_hi, x_new};
                rv.sfc[v] = DeltaSrc<T>{(const T *)dsfc_b[v], lerp ? (const T *)dsfc_a[v] : nullptr, x_hi, x_new};
            }
            rv.era0 = (const T *)T_era; rv.era1 = (const T *)RELHUM_era;
            rv.out[0] = (T *)ta_pgw; rv.out[1] = (T *)hur_pgw;
            rv.evap = (T *)evap;
            DeltaSrc<T> p{(const T *)pshist_b, lerp ? (const T *)pshist_a : nullptr, x_hi, x_new};
            launch_reinterp_pair<T>(ctx, lv, ntime, nplev, ncol, rv, p, (const T *)PS, (const T *)ps_pgw, true, ignore_top ? 0 : 1);
        });
    }
    // :262-308: the pass on the re-interpolated fields (delta_ps already carries this pass's increment)
    launch_step(ctx, dtype, ntime, ncol, ta_pgw, evap, PS, FIS, phi_ref_era, dphi_clim, delta_ps, adj_ps, p_ref, nullptr,
                adj_factor, ctx->opt[PGW_OPT_FULL_COLUMN], 0);
    HIPCHK(ctx, hipGetLastError());
    rc = status_check(ctx);
    if (max_abs_err) *max_abs_err = max_err_of(ctx);
    if (rc) return rc;
    if (!ignore_top) {                                     // functions.py:417-425
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
    }
    return PGW_OK;
}

extern "C" int pgw_update_ps(pgw_ctx *ctx, int dtype, long long n, const void *PS, double *delta_ps,
                             const double *adj_ps, void *ps_pgw) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, n >= 1 && PS && delta_ps && adj_ps && ps_pgw, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_update_ps<T>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, n,
                                         (const T *)PS, delta_ps, adj_ps, (T *)ps_pgw));
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_phi_ref_hybrid(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *T, const void *QV,
                                  const void *PS, const void *FIS, double p_ref, const double *p_ref_field,
                                  double *phi_ref) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, T && QV && PS && FIS && phi_ref, "null pointer");
    int rc = status_reset(ctx);
    if (rc) return rc;
    launch_phi_ref_hybrid(ctx, dtype, ntime, ncol, T, QV, PS, FIS, p_ref, phi_ref, ctx->opt[PGW_OPT_FULL_COLUMN], p_ref_field);
    HIPCHK(ctx, hipGetLastError());
    return status_check(ctx);
}

// ---- latitude-band sharding (pgw_set_reduce_hook).  One vector per loop launch: [status of the kernels before the
// loop | per pass: status, any-valid flag, max |err| (-inf when the band has no valid column)].
static int first_launch_passes(pgw_ctx *ctx, int max_n_iter) {
    int np = ctx->opt[PGW_OPT_LOOP_GUESS];
    if (np > MULTI_MAX_PASS) np = MULTI_MAX_PASS;
    if (np > max_n_iter) np = max_n_iter;
    return np < 1 ? 1 : np;
}

static int call_reduce(pgw_ctx *ctx, double *v, int n) {
    ctx->band_reduces += 1;
    if (ctx->reduce_fn(v, n, ctx->reduce_user) != 0) {
        ctx->band_agreed = true;                    // the exchange itself is broken: nobody is met by sending more
        ctx->err = "the reduce hook (pgw_set_reduce_hook) failed";
        ctx->err_col = -1;
        return PGW_ERR_REDUCE;
    }
    return PGW_OK;
}

// An error of THIS band only (a data error found before the loop's first launch, a failed allocation, a HIP error - at
// any point of the file): the other bands are waiting, or about to wait, in their next reduce.  Meet them there with the
// status - in the slot of the kernels before the loop while no reduce has been made, in the first pass's status slot of a
// continuation launch afterwards - so every rank returns it instead of blocking until the backend's timeout.
static int band_fail(pgw_ctx *ctx, int code, int max_n_iter) {
    if (!ctx->reduce_fn || code == PGW_OK || ctx->band_agreed) return code;
    const std::string text = ctx->err;
    const long long col = ctx->err_col;
    const bool first = ctx->band_reduces == 0;
    int np = first ? first_launch_passes(ctx, max_n_iter) : ctx->band_next_np;
    if (np < 1) np = 1;
    double v[1 + 3 * MULTI_MAX_PASS];
    v[0] = first ? (double)code : 0.0;
    for (int k = 0; k < np; ++k) { v[1 + 3 * k] = 0.0; v[2 + 3 * k] = 0.0; v[3 + 3 * k] = -INFINITY; }
    if (!first) v[1] = (double)code;
    call_reduce(ctx, v, 1 + 3 * np);
    ctx->band_agreed = true;
    ctx->err = text;
    ctx->err_col = col;
    return code;
}

extern "C" int pgw_band_abort(pgw_ctx *ctx, int code, int max_n_iter) {
    if (!ctx) return PGW_ERR_ARG;
    ctx->band_reduces = 0; ctx->band_next_np = 0; ctx->band_agreed = false;
    band_fail(ctx, code == PGW_OK ? PGW_ERR_ARG : code, max_n_iter);
    return PGW_OK;
}

extern "C" int pgw_set_reduce_hook(pgw_ctx *ctx, pgw_reduce_max_fn fn, void *user) {
    if (!ctx) return PGW_ERR_ARG;
    ctx->reduce_fn = fn;
    ctx->reduce_user = fn ? user : nullptr;
    return PGW_OK;
}

// The loop of step_03_apply_to_era.py:182-319 given the iterate-independent vapour pressure
// `evap` = hur_pgw/100 * e_sat(ta_pgw) (functions.py:123).  Shared by pgw_adjust_ps_loop and
// pgw_step03_file.  The host reads max|err| after every pass (status copy + stream synchronisation) before it
// launches the next - the reference's control flow literally (four device-assisted variants, with the passes
// enqueued ahead of the host, all measured slower: DESIGN.md section 4, "tried and dropped").
// local_nplev > 0 selects p_ref_inp = None (step_03:219-253): dzg_b/dzg_a are then the full
// (ntime, nplev, ncol) zg records and `plev_file` the plev coordinate in file order.
static int run_ps_loop(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *PS, const void *FIS,
                       const void *T, const void *QV, const void *ta_pgw, const void *evap,
                       const void *dzg_b, const void *dzg_a, double x_hi, double x_new, double p_ref,
                       double adj_factor, double thresh, int max_n_iter, void *ps_pgw, void *hus_pgw, int *n_iter,
                       double *max_err_hist, int hist_len, int local_nplev = 0, const double *plev_file = nullptr,
                       bool status_armed = false, int qv_done_levels = 0, bool ref = false) {
    const long long n2 = (long long)ntime * ncol;
    void *state = nullptr;
    int rc;
    if (ctx->opt[PGW_OPT_TEST_FAIL] == 1) return fail(ctx, PGW_ERR_HIP, "PGW_OPT_TEST_FAIL = 1: forced workspace failure (ws_get)");
    if ((rc = ws_get(ctx, 1, (size_t)n2 * 6 * sizeof(double), &state))) return rc;
    double *phi_era = (double *)state, *dphi = phi_era + n2, *delta_ps = dphi + n2, *adj_ps = delta_ps + n2;
    double *pref_f = adj_ps + n2;
    int *pref_idx = (int *)(pref_f + n2);
    const int full_column = ctx->opt[PGW_OPT_FULL_COLUMN];
    const bool local = local_nplev > 0;
    // several passes per launch: fixed p_ref, wave-level early exit (the full-column option is a per-pass traffic probe)
    const bool multipass = ctx->opt[PGW_OPT_MULTIPASS] && !full_column;
    NEED(ctx, !ctx->reduce_fn || multipass, "a reduce hook (latitude-band sharding) needs the multi-pass loop: "
                                            "PGW_OPT_MULTIPASS = 1, PGW_OPT_FULL_COLUMN = 0");
    PlevTable ptf;
    memset(&ptf, 0, sizeof(ptf));
    if (local) {
        ptf.n = local_nplev;
        for (int i = 0; i < local_nplev; ++i) ptf.p[i] = plev_file[i];
    }
    if (multipass) {
        // phi_ref of the ERA state, g * dzg and the zeroed state are produced by the first k_ps_loop_multi launch
    } else if (local) {
        HIPCHK(ctx, hipMemsetAsync(delta_ps, 0, sizeof(double) * 2 * n2, ctx->stream));    // :182-184
    } else {
        // phi_ref_era: constant over the iterations for a fixed p_ref (step_03:280-287 recomputes it).
        if (!status_armed && (rc = status_reset(ctx))) return rc;
        launch_phi_ref_hybrid(ctx, dtype, ntime, ncol, T, QV, PS, FIS, p_ref, phi_era, full_column, nullptr, ref);
        HIPCHK(ctx, hipGetLastError());
        // status_armed (pgw_step03_file with the model-top check off): nothing is read back before the first pass;
        // the status block keeps the first error any kernel reported, in stream order, so the first pass's check
        // raises what an immediate check would have raised
        if (!status_armed && (rc = status_check(ctx))) return rc;
        // g * (time-interpolated zg delta at p_ref)   step_03:292-295
        DISPATCH_TLV(dtype, ref, 1, {
            DeltaSrc<T> z{(const T *)dzg_b, (x_hi == 0.0) ? nullptr : (const T *)dzg_a, x_hi, x_new};
            hipLaunchKernelGGL((k_dphi_clim<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, n2, z, CON_G, dphi,
                               delta_ps, adj_ps);                                       // + delta_ps = adj_ps = 0  :182-184
            (void)V;
        });
    }

    if (multipass) {
        // ---- several passes per launch (k_ps_loop_multi); the reference's control flow is applied to the recorded maxima
        void *histv = nullptr;
        if ((rc = ws_get(ctx, 6, (size_t)n2 * MULTI_MAX_PASS * sizeof(double), &histv))) return rc;
        double *dps_hist = (double *)histv;
        DevStatus *mst = ctx->d_status + 2;                                // device per-pass blocks
        DevStatus *hback = ctx->h_status + 1;                              // [0] = block 0 (ERA-state scan / earlier kernels), [1..] passes
        DevStatus *hzero = ctx->h_status + 2 + MULTI_MAX_PASS;
        for (int k = 0; k < MULTI_MAX_PASS; ++k) {
            memset(&hzero[k], 0, sizeof(DevStatus));
            hzero[k].col = ~0ull; hzero[k].min_targ_bits = ~0ull; hzero[k].min_src_bits = ~0ull;
        }
        if (!status_armed && (rc = status_reset(ctx))) return rc;
        int it = 1;
        bool first = true;
        unsigned long long touched = 0;
        const double *conv = nullptr;
        int launched = 0;
        const void *era_T = T, *era_QV = QV;                               // `T` names the storage type inside the dispatch macro
        while (!conv) {
            const int allowed = max_n_iter - (it - 1);                     // passes it .. max_n_iter may still run (:313-319)
            int np = first ? first_launch_passes(ctx, max_n_iter) : 2;    // (a caller-set guess <= 0 counts as 1)
            if (np > allowed) np = allowed;
            if (np < 1) np = 1;
            ctx->band_next_np = np;
            if (first && ctx->opt[PGW_OPT_TEST_FAIL] == 2) return fail(ctx, PGW_ERR_HIP, "PGW_OPT_TEST_FAIL = 2: forced failure before the first loop launch");
            if (!first && ctx->opt[PGW_OPT_TEST_FAIL] == 3) return fail(ctx, PGW_ERR_HIP, "PGW_OPT_TEST_FAIL = 3: forced failure before a continuation launch");
            HIPCHK(ctx, hipMemcpyAsync(mst, hzero, sizeof(DevStatus) * np, hipMemcpyHostToDevice, ctx->stream));
            {
                // one column per lane (two columns: 168 VGPRs + scratch; measured 1.36 vs 1.39 ms before the log table)
                constexpr int MULTI_MAXV = 1;
                int vec = pick_vec(ctx, dtype, ncol, {ta_pgw, evap, era_T, era_QV, PS, FIS, phi_era, dphi, delta_ps, adj_ps, dps_hist}, MULTI_MAXV);
                Levels lv = levels_of(ctx);
                Prof pr(ctx, PGW_K_PS_LOOP_MULTI);
                const LocalPRef loc{ptf, ctx->h_akN, ctx->h_bkN, pref_f, pref_idx};
                DISPATCH_TLV(dtype, ref, vec, {
                    DeltaSrc<T> z{(const T *)dzg_b, (x_hi == 0.0) ? nullptr : (const T *)dzg_a, x_hi, x_new};
                    if (!local)
                        hipLaunchKernelGGL((k_ps_loop_multi<T, TL, V, STEP_U, REF, false>), dim3(nblocks(n2 / V, BLOCK)), dim3(BLOCK), 0,
                                           ctx->stream, lv, ntime, ncol, (const T *)era_T, (const T *)era_QV, (const TL *)ta_pgw,
                                           (const TL *)evap, (const T *)PS, (const T *)FIS, z, phi_era, dphi, delta_ps, adj_ps, dps_hist,
                                           p_ref, adj_factor, first ? 1 : 0, np, ctx->d_status, mst, loc);
                    else if constexpr (V == 1)                                 // MULTI_MAXV = 1: always this branch
                        hipLaunchKernelGGL((k_ps_loop_multi<T, TL, 1, STEP_U, REF, true>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0,
                                           ctx->stream, lv, ntime, ncol, (const T *)era_T, (const T *)era_QV, (const TL *)ta_pgw,
                                           (const TL *)evap, (const T *)PS, (const T *)FIS, z, phi_era, dphi, delta_ps, adj_ps, dps_hist,
                                           0.0, adj_factor, first ? 1 : 0, np, ctx->d_status, mst, loc);
                });
            }
            HIPCHK(ctx, hipGetLastError());
            launched += np;
            HIPCHK(ctx, hipMemcpyAsync(hback, ctx->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(hback + 1, mst, sizeof(DevStatus) * np, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            // figures of this launch; with a reduce hook: their maxima over the bands of the file
            double red[1 + 3 * MULTI_MAX_PASS];
            red[0] = first ? (double)hback[0].code : 0.0;
            for (int k = 0; k < np; ++k) {
                const DevStatus &h = hback[1 + k];
                double e = -INFINITY;
                if (h.valid) memcpy(&e, &h.max_bits, 8);
                red[1 + 3 * k] = (double)h.code; red[2 + 3 * k] = h.valid ? 1.0 : 0.0; red[3 + 3 * k] = e;
            }
            if (ctx->reduce_fn && (rc = call_reduce(ctx, red, 1 + 3 * np))) return rc;
            if (first && red[0] != 0.0) {                                  // earlier kernels of the file / the ERA-state scan
                ctx->band_agreed = true;                                   // every band has this status now
                const bool mine = hback[0].code != 0;                      // (else: another band's status)
                if (mine) *ctx->h_status = hback[0];
                const int code = mine ? (int)hback[0].code : (int)red[0];
                ctx->err_col = mine ? (long long)hback[0].col : -1;
                ctx->err = status_text(code);
                return code;
            }
            for (int k = 0; k < np && !conv; ++k) {
                const DevStatus &h = hback[1 + k];
                if (red[1 + 3 * k] != 0.0) {
                    ctx->band_agreed = true;
                    const int code = h.code != 0 ? (int)h.code : (int)red[1 + 3 * k];
                    ctx->err_col = h.code != 0 ? (long long)h.col : -1;
                    ctx->err = status_text(code);
                    return code;
                }
                const double err_k = red[2 + 3 * k] != 0.0 ? red[3 + 3 * k] : NAN;   // NaN: xarray .max() of an all-NaN field
                touched += h.levels_touched;
                if (max_err_hist && it - 1 < hist_len) max_err_hist[it - 1] = err_k;
                it += 1;                                                   // :313
                if (it > max_n_iter) {                                     // :315-319
                    ctx->band_agreed = true;                               // decided from reduced figures: all bands stop here
                    if (n_iter) *n_iter = it - 1;
                    ctx->last_passes_launched = launched;
                    ctx->err = status_text(PGW_ERR_NOT_CONVERGED);
                    ctx->err_col = -1;
                    return PGW_ERR_NOT_CONVERGED;
                }
                if (!(err_k > thresh)) conv = dps_hist + (size_t)k * n2;   // :189  (NaN stops the loop too)
            }
            first = false;
        }
        ctx->band_agreed = true;       // every band leaves the loop here: nobody waits in a reduce of this file any more
        ctx->opt[PGW_OPT_LOOP_GUESS] = (it - 1) < 1 ? 1 : ((it - 1) > MULTI_MAX_PASS ? MULTI_MAX_PASS : (it - 1));
        ctx->last_levels_touched = touched;
        ctx->last_passes_launched = launched;
        if (n_iter) *n_iter = it - 1;
        if (ps_pgw || hus_pgw) {
            int vec = pick_vec(ctx, dtype, ncol, {PS, evap, ps_pgw, hus_pgw, conv});
            Levels lv = levels_of(ctx);
            Prof pr(ctx, PGW_K_FINALIZE);
            DISPATCH_TLV(dtype, ref, vec, hipLaunchKernelGGL((k_finalize_ps_hus<T, TL, V, REF>), dim3(nblocks(n2 / V, BLOCK)), dim3(BLOCK), 0,
                                                             ctx->stream, lv, ntime, ncol, (const T *)PS, conv, (const TL *)evap,
                                                             (T *)ps_pgw, (TL *)hus_pgw, qv_done_levels));
        }
        HIPCHK(ctx, hipGetLastError());
        return PGW_OK;
    }

    double phi_ref_max_error = INFINITY;                                   // :186
    int it = 1;                                                            // :188
    unsigned long long touched = 0;
    // Fixed p_ref: the passes alternate between the two status blocks and each pass clears the other one for its
    // successor, so no reset copy is enqueued per pass (the read-back of the block being cleared was enqueued
    // before this pass was launched).
    DevStatus *blk[2] = {ctx->d_status, ctx->d_status + 1};
    while (phi_ref_max_error > thresh) {                                   // :189
        const bool reset_here = local || it == 1;
        if (reset_here && !(status_armed && it == 1) && (rc = status_reset(ctx))) return rc;
        DevStatus *cur = local ? ctx->d_status : blk[(it - 1) & 1];
        if (local) {
            // delta_ps += adj_ps ; per-column p_ref (never lower than last pass) ; g*zg at that level
            DISPATCH_TLV(dtype, ref, 1, {
                DeltaSrc<T> z{(const T *)dzg_b, (x_hi == 0.0) ? nullptr : (const T *)dzg_a, x_hi, x_new};
                hipLaunchKernelGGL((k_local_p_ref<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, ptf, ctx->h_akN,
                                   ctx->h_bkN, n2, (const T *)PS, delta_ps, adj_ps, z, ncol, it == 1 ? 1 : 0, pref_f,
                                   pref_idx, dphi, ctx->d_status);
                (void)V;
            });
            launch_phi_ref_hybrid(ctx, dtype, ntime, ncol, T, QV, PS, FIS, 0.0, phi_era, full_column, pref_f, ref);   // :280-287
            launch_step(ctx, dtype, ntime, ncol, ta_pgw, evap, PS, FIS, phi_era, dphi, delta_ps, adj_ps, 0.0, pref_f,
                        adj_factor, full_column, 0, nullptr, nullptr, ref);
        } else {
            launch_step(ctx, dtype, ntime, ncol, ta_pgw, evap, PS, FIS, phi_era, dphi, delta_ps, adj_ps, p_ref, nullptr,
                        adj_factor, full_column, 1, cur, blk[it & 1], ref);
        }
        HIPCHK(ctx, hipGetLastError());
        if ((rc = status_check(ctx, cur))) return rc;
        phi_ref_max_error = max_err_of(ctx);                               // :308
        touched += ctx->h_status->levels_touched;
        if (max_err_hist && it - 1 < hist_len) max_err_hist[it - 1] = phi_ref_max_error;
        it += 1;                                                           // :313
        if (it > max_n_iter) {                                             // :315-319
            if (n_iter) *n_iter = it - 1;
            ctx->err = status_text(PGW_ERR_NOT_CONVERGED);
            ctx->err_col = -1;
            return PGW_ERR_NOT_CONVERGED;
        }
    }
    ctx->last_levels_touched = touched;
    if (n_iter) *n_iter = it - 1;
    if (ps_pgw || hus_pgw) {
        int vec = pick_vec(ctx, dtype, ncol, {PS, evap, ps_pgw, hus_pgw, delta_ps});
        Levels lv = levels_of(ctx);
        Prof pr(ctx, PGW_K_FINALIZE);
        DISPATCH_TLV(dtype, ref, vec, hipLaunchKernelGGL((k_finalize_ps_hus<T, TL, V, REF>), dim3(nblocks(n2 / V, BLOCK)), dim3(BLOCK), 0,
                                                         ctx->stream, lv, ntime, ncol, (const T *)PS, delta_ps, (const TL *)evap,
                                                         (T *)ps_pgw, (TL *)hus_pgw, qv_done_levels));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_adjust_ps_loop(pgw_ctx *ctx, int dtype, int ntime, long long ncol, const void *PS,
                                  const void *FIS, const void *T, const void *QV, const void *ta_pgw,
                                  const void *hur_pgw, const void *dzg_pref, double p_ref, double adj_factor,
                                  double thresh, int max_n_iter, void *ps_pgw, void *hus_pgw, int *n_iter,
                                  double *max_err_hist) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0, "pgw_set_levels has not been called");
    NEED(ctx, PS && FIS && T && QV && ta_pgw && hur_pgw && dzg_pref, "null pointer");
    const size_t es = dtype == PGW_F64 ? 8 : 4;
    void *evap = nullptr;
    int rc;
    if ((rc = ws_get(ctx, 0, (size_t)ntime * ctx->nlev * ncol * es, &evap))) return rc;
    // e = hur_pgw/100 * e_sat(ta_pgw): iterate-independent part of :262-266
    if ((rc = humidity_hybrid<2>(ctx, PGW_K_RH_TO_Q, dtype, ntime, ncol, hur_pgw, PS, ta_pgw, evap))) return rc;
    return run_ps_loop(ctx, dtype, ntime, ncol, PS, FIS, T, QV, ta_pgw, evap, dzg_pref, nullptr, 0.0, 0.0, p_ref,
                       adj_factor, thresh, max_n_iter, ps_pgw, hus_pgw, n_iter, max_err_hist, max_n_iter);
}

// ------------------------------------------------------------------ whole file, settings.i_reinterp = 1
// step_03_apply_to_era.py:182-343 with i_reinterp = 1: in every pass the ERA ta / hur fields and their deltas are
// interpolated onto the CURRENT model-level pressures (:202-216), ua / va once after convergence (:330-343).  Fixed or local
// reference level (p_ref_inp = None, :219-253), float64 / float32 storage, reference-dtype mode on float32 files.  One launch
// sequence and one host read-back (max |err|, status) per pass - the loop control is the reference's.
static int run_reinterp_file(pgw_ctx *ctx, pgw_file_args *a, int check_top) {
    const int dtype = a->dtype, ntime = a->ntime, N = a->nlev, S = a->nplev;
    const long long ncol = a->ncol, n2 = (long long)ntime * ncol;
    const bool ref = a->ref_dtype != 0, local = a->local_p_ref != 0, lerp = (a->x_hi != 0.0);
    const size_t es = (dtype == PGW_F64 || ref) ? 8 : 4;       // element size of the level arrays the loop produces
    NEED(ctx, !ctx->reduce_fn, "latitude-band sharding needs the multi-pass loop (i_reinterp = 0)");
    int rc;
    void *evap = nullptr, *relhum = nullptr, *hur = a->hur_pgw_out, *state = nullptr;
    const size_t field = (size_t)ntime * N * ncol * es;
    if ((rc = ws_get(ctx, 0, field, &evap))) return rc;
    if ((rc = ws_get(ctx, 2, field, &relhum))) return rc;
    if (!hur && (rc = ws_get(ctx, 7, field, &hur))) return rc;
    if ((rc = ws_get(ctx, 1, (size_t)n2 * 6 * sizeof(double), &state))) return rc;
    double *phi_era = (double *)state, *dphi = phi_era + n2, *delta_ps = dphi + n2, *adj_ps = delta_ps + n2;
    double *pref_f = adj_ps + n2;
    int *pref_idx = (int *)(pref_f + n2);
    Levels lv = levels_of(ctx);
    const int full_column = ctx->opt[PGW_OPT_FULL_COLUMN];
    const double zx_hi = a->per_var_time ? a->zg_x_hi : a->x_hi, zx_new = a->per_var_time ? a->zg_x_new : a->x_new;
    PlevTable ptf;
    memset(&ptf, 0, sizeof(ptf));
    if (local) {
        NEED(ctx, a->zg3_b != nullptr, "local_p_ref needs the full zg records (zg3_b / zg3_a)");
        ptf.n = S;
        for (int i = 0; i < S; ++i) ptf.p[i] = a->plev[i];
    }
    if ((rc = status_reset(ctx))) return rc;
    // ---- ERA state: RELHUM (step_03:87-94)
    {
        Prof pr(ctx, PGW_K_Q_TO_RH);
        if (ref) {
            hipLaunchKernelGGL(k_relhum_ref, dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, lv, ntime, ncol,
                               (const float *)a->QV, (const float *)a->PS, (const float *)a->T, (double *)relhum);
        } else {
            int vec = pick_vec(ctx, dtype, ncol, {a->QV, a->PS, a->T, relhum});
            DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((k_humidity_hybrid<T, V, 0>), dim3(nblocks(n2 / V, BLOCK)), dim3(BLOCK), 0,
                                                        ctx->stream, lv, ntime, ncol, (const T *)a->QV, (const T *)a->PS,
                                                        (const T *)a->T, (T *)relhum));
        }
    }
    if (!local) {
        // phi_ref of the ERA state (constant for a fixed p_ref; :280-287 recomputes it) and g * dzg (:292-295); zeroed state
        launch_phi_ref_hybrid(ctx, dtype, ntime, ncol, a->T, a->QV, a->PS, a->FIS, a->p_ref, phi_era, full_column, nullptr, ref);
        DISPATCH_TLV(dtype, ref, 1, {
            DeltaSrc<T> z{(const T *)a->zg_b, (zx_hi == 0.0) ? nullptr : (const T *)a->zg_a, zx_hi, zx_new};
            hipLaunchKernelGGL((k_dphi_clim<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, n2, z, CON_G, dphi,
                               delta_ps, adj_ps);
            (void)V;
        });
    } else {
        HIPCHK(ctx, hipMemsetAsync(delta_ps, 0, sizeof(double) * 2 * n2, ctx->stream));    // :182-184
    }
    HIPCHK(ctx, hipGetLastError());
    if ((rc = status_check(ctx))) return rc;

    // one pair of variables (ERA fields + deltas) onto the levels of ps_pgw; thermo: ta + hur with e, else ua + va
    auto reinterp = [&](bool thermo) {
        Prof pr(ctx, PGW_K_VERT_INTERP_DELTA);
        const void *b0 = thermo ? a->ta_b : a->ua_b, *a0 = thermo ? a->ta_a : a->ua_a;
        const void *b1 = thermo ? a->hur_b : a->va_b, *a1 = thermo ? a->hur_a : a->va_a;
#define PGW_FILL_PAIR(rv)                                                                                                  \
        rv.d[0] = DeltaSrc<T>{(const T *)b0, lerp ? (const T *)a0 : nullptr, a->x_hi, a->x_new};                           \
        rv.d[1] = DeltaSrc<T>{(const T *)b1, lerp ? (const T *)a1 : nullptr, a->x_hi, a->x_new};                           \
        rv.sfc[0] = DeltaSrc<T>{thermo ? (const T *)a->tas_b : nullptr, (thermo && lerp) ? (const T *)a->tas_a : nullptr, a->x_hi, a->x_new};   \
        rv.sfc[1] = DeltaSrc<T>{thermo ? (const T *)a->hurs_b : nullptr, (thermo && lerp) ? (const T *)a->hurs_a : nullptr, a->x_hi, a->x_new}; \
        DeltaSrc<T> p{(const T *)a->pshist_b, lerp ? (const T *)a->pshist_a : nullptr, a->x_hi, a->x_new};
        if (ref) {
            typedef float T;
            if (thermo) {
                ReinterpPair<float, float, double, double> rv;
                PGW_FILL_PAIR(rv)
                rv.era0 = (const float *)a->T; rv.era1 = (const double *)relhum;
                rv.out[0] = (double *)a->T_out; rv.out[1] = (double *)hur; rv.evap = (double *)evap;
                launch_reinterp_pair<float, float, double, double, true>(ctx, lv, ntime, S, ncol, rv, p, (const float *)a->PS,
                                                                         (const float *)a->PS_out, true, check_top);
            } else {
                ReinterpPair<float, float, float, double> rv;
                PGW_FILL_PAIR(rv)
                rv.era0 = (const float *)a->U; rv.era1 = (const float *)a->V;
                rv.out[0] = (double *)a->U_out; rv.out[1] = (double *)a->V_out; rv.evap = nullptr;
                launch_reinterp_pair<float, float, float, double, true>(ctx, lv, ntime, S, ncol, rv, p, (const float *)a->PS,
                                                                        (const float *)a->PS_out, false, check_top);
            }
        } else {
            DISPATCH_T(dtype, {
                ReinterpPair<T> rv;
                PGW_FILL_PAIR(rv)
                rv.era0 = (const T *)(thermo ? a->T : a->U); rv.era1 = (const T *)(thermo ? relhum : a->V);
                rv.out[0] = (T *)(thermo ? a->T_out : a->U_out); rv.out[1] = (T *)(thermo ? hur : a->V_out);
                rv.evap = thermo ? (T *)evap : nullptr;
                launch_reinterp_pair<T>(ctx, lv, ntime, S, ncol, rv, p, (const T *)a->PS, (const T *)a->PS_out, thermo, check_top);
            });
        }
#undef PGW_FILL_PAIR
    };
    auto top_check = [&]() -> int {                       // functions.py:417-425
        if (!check_top) return PGW_OK;
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
        return PGW_OK;
    };

    double err = INFINITY;                                                  // :186
    int it = 1;                                                             // :188
    a->n_iter = 0;
    for (int i = 0; i < 32; ++i) a->max_err_hist[i] = NAN;
    while (err > a->thresh) {                                               // :189
        if ((rc = status_reset(ctx))) return rc;
        if (local) {
            // delta_ps += adj_ps; the column's reference level (never lower than the last pass's); g * zg there   :192, 219-253, 292-295
            DISPATCH_TLV(dtype, ref, 1, {
                DeltaSrc<T> z{(const T *)a->zg3_b, (zx_hi == 0.0) ? nullptr : (const T *)a->zg3_a, zx_hi, zx_new};
                hipLaunchKernelGGL((k_local_p_ref<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, ptf, ctx->h_akN,
                                   ctx->h_bkN, n2, (const T *)a->PS, delta_ps, adj_ps, z, ncol, it == 1 ? 1 : 0, pref_f,
                                   pref_idx, dphi, ctx->d_status);
                hipLaunchKernelGGL((k_update_ps<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, n2,
                                   (const T *)a->PS, delta_ps, adj_ps, (T *)a->PS_out, 0);                 // :193
                (void)V;
            });
            launch_phi_ref_hybrid(ctx, dtype, ntime, ncol, a->T, a->QV, a->PS, a->FIS, 0.0, phi_era, full_column, pref_f, ref);   // :280-287
        } else {
            DISPATCH_TLV(dtype, ref, 1, {                                                                  // :192-193
                hipLaunchKernelGGL((k_update_ps<T, REF>), dim3(nblocks(n2, BLOCK)), dim3(BLOCK), 0, ctx->stream, n2,
                                   (const T *)a->PS, delta_ps, adj_ps, (T *)a->PS_out, 1);
                (void)V;
            });
        }
        reinterp(true);                                                     // :202-216 + e of functions.py:123
        // :262-308 on the re-interpolated fields (delta_ps already carries this pass's increment)
        launch_step(ctx, dtype, ntime, ncol, a->T_out, evap, a->PS, a->FIS, phi_era, dphi, delta_ps, adj_ps, a->p_ref,
                    local ? pref_f : nullptr, a->adj_factor, full_column, 0, nullptr, nullptr, ref);
        HIPCHK(ctx, hipGetLastError());
        if ((rc = status_check(ctx))) return rc;
        if ((rc = top_check())) return rc;
        err = max_err_of(ctx);                                              // :308
        if (it - 1 < 32) a->max_err_hist[it - 1] = err;
        it += 1;                                                            // :313
        if (it > a->max_n_iter) {                                           // :315-319
            a->n_iter = it - 1;
            ctx->err = status_text(PGW_ERR_NOT_CONVERGED);
            ctx->err_col = -1;
            return PGW_ERR_NOT_CONVERGED;
        }
    }
    a->n_iter = it - 1;
    a->passes_launched = it - 1;
    a->levels_touched = 0;
    if ((rc = status_reset(ctx))) return rc;
    reinterp(false);                                                        // ua, va on the final levels   :330-343
    {   // hus of the last pass (:262-266, 370) from its e; PS_out already holds ps_pgw of the last pass
        int vec = pick_vec(ctx, dtype, ncol, {a->PS, evap, a->PS_out, a->QV_out, delta_ps});
        Prof pr(ctx, PGW_K_FINALIZE);
        DISPATCH_TLV(dtype, ref, vec, hipLaunchKernelGGL((k_finalize_ps_hus<T, TL, V, REF>), dim3(nblocks(n2 / V, BLOCK)), dim3(BLOCK), 0,
                                                         ctx->stream, lv, ntime, ncol, (const T *)a->PS, delta_ps, (const TL *)evap,
                                                         (T *)a->PS_out, (TL *)a->QV_out, 0));
    }
    HIPCHK(ctx, hipGetLastError());
    if ((rc = status_check(ctx))) return rc;
    return top_check();
}

// ------------------------------------------------------------------ whole file
static int step03_file(pgw_ctx *ctx, pgw_file_args *a);

extern "C" int pgw_step03_file(pgw_ctx *ctx, pgw_file_args *a) {
    if (!ctx) return PGW_ERR_ARG;
    ctx->band_reduces = 0; ctx->band_next_np = 0; ctx->band_agreed = false;
    const int rc = step03_file(ctx, a);
    // latitude-band mode: whatever made this band stop on its own - an argument check, a failed allocation, a data error
    // before the loop, a HIP error between two loop launches - reaches the other bands through their next reduce
    return (rc != PGW_OK && ctx->reduce_fn) ? band_fail(ctx, rc, a ? a->max_n_iter : 1) : rc;
}

static int step03_file(pgw_ctx *ctx, pgw_file_args *a) {
    NEED(ctx, a != nullptr, "null args");
    const int dtype = a->dtype, ntime = a->ntime;
    const long long ncol = a->ncol;
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, ctx->nlev > 0 && a->nlev == ctx->nlev, "nlev must match pgw_set_levels");
    NEED(ctx, a->nplev >= 2 && a->nplev <= MAX_PLEV, "nplev must be in [2, 64]");
    NEED(ctx, a->PS && a->FIS && a->T && a->QV && a->U && a->V, "ERA5 field pointer is NULL");
    NEED(ctx, a->plev && a->ta_b && a->hur_b && a->ua_b && a->va_b && (a->zg_b || a->local_p_ref) && a->tas_b && a->hurs_b &&
         a->pshist_b, "delta record pointer is NULL");
    NEED(ctx, a->PS_out && a->T_out && a->QV_out && a->U_out && a->V_out, "output pointer is NULL");
    NEED(ctx, a->max_n_iter >= 1 && a->max_n_iter <= 1000, "bad max_n_iter");
    NEED(ctx, !ctx->reduce_fn || (ctx->opt[PGW_OPT_MULTIPASS] && !ctx->opt[PGW_OPT_FULL_COLUMN]),
         "a reduce hook (latitude-band sharding) needs the multi-pass loop: PGW_OPT_MULTIPASS = 1, PGW_OPT_FULL_COLUMN = 0");
    const bool exact = (a->x_hi == 0.0);
    const bool ref = a->ref_dtype != 0;
    NEED(ctx, !ref || dtype == PGW_F32, "ref_dtype = 1 is the float32-file mode: dtype must be PGW_F32");
    NEED(ctx, !ref || ctx->opt[PGW_OPT_QUAD], "ref_dtype = 1 needs the quad kernel (PGW_OPT_QUAD = 1)");
    const size_t es = (dtype == PGW_F64 || ref) ? 8 : 4;      // element size of the PGW level arrays (evap, 4-D outputs)
    const int N = a->nlev;
    int rc;
    int qv_done = 0;          // leading levels whose final QV the quad kernel has already written
    if ((rc = plev_table(ctx, a->nplev, a->plev))) return rc;
    void *evap = nullptr;
    if ((rc = ws_get(ctx, 0, (size_t)ntime * N * ncol * es, &evap))) return rc;
    Levels lv = levels_of(ctx);
    const int check_top = a->ignore_top ? 0 : 1;
    auto top_check = [&]() -> int {                       // functions.py:417-425
        if (!check_top) return PGW_OK;
        DevStatus *h = ctx->h_status;
        if (!h->nan_seen && h->min_targ_bits != ~0ull && h->min_src_bits != ~0ull) {
            double mt, ms;
            memcpy(&mt, &h->min_targ_bits, 8);
            memcpy(&ms, &h->min_src_bits, 8);
            if (mt < ms) { ctx->err = status_text(PGW_ERR_TOP_PRESSURE); ctx->err_col = -1; return PGW_ERR_TOP_PRESSURE; }
        }
        return PGW_OK;
    };

    // ---- surface riders (step_03:103-146)
    if (a->FR_SEA_ICE && a->siconc_b && a->FR_SEA_ICE_out) {
        NEED(ctx, a->ts_b && a->tos_b && a->FR_LAND && a->T_SKIN && a->T_SKIN_out, "surface rider pointer is NULL");
        NEED(ctx, a->nsoil >= 0 && a->nsoil <= MAX_SOIL, "nsoil must be in [0, 16]");
        NEED(ctx, a->nsoil == 0 || (a->T_SO && a->T_SO_out && a->ts_clim && a->soil_depth), "soil pointers missing");
        SoilTable st;
        memset(&st, 0, sizeof(st));
        st.n = a->nsoil;
        for (int s = 0; s < a->nsoil; ++s) st.w[s] = exp(-a->soil_depth[s] / 2.8);      // step_03:140
        long long n = (long long)ntime * ncol;
        Prof pr(ctx, PGW_K_SURFACE);
        DISPATCH_TLV(dtype, ref, 1, {
            // each delta file has its own time axis in the reference (load_delta per variable): own bracket, own abscissae
            const double sx = a->per_var_time ? a->siconc_x_hi : a->x_hi, sn = a->per_var_time ? a->siconc_x_new : a->x_new;
            const double tx = a->per_var_time ? a->ts_x_hi : a->x_hi, tn = a->per_var_time ? a->ts_x_new : a->x_new;
            const double ox = a->per_var_time ? a->tos_x_hi : a->x_hi, on = a->per_var_time ? a->tos_x_new : a->x_new;
            DeltaSrc<T> dsic{(const T *)a->siconc_b, sx == 0.0 ? nullptr : (const T *)a->siconc_a, sx, sn};
            DeltaSrc<T> dts{(const T *)a->ts_b, tx == 0.0 ? nullptr : (const T *)a->ts_a, tx, tn};
            DeltaSrc<T> dtos{(const T *)a->tos_b, ox == 0.0 ? nullptr : (const T *)a->tos_a, ox, on};
            hipLaunchKernelGGL((k_surface_update_lerp<T, REF>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, ntime, ncol, st,
                               (const T *)a->FR_SEA_ICE, dsic, dtos, dts, (const T *)a->FR_LAND, (const T *)a->ts_clim,
                               (const T *)a->T_SKIN, (const T *)a->T_SO, (T *)a->FR_SEA_ICE_out, (T *)a->T_SKIN_out,
                               (T *)a->T_SO_out);
            (void)V;
        });
    }

    if (a->i_reinterp) {                                  // settings.i_reinterp = 1: the loop re-interpolates in every pass
        a->levels_touched = 0; a->passes_launched = 0;
        return run_reinterp_file(ctx, a, check_top);
    }

    // ---- ta + hur -> T_pgw, e_pgw   and   ua + va -> U_pgw, V_pgw
    // One column per thread: these kernels are bound by fp64 VALU work and register footprint, not by load width.
    {
        const int S = a->nplev;
        // one host synchronisation per file: without the model-top check nothing has to be read back
        // between the kernels (errors stay in the status block until the loop's first check)
        const bool defer = !check_top && !a->local_p_ref;
        if ((rc = status_reset(ctx))) return rc;
        if (ctx->opt[PGW_OPT_QUAD]) {
            // ---- all four variables in one kernel (production)
            const size_t qlds = (size_t)3 * N * sizeof(double);
            qv_done = ctx->opt[PGW_OPT_FULL_COLUMN] ? 0 : ctx->n_pure;     // full-column passes read e at every level
            {
                Prof pr(ctx, PGW_K_QUAD_DELTA);
#define LAUNCH_QUAD(OT, LERP_)                                                                                        \
                    hipLaunchKernelGGL((k_delta_quad<T, TL, QUAD_U, QUAD_TPB, OT, LERP_, REF>), dim3(nblocks((long long)ntime * ncol, QUAD_TPB)), \
                                       dim3(QUAD_TPB), qlds, ctx->stream, ctx->plev_tab, lv, ntime, ncol, (const T *)a->T,     \
                                       (const T *)a->QV, (const T *)a->U, (const T *)a->V, (const T *)a->PS, dth, ds, ph, \
                                       dwd, check_top, (TL *)a->T_out, (TL *)evap, (TL *)a->hur_pgw_out, (TL *)a->U_out,      \
                                       (TL *)a->V_out, (TL *)a->QV_out, qv_done, ctx->n_pure, ctx->d_status)
                DISPATCH_TLV(dtype, ref, 1, {
                    PairSrc<T> dth{{(const T *)a->ta_b, exact ? nullptr : (const T *)a->ta_a, a->x_hi, a->x_new},
                                   {(const T *)a->hur_b, exact ? nullptr : (const T *)a->hur_a, a->x_hi, a->x_new}};
                    PairSrc<T> ds{{(const T *)a->tas_b, exact ? nullptr : (const T *)a->tas_a, a->x_hi, a->x_new},
                                  {(const T *)a->hurs_b, exact ? nullptr : (const T *)a->hurs_a, a->x_hi, a->x_new}};
                    DeltaSrc<T> ph{(const T *)a->pshist_b, exact ? nullptr : (const T *)a->pshist_a, a->x_hi, a->x_new};
                    PairSrc<T> dwd{{(const T *)a->ua_b, exact ? nullptr : (const T *)a->ua_a, a->x_hi, a->x_new},
                                   {(const T *)a->va_b, exact ? nullptr : (const T *)a->va_a, a->x_hi, a->x_new}};
                    // arrays below 4 GiB (a 0.25 deg L137 field is 1.1 GB): 32-bit byte offsets from uniform bases
                    const bool o32 = !ctx->opt[PGW_OPT_FORCE_OFF64] &&
                                     (unsigned long long)ntime * (N > S ? N : S) * ncol * sizeof(TL) < (1ull << 32);
                    // LERP: the instant lies between two records (false: it is a record, `exact`)
                    if (o32) { if (exact) LAUNCH_QUAD(boff32, false); else LAUNCH_QUAD(boff32, true); }
                    else { if (exact) LAUNCH_QUAD(boff64, false); else LAUNCH_QUAD(boff64, true); }
                    (void)V;
                });
#undef LAUNCH_QUAD
            }
            HIPCHK(ctx, hipGetLastError());
        } else {
            // ---- PGW_OPT_QUAD = 0: the two pair kernels the quad kernel replaced (kept as an independently written
            // cross-check of the same arithmetic: tests/test_hip_parity.py::test_kernel_variants_are_bit_identical)
            const size_t lds = (size_t)2 * N * sizeof(double);            // akm | bkm
            const unsigned int grid = nblocks((long long)ntime * ncol, 128);
#define LAUNCH_PAIR(THERMO, FA, FB, D3, DS, PH, OA, OB, OH)                                                                   \
        hipLaunchKernelGGL((k_delta_pair<T, 1, THERMO, (THERMO ? 2 : PAIR_U), 128>), dim3(grid), dim3(128), lds, ctx->stream,  \
                           ctx->plev_tab, lv, ntime, ncol, FA, FB, (const T *)a->PS, D3, DS, PH, check_top, OA, OB, OH, \
                           ctx->d_status)
            {
                Prof pr(ctx, PGW_K_THERMO_DELTA);
                DISPATCH_T(dtype, {
                    PairSrc<T> d3{{(const T *)a->ta_b, exact ? nullptr : (const T *)a->ta_a, a->x_hi, a->x_new},
                                  {(const T *)a->hur_b, exact ? nullptr : (const T *)a->hur_a, a->x_hi, a->x_new}};
                    PairSrc<T> ds{{(const T *)a->tas_b, exact ? nullptr : (const T *)a->tas_a, a->x_hi, a->x_new},
                                  {(const T *)a->hurs_b, exact ? nullptr : (const T *)a->hurs_a, a->x_hi, a->x_new}};
                    DeltaSrc<T> ph{(const T *)a->pshist_b, exact ? nullptr : (const T *)a->pshist_a, a->x_hi, a->x_new};
                    LAUNCH_PAIR(true, (const T *)a->T, (const T *)a->QV, d3, ds, ph, (T *)a->T_out, (T *)evap, (T *)a->hur_pgw_out);
                });
            }
            {
                Prof pr(ctx, PGW_K_WIND_DELTA);
                DISPATCH_T(dtype, {
                    PairSrc<T> d3{{(const T *)a->ua_b, exact ? nullptr : (const T *)a->ua_a, a->x_hi, a->x_new},
                                  {(const T *)a->va_b, exact ? nullptr : (const T *)a->va_a, a->x_hi, a->x_new}};
                    PairSrc<T> ds{{nullptr, nullptr, 0.0, 0.0}, {nullptr, nullptr, 0.0, 0.0}};
                    DeltaSrc<T> ph{nullptr, nullptr, 0.0, 0.0};
                    LAUNCH_PAIR(false, (const T *)a->U, (const T *)a->V, d3, ds, ph, (T *)a->U_out, (T *)a->V_out, (T *)nullptr);
                });
            }
#undef LAUNCH_PAIR
            HIPCHK(ctx, hipGetLastError());
        }
        if (!defer) {
            if ((rc = status_check(ctx))) return rc;            // (latitude-band mode: pgw_step03_file meets the other bands)
            if ((rc = top_check())) return rc;
        }
    }

    // ---- fixed-point loop + final PS, QV
    a->n_iter = 0;
    for (int i = 0; i < 32; ++i) a->max_err_hist[i] = NAN;
    if (a->local_p_ref) NEED(ctx, a->zg3_b != nullptr, "local_p_ref needs the full zg records (zg3_b / zg3_a)");
    rc = run_ps_loop(ctx, dtype, ntime, ncol, a->PS, a->FIS, a->T, a->QV, a->T_out, evap,
                     a->local_p_ref ? a->zg3_b : a->zg_b, a->local_p_ref ? a->zg3_a : a->zg_a,
                     a->per_var_time ? a->zg_x_hi : a->x_hi, a->per_var_time ? a->zg_x_new : a->x_new, a->p_ref, a->adj_factor, a->thresh, a->max_n_iter, a->PS_out, a->QV_out, &a->n_iter,
                     a->max_err_hist, 32, a->local_p_ref ? a->nplev : 0, a->plev,
                     !check_top && !a->local_p_ref, qv_done, ref);
    a->levels_touched = ctx->last_levels_touched;
    a->passes_launched = ctx->last_passes_launched;
    return rc;
}

extern "C" unsigned long long pgw_last_levels_touched(pgw_ctx *ctx) { return ctx->last_levels_touched; }

extern "C" int pgw_test_exp(pgw_ctx *ctx, long long n, const double *in, double *out, double *ref) {
    NEED(ctx, n >= 1 && in && out && ref, "bad argument");
    hipLaunchKernelGGL(k_test_exp, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, n, in, out, ref);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_test_rh_f32(pgw_ctx *ctx, long long n, const float *hus, const double *pa, const float *ta,
                               double *out, double *lit, float *es, float *es_lit) {
    NEED(ctx, n >= 1 && hus && pa && ta && out && lit && es && es_lit, "bad argument");
    hipLaunchKernelGGL(k_test_rh_f32, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, n, hus, pa, ta, out, lit, es, es_lit);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_test_shared_div(pgw_ctx *ctx, long long n, const double *num, const double *den, double *out) {
    NEED(ctx, n >= 1 && num && den && out, "bad argument");
    hipLaunchKernelGGL(k_test_shared_div, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, n, num, den, out);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_harmonic_smooth(pgw_ctx *ctx, int dtype, int ntime, long long inner, const double *cos_tab,
                                   const double *sin_tab, const void *in, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, inner >= 1 && in && out && cos_tab && sin_tab, "bad argument");
    // functions.py:724-737: the first three harmonics need 3 < floor(ntime / 2)
    if (!(3 < ntime / 2))
        return fail(ctx, PGW_ERR_ARG, "Whooops that should not be the case for a yearly timeseries! i (reconstruction grade) "
                                      "is larger than the number of timeseries elements / 2.");
    const size_t lds = sizeof(double) * 6 * (size_t)ntime;
    NEED(ctx, lds <= 64 * 1024, "time series longer than 1365 steps are not supported");
    void *tab = nullptr;
    int rc = ws_get(ctx, 5, lds, &tab);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(tab, cos_tab, lds / 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync((char *)tab + lds / 2, sin_tab, lds / 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                  // the host tables may be freed after the call
    {
        Prof pr(ctx, PGW_K_HARMONIC);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_harmonic_smooth<T, 8>), dim3(nblocks(inner, BLOCK)), dim3(BLOCK), lds, ctx->stream,
                                             ntime, inner, (const double *)tab, (const T *)in, (T *)out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_gauss_interp(pgw_ctx *ctx, long long ntarg, const double *tx, const double *ty, int ncx, int ncy,
                                double x0, double y0, double cell, const int *cell_start, long long nsrc, const double *sx,
                                const double *sy, const double *sval, int nfield, double radius, double sharpness, double *out) {
    NEED(ctx, ntarg >= 1 && nsrc >= 0 && tx && ty && cell_start && out, "bad argument");
    NEED(ctx, nsrc == 0 || (sx && sy && sval), "null source pointer");
    NEED(ctx, ncx >= 1 && ncy >= 1 && cell > 0.0 && radius > 0.0, "bad cell grid");
    NEED(ctx, cell >= radius, "cells must be at least one kernel radius wide (3 x 3 block search)");
    NEED(ctx, nfield >= 1 && nfield <= GAUSS_MAX_FIELDS, "nfield must be in [1, 16]");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double r2 = radius * radius, f2 = (sharpness * sharpness) / (radius * radius);     // vtkGaussianKernel: F2 = (Sharpness / Radius)^2
    {
        Prof pr(ctx, PGW_K_GAUSS_INTERP);
        hipLaunchKernelGGL((k_gauss_interp<GAUSS_MAX_FIELDS>), dim3(nblocks(ntarg, BLOCK)), dim3(BLOCK), 0, ctx->stream, ntarg, tx, ty,
                           ncx, ncy, x0, y0, 1.0 / cell, cell_start, sx, sy, sval, nfield, r2, f2, out);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_planar_metres(pgw_ctx *ctx, long long n, const double *lat, const double *lon, double *lat_m, double *lon_m,
                                 double *lon_off) {
    NEED(ctx, n >= 0 && (n == 0 || (lat && lon && lat_m && lon_m && lon_off)), "bad argument");
    if (n == 0) return PGW_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_planar_metres, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, n, lat, lon, lat_m, lon_m, lon_off);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_placement_probe(pgw_ctx *ctx, int n_src, const void *const *src, int n_dst, void *const *dst, long long rows,
                                   long long ncol, int reps, double *gbps) {
    NEED(ctx, n_src >= 0 && n_src <= 4 && n_dst >= 0 && n_dst <= 4 && n_src + n_dst > 0, "1 to 4 + 4 streams");
    NEED(ctx, rows > 0 && ncol > 0 && reps > 0 && gbps, "bad argument");
    ProbeStreams s;
    s.ns = n_src; s.nd = n_dst;
    for (int i = 0; i < 4; ++i) {
        s.src[i] = i < n_src ? (const double *)src[i] : nullptr;
        s.dst[i] = i < n_dst ? (double *)dst[i] : nullptr;
        NEED(ctx, (i >= n_src || (s.src[i] && ((uintptr_t)s.src[i] % 8) == 0)) && (i >= n_dst || (s.dst[i] && ((uintptr_t)s.dst[i] % 8) == 0)),
             "stream pointers must be non-null and 8-byte aligned");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;             // its own pair: pgw_timer_start / _stop of the caller stay untouched
    HIPCHK(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return fail(ctx, PGW_ERR_HIP, "hipEventCreate failed"); }
    const unsigned int nb = nblocks(ncol, 128);
    hipLaunchKernelGGL(k_placement_probe, dim3(nb), dim3(128), 0, ctx->stream, rows, ncol, s);       // warm-up
    hipError_t e = hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_placement_probe, dim3(nb), dim3(128), 0, ctx->stream, rows, ncol, s);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) return fail(ctx, PGW_ERR_HIP, "placement probe failed: %s", hipGetErrorString(e));
    *gbps = ms > 0.f ? (double)rows * (double)ncol * 8.0 * (n_src + n_dst) * reps / ((double)ms * 1e6) : 0.0;
    return PGW_OK;
}

extern "C" int pgw_ws_adopt(pgw_ctx *ctx, int slot, void *dptr, size_t bytes) {
    NEED(ctx, slot >= 0 && slot < 8, "workspace slot out of range");
    NEED(ctx, (dptr != nullptr) == (bytes > 0), "a buffer and its size, or neither (release the slot)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->ws[slot] && ctx->ws[slot] != dptr) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(ctx->ws[slot]));
    }
    ctx->ws[slot] = dptr; ctx->ws_bytes[slot] = bytes;
    return PGW_OK;
}

extern "C" int pgw_byteswap(pgw_ctx *ctx, int elem_bytes, long long n, const void *src, void *dst) {
    NEED(ctx, elem_bytes == 4 || elem_bytes == 8, "elem_bytes must be 4 or 8");
    NEED(ctx, n >= 0 && (n == 0 || (src && dst)), "bad argument");
    if (n == 0) return PGW_OK;
    NEED(ctx, ((uintptr_t)src % elem_bytes) == 0 && ((uintptr_t)dst % elem_bytes) == 0, "pointers must be element-aligned");
    const bool al16 = ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0;
    const long long n16 = al16 ? n / (16 / elem_bytes) : 0;
    const long long work = al16 ? (n16 ? n16 : 1) : n;
    unsigned int nb = nblocks(work, BLOCK);
    if (nb > 256 * 16) nb = 256 * 16;
    {
        Prof pr(ctx, PGW_K_BYTESWAP);
        if (elem_bytes == 4)
            hipLaunchKernelGGL((k_byteswap<4>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n16, n, (const uint4 *)src, (uint4 *)dst);
        else
            hipLaunchKernelGGL((k_byteswap<8>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n16, n, (const uint4 *)src, (uint4 *)dst);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_narrow_f64_f32(pgw_ctx *ctx, long long n, const double *src, void *dst, int big_endian) {
    NEED(ctx, n >= 0 && (n == 0 || (src && dst)), "bad argument");
    if (n == 0) return PGW_OK;
    NEED(ctx, ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 4) == 0, "pointers must be element-aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool al = ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0;
    const long long n2 = al ? n / 2 : 0;
    unsigned int nb = nblocks(n2 ? n2 : n, BLOCK);
    if (nb > 256 * 16) nb = 256 * 16;
    {
        Prof pr(ctx, PGW_K_BYTESWAP);
        if (big_endian) hipLaunchKernelGGL((k_narrow_f64_f32<true>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n2, n, src, (unsigned int *)dst);
        else hipLaunchKernelGGL((k_narrow_f64_f32<false>), dim3(nb), dim3(BLOCK), 0, ctx->stream, n2, n, src, (unsigned int *)dst);
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_test_log(pgw_ctx *ctx, long long n, const double *in, double *out) {
    NEED(ctx, n >= 1 && in && out, "bad argument");
    hipLaunchKernelGGL(k_test_log, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, n, in, out, 0);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_test_log_table(pgw_ctx *ctx, long long n, const double *in, double *out) {
    NEED(ctx, n >= 1 && in && out, "bad argument");
    hipLaunchKernelGGL(k_test_log, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, n, in, out, 1);
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

// ------------------------------------------------------------------ regridding
extern "C" int pgw_regrid_bilinear(pgw_ctx *ctx, int dtype, long long nfield, int nlat_s, int nlon_s, int nlat_t,
                                   int nlon_t, const void *src, const int *lat_lo, const int *lat_hi,
                                   const double *lat_dx, const double *lat_Dx, const int *lat_oob, const int *lon_lo,
                                   const int *lon_hi, const double *lon_dx, const double *lon_Dx, const int *lon_oob,
                                   int south_row, int north_row, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, nfield >= 1 && nlat_s >= 2 && nlon_s >= 2 && nlat_t >= 1 && nlon_t >= 1, "bad shape");
    NEED(ctx, src && out && lat_lo && lat_hi && lat_dx && lat_Dx && lat_oob && lon_lo && lon_hi && lon_dx && lon_Dx && lon_oob,
         "null pointer");
    NEED(ctx, south_row < nlat_s && north_row < nlat_s, "bad pole row");
    for (int j = 0; j < nlat_t; ++j)
        NEED(ctx, lat_oob[j] || (lat_lo[j] >= -1 && lat_lo[j] <= nlat_s && lat_hi[j] >= -1 && lat_hi[j] <= nlat_s), "lat index out of range");
    for (int i = 0; i < nlon_t; ++i)
        NEED(ctx, lon_oob[i] || (lon_lo[i] >= 0 && lon_lo[i] < nlon_s && lon_hi[i] >= 0 && lon_hi[i] < nlon_s), "lon index out of range");
    // tables -> device workspace slot 3
    size_t ti = sizeof(int) * (3 * (size_t)nlat_t + 3 * (size_t)nlon_t);
    size_t td = sizeof(double) * (2 * (size_t)nlat_t + 2 * (size_t)nlon_t);
    size_t tp = sizeof(double) * 2 * (size_t)nfield;
    void *tab = nullptr;
    int rc = ws_get(ctx, 3, td + tp + ti + 64, &tab);
    if (rc) return rc;
    std::vector<char> h(td + ti);
    double *hd = (double *)h.data();
    memcpy(hd, lat_dx, 8 * nlat_t); memcpy(hd + nlat_t, lat_Dx, 8 * nlat_t);
    memcpy(hd + 2 * nlat_t, lon_dx, 8 * nlon_t); memcpy(hd + 2 * nlat_t + nlon_t, lon_Dx, 8 * nlon_t);
    int *hi = (int *)(h.data() + td);
    memcpy(hi, lat_lo, 4 * nlat_t); memcpy(hi + nlat_t, lat_hi, 4 * nlat_t); memcpy(hi + 2 * nlat_t, lat_oob, 4 * nlat_t);
    int *hl = hi + 3 * nlat_t;
    memcpy(hl, lon_lo, 4 * nlon_t); memcpy(hl + nlon_t, lon_hi, 4 * nlon_t); memcpy(hl + 2 * nlon_t, lon_oob, 4 * nlon_t);
    double *dd = (double *)tab;
    double *dpole = dd + 2 * nlat_t + 2 * nlon_t;
    int *di = (int *)(dpole + 2 * nfield);
    HIPCHK(ctx, hipMemcpyAsync(dd, hd, td, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(di, hi, ti, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    RegridTables tb;
    tb.lat_dx = dd; tb.lat_Dx = dd + nlat_t; tb.lon_dx = dd + 2 * nlat_t; tb.lon_Dx = dd + 2 * nlat_t + nlon_t;
    tb.lat_lo = di; tb.lat_hi = di + nlat_t; tb.lat_oob = di + 2 * nlat_t;
    tb.lon_lo = di + 3 * nlat_t; tb.lon_hi = tb.lon_lo + nlon_t; tb.lon_oob = tb.lon_lo + 2 * nlon_t;
    {
        Prof pr(ctx, PGW_K_REGRID);
        DISPATCH_T(dtype, {
            if (south_row >= 0 || north_row >= 0)
                hipLaunchKernelGGL((k_zonal_mean_rows<T>), dim3(nblocks(nfield * 2 * 64, BLOCK)), dim3(BLOCK), 0, ctx->stream, nfield,
                                   nlat_s, nlon_s, (const T *)src, south_row, north_row, dpole);
            // two target longitudes per thread (one 16-B / 8-B store per plane) when the row length and the output
            // alignment allow; z-slices: enough blocks to fill 256 CUs several times over even for small target grids
            const int W = (nlon_t % 2 == 0 && ((uintptr_t)out % (2 * sizeof(T))) == 0 && !ctx->opt[PGW_OPT_FORCE_VEC1]) ? 2 : 1;
            const unsigned int bx = nblocks(nlon_t, BLOCK * W);
            long long xy = (long long)bx * nlat_t;
            long long want = (8192 + xy - 1) / xy;
            unsigned int gz = (unsigned int)(want < 1 ? 1 : (want > nfield ? nfield : want));
            // z-slices in multiples of 8 where there are planes for it: one XCD per slice group (k_regrid)
            if (nfield >= 8) gz = (gz + 7u) / 8u * 8u;
            if (gz > nfield) gz = (unsigned int)nfield;
            NEED(ctx, xy * gz < (1ll << 31), "regrid: too many blocks");
            const unsigned int nb = (unsigned int)(xy * gz);
            // 4 planes per step (2: 6 % slower, 8: the same)
            if (W == 2)
                hipLaunchKernelGGL((k_regrid<T, 4, 2>), dim3(nb), dim3(BLOCK), 0, ctx->stream, nfield, nlat_s,
                                   nlon_s, nlat_t, nlon_t, bx, gz, (const T *)src, tb, dpole, (T *)out);
            else
                hipLaunchKernelGGL((k_regrid<T, 4, 1>), dim3(nb), dim3(BLOCK), 0, ctx->stream, nfield, nlat_s,
                                   nlon_s, nlat_t, nlon_t, bx, gz, (const T *)src, tb, dpole, (T *)out);
        });
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

// ------------------------------------------------------------------ surface riders
extern "C" int pgw_integrate_tos(pgw_ctx *ctx, int dtype, long long n, const void *tos, const void *ts,
                                 const void *land, const void *ice, void *out) {
    NEED(ctx, dtype == PGW_F32 || dtype == PGW_F64, "dtype must be PGW_F32 or PGW_F64");
    NEED(ctx, n >= 1 && tos && ts && land && ice && out, "bad argument");
    {
        Prof pr(ctx, PGW_K_SURFACE);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_integrate_tos<T>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, n,
                                             (const T *)tos, (const T *)ts, (const T *)land, (const T *)ice, (T *)out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}

extern "C" int pgw_surface_update(pgw_ctx *ctx, int dtype, int ntime, long long ncol, int nsoil,
                                  const double *soil_depth, const void *sic, const void *dsic, const void *dtos,
                                  const void *dts, const void *land, const void *ts_clim, const void *tskin,
                                  const void *tso, void *sic_out, void *dts_comb_out, void *tskin_out, void *tso_out) {
    CHECK_COMMON(ctx, dtype, ntime, ncol);
    NEED(ctx, nsoil >= 0 && nsoil <= MAX_SOIL, "nsoil must be in [0, 16]");
    NEED(ctx, sic && dsic && dtos && dts && land, "null pointer");
    NEED(ctx, !tskin_out || tskin, "tskin required for tskin_out");
    NEED(ctx, !tso_out || (tso && ts_clim && soil_depth && nsoil > 0), "tso, ts_clim, soil_depth required for tso_out");
    SoilTable st;
    memset(&st, 0, sizeof(st));
    st.n = nsoil;
    for (int s = 0; s < nsoil; ++s) st.w[s] = exp(-soil_depth[s] / 2.8);      // step_03:140
    long long n = (long long)ntime * ncol;
    {
        Prof pr(ctx, PGW_K_SURFACE);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_surface_update<T>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, ntime, ncol,
                                              st, (const T *)sic, (const T *)dsic, (const T *)dtos, (const T *)dts, (const T *)land,
                                              (const T *)ts_clim, (const T *)tskin, (const T *)tso, (T *)sic_out, (T *)dts_comb_out,
                                              (T *)tskin_out, (T *)tso_out));
    }
    HIPCHK(ctx, hipGetLastError());
    return PGW_OK;
}
