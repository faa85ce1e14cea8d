// Micro-benchmark (not part of the product): the HBM rate the box at hand gives the ACCESS PATTERN of k_delta_quad with no
// arithmetic - one thread per column, blocks of 128 columns, NS read streams and NS write streams of (level, column) arrays,
// two levels per step with the next step's rows requested one step ahead, streaming (nt) loads and stores.  bench.py runs it
// beside the file path (extras.pattern_ceiling) so that the quad kernel's fraction of the 8 TB/s peak can be read against what
// the same box delivers to the bare pattern.  Prints one JSON line.  Build: hipcc --offload-arch=gfx950 -O3 -o quad_pattern quad_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NS = 4;                    // ta, hus, ua, va in; T_pgw, e, U_pgw, V_pgw out
template <typename TI> struct Streams { const TI *in[NS]; double *out[NS]; };

template <typename TI>
__global__ __launch_bounds__(128) void k_pattern(int nlev, int ncol, Streams<TI> s) {
    extern __shared__ double s_occupancy[];          // dynamic LDS only to set the blocks per CU (argv[5]); never touched
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    TI cur[2][NS], nxt[2][NS];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < NS; ++i) cur[u][i] = __builtin_nontemporal_load(s.in[i] + (size_t)u * ncol + c);
    for (int l = 0; l + 1 < nlev; l += 2) {
        const bool more = l + 3 < nlev;
        if (more) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) nxt[u][i] = __builtin_nontemporal_load(s.in[i] + (size_t)(l + 2 + u) * ncol + c);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < NS; ++i) __builtin_nontemporal_store((double)cur[u][i] + 1.0, s.out[i] + (size_t)(l + u) * ncol + c);
        if (more) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) cur[u][i] = nxt[u][i];
        }
    }
    if (nlev & 1) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
            __builtin_nontemporal_store((double)__builtin_nontemporal_load(s.in[i] + (size_t)(nlev - 1) * ncol + c) + 1.0,
                                        s.out[i] + (size_t)(nlev - 1) * ncol + c);
    }
}

template <typename TI>
static void run(const char *name, int nlev, int ncol, int reps, bool last, int lds) {
    Streams<TI> s;
    const size_t n = (size_t)nlev * ncol;
    for (int i = 0; i < NS; ++i) {
        TI *p; CK(hipMalloc(&p, n * sizeof(TI))); CK(hipMemset(p, 0, n * sizeof(TI))); s.in[i] = p;
        CK(hipMalloc(&s.out[i], n * sizeof(double)));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((ncol + 127) / 128);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_pattern<TI>, dim3(nb), dim3(128), lds, 0, nlev, ncol, s);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_pattern<TI>, dim3(nb), dim3(128), lds, 0, nlev, ncol, s);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gb = (double)n * NS * (sizeof(TI) + 8) / 1e9;
    printf("\"%s\": {\"ms\": %.4f, \"GB\": %.3f, \"GBps\": %.0f}%s", name, ms, gb, gb / ms * 1e3, last ? "" : ", ");
    for (int i = 0; i < NS; ++i) { CK(hipFree((void *)s.in[i])); CK(hipFree(s.out[i])); }
}

int main(int argc, char **argv) {
    const int nlev = argc > 1 ? atoi(argv[1]) : 137, nlat = argc > 2 ? atoi(argv[2]) : 721, nlon = argc > 3 ? atoi(argv[3]) : 1440;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int bpc = argc > 5 ? atoi(argv[5]) : 0;      // blocks of 128 threads per CU (0 = as many as fit: 16); k_delta_quad runs 5 - 8
    const int lds = bpc > 0 ? (160 * 1024 / bpc) & ~255 : 0;
    if (lds > 64 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_pattern<double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_pattern<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    printf("{\"pattern\": \"%d read + %d write streams of (level, column), block of 128 columns, 2 levels per step, nt\", \"nlev\": %d, \"ncol\": %d, \"blocks_per_cu\": %d, ",
           NS, NS, nlev, nlat * nlon, bpc);
    run<double>("f64_in_f64_out", nlev, nlat * nlon, reps, false, lds);
    run<float>("f32_in_f64_out", nlev, nlat * nlon, reps, true, lds);
    printf("}\n");
    return 0;
}
