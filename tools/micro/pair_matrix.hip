// Micro-benchmark (not part of the product): which PAIRS of arrays copy fast?  M arrays of one ERA5 level field each
// (137 x 721 x 1440 doubles, separate hipMallocs, held for the life of the process); a column-pattern copy i -> j for every
// ordered pair, and the read-only / write-only rate of each array alone on the diagonal lines.  See alloc_lottery.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(128) void k_copy(int nlev, int ncol, const double *__restrict__ in, double *__restrict__ out) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    double a = __builtin_nontemporal_load(in + c), b = __builtin_nontemporal_load(in + (size_t)ncol + c);
    for (int l = 0; l + 1 < nlev; l += 2) {
        double a2 = 0, b2 = 0;
        if (l + 3 < nlev) { a2 = __builtin_nontemporal_load(in + (size_t)(l + 2) * ncol + c); b2 = __builtin_nontemporal_load(in + (size_t)(l + 3) * ncol + c); }
        __builtin_nontemporal_store(a, out + (size_t)l * ncol + c);
        __builtin_nontemporal_store(b, out + (size_t)(l + 1) * ncol + c);
        a = a2; b = b2;
    }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12, nlev = 136, ncol = 721 * 1440, T = 4;
    const size_t n = (size_t)nlev * ncol;
    std::vector<double *> a(M);
    for (int i = 0; i < M; ++i) { CK(hipMalloc(&a[i], n * 8)); CK(hipMemset(a[i], 0, n * 8)); printf("a[%2d] = %p\n", i, (void *)a[i]); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((ncol + 127) / 128);
    printf("copy rate GB/s, row = source, column = destination\n      ");
    for (int j = 0; j < M; ++j) printf("%5d", j);
    printf("\n");
    for (int i = 0; i < M; ++i) {
        printf("%4d :", i);
        for (int j = 0; j < M; ++j) {
            if (i == j) { printf("    -"); continue; }
            hipLaunchKernelGGL(k_copy, dim3(nb), dim3(128), 0, 0, nlev, ncol, a[i], a[j]);
            CK(hipEventRecord(e0));
            for (int t = 0; t < T; ++t) hipLaunchKernelGGL(k_copy, dim3(nb), dim3(128), 0, 0, nlev, ncol, a[i], a[j]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
            printf("%5.0f", (double)n * 16 / 1e9 / ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
