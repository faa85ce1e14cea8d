// Micro-benchmark (not part of the product): a map of the card's memory as the copy probe sees it.  M chunks of 1 GiB
// (separate hipMallocs, all held), column-pattern copy chunk 0 -> chunk k, chunk k -> chunk 0 and chunk k-1 -> chunk k for
// every k: chunks of the same "stretch" as chunk 0 copy slowly, others fast (see pair_matrix.hip, alloc_lottery.hip).
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
    const int M = argc > 1 ? atoi(argv[1]) : 200, nlev = 128, ncol = 1 << 20, T = 3;     // 128 x 1 Mi doubles = 1 GiB
    const size_t n = (size_t)nlev * ncol;
    std::vector<double *> a(M);
    for (int i = 0; i < M; ++i) { CK(hipMalloc(&a[i], n * 8)); }
    for (int i = 0; i < M; ++i) CK(hipMemsetAsync(a[i], 0, n * 8));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((ncol + 127) / 128);
    auto rate = [&](int i, int j) {
        hipLaunchKernelGGL(k_copy, dim3(nb), dim3(128), 0, 0, nlev, ncol, a[i], a[j]);
        CK(hipEventRecord(e0));
        for (int t = 0; t < T; ++t) hipLaunchKernelGGL(k_copy, dim3(nb), dim3(128), 0, 0, nlev, ncol, a[i], a[j]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
        return (double)n * 16 / 1e9 / ms * 1e3;
    };
    printf("  k  virtual address     0->k   k->0  (k-1)->k   GB/s\n");
    for (int k = 1; k < M; ++k) printf("%3d  %p  %5.0f  %5.0f  %5.0f\n", k, (void *)a[k], rate(0, k), rate(k, 0), rate(k - 1, k));
    return 0;
}
