// Micro-benchmark (not part of the product): is the rate the bare quad access pattern gets (quad_pattern.hip) a property of
// the BOX, of the moment, or of where hipMalloc put the arrays?  One process: R rounds; each round allocates the eight arrays
// anew (a dummy allocation of a different size is held across the round, so the arrays land elsewhere), and times the same
// kernel three times, T launches each.  Prints one line per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NS = 4;
struct Streams { const double *in[NS]; double *out[NS]; };

__global__ __launch_bounds__(128) void k_pattern(int nlev, int ncol, Streams s) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    double cur[2][NS], nxt[2][NS];
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
            for (int i = 0; i < NS; ++i) __builtin_nontemporal_store(cur[u][i] + 1.0, s.out[i] + (size_t)(l + u) * ncol + c);
        if (more) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) cur[u][i] = nxt[u][i];
        }
    }
}

__global__ __launch_bounds__(128) void k_read1(int nlev, int ncol, const double *in, double *sink) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    double acc = 0.0;
#pragma unroll 4
    for (int l = 0; l < nlev; ++l) acc += __builtin_nontemporal_load(in + (size_t)l * ncol + c);
    if (acc == 1234.5) *sink = acc;
}
__global__ __launch_bounds__(128) void k_write1(int nlev, int ncol, double *out) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= ncol) return;
    for (int l = 0; l < nlev; ++l) __builtin_nontemporal_store((double)l, out + (size_t)l * ncol + c);
}

int main(int argc, char **argv) {
    const int nlev = 136, ncol = 721 * 1440, rounds = argc > 1 ? atoi(argv[1]) : 10, T = 20, arena_mode_arg = argc > 2 ? atoi(argv[2]) : 0;
    const bool contiguous = arena_mode_arg >= 3;          // 3: eight physically contiguous allocations per round; 4: one contiguous arena, spacings
    const int arena_mode = arena_mode_arg == 3 ? 0 : (arena_mode_arg == 4 ? 2 : arena_mode_arg);
    const size_t n = (size_t)nlev * ncol;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((ncol + 127) / 128);
    if (arena_mode == 2) {
        // ONE arena for the life of the process; the eight arrays at base + k * spacing for a list of spacings, twice over
        const size_t MB = 1u << 20, field = (n * 8 + 2 * MB - 1) / (2 * MB) * (2 * MB);
        double *arena; const size_t bytes = 8 * (field + 600 * MB);
        if (contiguous) CK(hipExtMallocWithFlags((void **)&arena, bytes, hipDeviceMallocContiguous)); else CK(hipMalloc(&arena, bytes));
        CK(hipMemset(arena, 0, bytes));
        const size_t extra[] = {0, 2 * MB, 4 * MB, 6 * MB, 8 * MB, 16 * MB, 32 * MB, 34 * MB, 64 * MB, 128 * MB, 130 * MB, 256 * MB, 258 * MB, 512 * MB,
                                4096, 65536, 262144, 1 * MB, 3 * MB, 33 * MB + 4096, 128, 256, 512, 768, 1024, 1280, 2048, 2304, 4096 + 256, 8192 + 512, 16384 + 1024, 65536 + 256, 2 * MB + 256};
        for (int pass = 0; pass < 2; ++pass)
            for (size_t e : extra) {
                Streams s; const size_t sp = (field + e) / 8;
                for (int i = 0; i < NS; ++i) { s.in[i] = arena + (size_t)i * sp; s.out[i] = arena + (size_t)(NS + i) * sp; }
                hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(128), 0, 0, nlev, ncol, s);
                CK(hipEventRecord(e0));
                for (int i = 0; i < T; ++i) hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(128), 0, 0, nlev, ncol, s);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
                printf("round pass %d spacing field + %10zu B: %.3f ms %4.0f GB/s\n", pass, e, ms, (double)n * 16 * NS / 1e9 / ms * 1e3);
            }
        return 0;
    }
    for (int r = 0; r < rounds; ++r) {
        void *dummy = nullptr;
        const size_t dummy_bytes = (size_t)(r * 37 % 11) * (3u << 20) + (size_t)(r % 3) * 4096;
        if (dummy_bytes) CK(hipMalloc(&dummy, dummy_bytes));
        Streams s;
        double *arena = nullptr;
        const size_t stride = (n * 8 + (2u << 20) - 1) / (2u << 20) * (2u << 20) / 8;      // arrays 2 MiB-aligned inside the arena
        if (arena_mode) {
            CK(hipMalloc(&arena, stride * 8 * 2 * NS)); CK(hipMemset(arena, 0, stride * 8 * 2 * NS));
            for (int i = 0; i < NS; ++i) { s.in[i] = arena + (size_t)i * stride; s.out[i] = arena + (size_t)(NS + i) * stride; }
        } else {
            for (int i = 0; i < NS; ++i) {
                double *p;
                if (contiguous) { CK(hipExtMallocWithFlags((void **)&p, n * 8, hipDeviceMallocContiguous)); CK(hipExtMallocWithFlags((void **)&s.out[i], n * 8, hipDeviceMallocContiguous)); }
                else { CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&s.out[i], n * 8)); }
                CK(hipMemset(p, 0, n * 8)); s.in[i] = p;
            }
        }
        printf("round %2d  dummy %9zu B  in0 %p out0 %p :", r, dummy_bytes, (void *)s.in[0], (void *)s.out[0]);
        for (int t = 0; t < 3; ++t) {
            hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(128), 0, 0, nlev, ncol, s);
            CK(hipEventRecord(e0));
            for (int i = 0; i < T; ++i) hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(128), 0, 0, nlev, ncol, s);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
            printf("  %.3f ms %4.0f GB/s", ms, (double)n * 16 * NS / 1e9 / ms * 1e3);
        }
        {   // one stream alone: read in[0], write out[0]
            float ms;
            hipLaunchKernelGGL(k_read1, dim3(nb), dim3(128), 0, 0, nlev, ncol, s.in[0], s.out[0]);
            CK(hipEventRecord(e0));
            for (int i = 0; i < T; ++i) hipLaunchKernelGGL(k_read1, dim3(nb), dim3(128), 0, 0, nlev, ncol, s.in[0], s.out[0]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
            printf("  | read1 %4.0f", (double)n * 8 / 1e9 / ms * 1e3);
            hipLaunchKernelGGL(k_write1, dim3(nb), dim3(128), 0, 0, nlev, ncol, s.out[0]);
            CK(hipEventRecord(e0));
            for (int i = 0; i < T; ++i) hipLaunchKernelGGL(k_write1, dim3(nb), dim3(128), 0, 0, nlev, ncol, s.out[0]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= T;
            printf(" write1 %4.0f GB/s", (double)n * 8 / 1e9 / ms * 1e3);
        }
        printf("\n");
        if (arena_mode) CK(hipFree(arena));
        else for (int i = 0; i < NS; ++i) { CK(hipFree((void *)s.in[i])); CK(hipFree(s.out[i])); }
        if (dummy) CK(hipFree(dummy));
    }
    return 0;
}
