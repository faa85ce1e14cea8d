// Micro-benchmark (not part of the product): how the size of the contiguous chunk a block touches per level affects the
// achieved HBM rate of a column kernel's access pattern - arrays (level, column), a block owns `TPB * V` columns and
// marches over the levels, level stride = ncol elements.  Modes: write-only, read-only (sum), copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
struct Vec { double v[V]; };

template <int V, int MODE>   // MODE 0 write, 1 read, 2 copy, 3 copy with non-temporal stores, 4 copy with non-temporal loads and stores
__global__ void k(int nlev, long long ncol, const double *__restrict__ in, double *__restrict__ out, double *sink) {
    long long c = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (c >= ncol) return;
    double acc = 0.0;
    for (int l = 0; l < nlev; ++l) {
        if (MODE == 0) {
            Vec<V> r;
#pragma unroll
            for (int v = 0; v < V; ++v) r.v[v] = (double)l + (double)v;
            *reinterpret_cast<Vec<V> *>(out + (long long)l * ncol + c) = r;
        } else if (MODE == 1) {
            Vec<V> r = *reinterpret_cast<const Vec<V> *>(in + (long long)l * ncol + c);
#pragma unroll
            for (int v = 0; v < V; ++v) acc += r.v[v];
        } else if (MODE == 2) {
            Vec<V> r = *reinterpret_cast<const Vec<V> *>(in + (long long)l * ncol + c);
            *reinterpret_cast<Vec<V> *>(out + (long long)l * ncol + c) = r;
        } else {
            double r[V];
#pragma unroll
            for (int v = 0; v < V; ++v)
                r[v] = MODE == 4 ? __builtin_nontemporal_load(in + (long long)l * ncol + c + v) : in[(long long)l * ncol + c + v];
#pragma unroll
            for (int v = 0; v < V; ++v) __builtin_nontemporal_store(r[v], out + (long long)l * ncol + c + v);
        }
    }
    if (MODE == 1 && acc == 12345.678) *sink = acc;
}

template <int V, int MODE>
static void run(const char *name, int tpb, int nlev, long long ncol, const double *in, double *out, double *sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned int nb = (unsigned int)((ncol / V + tpb - 1) / tpb);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<V, MODE>), dim3(nb), dim3(tpb), 0, 0, nlev, ncol, in, out, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<V, MODE>), dim3(nb), dim3(tpb), 0, 0, nlev, ncol, in, out, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double gb = (double)nlev * ncol * 8 * (MODE >= 2 ? 2 : 1) / 1e9;
    printf("%-6s V=%d tpb=%4d chunk=%5d B  %.3f ms  %.0f GB/s\n", name, V, tpb, tpb * V * 8, ms, gb / ms * 1e3);
}

int main() {
    const int nlev = 137; const long long ncol = 1038240;
    double *in, *out, *sink;
    CK(hipMalloc(&in, sizeof(double) * nlev * ncol)); CK(hipMalloc(&out, sizeof(double) * nlev * ncol)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(in, 0, sizeof(double) * nlev * ncol));
    int tpbs[] = {64, 128, 256, 512, 1024};
    for (int t : tpbs) { run<1, 0>("write", t, nlev, ncol, in, out, sink); run<2, 0>("write", t, nlev, ncol, in, out, sink); }
    for (int t : tpbs) { run<1, 1>("read", t, nlev, ncol, in, out, sink); run<2, 1>("read", t, nlev, ncol, in, out, sink); }
    for (int t : tpbs) { run<1, 2>("copy", t, nlev, ncol, in, out, sink); run<2, 2>("copy", t, nlev, ncol, in, out, sink); }
    for (int t : tpbs) { run<1, 3>("cp-nts", t, nlev, ncol, in, out, sink); run<2, 3>("cp-nts", t, nlev, ncol, in, out, sink); }
    for (int t : tpbs) { run<1, 4>("cp-ntb", t, nlev, ncol, in, out, sink); run<2, 4>("cp-ntb", t, nlev, ncol, in, out, sink); }
    return 0;
}
