// What a plain streaming read (and a copy) achieves on this MI355X, to set beside the kernels' HBM fractions.
//   hipcc -O3 --offload-arch=gfx950 profiles/microbench/hbm_read.hip -o build/hbm_read && build/hbm_read
// Every workgroup reads contiguous 16-byte-per-lane pieces, grid-strided, of a buffer far larger than the 256 MB last-level cache;
// variants: bytes per lane per load (16), loads in flight per lane (UNROLL), workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const d2* __restrict__ p, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0] + v[u][1];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// each workgroup streams its own contiguous chunk (the layout of the row-chunked kernels)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_read_chunk(const d2* __restrict__ p, size_t n, double* out) {
    const size_t per = n / gridDim.x;
    const d2* q = p + (size_t)blockIdx.x * per;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per; i += UNROLL * 256) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(q + i + u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0] + v[u][1];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_copy(const d2* __restrict__ p, d2* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[i + u * stride] = v[u];
    }
}

template <typename F> static double timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    d2 *p, *o; double* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(o, 0, bytes));
    for (int wgs : {256, 512, 1024, 2048, 4096, 16384}) {
        double t4 = timeit([&] { hipLaunchKernelGGL(k_read<4>, dim3(wgs), dim3(256), 0, 0, p, n, out); }, 5);
        double t8 = timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(wgs), dim3(256), 0, 0, p, n, out); }, 5);
        double tc = timeit([&] { hipLaunchKernelGGL(k_read_chunk<8>, dim3(wgs), dim3(256), 0, 0, p, n, out); }, 5);
        double tk = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, p, o, n); }, 5);
        printf("%6d workgroups: read, 4 loads in flight %6.0f GB/s | 8 in flight %6.0f GB/s | own chunk per workgroup, 8 in flight %6.0f GB/s | copy %6.0f GB/s read + as much written\n",
               wgs, bytes / t4 * 1e-6, bytes / t8 * 1e-6, bytes / tc * 1e-6, bytes / tk * 1e-6);
    }
    return 0;
}
