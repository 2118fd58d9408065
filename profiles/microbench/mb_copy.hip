// HBM copy ceiling of an MI355X in isolation: dst[i] = src[i] over 4 GiB + 4 GiB, 16 or 32 bytes per lane, several grid
// sizes, and the access shape of the sweeps (every wavefront streaming its own rows, 128-byte pieces of 16 rows per
// load instruction).  hipcc -O3 --offload-arch=gfx950 mb_copy.hip -o mb_copy && ./mb_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <class V>
__global__ void __launch_bounds__(256) k_copy(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// the sweeps' shape: wavefront w owns rows [w*T, (w+1)*T) of 64 doubles; per step it reads 16 rows that are T/16 apart
// (lane = 16 rows x 4 pieces of 32 bytes, four instructions per row set) and writes them elsewhere
__global__ void __launch_bounds__(64) k_rows(const double* __restrict__ src, double* __restrict__ dst, int T) {
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const size_t base = (size_t)blockIdx.x * T * 64;
    const int L = T / 16;
    for (int j = 0; j < L; ++j) {
        const size_t row = base + (size_t)(c * L + j) * 64;
        d4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const d4*>(src + row + (m * 4 + q) * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<d4*>(dst + row + (m * 4 + q) * 4) = v[m];
    }
}

// U loads in flight per lane, then U stores; NT: non-temporal loads and stores (streaming: no reuse expected)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy_u(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// every workgroup copies one contiguous slab (instead of a grid-stride walk)
template <int U>
__global__ void __launch_bounds__(256) k_copy_slab(const f4* __restrict__ src, f4* __restrict__ dst, size_t per_block) {
    const f4* s = src + (size_t)blockIdx.x * per_block;
    f4* d = dst + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per_block; i += U * 256) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
    }
}
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ src, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.0f;      // never true: keeps the loads
}
__global__ void __launch_bounds__(256) k_fill(f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t nb = (size_t)4 << 30;
    void *a, *b; CK(hipMalloc(&a, nb)); CK(hipMalloc(&b, nb));
    CK(hipMemset(a, 1, nb)); CK(hipMemset(b, 0, nb));
    for (int mult : {4, 8, 16, 32, 64}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_copy<d2>, dim3(cus * mult), dim3(256), 0, 0, (const d2*)a, (d2*)b, nb / 16); }, 10);
        printf("copy 16 B/lane, %4d workgroups: %.3f ms  %.2f TB/s (read + write)\n", cus * mult, ms, 2.0 * nb / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy<d4>, dim3(cus * mult), dim3(256), 0, 0, (const d4*)a, (d4*)b, nb / 32); }, 10);
        printf("copy 32 B/lane, %4d workgroups: %.3f ms  %.2f TB/s\n", cus * mult, ms, 2.0 * nb / (ms * 1e-3) / 1e12);
    }
    // the guide's figure (MI355X_MICROARCH.md: "6.29 TB/s measured, float4 copy"): float4 per lane, with several loads in flight
    // per lane, default and non-temporal, grid-stride and slab-per-workgroup; and read-only / write-only streams
    const size_t n4 = nb / 16;
    for (int mult : {2, 4, 8, 16}) {
        const int g = cus * mult;
        double ms = time_ms([&] { hipLaunchKernelGGL((k_copy_u<4, false>), dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); }, 10);
        printf("float4 copy, 4 in flight, %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 2.0 * nb / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL((k_copy_u<8, false>), dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); }, 10);
        printf("float4 copy, 8 in flight, %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 2.0 * nb / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL((k_copy_u<4, true>), dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); }, 10);
        printf("float4 copy, 4 in flight, non-temporal, %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 2.0 * nb / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL((k_copy_slab<4>), dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4 / g / 1024 * 1024); }, 10);
        printf("float4 copy, slab per workgroup, %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 2.0 * (double)(n4 / g / 1024 * 1024) * g * 16 / (ms * 1e-3) / 1e12);
    }
    for (int mult : {4, 16}) {
        const int g = cus * mult;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const f4*)a, (float*)b, n4); }, 10);
        printf("read only,  %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 1.0 * nb / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, (f4*)b, n4); }, 10);
        printf("write only, %5d workgroups: %.3f ms  %.2f TB/s\n", g, ms, 1.0 * nb / (ms * 1e-3) / 1e12);
    }
    {
        double ms = time_ms([&] { CK(hipMemcpyAsync(b, a, nb, hipMemcpyDeviceToDevice, 0)); }, 10);
        printf("hipMemcpyAsync device to device: %.3f ms  %.2f TB/s\n", ms, 2.0 * nb / (ms * 1e-3) / 1e12);
    }
    for (int waves : {1024, 2048, 4096}) {
        const int T = (int)(nb / 8 / 64 / waves) / 16 * 16;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_rows, dim3(waves), dim3(64), 0, 0, (const double*)a, (double*)b, T); }, 10);
        printf("sweep-shaped rows, %4d wavefronts x %d rows: %.3f ms  %.2f TB/s\n", waves, T, ms, 2.0 * waves * (double)T * 512 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
