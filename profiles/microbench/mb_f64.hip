// Microbenchmarks that size the fp64 design on gfx950 (MI355X):
//   1. v_mfma_f64_16x16x4_f64 issue rate (independent / dependent accumulators, 1-2 waves per SIMD)
//   2. v_fma_f64 issue rate
//   3. both at once (does the matrix pipe add to the vector pipe for fp64?)
//   4. operand/accumulator lane maps of the f64 MFMA (checked against a host product)
// Build: hipcc -O3 --offload-arch=gfx950 mb_f64.hip -o mb_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// NACC accumulators, each updated CH times back to back before the next (CH = 1: round robin).
// Operands change every pass so that nothing is loop invariant.  WPS = waves per SIMD the launch is sized for.
template <int NACC, int CH, int WPS>
__global__ void __launch_bounds__(256, WPS) k_mfma_many(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a[CH], b[CH];
    for (int c = 0; c < CH; ++c) { a[c] = a0 + threadIdx.x * 1e-9 + c; b[c] = b0 + c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], b[c], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c] += 1e-9;
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The MFMA stream of k_stats' second wavefront (16 tiles x1 x^T + 5 tiles y x^T per k-step, operands in
// ten distinct registers), without memory (LOADS = 0) or with the ten operand loads of a k-step from an
// L1-resident buffer, issued in a burst behind the MFMAs (LOADS = 1) or one between every two (LOADS = 2).
template <int WPS, int LOADS>
__global__ void __launch_bounds__(256, WPS) k_mfma_stats(double* out, const double* buf, int iters) {
    d4 acc[21];
    for (int i = 0; i < 21; ++i) acc[i] = d4{0, 0, 0, 0};
    double op[2][10];
    const double* p = buf + (threadIdx.x & 63);
    for (int j = 0; j < 10; ++j) { op[0][j] = p[64 * j]; op[1][j] = p[64 * (j + 10)]; }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 21; ++i) {
                const int m = i < 16 ? 4 + i / 4 : 8 + (i - 16) / 4, k = i % 4;       // x1 in 4..7, y in 8..9, x in 0..3
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[h][m], op[h][k], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LOADS == 2 && (i & 1) && i / 2 < 10) { op[h ^ 1][i / 2] = p[64 * (i / 2) + 640 * ((it + h) & 1)]; __builtin_amdgcn_sched_barrier(0); }
            }
            if (LOADS == 1) {
#pragma unroll
                for (int j = 0; j < 10; ++j) op[h ^ 1][j] = p[64 * j + 640 * ((it + h) & 1)];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 21; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_stream(const d4* __restrict__ src, d4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double a = a0, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves with (wave index & 1) == 0 do MFMA, the others VALU FMA; 8 waves per block -> 2 per SIMD
__global__ void __launch_bounds__(512) k_mixed(double* out, int iters, double a0, double b0) {
    int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave >= 4) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a0, b0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one MFMA on known data: D = A(16x4) * B(4x16); records what each lane holds
__global__ void k_layout(const double* A, const double* B, double* Dout) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];      // A[row l&15][k l>>4]
    double b = B[(l >> 4) * 16 + (l & 15)];     // B[k l>>4][col l&15]
    d4 acc = d4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dout[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];   // row (l>>4)+4r, col l&15
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    double* out; CK(hipMalloc(&out, 1 << 24));
    const int iters = 4000;
    const int cus = p.multiProcessorCount;
    // --- layout check
    {
        std::vector<double> A(64), B(64), D(256), Dref(256, 0.0);
        for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11 + (i % 7); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Dref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
        k_layout<<<1, 64>>>(dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - Dref[i]));
        printf("layout check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): max err %.3e -> %s\n", err, err < 1e-9 ? "OK" : "MISMATCH");
    }
    for (int wps = 1; wps <= 2; ++wps) {
        int blocks = cus * wps;   // 256 threads = 4 waves = 1 per SIMD
        double fl = (double)blocks * 4 * iters * 16 * 2048.0;
        double ms;
        ms = time_ms([&] { k_mfma<4><<<blocks, 256>>>(out, iters, 1.0, 1e-3); }, 5);
        printf("mfma_f64_16x16x4 indep4  waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wps, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * wps));
        ms = time_ms([&] { k_mfma<2><<<blocks, 256>>>(out, iters, 1.0, 1e-3); }, 5);
        printf("mfma_f64_16x16x4 indep2  waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n", wps, ms, fl / ms * 1e-9);
        ms = time_ms([&] { k_mfma<1><<<blocks, 256>>>(out, iters, 1.0, 1e-3); }, 5);
        printf("mfma_f64_16x16x4 dep1    waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", wps, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * wps));
        double flv = (double)blocks * 256 * iters * 16 * 2.0;
        ms = time_ms([&] { k_fma<<<blocks, 256>>>(out, iters, 1.0000001, 1e-3); }, 5);
        printf("v_fma_f64 indep8         waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  (%.2f cyc/FMA/SIMD @2.4GHz)\n", wps, ms, flv / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * wps));
    }
    // sustained rates: ~1 s of back-to-back launches each (the short runs above see boost clocks)
    {
        const int it2 = 2000;
        auto sustained = [&](const char* name, auto launch, double flops_per_launch) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            double ms1 = time_ms(launch, 5);
            int reps = (int)(1000.0 / ms1) + 1;
            double ms = time_ms(launch, reps);
            printf("sustained %-44s %.3f ms/launch x %d  %.2f TFLOP/s\n", name, ms, reps, flops_per_launch / ms * 1e-9);
        };
        double* buf; CK(hipMalloc(&buf, 1 << 16));
        {   // random operands: an all-zero stream draws less power and sustains a higher clock than real data
            std::vector<double> hb((1 << 16) / 8);
            unsigned long long st = 88172645463325252ull;
            for (auto& v : hb) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = getenv("MB_ZERO") ? 0.0 : (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
            CK(hipMemcpy(buf, hb.data(), 1 << 16, hipMemcpyHostToDevice));
        }
        const double fs1 = (double)cus * 4 * it2 * 21 * 2048.0;
        sustained("stats stream, no loads, 1 wave/SIMD", [&] { k_mfma_stats<1, 0><<<cus, 256>>>(out, buf, it2); }, fs1);
        sustained("stats stream, no loads, 2 waves/SIMD", [&] { k_mfma_stats<2, 0><<<cus * 2, 256>>>(out, buf, it2); }, 2 * fs1);
        sustained("stats stream, load burst, 1 wave/SIMD", [&] { k_mfma_stats<1, 1><<<cus, 256>>>(out, buf, it2); }, fs1);
        sustained("stats stream, load burst, 2 waves/SIMD", [&] { k_mfma_stats<2, 1><<<cus * 2, 256>>>(out, buf, it2); }, 2 * fs1);
        sustained("stats stream, loads spread, 1 wave/SIMD", [&] { k_mfma_stats<1, 2><<<cus, 256>>>(out, buf, it2); }, fs1);
        sustained("stats stream, loads spread, 2 waves/SIMD", [&] { k_mfma_stats<2, 2><<<cus * 2, 256>>>(out, buf, it2); }, 2 * fs1);
        {   // does HBM traffic between the MFMA launches lower the clock the MFMA launches see?
            const size_t nb = (size_t)4 << 30;      // 4 GiB read + 4 GiB written per k_stream launch
            d4 *sa, *sb; CK(hipMalloc(&sa, nb)); CK(hipMalloc(&sb, nb)); CK(hipMemset(sa, 1, nb));
            hipEvent_t ev[4]; for (auto& x : ev) CK(hipEventCreate(&x));
            double tm = 0, ts = 0; const int reps = 200;
            for (int i = 0; i < reps + 20; ++i) {
                CK(hipEventRecord(ev[0]));
                k_mfma_stats<2, 2><<<cus * 2, 256>>>(out, buf, it2);
                CK(hipEventRecord(ev[1]));
                k_stream<<<cus * 8, 256>>>(sa, sb, nb / 32);
                CK(hipEventRecord(ev[2]));
                CK(hipEventSynchronize(ev[2]));
                float a, b; CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&b, ev[1], ev[2]));
                if (i >= 20) { tm += a; ts += b; }
            }
            printf("alternating with an 8 GiB stream kernel: stats stream (loads spread, 2 waves/SIMD) %.3f ms  %.2f TFLOP/s;  stream %.3f ms  %.2f TB/s\n",
                   tm / reps, 2 * fs1 / (tm / reps) * 1e-9, ts / reps, 2.0 * nb / (ts / reps) * 1e-9);
            CK(hipFree(sa)); CK(hipFree(sb));
        }
        sustained("21 acc round robin, 1 wave/SIMD", [&] { k_mfma_many<21, 1, 1><<<cus, 256>>>(out, it2, 1.0, 1e-3); }, (double)cus * 4 * it2 * 21 * 2048.0);
        sustained("21 acc chains of 4, 1 wave/SIMD", [&] { k_mfma_many<21, 4, 1><<<cus, 256>>>(out, it2 / 4, 1.0, 1e-3); }, (double)cus * 4 * (it2 / 4) * 21 * 4 * 2048.0);
        sustained("21 acc round robin, 2 waves/SIMD", [&] { k_mfma_many<21, 1, 2><<<cus * 2, 256>>>(out, it2, 1.0, 1e-3); }, (double)cus * 2 * 4 * it2 * 21 * 2048.0);
        sustained("21 acc chains of 4, 2 waves/SIMD", [&] { k_mfma_many<21, 4, 2><<<cus * 2, 256>>>(out, it2 / 4, 1.0, 1e-3); }, (double)cus * 2 * 4 * (it2 / 4) * 21 * 4 * 2048.0);
        sustained("4 acc chains of 16, 1 wave/SIMD", [&] { k_mfma_many<4, 16, 1><<<cus, 256>>>(out, it2 / 4, 1.0, 1e-3); }, (double)cus * 4 * (it2 / 4) * 4 * 16 * 2048.0);
        sustained("4 acc round robin, 1 wave/SIMD", [&] { k_mfma_many<4, 1, 1><<<cus, 256>>>(out, it2 * 4, 1.0, 1e-3); }, (double)cus * 4 * (it2 * 4) * 4 * 2048.0);
        sustained("1 acc (dependent), 1 wave/SIMD", [&] { k_mfma_many<1, 16, 1><<<cus, 256>>>(out, it2, 1.0, 1e-3); }, (double)cus * 4 * it2 * 16 * 2048.0);
        sustained("1 acc (dependent), 2 waves/SIMD", [&] { k_mfma_many<1, 16, 2><<<cus * 2, 256>>>(out, it2, 1.0, 1e-3); }, (double)cus * 2 * 4 * it2 * 16 * 2048.0);
        sustained("4 acc chains of 16, 2 waves/SIMD", [&] { k_mfma_many<4, 16, 2><<<cus * 2, 256>>>(out, it2 / 4, 1.0, 1e-3); }, (double)cus * 2 * 4 * (it2 / 4) * 4 * 16 * 2048.0);
    }
    {
        int blocks = cus;
        double fl_m = (double)blocks * 4 * iters * 16 * 2048.0, fl_v = (double)blocks * 256 * iters * 16 * 2.0;
        double ms = time_ms([&] { k_mixed<<<blocks, 512>>>(out, iters, 1.0000001, 1e-3); }, 5);
        printf("mixed (1 MFMA wave + 1 FMA wave per SIMD): %.3f ms  mfma %.2f + valu %.2f TFLOP/s\n", ms, fl_m / ms * 1e-9, fl_v / ms * 1e-9);
    }
    return 0;
}
