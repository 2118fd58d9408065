// What does the fp64 matrix pipe of an MI355X sustain under its power cap, and does it depend on the DATA?
//   hipcc -O3 --offload-arch=gfx950 mb_power.hip -o mb_power && ./mb_power
// A pure v_mfma_f64_16x16x4_f64 loop (4 accumulators, chains of 16, 1 or 2 wavefronts per SIMD, all 256 CUs) is run
// for ~3 s per case while `rocm-smi --showclocks --showpower` is polled: once with near-constant operands (what
// profiles/microbench/mb_f64.hip uses), once with operands whose mantissas and signs are are random per lane and differ from one MFMA to the next
// (16 operand pairs held in registers).  Output: TFLOP/s, shader clock, socket power.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double rnd_operand(unsigned& s0, unsigned& s1) {
    // xorshift on two 32-bit halves; result: random sign, exponent of 1.0, random 52-bit mantissa
    s0 ^= s0 << 13; s0 ^= s0 >> 17; s0 ^= s0 << 5;
    s1 ^= s1 << 15; s1 ^= s1 >> 13; s1 ^= s1 << 7;
    const unsigned hi = (s0 & 0x800FFFFFu) | 0x3FF00000u;
    return __hiloint2double((int)hi, (int)s1);
}

template <bool RANDOM, int WPS>
__global__ void __launch_bounds__(256, WPS) k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    unsigned s0 = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x, s1 = 0x85EBCA6Bu * (threadIdx.x + 7) + 31 * blockIdx.x;
    // 16 operand pairs per lane, fixed for the launch: consecutive MFMAs see unrelated bit patterns (RANDOM) or
    // nearly the same one (not RANDOM); no VALU work inside the loop either way
    double a[16], b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        a[u] = RANDOM ? rnd_operand(s0, s1) : a0 + threadIdx.x * 1e-9 + u * 1e-9;
        b[u] = RANDOM ? rnd_operand(s1, s0) : b0 + u * 1e-9;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static std::string smi() {
    std::string r;
    FILE* f = popen("rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card0", "r");
    if (!f) return "popen failed";
    char buf[512];
    while (fgets(buf, sizeof buf, f)) r += buf;
    pclose(f);
    // card0,(fclk),lvl,(mclk),lvl,(sclk),lvl,(socclk),lvl,power
    std::vector<std::string> fld; size_t p = 0;
    while (true) { size_t q = r.find(',', p); fld.push_back(r.substr(p, q == std::string::npos ? q : q - p)); if (q == std::string::npos) break; p = q + 1; }
    if (fld.size() < 10) return r;
    std::string pw = fld.back(); while (!pw.empty() && (pw.back() == '\n' || pw.back() == '\r')) pw.pop_back();
    return "sclk " + fld[5] + " power " + pw + " W";
}

template <bool RANDOM, int WPS>
static void run(const char* name, double* out, int blocks) {
    const int iters = 400;                      // 400 * 64 MFMAs per wave per launch
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // calibrate one launch
    hipLaunchKernelGGL((k_mfma<RANDOM, WPS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mfma<RANDOM, WPS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms1 = 0; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int launches = (int)(3000.0 / ms1) + 1;
    CK(hipEventRecord(e0));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k_mfma<RANDOM, WPS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    CK(hipEventRecord(e1));
    std::string samples;
    for (int s = 0; s < 6; ++s) {
        std::this_thread::sleep_for(std::chrono::milliseconds(350));
        samples += "\n      t=" + std::to_string(0.35 * (s + 1)).substr(0, 4) + " s  " + smi();
    }
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)launches * blocks * 4.0 * iters * 64.0 * 2048.0;
    printf("%-46s %7.2f TFLOP/s over %.2f s%s\n", name, flops / (ms * 1e-3) / 1e12, ms * 1e-3, samples.c_str());
    fflush(stdout);
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device CUs %d clock %d kHz; idle: %s\n", p.multiProcessorCount, p.clockRate, smi().c_str());
    double* out; CK(hipMalloc(&out, (size_t)1 << 24));
    const int cus = p.multiProcessorCount;
    run<false, 1>("near-constant operands, 1 wave/SIMD", out, cus);
    run<true, 1>("random mantissas/signs,  1 wave/SIMD", out, cus);
    run<false, 2>("near-constant operands, 2 waves/SIMD", out, 2 * cus);
    run<true, 2>("random mantissas/signs,  2 waves/SIMD", out, 2 * cus);
    CK(hipFree(out));
    return 0;
}
