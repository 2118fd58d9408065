// Unit test + timing of the blocked MFMA Gauss-Jordan (profiles/experiments/bgj.h) against the scalar register
// Gauss-Jordan (pyvb_amd/csrc/gj.h) and a host inverse.   hipcc --offload-arch=gfx950 -O3 -I pyvb_amd/csrc -I profiles/experiments -o mb_bgj mb_bgj.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gj.h"
#include "bgj.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NP>
__global__ __launch_bounds__(256) void k_old(const double* in, double* out, double* piv, int D) {
    __shared__ double gjbuf[NP * 2 * GJ_BUF + NP * 64];
    const int tid = threadIdx.x, a = tid >> 4, b = tid & 15;
    const double* src = in + (size_t)blockIdx.x * NP * 4096;
    double v[NP][16];
    for (int c = 0; c < NP; ++c)
        for (int ra = 0; ra < 4; ++ra)
            for (int cb = 0; cb < 4; ++cb) v[c][4 * ra + cb] = src[c * 4096 + (4 * a + ra) * 64 + 4 * b + cb];
    gj_inverse<NP>(v, D, tid, gjbuf, gjbuf + NP * 2 * GJ_BUF);
    double* dst = out + (size_t)blockIdx.x * NP * 4096;
    for (int c = 0; c < NP; ++c)
        for (int ra = 0; ra < 4; ++ra)
            for (int cb = 0; cb < 4; ++cb) dst[c * 4096 + (4 * a + ra) * 64 + 4 * b + cb] = v[c][4 * ra + cb];
    if (tid < 64) for (int c = 0; c < NP; ++c) piv[(size_t)blockIdx.x * NP * 64 + c * 64 + tid] = gjbuf[NP * 2 * GJ_BUF + c * 64 + tid];
}

template <int NP>
__global__ __launch_bounds__(256) void k_new(const double* in, double* out, double* piv, int D, int* status) {
    __shared__ double lds[NP * BGJ_LDS_PER + NP * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r = lane & 15;
    const double* src = in + (size_t)blockIdx.x * NP * 4096;
    d4 t[NP][4];
    for (int c = 0; c < NP; ++c)
        for (int J = 0; J < 4; ++J)
            for (int e = 0; e < 4; ++e) t[c][J][e] = src[c * 4096 + (16 * wave + 4 * e + q) * 64 + 16 * J + r];
    bgj_inverse<NP>(t, (D + 15) / 16, tid, lds, lds + NP * BGJ_LDS_PER, status);
    double* dst = out + (size_t)blockIdx.x * NP * 4096;
    for (int c = 0; c < NP; ++c)
        for (int J = 0; J < 4; ++J)
            for (int e = 0; e < 4; ++e) dst[c * 4096 + (16 * wave + 4 * e + q) * 64 + 16 * J + r] = t[c][J][e];
    if (tid < 64) for (int c = 0; c < NP; ++c) piv[(size_t)blockIdx.x * NP * 64 + c * 64 + tid] = lds[NP * BGJ_LDS_PER + c * 64 + tid];
}

static void host_inverse(const double* A, int D, double* inv, double* logdet) {   // plain Gauss-Jordan on the D x D corner
    std::vector<double> M(A, A + 4096);
    *logdet = 0;
    for (int p = 0; p < D; ++p) {
        double d = M[p * 64 + p]; *logdet += std::log(d);
        double dinv = 1.0 / d;
        for (int i = 0; i < D; ++i) if (i != p) {
            double f = M[i * 64 + p] * dinv;
            for (int j = 0; j < D; ++j) if (j != p) M[i * 64 + j] -= f * M[p * 64 + j];
        }
        for (int j = 0; j < D; ++j) if (j != p) M[p * 64 + j] *= dinv;
        for (int i = 0; i < D; ++i) if (i != p) M[i * 64 + p] *= -dinv;
        M[p * 64 + p] = dinv;
    }
    for (int i = 0; i < 4096; ++i) inv[i] = M[i];
}

template <int NP>
static int run(int D, int nblk) {
    const size_t n = (size_t)nblk * NP * 4096;
    std::vector<double> h(n, 0.0);
    srand(7 + D);
    for (int m = 0; m < nblk * NP; ++m) {                       // B B^T / D + I on the D x D corner, identity beyond
        double* A = &h[(size_t)m * 4096];
        std::vector<double> B(D * D);
        for (auto& x : B) x = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < 64; ++i) A[i * 64 + i] = 1.0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                double s = 0; for (int k = 0; k < D; ++k) s += B[i * D + k] * B[j * D + k];
                A[i * 64 + j] = s / D * 4 + (i == j ? 0.3 : 0.0);
            }
    }
    double *din, *dold, *dnew, *pold, *pnew; int* st;
    CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dold, n * 8)); CK(hipMalloc(&dnew, n * 8));
    CK(hipMalloc(&pold, (size_t)nblk * NP * 64 * 8)); CK(hipMalloc(&pnew, (size_t)nblk * NP * 64 * 8)); CK(hipMalloc(&st, 4));
    CK(hipMemset(st, 0, 4));
    CK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_old = 0, ms_new = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); k_old<NP><<<nblk, 256>>>(din, dold, pold, D); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_old, e0, e1));
        CK(hipEventRecord(e0)); k_new<NP><<<nblk, 256>>>(din, dnew, pnew, D, st); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_new, e0, e1));
    }
    CK(hipDeviceSynchronize());
    std::vector<double> o(n), w(n), po((size_t)nblk * NP * 64), pw((size_t)nblk * NP * 64);
    int hst;
    CK(hipMemcpy(o.data(), dold, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(w.data(), dnew, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(po.data(), pold, po.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(pw.data(), pnew, pw.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hst, st, 4, hipMemcpyDeviceToHost));
    double worst_old = 0, worst_new = 0, worst_ld_old = 0, worst_ld_new = 0;
    std::vector<double> inv(4096);
    for (int m = 0; m < (nblk < 8 ? nblk : 8) * NP; ++m) {
        double ld; host_inverse(&h[(size_t)m * 4096], D, inv.data(), &ld);
        double scale = 0; for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) scale = fmax(scale, fabs(inv[i * 64 + j]));
        double lo = 0, ln = 0;
        for (int k = 0; k < D; ++k) { lo += std::log(po[(size_t)m * 64 + k]); ln += std::log(pw[(size_t)m * 64 + k]); }
        for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) {
            worst_old = fmax(worst_old, fabs(o[(size_t)m * 4096 + i * 64 + j] - inv[i * 64 + j]) / scale);
            worst_new = fmax(worst_new, fabs(w[(size_t)m * 4096 + i * 64 + j] - inv[i * 64 + j]) / scale);
        }
        worst_ld_old = fmax(worst_ld_old, fabs(lo - ld)); worst_ld_new = fmax(worst_ld_new, fabs(ln - ld));
    }
    printf("NP=%d D=%2d blocks=%d  scalar %.3f ms  blocked %.3f ms  inverse err scalar %.2e blocked %.2e  logdet err %.2e %.2e  status %d\n",
           NP, D, nblk, ms_old, ms_new, worst_old, worst_new, worst_ld_old, worst_ld_new, hst);
    CK(hipFree(din)); CK(hipFree(dold)); CK(hipFree(dnew)); CK(hipFree(pold)); CK(hipFree(pnew)); CK(hipFree(st));
    return !(worst_new < 1e-11 && worst_ld_new < 1e-10 && hst == 0);
}

int main() {
    int bad = 0;
    const int Ds[] = {64, 50, 48, 33, 32, 17, 16, 10, 3, 1};
    for (int D : Ds) bad += run<3>(D, 1024);
    bad += run<4>(64, 1024); bad += run<4>(20, 1024); bad += run<2>(64, 1024); bad += run<1>(64, 2048);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
