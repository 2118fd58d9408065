// k_stats: the sums over t that every parameter update reads
//   Sxx  = sum_t mu_t mu_t^T          Sx1x = sum_t mu_{t+1} mu_t^T        Syx = sum_t y_t mu_t^T
// (the covariance parts of <x x^T> = qmu qmu^T + qcov, gaussian.py:162-168, are added per class
// by the consumers).  In the reference these sums are re-gathered message by message for every
// column of A and C (hstack.pass_up_m1_m2 nodes_todo.py:43-62, Multiplication.pass_up_m1_m2
// node.py:193-202) and for Q and R (nodes_todo.py:187-190).
//
// Mapping: T is the K dimension of v_mfma_f64_16x16x4_f64; a k-step is 4 consecutive time steps.
// Lane (r = lane%16, q = lane/16) loads element 16m + r of row t + q, which is at once the A
// operand for row tile m and the B operand for column tile m, so Sxx needs no second load.
// One wavefront per (replicate, time chunk); partial sums per chunk are reduced by the consumers.
#include "common.h"

struct StatsArgs {
    const double* X; const double* Y; double* part;
    int N, T, D, K, nchunk, chunk_len;
    Layout L;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int DT, int KT>
__global__ void __launch_bounds__(64) k_stats(StatsArgs a) {
    constexpr int DP = 16 * DT;
    const int ch = blockIdx.x, n = blockIdx.y, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int T = a.T, D = a.D, K = a.K;
    const double* X = a.X + (size_t)n * T * D;
    const double* Y = a.Y + (size_t)n * T * K;
    const int t0 = ch * a.chunk_len;
    const int t1 = (t0 + a.chunk_len < T) ? t0 + a.chunk_len : T;

    d4 sxx[DT][DT], sx1[DT][DT], syx[KT][DT];
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int k = 0; k < DT; ++k) { sxx[m][k] = d4{0, 0, 0, 0}; sx1[m][k] = d4{0, 0, 0, 0}; }
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int k = 0; k < DT; ++k) syx[m][k] = d4{0, 0, 0, 0};

    double xa[DT], xb[DT], ya[KT];
    auto load = [&](int tb, double* pxa, double* pxb, double* pya) {
        const int t = tb + q;
        const bool v = t < t1;
        const bool v1 = v && (t + 1 < T);
#pragma unroll
        for (int m = 0; m < DT; ++m) {
            const int dim = 16 * m + r;
            pxa[m] = (v && dim < D) ? X[(size_t)t * D + dim] : 0.0;
            pxb[m] = (v1 && dim < D) ? X[(size_t)(t + 1) * D + dim] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < KT; ++m) {
            const int dim = 16 * m + r;
            pya[m] = (v && dim < K) ? Y[(size_t)t * K + dim] : 0.0;
        }
    };
    load(t0, xa, xb, ya);
    for (int tb = t0; tb < t1; tb += 4) {
        double xa_n[DT], xb_n[DT], ya_n[KT];
        load(tb + 4, xa_n, xb_n, ya_n);     // rows >= t1 read as zero
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                if (k >= m) sxx[m][k] = MFMA(xa[m], xa[k], sxx[m][k]);     // symmetric: upper tiles only
                sx1[m][k] = MFMA(xb[m], xa[k], sx1[m][k]);
            }
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) syx[m][k] = MFMA(ya[m], xa[k], syx[m][k]);
#pragma unroll
        for (int m = 0; m < DT; ++m) { xa[m] = xa_n[m]; xb[m] = xb_n[m]; }
#pragma unroll
        for (int m = 0; m < KT; ++m) ya[m] = ya_n[m];
    }

    // accumulator element: row = 16m + 4*reg + q, col = 16k + r
    double* P = a.part + ((size_t)n * a.nchunk + ch) * a.L.stats_total;
#pragma unroll
    for (int m = 0; m < DT; ++m)
#pragma unroll
        for (int k = 0; k < DT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * m + 4 * e + q, col = 16 * k + r;
                if (k >= m) {
                    P[a.L.oSxx + (size_t)row * DP + col] = sxx[m][k][e];
                    if (k > m) P[a.L.oSxx + (size_t)col * DP + row] = sxx[m][k][e];
                }
                P[a.L.oSx1x + (size_t)row * DP + col] = sx1[m][k][e];
            }
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int k = 0; k < DT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                P[a.L.oSyx + (size_t)(16 * m + 4 * e + q) * DP + 16 * k + r] = syx[m][k][e];
}

// Syy[n][k] = sum_t y_t[k]^2: the observations never change, so this runs once per set_observations.
struct SyyArgs { const double* Y; double* Syy; int N, T, K; };

__global__ void __launch_bounds__(256) k_syy(SyyArgs a) {
    __shared__ double red[256];
    const int n = blockIdx.x, tid = threadIdx.x, K = a.K;
    const double* Y = a.Y + (size_t)n * a.T * K;
    // thread owns component tid % K of rows tid / K, tid / K + 256 / K ...
    const int per = 256 / K, k = tid % K, r0 = tid / K;
    double s = 0.0;
    if (r0 < per)
        for (int t = r0; t < a.T; t += per) { double v = Y[(size_t)t * K + k]; s += v * v; }
    red[tid] = (r0 < per) ? s : 0.0;
    __syncthreads();
    if (tid < K) {
        double tot = 0.0;
        for (int i = 0; i < per; ++i) tot += red[i * K + tid];
        a.Syy[(size_t)n * K + tid] = tot;
    }
}

template <int DT, int KT>
static void launch_stats_t(pyvb_lds* h, const StatsArgs& a) {
    hipLaunchKernelGGL((k_stats<DT, KT>), dim3(h->nchunk, h->N), dim3(64), 0, h->stream, a);
}

int launch_stats(pyvb_lds* h) {
    StatsArgs a;
    a.X = h->X[h->cur]; a.Y = h->Y; a.part = h->stats;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.nchunk = h->nchunk; a.chunk_len = h->chunk_len; a.L = h->L;
    {
        TimedLaunch tl(h, PYVB_K_STATS);
        switch (h->L.DT * 10 + h->L.KT) {
            case 11: launch_stats_t<1, 1>(h, a); break;
            case 12: launch_stats_t<1, 2>(h, a); break;
            case 14: launch_stats_t<1, 4>(h, a); break;
            case 21: launch_stats_t<2, 1>(h, a); break;
            case 22: launch_stats_t<2, 2>(h, a); break;
            case 24: launch_stats_t<2, 4>(h, a); break;
            case 41: launch_stats_t<4, 1>(h, a); break;
            case 42: launch_stats_t<4, 2>(h, a); break;
            case 44: launch_stats_t<4, 4>(h, a); break;
            default: pyvb_set_error("unsupported tile shape"); return PYVB_E_ARG;
        }
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_syy(pyvb_lds* h) {
    SyyArgs a; a.Y = h->Y; a.Syy = h->Syy; a.N = h->N; a.T = h->T; a.K = h->K;
    hipLaunchKernelGGL(k_syy, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
