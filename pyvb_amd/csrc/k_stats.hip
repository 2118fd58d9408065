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
#include <type_traits>

struct StatsArgs {
    const double* X; const double* Y; double* part;
    int N, T, D, K, nchunk, chunk_len;
    Layout L;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// The 42 accumulator tiles of the D = K = 64 case (10 of Sxx by symmetry, 16 of Sx1x, 16 of Syx) do not
// fit one wavefront's registers next to the operands, so a workgroup is two wavefronts that split them:
//   wave 0: Sxx (upper tiles) + the first KA row tiles of Syx      wave 1: Sx1x + the remaining Syx row tiles
// with KA chosen to balance the MFMA counts (22 / 20 at D = K = 64).  Both read the same rows at the same
// time (a barrier per k-step keeps them together), so the second read of a row is an L1 hit.
template <int DT, int KT>
struct StatsSplit {
    static constexpr int NXX = DT * (DT + 1) / 2, TOTAL = NXX + DT * DT + KT * DT;
    static constexpr int KA_RAW = (TOTAL - 2 * NXX + DT) / (2 * DT);      // round((TOTAL/2 - NXX) / DT)
    static constexpr int KA = KA_RAW < 0 ? 0 : (KA_RAW > KT ? KT : KA_RAW);
};

template <int DT, int KT>
__global__ void __launch_bounds__(128, 2) k_stats(StatsArgs a) {
    constexpr int DP = 16 * DT;
    constexpr int KA = StatsSplit<DT, KT>::KA, KB = KT - KA;
    const int ch = blockIdx.x, n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int T = a.T, D = a.D, K = a.K;
    const double* X = a.X + (size_t)n * T * DP;        // rows: stride DP, accumulator order
    const double* Y = a.Y + (size_t)n * T * K;
    const int t0 = ch * a.chunk_len;
    const int t1 = (t0 + a.chunk_len < T) ? t0 + a.chunk_len : T;
    double* P = a.part + ((size_t)n * a.nchunk + ch) * a.L.stats_total;

    // Rows are read from clamped (always valid) addresses and zeroed by selects when they lie outside
    // [t0, t1) (or, for mu_{t+1}, beyond the chain).  Padded dimensions are not masked: they only reach
    // padded rows/columns of the results, which no consumer reads.
    // raw (unmasked) loads are issued ahead of a k-step's MFMAs, the masks are applied after them
    auto row_x = [&](int t, bool valid, double* dst) {
        const int tc = valid ? t : t0;
#pragma unroll
        for (int m = 0; m < DT; ++m) {
            const int dim = 16 * m + r;
            dst[m] = X[(size_t)tc * DP + xpos(dim)];      // padded positions hold zeros
        }
    };
    auto row_y = [&](int t, bool valid, int m0, int cnt, double* dst) {
        const int tc = valid ? t : t0;
        for (int m = 0; m < cnt; ++m) {
            const int dim = 16 * (m0 + m) + r;
            dst[m] = Y[(size_t)tc * K + (dim < K ? dim : K - 1)];
        }
    };
    auto store_tile = [&](size_t base, int m, int k, const d4& acc, bool mirror) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * m + 4 * e + q, col = 16 * k + r;     // accumulator element (row, col)
            P[base + (size_t)row * DP + col] = acc[e];
            if (mirror) P[base + (size_t)col * DP + row] = acc[e];
        }
    };

    if (wave == 0) {
        d4 sxx[DT][DT], syx[KA > 0 ? KA : 1][DT];
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) sxx[m][k] = d4{0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < KA; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) syx[m][k] = d4{0, 0, 0, 0};
        double xa[DT], ya[KA > 0 ? KA : 1];
        {
            const bool v = t0 + q < t1;
            row_x(t0 + q, v, xa);
            row_y(t0 + q, v, 0, KA, ya);
#pragma unroll
            for (int m = 0; m < DT; ++m) xa[m] = v ? xa[m] : 0.0;
#pragma unroll
            for (int m = 0; m < KA; ++m) ya[m] = v ? ya[m] : 0.0;
        }
        for (int tb = t0; tb < t1; tb += 4) {
            double xa_n[DT], ya_n[KA > 0 ? KA : 1];
            const bool v = tb + 4 + q < t1;
            row_x(tb + 4 + q, v, xa_n);
#pragma unroll
            for (int m = 0; m < KA; ++m) { const int dim = 16 * m + r; ya_n[m] = Y[(size_t)(v ? tb + 4 + q : t0) * K + (dim < K ? dim : K - 1)]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int k = m; k < DT; ++k) sxx[m][k] = MFMA(xa[m], xa[k], sxx[m][k]);
#pragma unroll
            for (int m = 0; m < KA; ++m)
#pragma unroll
                for (int k = 0; k < DT; ++k) syx[m][k] = MFMA(ya[m], xa[k], syx[m][k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < DT; ++m) xa[m] = v ? xa_n[m] : 0.0;
#pragma unroll
            for (int m = 0; m < KA; ++m) ya[m] = v ? ya_n[m] : 0.0;
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = m; k < DT; ++k) store_tile(a.L.oSxx, m, k, sxx[m][k], k > m);
#pragma unroll
        for (int m = 0; m < KA; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) store_tile(a.L.oSyx, m, k, syx[m][k], false);
    } else {
        d4 sx1[DT][DT], syx[KB > 0 ? KB : 1][DT];
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) sx1[m][k] = d4{0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < KB; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) syx[m][k] = d4{0, 0, 0, 0};
        double xa[DT], xb[DT], ya[KB > 0 ? KB : 1];
        {
            const bool v = t0 + q < t1, v1 = v && (t0 + q + 1 < T);
            row_x(t0 + q, v, xa);
            row_x(t0 + q + 1, v1, xb);
            row_y(t0 + q, v, KA, KB, ya);
#pragma unroll
            for (int m = 0; m < DT; ++m) { xa[m] = v ? xa[m] : 0.0; xb[m] = v1 ? xb[m] : 0.0; }
#pragma unroll
            for (int m = 0; m < KB; ++m) ya[m] = v ? ya[m] : 0.0;
        }
        for (int tb = t0; tb < t1; tb += 4) {
            double xa_n[DT], xb_n[DT], ya_n[KB > 0 ? KB : 1];
            const bool v = tb + 4 + q < t1, v1 = v && (tb + 5 + q < T);
            row_x(tb + 4 + q, v, xa_n);
            row_x(tb + 5 + q, v1, xb_n);
#pragma unroll
            for (int m = 0; m < KB; ++m) { const int dim = 16 * (KA + m) + r; ya_n[m] = Y[(size_t)(v ? tb + 4 + q : t0) * K + (dim < K ? dim : K - 1)]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int k = 0; k < DT; ++k) sx1[m][k] = MFMA(xb[m], xa[k], sx1[m][k]);
#pragma unroll
            for (int m = 0; m < KB; ++m)
#pragma unroll
                for (int k = 0; k < DT; ++k) syx[m][k] = MFMA(ya[m], xa[k], syx[m][k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < DT; ++m) { xa[m] = v ? xa_n[m] : 0.0; xb[m] = v1 ? xb_n[m] : 0.0; }
#pragma unroll
            for (int m = 0; m < KB; ++m) ya[m] = v ? ya_n[m] : 0.0;
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) store_tile(a.L.oSx1x, m, k, sx1[m][k], false);
#pragma unroll
        for (int m = 0; m < KB; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) store_tile(a.L.oSyx, KA + m, k, syx[m][k], false);
    }
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
    hipLaunchKernelGGL((k_stats<DT, KT>), dim3(h->nchunk, h->N), dim3(128), 0, h->stream, a);
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
