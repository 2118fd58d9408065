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
    const double* zeros;    // 64 zeros
    int N, T, D, K, nchunk, chunk_len;
    Layout L;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// The 42 accumulator tiles of the D = K = 64 case (10 of Sxx by symmetry, 16 of Sx1x, 16 of Syx) do not
// fit one wavefront's registers next to the operands, so a workgroup is two wavefronts that split them:
//   wave 0: Sxx (upper tiles) + the first NA tiles of Syx      wave 1: Sx1x + the remaining Syx tiles
// (Syx tiles counted row tile by row tile), NA chosen to give both the same number of MFMAs (21 / 21 at
// D = K = 64).  Both read the same rows at about the same time (a barrier per PF k-steps keeps them
// together), so the second read of a row is an L1 hit.
// XX = false: Sxx is not computed here (the backward sweep has it, k_sweep.hip MODE 2): 16 + 16 tiles.
template <int DT, int KT, bool XX>
struct StatsSplit {
    static constexpr int NXX = XX ? DT * (DT + 1) / 2 : 0, NYX = KT * DT, TOTAL = NXX + DT * DT + NYX;
    static constexpr int NA_RAW = (TOTAL + 1) / 2 - NXX;
    static constexpr int NA = NA_RAW < 0 ? 0 : (NA_RAW > NYX ? NYX : NA_RAW);     // Syx tiles of wave 0
    static constexpr int MA = (NA + DT - 1) / DT;       // wave 0 needs y row tiles [0, MA)
    static constexpr int MB0 = NA / DT;                 // wave 1 needs y row tiles [MB0, KT)
};

#ifndef STATS_OCC
#define STATS_OCC 2      // two wavefronts per SIMD run the fp64 matrix pipe faster than one (profiles/r01/microbench_f64.txt)
#endif
#ifndef STATS_PF
#define STATS_PF 1      // k-steps per half of the operand ring (registers: 2 * PF * 10 doubles next to 168 of accumulators)
#endif

template <int DT, int KT, bool XX>
__global__ void __launch_bounds__(128, STATS_OCC) k_stats(StatsArgs a) {
    constexpr int DP = 16 * DT, PF = STATS_PF;
    using SP = StatsSplit<DT, KT, XX>;
    constexpr int NA = SP::NA, MA = SP::MA, MB0 = SP::MB0, MB = KT - MB0;
    const int ch = blockIdx.x, n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int T = a.T, K = a.K;
    const double* X = a.X + (size_t)n * T * DP;        // rows: stride DP, accumulator order
    const double* Y = a.Y + (size_t)n * T * K;
    const double* Z = a.zeros;                         // a row of zeros: what rows outside the chunk read as
    const int t0 = ch * a.chunk_len;
    const int t1 = (t0 + a.chunk_len < T) ? t0 + a.chunk_len : T;
    double* P = a.part + ((size_t)n * a.nchunk + ch) * a.L.stats_total;

    // Rows are fetched two halves of PF k-steps ahead into a ring of registers (the loop is unrolled so
    // that every load has a fixed destination) and feed the MFMAs straight from there: a row outside
    // [t0, t1) (or, for mu_{t+1}, beyond the chain) is read from the zero row instead, so no operand
    // needs masking.  Padded dimensions are not masked either: they only reach padded rows/columns of the
    // results, which no consumer reads.  A slot is refilled right behind the MFMAs that read it.
    int xoff[DT];
#pragma unroll
    for (int m = 0; m < DT; ++m) xoff[m] = xpos(16 * m + r);      // padded positions hold zeros
    auto row_x = [&](int t, double* dst) {
        const double* p = t < t1 ? X + (size_t)t * DP : Z;
#pragma unroll
        for (int m = 0; m < DT; ++m) dst[m] = p[xoff[m]];
    };
    auto row_x1 = [&](int t, double* dst) {             // row t + 1 for the lanes of row t
        const double* p = (t < t1 && t + 1 < T) ? X + (size_t)(t + 1) * DP : Z;
#pragma unroll
        for (int m = 0; m < DT; ++m) dst[m] = p[xoff[m]];
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
        constexpr int MY = MA > 0 ? MA : 1;
        d4 sxx[DT][DT], syx[MY][DT];
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) sxx[m][k] = d4{0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MY; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) syx[m][k] = d4{0, 0, 0, 0};
        int yoff[MY];
#pragma unroll
        for (int m = 0; m < MY; ++m) { const int dim = 16 * m + r; yoff[m] = dim < K ? dim : K - 1; }
        auto row_y = [&](int t, double* dst) {
            const double* p = t < t1 ? Y + (size_t)t * K : Z;
#pragma unroll
            for (int m = 0; m < MA; ++m) dst[m] = p[yoff[m]];
        };
        double xr[2][PF][DT], yr[2][PF][MY];
        auto fill = [&](int h, int tb) {        // the PF k-steps that start at row tb, into half h of the ring
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                row_x(tb + 4 * p + q, xr[h][p]);
                row_y(tb + 4 * p + q, yr[h][p]);
            }
        };
        // same order of loads as the loop issues them: the compiler's in-order vmcnt waits at the loop head
        // are the minimum over both incoming edges
        fill(0, t0);
        __builtin_amdgcn_sched_barrier(0);
        fill(1, t0 + 4 * PF);
        __builtin_amdgcn_sched_barrier(0);
        for (int tb = t0; tb < t1; tb += 8 * PF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // tile by tile, the PF k-steps of a tile back to back: a chain of dependent MFMAs on one
                // accumulator runs faster than the same MFMAs spread over many (profiles/r01/microbench_f64.txt)
                if constexpr (XX) {
#pragma unroll
                    for (int m = 0; m < DT; ++m)
#pragma unroll
                        for (int k = m; k < DT; ++k)
#pragma unroll
                            for (int p = 0; p < PF; ++p) sxx[m][k] = MFMA(xr[h][p][m], xr[h][p][k], sxx[m][k]);
                }
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int k = 0; k < DT; ++k)
                        if (m * DT + k < NA) {
#pragma unroll
                            for (int p = 0; p < PF; ++p) syx[m][k] = MFMA(yr[h][p][m], xr[h][p][k], syx[m][k]);
                        }
                __builtin_amdgcn_sched_barrier(0);
                fill(h, tb + (h + 2) * 4 * PF);     // refilled behind the MFMAs that read it; a whole half to arrive
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();     // lock-step only (L1 reuse between the two waves): no fence, the loads in flight stay in flight
        }
        if constexpr (XX) {
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int k = m; k < DT; ++k) store_tile(a.L.oSxx, m, k, sxx[m][k], k > m);
        }
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k)
                if (m * DT + k < NA) store_tile(a.L.oSyx, m, k, syx[m][k], false);
    } else {
        constexpr int MY = MB > 0 ? MB : 1;
        d4 sx1[DT][DT], syx[MY][DT];
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) sx1[m][k] = d4{0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < MY; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) syx[m][k] = d4{0, 0, 0, 0};
        int yoff[MY];
#pragma unroll
        for (int m = 0; m < MY; ++m) { const int dim = 16 * (MB0 + m) + r; yoff[m] = dim < K ? dim : K - 1; }
        auto row_y = [&](int t, double* dst) {
            const double* p = t < t1 ? Y + (size_t)t * K : Z;
#pragma unroll
            for (int m = 0; m < MB; ++m) dst[m] = p[yoff[m]];
        };
        double xr[2][PF][DT], x1r[2][PF][DT], yr[2][PF][MY];
        auto fill = [&](int h, int tb) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                row_x(tb + 4 * p + q, xr[h][p]);
                row_x1(tb + 4 * p + q, x1r[h][p]);
                row_y(tb + 4 * p + q, yr[h][p]);
            }
        };
        fill(0, t0);
        __builtin_amdgcn_sched_barrier(0);
        fill(1, t0 + 4 * PF);
        __builtin_amdgcn_sched_barrier(0);
        for (int tb = t0; tb < t1; tb += 8 * PF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int m = 0; m < DT; ++m)
#pragma unroll
                    for (int k = 0; k < DT; ++k)
#pragma unroll
                        for (int p = 0; p < PF; ++p) sx1[m][k] = MFMA(x1r[h][p][m], xr[h][p][k], sx1[m][k]);
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int k = 0; k < DT; ++k)
                        if ((MB0 + m) * DT + k >= NA) {
#pragma unroll
                            for (int p = 0; p < PF; ++p) syx[m][k] = MFMA(yr[h][p][m], xr[h][p][k], syx[m][k]);
                        }
                __builtin_amdgcn_sched_barrier(0);
                fill(h, tb + (h + 2) * 4 * PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k) store_tile(a.L.oSx1x, m, k, sx1[m][k], false);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int k = 0; k < DT; ++k)
                if ((MB0 + m) * DT + k >= NA) store_tile(a.L.oSyx, MB0 + m, k, syx[m][k], false);
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
static void launch_stats_t(pyvb_lds* h, const StatsArgs& a, bool with_sxx) {
    if (with_sxx) hipLaunchKernelGGL((k_stats<DT, KT, true>), dim3(h->nchunk, h->N), dim3(128), 0, h->stream, a);
    else hipLaunchKernelGGL((k_stats<DT, KT, false>), dim3(h->nchunk, h->N), dim3(128), 0, h->stream, a);
}

int launch_stats(pyvb_lds* h, bool with_sxx) {
    if (h->big) return launch_stats_big(h);
    StatsArgs a;
    a.X = h->X[h->cur]; a.Y = h->Y; a.part = h->stats; a.zeros = h->zeros;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.nchunk = h->nchunk; a.chunk_len = h->chunk_len; a.L = h->L;
    {
        TimedLaunch tl(h, PYVB_K_STATS);
        switch (h->L.DT * 10 + h->L.KT) {
            case 11: launch_stats_t<1, 1>(h, a, with_sxx); break;
            case 12: launch_stats_t<1, 2>(h, a, with_sxx); break;
            case 14: launch_stats_t<1, 4>(h, a, with_sxx); break;
            case 21: launch_stats_t<2, 1>(h, a, with_sxx); break;
            case 22: launch_stats_t<2, 2>(h, a, with_sxx); break;
            case 24: launch_stats_t<2, 4>(h, a, with_sxx); break;
            case 41: launch_stats_t<4, 1>(h, a, with_sxx); break;
            case 42: launch_stats_t<4, 2>(h, a, with_sxx); break;
            case 44: launch_stats_t<4, 4>(h, a, with_sxx); break;
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
