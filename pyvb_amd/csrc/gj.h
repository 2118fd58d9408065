// Gauss-Jordan inversion of symmetric positive definite matrices in registers, shared by k_prep.hip and k_wishart.hip.
#pragma once
#include "common.h"

// Inverses of NP symmetric positive definite D x D matrices at once, by Gauss-Jordan elimination
// without pivoting, entirely in registers.  Thread (a = tid/16, b = tid%16) owns the 4 x 4 tile of
// elements (4a + ra, 4b + cb) of every matrix (padded to 64 x 64 with the identity), v[c][4 ra + cb].
// Step p of each:  P_ij -= P_ip P_pj / piv  off row and column p, row p *= 1/piv, column p *= -1/piv,
// pivot -> 1/piv.  A thread needs 4 entries of column p and 4 of row p per step (two 32-byte LDS reads
// each); the loop over p is unrolled by four so that which of its rows / columns is the pivot one is
// a compile-time index.  Row p+1, column p+1 and the reciprocal of the next pivot (computed once, by the
// thread that owns it) are stashed into double-buffered LDS vectors as they are produced, so a step
// costs one barrier, and the NP independent eliminations interleave to cover its latency.
// rc: scratch [NP][2][GJ_BUF], pivs [NP][64].  On return v holds the inverses and pivs the pivots,
// whose logs sum to 2 * sum log diag(chol(P)).
// (Layout of a pivot's LDS vectors.  Shifting the second half of the row by two doubles -- GJ_ROW(j) = j + 2 (j / 32), against the
// two-way bank conflict of threads whose b differ by eight, by four in gj_wave -- was measured and is NOT used: no difference in
// k_prep, and the Wishart column eliminations got slower, 5.57 against 5.36 ms.  The four-way conflict of gj_wg128 below is another
// matter: there the padded row pays.)
#define GJ_ROW(j) (j)
#define GJ_COL 64
#define GJ_D 128
#define GJ_BUF 136      // row (64), column (64), 1/pivot, padding
template <int NP>
__device__ __forceinline__ void gj_inverse(double (&v)[NP][16], int D, int tid, double* rc, double* pivs) {
    const int a = tid >> 4, b = tid & 15;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        double* row = rc + (c * 2) * GJ_BUF;
        if (a == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) row[GJ_ROW(4 * b) + cb] = v[c][cb];   // row 0
        }
        if (b == 0) {
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) row[GJ_COL + 4 * a + ra] = v[c][4 * ra];  // column 0
        }
        if (tid == 0) row[GJ_D] = 1.0 / v[c][0];
    }
    int cur = 0;
    for (int P = 0; 4 * P < D; ++P) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int p = 4 * P + pp;
            if (p >= D) break;                                  // block-uniform
            const int P1 = (pp == 3) ? P + 1 : P, q1 = (pp + 1) & 3;    // where row / column p + 1 live
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const double* row = rc + (c * 2 + cur) * GJ_BUF;
                const double* col = row + GJ_COL;
                double* nrow = rc + (c * 2 + (cur ^ 1)) * GJ_BUF;
                double* ncol = nrow + GJ_COL;
                const double d = row[GJ_D];
                if (tid == 0) pivs[c * 64 + p] = row[GJ_ROW(p)];
                double rj[4], ci[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { rj[k] = row[GJ_ROW(4 * b) + k] * d; ci[k] = col[4 * a + k]; }
#pragma unroll
                for (int ra = 0; ra < 4; ++ra)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) v[c][4 * ra + cb] -= ci[ra] * rj[cb];
                if (b == P) {                                   // column p
#pragma unroll
                    for (int ra = 0; ra < 4; ++ra) v[c][4 * ra + pp] = -ci[ra] * d;
                }
                if (a == P) {                                   // row p
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) v[c][4 * pp + cb] = (b == P && cb == pp) ? d : rj[cb];
                }
                if (a == P1) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) nrow[GJ_ROW(4 * b) + cb] = v[c][4 * q1 + cb];
                }
                if (b == P1) {
#pragma unroll
                    for (int ra = 0; ra < 4; ++ra) ncol[4 * a + ra] = v[c][4 * ra + q1];
                }
                if (a == P1 && b == P1) nrow[GJ_D] = 1.0 / v[c][4 * q1 + q1];
            }
            cur ^= 1;
        }
    }
    __syncthreads();
}


// ---------------------------------------------------------------------------------------------------------------------
// The same elimination by ONE wavefront per matrix: lane (a = lane / 8, b = lane % 8) owns the 8 x 8 tile of elements
// (8a + ra, 8b + cb), v[ra][cb] (64 doubles per lane, the matrix padded to 64 x 64 with the identity).  A step is then 64
// FMAs per lane against 8 + 8 broadcast values -- four times the arithmetic per exchanged value of the 4 x 4 tiling above --
// and involves no other wavefront: the pivot row, the pivot column and the reciprocal of the pivot go through a scratch
// area in LDS that only this wavefront touches (LDS operations of one wavefront execute in order), so there is no
// workgroup barrier in the loop and the wavefronts of a workgroup invert different matrices independently.
// rc: this wavefront's scratch [2][GJW_BUF]; pivs: [64] pivots (their logs sum to ln det).  Used for the column
// covariances under Wishart noise (k_wishart.hip), where a replicate needs 2 D inversions per iteration.
#define GJW_BUF GJ_BUF   // the same layout
__device__ __forceinline__ double gjw_recip(double x) {
    // v_rcp_f64 and two Newton steps: the IEEE division is ~40 instructions that all 64 lanes would sit through for one pivot
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void gjw_sync() {
    // orders this wavefront's LDS writes before its later LDS reads for the COMPILER (the hardware keeps them in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void gj_wave(double (&v)[8][8], int D, int lane, double* rc, double* pivs) {
    const int a = lane >> 3, b = lane & 7;
    {
        double* row = rc;
        if (a == 0) {
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) row[GJ_ROW(8 * b) + cb] = v[0][cb];   // row 0
        }
        if (b == 0) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) row[GJ_COL + 8 * a + ra] = v[ra][0];  // column 0
        }
        if (lane == 0) row[GJ_D] = gjw_recip(v[0][0]);
    }
    int cur = 0;
    for (int P = 0; 8 * P < D; ++P) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int p = 8 * P + pp;
            if (p >= D) continue;                               // wave-uniform (no break: the eight steps stay unrolled, pp a constant)
            const int P1 = (pp == 7) ? P + 1 : P, q1 = (pp + 1) & 7;    // where row / column p + 1 live
            gjw_sync();
            const double* row = rc + cur * GJW_BUF;
            const double* col = row + GJ_COL;
            double* nrow = rc + (cur ^ 1) * GJW_BUF;
            double* ncol = nrow + GJ_COL;
            const double d = row[GJ_D];
            if (lane == 0) pivs[p] = row[GJ_ROW(p)];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[GJ_ROW(8 * b) + k] * d; ci[k] = col[8 * a + k]; }
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[ra][cb] = __builtin_fma(-ci[ra], rj[cb], v[ra][cb]);
            if (b == P) {                                       // column p
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) v[ra][pp] = -ci[ra] * d;
            }
            if (a == P) {                                       // row p
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[pp][cb] = (b == P && cb == pp) ? d : rj[cb];
            }
            if (a == P1) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) nrow[GJ_ROW(8 * b) + cb] = v[q1][cb];
            }
            if (b == P1) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) ncol[8 * a + ra] = v[ra][q1];
            }
            if (a == P1 && b == P1) nrow[GJ_D] = gjw_recip(v[q1][q1]);
            cur ^= 1;
        }
    }
    gjw_sync();
}

// The same step applied once more to the pivots in `mask` (bit p set) of a matrix that gj_wave has already inverted.  The step
// is an exchange operator (an involution): exchanging back the indices o of the mask leaves, for the others (u),
//   block [u, u] = inv(P_uu),   block [u, o] = inv(P_uu) P_uo
// of the ORIGINAL matrix P -- the conditional covariance and the regression of a Gaussian conditioned on the entries o
// (gaussian.py:125-134), and sum_o ln(pivot) + ln det P = ln det P_uu.  The pivots need not be consecutive, so every step
// publishes its own row, column and reciprocal first (two wave-level syncs per step); pivs2 [64] gets the pivots met.
__device__ __forceinline__ void gj_wave_subset(double (&v)[8][8], unsigned long long mask, int lane, double* rc, double* pivs2) {
    const int a = lane >> 3, b = lane & 7;
    for (int P = 0; P < 8; ++P) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int p = 8 * P + pp;
            if (!((mask >> p) & 1ull)) continue;                // wave-uniform
            double* row = rc;
            double* col = rc + GJ_COL;
            gjw_sync();
            if (a == P) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) row[GJ_ROW(8 * b) + cb] = v[pp][cb];
            }
            if (b == P) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) col[8 * a + ra] = v[ra][pp];
            }
            if (a == P && b == P) { row[GJ_D] = gjw_recip(v[pp][pp]); pivs2[p] = v[pp][pp]; }
            gjw_sync();
            const double d = row[GJ_D];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[GJ_ROW(8 * b) + k] * d; ci[k] = col[8 * a + k]; }
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[ra][cb] = __builtin_fma(-ci[ra], rj[cb], v[ra][cb]);
            if (b == P) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) v[ra][pp] = -ci[ra] * d;
            }
            if (a == P) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[pp][cb] = (b == P && cb == pp) ? d : rj[cb];
            }
        }
    }
    gjw_sync();
}


// ---------------------------------------------------------------------------------------------------------------------
// 128-wide: the whole workgroup on one matrix (k_big.hip: k_prep_big; k_wishart_big.hip)
// Inverse of the symmetric positive definite matrix held as 8 x 8 tiles by the 16 x 16 threads of the workgroup (thread
// (a = tid / 16, b = tid % 16) owns elements (8a + ra, 8b + cb)), Gauss-Jordan without pivoting as gj.h; rc: [2][GJB_BUF] doubles
// of LDS (row, column, 1/pivot), pivs: [128].
// The pivot row sits in LDS with a stride of NINE doubles per thread column (element j at 9 (j / 8) + j % 8): the sixteen threads that
// differ in b read 8 b + k otherwise -- 64 bytes apart, four of the 64 banks, a four-way conflict on every one of the eight reads of a
// pivot (stamps: 830 of a pivot's 2 300 cycles went to these reads).
#define GJB_ROW(j) (9 * ((j) >> 3) + ((j) & 7))
#define GJB_COL 144
#define GJB_D 272
#define GJB_BUF 280
__device__ __forceinline__ void gj_wg128(double (&v)[8][8], int D, int tid, double* rc, double* pivs) {
    const int a = tid >> 4, b = tid & 15;
    if (a == 0) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) rc[9 * b + cb] = v[0][cb];
    }
    if (b == 0) {
#pragma unroll
        for (int ra = 0; ra < 8; ++ra) rc[GJB_COL + 8 * a + ra] = v[ra][0];
    }
    if (tid == 0) rc[GJB_D] = 1.0 / v[0][0];
    int cur = 0;
#ifdef GJ_STAMP     // (profiles/build_variant.sh k_big gjstamp "-DGJ_STAMP -DPREP_STAMP": what a pivot's cycles are made of)
    unsigned long long gs_t = __builtin_amdgcn_s_memtime(), gs_acc[5] = {0, 0, 0, 0, 0};
#define GSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gs_acc[i] += t_ - gs_t; gs_t = t_; } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
    for (int P = 0; 8 * P < D; ++P) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int p = 8 * P + pp;
            if (p >= D) continue;                                   // block-uniform
            const int P1 = (pp == 7) ? P + 1 : P, q1 = (pp + 1) & 7;
            GSTAMP(0);
            __syncthreads();
            GSTAMP(1);
            const double* row = rc + cur * GJB_BUF;
            const double* col = row + GJB_COL;
            double* nrow = rc + (cur ^ 1) * GJB_BUF;
            double* ncol = nrow + GJB_COL;
            const double d = row[GJB_D];
            if (tid == 0) pivs[p] = row[GJB_ROW(p)];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[9 * b + k] * d; ci[k] = col[8 * a + k]; }
            GSTAMP(2);
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[ra][cb] = __builtin_fma(-ci[ra], rj[cb], v[ra][cb]);
            GSTAMP(3);
            if (b == P) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) v[ra][pp] = -ci[ra] * d;
            }
            if (a == P) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[pp][cb] = (b == P && cb == pp) ? d : rj[cb];
            }
            if (a == P1) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) nrow[9 * b + cb] = v[q1][cb];
            }
            if (b == P1) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) ncol[8 * a + ra] = v[ra][q1];
            }
            if (a == P1 && b == P1) nrow[GJB_D] = 1.0 / v[q1][q1];
            cur ^= 1;
        }
    }
    __syncthreads();
#ifdef GJ_STAMP
    GSTAMP(0);
    if (blockIdx.x == 100 && (tid == 0 || tid == 200))
        printf("gj_wg128 thread %d: fix-ups, publishing, 1/pivot %llu | at the barrier %llu | row, column, 1/pivot from LDS %llu | 64 multiply-adds %llu\n", tid, gs_acc[0], gs_acc[1], gs_acc[2], gs_acc[3]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same with GJR pivots to a step (-DGJ_RANK=2 or 4; profiles/build_variant.sh): a step takes the GJR rows and columns of its pivots as
// they stand (published by their owners at the end of the step before), inverts the GJR x GJR pivot block B in every thread, and applies
//     M[i,j] -= C_i B^-1 R_j      M[B,j] = B^-1 R_j      M[i,B] = -C_i B^-1      M[B,B] = B^-1
// -- the arithmetic of GJR single-pivot steps behind one barrier; the pivots of the small elimination are those of the single steps.
// An experiment of round 4, not in the shipped library: correct with 2 and with 4 (all tests of the 128-wide class) and no faster --
// the three inversions of k_prep_big take 850 000 cycles per workgroup with a pivot per barrier, 881 000 with two, 1 017 000 with
// four (there the 128 + 64 + 64 live doubles per thread run through accumulator-register moves).  The barrier is not what a pivot's
// 2 200 cycles are made of (profiles/r04/prep_big_stamps.txt).
#ifdef GJ_RANK
#define GJR GJ_RANK
#undef GJB_BUF
#define GJB_BUF (2 * GJR * 128)
__device__ __forceinline__ void gj_wg128_blocked(double (&v)[8][8], int D, int tid, double* rc, double* pivs) {
    const int a = tid >> 4, b = tid & 15;
    constexpr int NH = 8 / GJR;             // steps per 8 x 8 tile
    auto publish = [&](double* buf, int P1, int h1) {       // rows and columns 8 P1 + GJR h1 .. of the matrix as it stands
        if (a == P1) {
#pragma unroll
            for (int r4 = 0; r4 < GJR; ++r4)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    double x = 0.0;
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) x = (hh == h1) ? v[GJR * hh + r4][cb] : x;
                    buf[r4 * 128 + 8 * b + cb] = x;
                }
        }
        if (b == P1) {
#pragma unroll
            for (int c4 = 0; c4 < GJR; ++c4)
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) {
                    double x = 0.0;
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) x = (hh == h1) ? v[ra][GJR * hh + c4] : x;
                    buf[GJR * 128 + c4 * 128 + 8 * a + ra] = x;
                }
        }
    };
    publish(rc, 0, 0);
    int cur = 0;
    for (int P = 0; 8 * P < D; ++P) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int p = 8 * P + GJR * h;
            if (p >= D) continue;                                   // block-uniform (rows beyond D carry the identity)
            __syncthreads();
            const double* row = rc + cur * GJB_BUF;
            const double* col = row + GJR * 128;
            double* nbuf = rc + (cur ^ 1) * GJB_BUF;
            double dm[GJR][GJR];
#pragma unroll
            for (int r4 = 0; r4 < GJR; ++r4)
#pragma unroll
                for (int c4 = 0; c4 < GJR; ++c4) dm[r4][c4] = row[r4 * 128 + p + c4];
#pragma unroll
            for (int k = 0; k < GJR; ++k) {
                const double piv = dm[k][k];
                if (tid == 0 && p + k < D) pivs[p + k] = piv;
                const double d = 1.0 / piv;
#pragma unroll
                for (int j = 0; j < GJR; ++j) if (j != k) dm[k][j] *= d;
#pragma unroll
                for (int i = 0; i < GJR; ++i) {
                    if (i == k) continue;
                    const double f = dm[i][k];
#pragma unroll
                    for (int j = 0; j < GJR; ++j) if (j != k) dm[i][j] = __builtin_fma(-f, dm[k][j], dm[i][j]);
                    dm[i][k] = -f * d;
                }
                dm[k][k] = d;
            }
            double T[GJR][8];
            {
                double rj[GJR][8];
#pragma unroll
                for (int r4 = 0; r4 < GJR; ++r4)
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) rj[r4][cb] = row[r4 * 128 + 8 * b + cb];
#pragma unroll
                for (int r4 = 0; r4 < GJR; ++r4)
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) {
                        double t = dm[r4][0] * rj[0][cb];
#pragma unroll
                        for (int m = 1; m < GJR; ++m) t = __builtin_fma(dm[r4][m], rj[m][cb], t);
                        T[r4][cb] = t;
                    }
            }
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                double ci[GJR];
#pragma unroll
                for (int c4 = 0; c4 < GJR; ++c4) ci[c4] = col[c4 * 128 + 8 * a + ra];
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    double x = v[ra][cb];
#pragma unroll
                    for (int c4 = 0; c4 < GJR; ++c4) x = __builtin_fma(-ci[c4], T[c4][cb], x);
                    v[ra][cb] = x;
                }
                if (b == P) {                                       // the pivots' columns: -C_i B^-1
#pragma unroll
                    for (int c4 = 0; c4 < GJR; ++c4) {
                        double t = ci[0] * dm[0][c4];
#pragma unroll
                        for (int m = 1; m < GJR; ++m) t = __builtin_fma(ci[m], dm[m][c4], t);
                        v[ra][GJR * h + c4] = -t;
                    }
                }
            }
            if (a == P) {                                           // the pivots' rows: B^-1 R_j, and B^-1 itself where they cross the columns
#pragma unroll
                for (int r4 = 0; r4 < GJR; ++r4)
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb)
                        v[GJR * h + r4][cb] = (b == P && cb >= GJR * h && cb < GJR * h + GJR) ? dm[r4][(cb - GJR * h) & (GJR - 1)] : T[r4][cb];
            }
            if (h + 1 < NH) publish(nbuf, P, h + 1); else publish(nbuf, P + 1, 0);
            cur ^= 1;
        }
    }
    __syncthreads();
}
#define gj_wg128 gj_wg128_blocked
#endif
