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
#define GJ_BUF 136      // row (64), column (64), 1/pivot, padding
template <int NP>
__device__ __forceinline__ void gj_inverse(double (&v)[NP][16], int D, int tid, double* rc, double* pivs) {
    const int a = tid >> 4, b = tid & 15;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        double* row = rc + (c * 2) * GJ_BUF;
        if (a == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) row[4 * b + cb] = v[c][cb];           // row 0
        }
        if (b == 0) {
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) row[64 + 4 * a + ra] = v[c][4 * ra];  // column 0
        }
        if (tid == 0) row[128] = 1.0 / v[c][0];
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
                const double* col = row + 64;
                double* nrow = rc + (c * 2 + (cur ^ 1)) * GJ_BUF;
                double* ncol = nrow + 64;
                const double d = row[128];
                if (tid == 0) pivs[c * 64 + p] = row[p];
                double rj[4], ci[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { rj[k] = row[4 * b + k] * d; ci[k] = col[4 * a + k]; }
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
                    for (int cb = 0; cb < 4; ++cb) nrow[4 * b + cb] = v[c][4 * q1 + cb];
                }
                if (b == P1) {
#pragma unroll
                    for (int ra = 0; ra < 4; ++ra) ncol[4 * a + ra] = v[c][4 * ra + q1];
                }
                if (a == P1 && b == P1) nrow[128] = 1.0 / v[c][4 * q1 + q1];
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
#define GJW_BUF 136     // row (64), column (64), 1/pivot, padding
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
            for (int cb = 0; cb < 8; ++cb) row[8 * b + cb] = v[0][cb];           // row 0
        }
        if (b == 0) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) row[64 + 8 * a + ra] = v[ra][0];      // column 0
        }
        if (lane == 0) row[128] = gjw_recip(v[0][0]);
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
            const double* col = row + 64;
            double* nrow = rc + (cur ^ 1) * GJW_BUF;
            double* ncol = nrow + 64;
            const double d = row[128];
            if (lane == 0) pivs[p] = row[p];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[8 * b + k] * d; ci[k] = col[8 * a + k]; }
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
                for (int cb = 0; cb < 8; ++cb) nrow[8 * b + cb] = v[q1][cb];
            }
            if (b == P1) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) ncol[8 * a + ra] = v[ra][q1];
            }
            if (a == P1 && b == P1) nrow[128] = gjw_recip(v[q1][q1]);
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
            double* col = rc + 64;
            gjw_sync();
            if (a == P) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) row[8 * b + cb] = v[pp][cb];
            }
            if (b == P) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) col[8 * a + ra] = v[ra][pp];
            }
            if (a == P && b == P) { row[128] = gjw_recip(v[pp][pp]); pivs2[p] = v[pp][pp]; }
            gjw_sync();
            const double d = row[128];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[8 * b + k] * d; ci[k] = col[8 * a + k]; }
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
// (a = tid / 16, b = tid % 16) owns elements (8a + ra, 8b + cb)), Gauss-Jordan without pivoting as gj.h; rc: [2][264] doubles
// of LDS (row 128, column 128, 1/pivot), pivs: [128].
#define GJB_BUF 264
__device__ __forceinline__ void gj_wg128(double (&v)[8][8], int D, int tid, double* rc, double* pivs) {
    const int a = tid >> 4, b = tid & 15;
    if (a == 0) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) rc[8 * b + cb] = v[0][cb];
    }
    if (b == 0) {
#pragma unroll
        for (int ra = 0; ra < 8; ++ra) rc[128 + 8 * a + ra] = v[ra][0];
    }
    if (tid == 0) rc[256] = 1.0 / v[0][0];
    int cur = 0;
    for (int P = 0; 8 * P < D; ++P) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int p = 8 * P + pp;
            if (p >= D) continue;                                   // block-uniform
            const int P1 = (pp == 7) ? P + 1 : P, q1 = (pp + 1) & 7;
            __syncthreads();
            const double* row = rc + cur * GJB_BUF;
            const double* col = row + 128;
            double* nrow = rc + (cur ^ 1) * GJB_BUF;
            double* ncol = nrow + 128;
            const double d = row[256];
            if (tid == 0) pivs[p] = row[p];
            double rj[8], ci[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { rj[k] = row[8 * b + k] * d; ci[k] = col[8 * a + k]; }
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
            if (a == P1) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) nrow[8 * b + cb] = v[q1][cb];
            }
            if (b == P1) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) ncol[8 * a + ra] = v[ra][q1];
            }
            if (a == P1 && b == P1) nrow[256] = 1.0 / v[q1][q1];
            cur ^= 1;
        }
    }
    __syncthreads();
}

