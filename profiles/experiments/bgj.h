// Blocked Gauss-Jordan inversion of NP symmetric positive definite matrices (up to 64 x 64) by one 256-thread workgroup,
// with the rank-16 updates on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// Ownership: wavefront w holds row tile w (rows 16w .. 16w+15) of every matrix in ACCUMULATOR layout --
//   t[c][J][e] = element (16 w + 4 e + q, 16 J + r),  q = lane / 16, r = lane % 16
// -- so the products below accumulate straight into the registers that hold the matrix, and a tile a wavefront owns is
// at once a B operand of the instruction (k = 4 s + q, column r  <->  register s), the trick the sweeps use.
// In-place block elimination without pivoting (the matrices are s.p.d.), block step p = 0 .. nb-1:
//   1. the 16 x 16 pivot tile T = A_pp goes through LDS to wavefront c % 4 (one matrix each), which inverts it by
//      16 scalar Gauss-Jordan steps on cross-lane shuffles; the scalar pivots are the squares of the Cholesky diagonal
//      (the caller's log-determinant) and are written to pivs[c][16 p + k]
//   2. wavefront p:      A_pJ <- T^-1 A_pJ  (J != p)          [A operand T^-1: symmetric, read from LDS; B: own registers]
//   3. wavefronts I != p: A_IJ <- A_IJ - A_Ip (T^-1 A_pJ),   A_Ip <- - A_Ip T^-1      [A operand: own tile through LDS]
//      wavefront p:      A_pp <- T^-1
// lds: NP * BGJ_LDS_PER doubles of scratch; pivs: [NP][64].  Tiles beyond nb are never touched.
#pragma once
#include "common.h"

#define BGJ_TS 272                       // a 16 x 16 tile with row stride 17
#define BGJ_LDS_PER (9 * BGJ_TS)         // per matrix: T / T^-1, four row-panel tiles, four per-wavefront staging tiles
#define BGJ_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double bgj_shfl(double v, int src) { return __shfl(v, src, 64); }

// 16 x 16 s.p.d. tile in accumulator layout (v[e] = T[4e+q][r]) -> its inverse, same layout; pivots to piv[0..15] (lane 0)
__device__ __forceinline__ void bgj_tile_inverse(d4& v, int lane, double* piv, int* status) {
    const int q = lane >> 4, r = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ek = k >> 2, qk = k & 3;
        const double d = bgj_shfl(v[ek], qk * 16 + k);
        const double rk = bgj_shfl(v[ek], qk * 16 + r);             // T[k][r]
        double ck[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ck[e] = bgj_shfl(v[e], q * 16 + k);   // T[4e+q][k]
        if (lane == 0) { piv[k] = d; if (!(d > 0.0)) atomicOr(status, 1); }
        const double dinv = 1.0 / d;
        const double rs = rk * dinv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool prow = (e == ek) && (q == qk), pcol = (r == k);
            const double upd = v[e] - ck[e] * rs;
            v[e] = prow ? (pcol ? dinv : rs) : (pcol ? -ck[e] * dinv : upd);
        }
    }
}

template <int NP>
__device__ __forceinline__ void bgj_inverse(d4 (&t)[NP][4], int nb, int tid, double* lds, double* pivs, int* status) {
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, r = lane & 15;
#pragma unroll
    for (int p = 0; p < 4; ++p) {                 // unrolled: the tile indices stay compile-time register names
        if (p >= nb) break;
        // ---- 1. pivot tiles to LDS, one wavefront per matrix inverts
        if (wave == p) {
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[c * BGJ_LDS_PER + (4 * e + q) * 17 + r] = t[c][p][e];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            if (wave == (c & 3)) {
                double* T = lds + c * BGJ_LDS_PER;
                d4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = T[(4 * e + q) * 17 + r];
                bgj_tile_inverse(v, lane, pivs + c * 64 + 16 * p, status);
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(4 * e + q) * 17 + r] = v[e];
            }
        }
        __syncthreads();
        // ---- 2. wavefront p: the row panel, kept and published for the others
        if (wave == p) {
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const double* T = lds + c * BGJ_LDS_PER;
                double ta[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) ta[s] = T[r * 17 + 4 * s + q];         // A operand: T^-1[r][4s+q]
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    if (J == p || J >= nb) continue;
                    d4 acc = d4{0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = BGJ_MFMA(ta[s], t[c][J][s], acc);    // B operand: own registers
                    t[c][J] = acc;
                    double* R = lds + c * BGJ_LDS_PER + (1 + J) * BGJ_TS;
#pragma unroll
                    for (int e = 0; e < 4; ++e) R[(4 * e + q) * 17 + r] = acc[e];
                }
            }
        }
        __syncthreads();
        // ---- 3. the other wavefronts: trailing update and the column panel; wavefront p takes T^-1
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const double* T = lds + c * BGJ_LDS_PER;
            if (wave == p) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c][p][e] = T[(4 * e + q) * 17 + r];
            } else if (wave < nb) {
                // own tile (wave, p) as an A operand: through this wavefront's staging tile (same-wavefront LDS traffic is ordered)
                double* S = lds + c * BGJ_LDS_PER + (5 + wave) * BGJ_TS;
#pragma unroll
                for (int e = 0; e < 4; ++e) S[(4 * e + q) * 17 + r] = t[c][p][e];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                double ca[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) ca[s] = -S[r * 17 + 4 * s + q];          // - A_Ip[r][4s+q]
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    if (J == p || J >= nb) continue;
                    const double* R = lds + c * BGJ_LDS_PER + (1 + J) * BGJ_TS;
                    d4 acc = t[c][J];
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = BGJ_MFMA(ca[s], R[(4 * s + q) * 17 + r], acc);
                    t[c][J] = acc;
                }
                d4 acc = d4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = BGJ_MFMA(ca[s], T[(4 * s + q) * 17 + r], acc);
                t[c][p] = acc;
            }
        }
        __syncthreads();
    }
}
