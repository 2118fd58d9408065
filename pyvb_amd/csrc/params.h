// Shared by the parameter-update translation units (k_params.hip, k_cols.hip).
#pragma once
#include "common.h"

#define LN2PI 1.8378770664093453

struct ParamArgs {
    // statistics
    const double* part; int nchunk; const double* Sigma; const double* qld_x; const double* X; const double* Syy;
    double* mom;            // [N][mom_total]: see k_moments
    const double* sxx;      // k_moments: [N][W][DP][DP] parts of the interior sum of mu mu^T from the backward sweep, or null (then from part)
    int W;
    // parameters
    double *A_mean, *A_var, *C_mean, *C_var, *Q_a, *Q_b, *R_a, *R_b, *qld_A, *qld_C;
    double *resQ, *resR, *elbo;
    const double* Yent;     // k_elbo: per replicate, what the outputs that are not fully observed subtract (k_missing.hip), or null
    Priors pri;
    int N, T, D, K, noise;
    int c0, c1;             // k_cols: columns [c0, c1) are updated
    int fuse;               // k_cols: bit 0 = residuals of the noise node too, bit 1 = and its update
    int which0;             // blockIdx.y + which0 selects the matrix / noise node (0: A, Q; 1: C, R)
    Layout L;
};

// layout of the per-replicate moment block written by k_moments (all row-major, no padding)
//   GA [D][D]  = sum_{t=0}^{T-2} <x x^T>      (children of hstack A: Mult(A, X_t))
//   GC [D][D]  = sum_{t=0}^{T-1} <x x^T>      (children of hstack C)
//   HA [D][D]  = sum_t mu_{t+1} mu_t^T        HC [K][D] = sum_t y_t mu_t^T
//   dp [D]     = diag sum_{t=1}^{T-1} <x x^T> (children of Q)
__host__ __device__ static inline size_t mom_total(int D, int K) { return (size_t)3 * D * D + (size_t)K * D + D; }
#define MOM_GA(D, K) ((size_t)0)
#define MOM_GC(D, K) ((size_t)(D) * (D))
#define MOM_HA(D, K) ((size_t)2 * (D) * (D))
#define MOM_HC(D, K) ((size_t)3 * (D) * (D))
#define MOM_DP(D, K) ((size_t)3 * (D) * (D) + (size_t)(K) * (D))

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// value of v in lane j (j wave-uniform) for every lane: two v_readlane_b32, no LDS
__device__ __forceinline__ double bcast(double v, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}

// digamma for positive and for negative non-integer arguments (recurrence up to x >= 10, then the asymptotic series)
__device__ static inline double digamma_pos(double x) {
    if (!(x - x == 0.0) || x < -4.5e15) return x > 0 ? x : __builtin_nan("");      // below -2^52 every double is an integer: a pole      // bounded recurrence: see tape_digamma (k_tape.hip)
    double r = 0.0;
    if (x < -64.0) { r = -M_PI / tan(M_PI * x); x = 1.0 - x; }
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double ser = f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
    return r + log(x) - 0.5 / x - ser;
}


ParamArgs make_args(pyvb_lds* h);
