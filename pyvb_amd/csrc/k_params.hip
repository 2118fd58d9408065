// Parameter updates and the lower bound, one wavefront per replicate, lane = row of the matrix.
//   k_cols   [a.update() for a in As] / [c.update() for c in Cs]:
//            Gaussian.update gaussian.py:102-123 fed by hstack.pass_up_m1_m2 nodes_todo.py:43-62
//   k_resid  sum over children of  1/2 diag<x x^T> + 1/2 diag<mu mu^T> - diag(<x><mu>^T)
//            (nodes_todo.py:138, :190 with Multiplication.pass_down_ExxT node.py:260-271)
//   k_noise  Gamma.update / DiagonalGamma.update  nodes_todo.py:130-138, :187-190
//   k_elbo   sum of log_lower_bound() per node class: gaussian.py:136-151, nodes_todo.py:149-157, :199-204
// With diagonal noise precisions and diagonal column priors every row of A (of C) only
// interacts with itself, so the Gauss-Seidel pass over the columns is sequential in the
// column index but parallel over rows.
#include "common.h"

#define LN2PI 1.8378770664093453

struct ParamArgs {
    // statistics
    const double* part; int nchunk; const double* Sigma; const double* qld_x; const double* X; const double* Syy;
    // parameters
    double *A_mean, *A_var, *C_mean, *C_var, *Q_a, *Q_b, *R_a, *R_b, *qld_A, *qld_C;
    double *resQ, *resR, *elbo;
    Priors pri;
    int N, T, D, K, noise;
    Layout L;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// second-moment matrix of the states over a range of t, into LDS Gm[D][D]
//   which = 0: t = 0..T-2 (children of hstack A)   1: t = 0..T-1 (children of hstack C)
__device__ static void build_G(const ParamArgs& a, int n, int which, double* Gm, int lane) {
    const int D = a.D, T = a.T, DP = a.L.DP;
    const double* P = a.part + (size_t)n * a.nchunk * a.L.stats_total;
    const double* S = a.Sigma + (size_t)n * 3 * D * D;
    const double* xL = a.X + ((size_t)n * T + (T - 1)) * D;
    const double nint = (double)(T - 2);
    for (int idx = lane; idx < D * D; idx += 64) {
        int i = idx / D, j = idx % D;
        double s = 0.0;
        for (int ch = 0; ch < a.nchunk; ++ch) s += P[(size_t)ch * a.L.stats_total + a.L.oSxx + (size_t)i * DP + j];
        s += S[idx] + nint * S[D * D + idx];
        if (which == 0) s -= xL[i] * xL[j]; else s += S[2 * D * D + idx];
        Gm[idx] = s;
    }
}

// H[row][col] summed over chunks (Sx1x for A, Syx for C)
__device__ __forceinline__ double stat_H(const ParamArgs& a, int n, int which, int row, int col) {
    const double* P = a.part + (size_t)n * a.nchunk * a.L.stats_total + (which == 0 ? a.L.oSx1x : a.L.oSyx);
    double s = 0.0;
    for (int ch = 0; ch < a.nchunk; ++ch) s += P[(size_t)ch * a.L.stats_total + (size_t)row * a.L.DP + col];
    return s;
}

template <int WHICH>
__global__ void __launch_bounds__(64) k_cols(ParamArgs a) {
    __shared__ double Gm[64 * 64];
    __shared__ double Mb[64 * 64];     // Mb[col * 64 + row]
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const bool live = lane < rows;
    build_G(a, n, WHICH, Gm, lane);
    double lam = 0.0;
    if (live) {
        lam = WHICH == 0 ? a.Q_a[(size_t)n * D + lane] / a.Q_b[(size_t)n * D + lane]
                         : a.R_a[(size_t)n * a.K + lane] / a.R_b[(size_t)n * a.K + lane];
        for (int j = 0; j < D; ++j) Mb[j * 64 + lane] = M[(size_t)lane * D + j];
    }
    __syncthreads();
    for (int i = 0; i < D; ++i) {
        double lp = 0.0;
        if (live) {
            const double* Gi = Gm + i * D;
            double acc = 0.0;
            for (int j = 0; j < D; ++j) acc += (j == i) ? 0.0 : Mb[j * 64 + lane] * Gi[j];
            const double p0 = pp[(size_t)i * rows + lane];
            const double prec = p0 + lam * Gi[i];                                  // qprec  gaussian.py:117
            const double num = p0 * pm[(size_t)lane * D + i] + lam * (stat_H(a, n, WHICH, lane, i) - acc);
            const double val = num / prec;                                          // qmu    gaussian.py:122-123
            Mb[i * 64 + lane] = val;
            M[(size_t)lane * D + i] = val;
            V[(size_t)i * rows + lane] = 1.0 / prec;                                // qcov (diagonal)
            lp = 0.5 * log(prec);                                                   // log of the Cholesky diagonal
        }
        lp = wave_sum(lp);
        if (lane == 0) qld[i] = 0.5 / lp;                                           // gaussian.py:120 (quirk Q1)
    }
}

template <int WHICH>
__global__ void __launch_bounds__(64) k_resid(ParamArgs a) {
    __shared__ double Gm[64 * 64];
    __shared__ double Mb[64 * 64];
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D, T = a.T;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    const bool live = lane < rows;
    build_G(a, n, WHICH, Gm, lane);
    if (live) for (int j = 0; j < D; ++j) Mb[j * 64 + lane] = M[(size_t)lane * D + j];
    __syncthreads();
    if (!live) return;
    // own second moment of the children: X_t, t = 1..T-1 (Q) or the observed Y_t (R)
    double own;
    if (WHICH == 0) {
        const double* S = a.Sigma + (size_t)n * 3 * D * D;
        const double* x0 = a.X + (size_t)n * T * D;
        double s = 0.0;
        const double* P = a.part + (size_t)n * a.nchunk * a.L.stats_total + a.L.oSxx;
        for (int ch = 0; ch < a.nchunk; ++ch) s += P[(size_t)ch * a.L.stats_total + (size_t)lane * a.L.DP + lane];
        own = s - x0[lane] * x0[lane] + (double)(T - 2) * S[D * D + lane * D + lane] + S[2 * D * D + lane * D + lane];
    } else {
        own = a.Syy[(size_t)n * a.K + lane];
    }
    // <mu mu^T>[k,k] = sum_ij M[k,i] G[i,j] M[k,j] + sum_i var_i[k] G[i,i]      node.py:260-271
    double e = 0.0, hm = 0.0;
    for (int i = 0; i < D; ++i) {
        const double* Gi = Gm + i * D;
        double t = 0.0;
        for (int j = 0; j < D; ++j) t += Gi[j] * Mb[j * 64 + lane];
        const double mi = Mb[i * 64 + lane];
        e += mi * t + V[(size_t)i * rows + lane] * Gi[i];
        hm += stat_H(a, n, WHICH, lane, i) * mi;
    }
    double* res = (WHICH == 0 ? a.resQ : a.resR) + (size_t)n * rows;
    res[lane] = 0.5 * own + 0.5 * e - hm;
}

template <int WHICH>
__global__ void __launch_bounds__(64) k_noise(ParamArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int dim = WHICH == 0 ? a.D : a.K;
    const double* res = (WHICH == 0 ? a.resQ : a.resR) + (size_t)n * dim;
    const double* b0 = WHICH == 0 ? a.pri.Q_b0 : a.pri.R_b0;
    double* qb = (WHICH == 0 ? a.Q_b : a.R_b) + (size_t)n * dim;
    const bool live = lane < dim;
    double r = live ? res[lane] : 0.0;
    if (a.noise == PYVB_NOISE_GAMMA) {
        r = wave_sum(r);                       // traces instead of diagonals  nodes_todo.py:138
        if (live) qb[lane] = b0[0] + r;
    } else if (live) {
        qb[lane] = b0[lane] + r;               // nodes_todo.py:188-190
    }
}

__device__ static double digamma_pos(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double ser = f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
    return r + log(x) - 0.5 / x - ser;
}

// log_lower_bound of one Gamma-family entry  nodes_todo.py:149-157 / :199-204
__device__ static double gamma_llb(double a0, double b0, double qa, double qb) {
    const double Elnx = digamma_pos(qa) - log(qb);
    double ret = (a0 - 1.0) * Elnx - lgamma(a0) + a0 * log(b0) - b0 * (qa / qb);
    ret -= (qa - 1.0) * Elnx - lgamma(qa) + qa * log(qb) - qb * (qa / qb);
    return ret;
}

__global__ void __launch_bounds__(64) k_elbo(ParamArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D, K = a.K, T = a.T;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* x0 = a.X + (size_t)n * T * D;
    const double* qx = a.qld_x + (size_t)n * 3;
    // --- noise expectations
    double qbar = 0.0, rbar = 0.0, lnq = 0.0, lnr = 0.0, rq = 0.0, rr = 0.0, lq = 0.0, lr = 0.0;
    if (lane < D) {
        const double qa = a.Q_a[(size_t)n * D + lane], qb = a.Q_b[(size_t)n * D + lane];
        qbar = qa / qb; lnq = log(qbar); rq = qbar * a.resQ[(size_t)n * D + lane];
        if (a.noise == PYVB_NOISE_DIAGONAL_GAMMA) lq = gamma_llb(a.pri.Q_a0[lane], a.pri.Q_b0[lane], qa, qb);
        else if (lane == 0) lq = gamma_llb(a.pri.Q_a0[0], a.pri.Q_b0[0], qa, qb);
    }
    if (lane < K) {
        const double ra = a.R_a[(size_t)n * K + lane], rb = a.R_b[(size_t)n * K + lane];
        rbar = ra / rb; lnr = log(rbar); rr = rbar * a.resR[(size_t)n * K + lane];
        if (a.noise == PYVB_NOISE_DIAGONAL_GAMMA) lr = gamma_llb(a.pri.R_a0[lane], a.pri.R_b0[lane], ra, rb);
        else if (lane == 0) lr = gamma_llb(a.pri.R_a0[0], a.pri.R_b0[0], ra, rb);
    }
    const double lndQ = wave_sum(lnq), lndR = wave_sum(lnr);       // pass_down_lndet (quirk Q2)
    const double trQ = wave_sum(rq), trR = wave_sum(rr);
    const double LQ = wave_sum(lq), LR = wave_sum(lr);
    // --- X_0 against its Constant parents
    double e0 = 0.0;
    if (lane < D) {
        const int i = lane;
        for (int j = 0; j < D; ++j) {
            const double ex = x0[j] * x0[i] + S0[j * D + i] + a.pri.x0_mean[j] * a.pri.x0_mean[i] - 2.0 * x0[j] * a.pri.x0_mean[i];
            e0 += a.pri.x0_prec[i * D + j] * ex;
        }
    }
    e0 = wave_sum(e0);
    const double nint = (double)(T - 2);
    double LX = -0.5 * D * LN2PI + 0.5 * a.pri.x0_lndet - 0.5 * e0;
    LX += (double)(T - 1) * (-0.5 * D * LN2PI + 0.5 * lndQ) - trQ;
    LX += (double)T * (0.5 * D * LN2PI + 0.5 * D) + 0.5 * (qx[0] + nint * qx[1] + qx[2]);
    const double LY = (double)T * (-0.5 * K * LN2PI + 0.5 * lndR) - trR;
    // --- columns of A and C against their Constant parents
    double la = 0.0, lc = 0.0;
    if (lane < D) {
        const int i = lane;   // column i
        double lndet = 0.0, tr = 0.0;
        for (int k = 0; k < D; ++k) {
            const double p = a.pri.A_pp[(size_t)i * D + k], m = a.A_mean[((size_t)n * D + k) * D + i], m0 = a.pri.A_pm[(size_t)k * D + i];
            lndet += log(p);
            tr += p * (m * m + a.A_var[((size_t)n * D + i) * D + k] + m0 * m0 - 2.0 * m * m0);
        }
        la = -0.5 * D * LN2PI + 0.5 * lndet - 0.5 * tr + 0.5 * D * LN2PI + 0.5 * a.qld_A[(size_t)n * D + i] + 0.5 * D;
        lndet = 0.0; tr = 0.0;
        for (int k = 0; k < K; ++k) {
            const double p = a.pri.C_pp[(size_t)i * K + k], m = a.C_mean[((size_t)n * K + k) * D + i], m0 = a.pri.C_pm[(size_t)k * D + i];
            lndet += log(p);
            tr += p * (m * m + a.C_var[((size_t)n * D + i) * K + k] + m0 * m0 - 2.0 * m * m0);
        }
        lc = -0.5 * K * LN2PI + 0.5 * lndet - 0.5 * tr + 0.5 * K * LN2PI + 0.5 * a.qld_C[(size_t)n * D + i] + 0.5 * K;
    }
    const double LA = wave_sum(la), LC = wave_sum(lc);
    if (lane == 0) {
        double* o = a.elbo + (size_t)n * 6;
        o[0] = LX; o[1] = LY; o[2] = LA; o[3] = LC; o[4] = LQ; o[5] = LR;
    }
}

struct SumArgs { const double* elbo; double* out; int N; };
__global__ void __launch_bounds__(256) k_elbo_sum(SumArgs a) {
    __shared__ double red[256 * 6];
    const int tid = threadIdx.x;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int n = tid; n < a.N; n += 256)
        for (int p = 0; p < 6; ++p) s[p] += a.elbo[(size_t)n * 6 + p];
    for (int p = 0; p < 6; ++p) red[p * 256 + tid] = s[p];
    __syncthreads();
    if (tid < 6) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += red[tid * 256 + i];
        a.out[tid] = t;
    }
}

static ParamArgs make_args(pyvb_lds* h) {
    ParamArgs a;
    a.part = h->stats; a.nchunk = h->nchunk; a.Sigma = h->Sigma; a.qld_x = h->qld_x; a.X = h->X[h->cur]; a.Syy = h->Syy;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var;
    a.Q_a = h->Q_a; a.Q_b = h->Q_b; a.R_a = h->R_a; a.R_b = h->R_b; a.qld_A = h->qld_A; a.qld_C = h->qld_C;
    a.resQ = h->resQ; a.resR = h->resR; a.elbo = h->elbo; a.pri = h->pri;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.noise = h->noise; a.L = h->L;
    return a;
}

int launch_cols(pyvb_lds* h, int which) {
    ParamArgs a = make_args(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    if (which == 0) hipLaunchKernelGGL(k_cols<0>, dim3(h->N), dim3(64), 0, h->stream, a);
    else hipLaunchKernelGGL(k_cols<1>, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_resid(pyvb_lds* h, int which) {
    ParamArgs a = make_args(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    if (which == 0) hipLaunchKernelGGL(k_resid<0>, dim3(h->N), dim3(64), 0, h->stream, a);
    else hipLaunchKernelGGL(k_resid<1>, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_noise(pyvb_lds* h, int which) {
    ParamArgs a = make_args(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    if (which == 0) hipLaunchKernelGGL(k_noise<0>, dim3(h->N), dim3(64), 0, h->stream, a);
    else hipLaunchKernelGGL(k_noise<1>, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo(pyvb_lds* h) {
    ParamArgs a = make_args(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_elbo, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo_sum(pyvb_lds* h) {
    SumArgs a; a.elbo = h->elbo; a.out = h->elbo_sum; a.N = h->N;
    hipLaunchKernelGGL(k_elbo_sum, dim3(1), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
