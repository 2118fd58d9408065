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
#include "params.h"

// k_moments: reduce the chunk partials of k_stats and add the covariance classes
// (<x x^T> = qmu qmu^T + qcov, gaussian.py:162-168), once per statistics pass, fully parallel.
__global__ void __launch_bounds__(256) k_moments(ParamArgs a) {
    const int n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K, T = a.T, DP = a.L.DP;
    const double* P = a.part + (size_t)n * a.nchunk * a.L.stats_total;
    const double* S = a.Sigma + (size_t)n * 3 * D * D;
    const double* x0 = a.X + (size_t)n * T * DP;       // state rows: stride DP, accumulator order (xpos)
    const double* xL = x0 + (size_t)(T - 1) * DP;
    double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double nint = (double)(T - 2);
    for (int idx = tid; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx % D;
        double xx = 0.0, h = 0.0;
        for (int ch = 0; ch < a.nchunk; ++ch) {
            const double* Pc = P + (size_t)ch * a.L.stats_total;
            if (!a.sxx) xx += Pc[a.L.oSxx + (size_t)i * DP + j];
            h += Pc[a.L.oSx1x + (size_t)i * DP + j];
        }
        // from the backward sweep: the interior nodes; the two boundary nodes are added here
        if (a.sxx) {
            xx = x0[xpos(i)] * x0[xpos(j)] + xL[xpos(i)] * xL[xpos(j)];
#pragma unroll 8
            for (int w = 0; w < a.W; ++w) xx += a.sxx[((size_t)n * a.W + w) * DP * DP + (size_t)i * DP + j];
        }
        const double s0 = S[idx], s1 = S[D * D + idx], s2 = S[2 * D * D + idx];
        mo[MOM_GA(D, K) + idx] = xx - xL[xpos(i)] * xL[xpos(j)] + s0 + nint * s1;
        mo[MOM_GC(D, K) + idx] = xx + s0 + nint * s1 + s2;
        mo[MOM_HA(D, K) + idx] = h;
        if (i == j) mo[MOM_DP(D, K) + i] = xx - x0[xpos(i)] * x0[xpos(i)] + nint * s1 + s2;
    }
    for (int idx = tid; idx < K * D; idx += 256) {
        const int k = idx / D, j = idx % D;
        double h = 0.0;
        for (int ch = 0; ch < a.nchunk; ++ch) h += P[(size_t)ch * a.L.stats_total + a.L.oSyx + (size_t)k * DP + j];
        mo[MOM_HC(D, K) + idx] = h;
    }
}

// sum over the workgroup: one wavefront (lane = row or column, D, K <= 64), or two in the second shape class (k_big.hip)
template <int NW>
__device__ __forceinline__ double blk_sum(double v, double* red) {
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    else {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1];
    }
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) k_noise(ParamArgs a) {
    __shared__ double red[2];
    const int WHICH = a.which0 + blockIdx.y;
    const int n = blockIdx.x, lane = threadIdx.x;
    const int dim = WHICH == 0 ? a.D : a.K;
    const double* res = (WHICH == 0 ? a.resQ : a.resR) + (size_t)n * dim;
    const double* b0 = WHICH == 0 ? a.pri.Q_b0 : a.pri.R_b0;
    double* qb = (WHICH == 0 ? a.Q_b : a.R_b) + (size_t)n * dim;
    const bool live = lane < dim;
    double r = live ? res[lane] : 0.0;
    if (a.noise == PYVB_NOISE_GAMMA) {
        r = blk_sum<NW>(r, red);               // traces instead of diagonals  nodes_todo.py:138
        if (live) qb[lane] = b0[0] + r;
    } else if (live) {
        qb[lane] = b0[lane] + r;               // nodes_todo.py:188-190
    }
}

// log_lower_bound of one Gamma-family entry  nodes_todo.py:149-157 / :199-204
__device__ __forceinline__ double gamma_llb(double a0, double b0, double qa, double qb) {
    const double Elnx = digamma_pos(qa) - log(qb);
    double ret = (a0 - 1.0) * Elnx - lgamma(a0) + a0 * log(b0) - b0 * (qa / qb);
    ret -= (qa - 1.0) * Elnx - lgamma(qa) + qa * log(qb) - qb * (qa / qb);
    return ret;
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) k_elbo(ParamArgs a) {
    __shared__ double red[2];
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D, K = a.K, T = a.T;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* x0 = a.X + (size_t)n * T * a.L.DP;
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
    const double lndQ = blk_sum<NW>(lnq, red), lndR = blk_sum<NW>(lnr, red);       // pass_down_lndet (quirk Q2)
    const double trQ = blk_sum<NW>(rq, red), trR = blk_sum<NW>(rr, red);
    const double LQ = blk_sum<NW>(lq, red), LR = blk_sum<NW>(lr, red);
    // --- X_0 against its Constant parents
    double e0 = 0.0;
    if (lane < D) {
        const int i = lane;
#pragma unroll 8
        for (int j = 0; j < D; ++j) {
            const double xj = x0[xpos(j)];
            const double ex = xj * x0[xpos(i)] + S0[j * D + i] + a.pri.x0_mean[j] * a.pri.x0_mean[i] - 2.0 * xj * a.pri.x0_mean[i];
            e0 += a.pri.x0_prec[i * D + j] * ex;
        }
    }
    e0 = blk_sum<NW>(e0, red);
    const double nint = (double)(T - 2);
    double LX = -0.5 * D * LN2PI + 0.5 * a.pri.x0_lndet - 0.5 * e0;
    LX += (double)(T - 1) * (-0.5 * D * LN2PI + 0.5 * lndQ) - trQ;
    LX += (double)T * (0.5 * D * LN2PI + 0.5 * D) + 0.5 * (qx[0] + nint * qx[1] + qx[2]);
    double LY = (double)T * (-0.5 * K * LN2PI + 0.5 * lndR) - trR;
    if (a.Yent) LY -= a.Yent[n];
    // --- columns of A and C against their Constant parents (gaussian.py:141-150).  The last term depends
    // on how much of the column was observed: nothing -> the q_ln_det form (:147), some entries -> the
    // covariance of the missing part (:150, with the reference's sign of the 2 pi term), all -> no term.
    double la = 0.0, lc = 0.0;
    if (lane < D) {
        const int i = lane;   // column i
        auto column = [&](int rows, const double* pp, const double* pm, const double* M, const double* V,
                          const double* obs, double qld, double lndet) {
            // this lane's rows of pp and V are contiguous, the columns of M, pm and obs are coalesced across lanes
            const double* ppi = pp + (size_t)i * rows;
            const double* Vi = V + (size_t)i * rows;
            double tr0 = 0.0, tr1 = 0.0, lvar = 0.0;
            int missing = 0;
            for (int k0 = 0; k0 < rows; k0 += 8) {
                double p[8], v[8], m[8], m0[8], ob[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u < rows ? k0 + u : rows - 1;
                    p[u] = ppi[k]; v[u] = Vi[k];
                    m[u] = M[(size_t)k * D + i]; m0[u] = pm[(size_t)k * D + i]; ob[u] = obs[(size_t)k * D + i];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u < rows) {
                        const double t = p[u] * (m[u] * m[u] + v[u] + m0[u] * m0[u] - 2.0 * m[u] * m0[u]);
                        if (u & 1) tr1 += t; else tr0 += t;
                        if (!(ob[u] == ob[u])) ++missing;
                    }
                }
            }
            double r = -0.5 * rows * LN2PI + 0.5 * lndet - 0.5 * (tr0 + tr1);
            if (missing == rows) r += 0.5 * rows * LN2PI + 0.5 * qld + 0.5 * rows;
            else if (missing > 0) {     // part of the column is known (rare): ln of the variances of the rest
                for (int k = 0; k < rows; ++k) {
                    const double ob = obs[(size_t)k * D + i];
                    if (!(ob == ob)) lvar += log(Vi[k]);
                }
                r -= 0.5 * missing * LN2PI - 0.5 * lvar - 0.5 * missing;
            }
            return r;
        };
        la = column(D, a.pri.A_pp, a.pri.A_pm, a.A_mean + (size_t)n * D * D, a.A_var + (size_t)n * D * D, a.pri.A_obs,
                    a.qld_A[(size_t)n * D + i], a.pri.A_pld[i]);
        lc = column(K, a.pri.C_pp, a.pri.C_pm, a.C_mean + (size_t)n * K * D, a.C_var + (size_t)n * D * K, a.pri.C_obs,
                    a.qld_C[(size_t)n * D + i], a.pri.C_pld[i]);
    }
    const double LA = blk_sum<NW>(la, red), LC = blk_sum<NW>(lc, red);
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

// Gaussian.observe on fully known columns (gaussian.py:97-100): value in, covariance zero, for every replicate
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_observe(ParamArgs a) {
    __shared__ double red[2];
    const int WHICH = blockIdx.y, n = blockIdx.x, lane = threadIdx.x, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs;
    for (int i = 0; i < D; ++i) {
        const double ob = lane < rows ? obs[(size_t)lane * D + i] : 0.0;
        const int nknown = (int)blk_sum<NW>((lane < rows && ob == ob) ? 1.0 : 0.0, red);
        if (nknown == rows && lane < rows) { M[(size_t)lane * D + i] = ob; V[(size_t)i * rows + lane] = 0.0; }
    }
}

ParamArgs make_args(pyvb_lds* h) {
    ParamArgs a;
    a.part = h->stats; a.nchunk = h->nchunk; a.mom = h->mom; a.Sigma = h->Sigma; a.qld_x = h->qld_x; a.X = h->X[h->cur]; a.Syy = h->Syy;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var;
    a.Q_a = h->Q_a; a.Q_b = h->Q_b; a.R_a = h->R_a; a.R_b = h->R_b; a.qld_A = h->qld_A; a.qld_C = h->qld_C;
    a.resQ = h->resQ; a.resR = h->resR; a.elbo = h->elbo; a.pri = h->pri; a.Yent = h->has_missing ? h->Yent : nullptr;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.noise = h->noise; a.L = h->L; a.c0 = 0; a.c1 = h->D; a.which0 = 0; a.fuse = 0; a.sxx = nullptr; a.W = 1;
    return a;
}

int launch_observe(pyvb_lds* h) {
    ParamArgs a = make_args(h);
    if (h->big) hipLaunchKernelGGL(k_observe<2>, dim3(h->N, 2), dim3(128), 0, h->stream, a);
    else hipLaunchKernelGGL(k_observe<1>, dim3(h->N, 2), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_moments(pyvb_lds* h, bool sxx_from_sweep) {
    ParamArgs a = make_args(h);
    a.sxx = sxx_from_sweep ? h->sxx : nullptr;
    a.W = h->W;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_moments, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

// which: 0 = A / Q, 1 = C / R, 2 = both in one launch (they are independent given the statistics)
int launch_noise(pyvb_lds* h, int which) {
    ParamArgs a = make_args(h);
    a.which0 = which == 1 ? 1 : 0;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    if (h->big) hipLaunchKernelGGL(k_noise<2>, dim3(h->N, which == 2 ? 2 : 1), dim3(128), 0, h->stream, a);
    else hipLaunchKernelGGL(k_noise<1>, dim3(h->N, which == 2 ? 2 : 1), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo(pyvb_lds* h, hipStream_t stream) {
    ParamArgs a = make_args(h);
    if (!stream) stream = h->stream;
    TimedLaunch tl(h, PYVB_K_ELBO, stream);
    if (h->big) hipLaunchKernelGGL(k_elbo<2>, dim3(h->N), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL(k_elbo<1>, dim3(h->N), dim3(64), 0, stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo_sum(pyvb_lds* h, double* out, hipStream_t stream) {
    SumArgs a; a.elbo = h->elbo; a.out = out ? out : h->elbo_sum; a.N = h->N;
    hipLaunchKernelGGL(k_elbo_sum, dim3(1), dim3(256), 0, stream ? stream : h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
