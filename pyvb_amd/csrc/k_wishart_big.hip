// Wishart noise precisions (k_wishart.hip; nodes_todo.py:205-234, Linear_Dynamic_System.py:55-56) on the second shape class of
// the fused LDS path, 64 < max(D, K) <= 128.
//
// The 64-wide kernels of k_wishart.hip keep a 64 x 64 matrix per wavefront in registers and their operands in 64 x 64 LDS
// arrays; neither exists at 128.  Here a (replicate, matrix) is a workgroup of 256 threads, the 128 x 128 work matrices stay in
// global memory (L2: they are read along rows by neighbouring threads), every inversion is the workgroup-wide Gauss-Jordan
// of k_prep_big (gj_wg128, 8 x 8 tile per thread, a barrier per pivot) and the columns of a matrix are taken one after the other:
//   k_wexpect_big        E[Q], E[R] and their log-determinants                       (Wishart.pass_down_Ex, :233-234)
//   k_dense_pre_big      E[Q]<A>, E[R]<C>, tr(S_i E[Q]), tr(S'_i E[R])               (node.py:213-227 with dense expectations)
//   k_cols_wishart_big   [a.update() for a in As] / Cs: D inversions of a rows x rows precision, the Gauss-Seidel chain of
//                        the means, sum_i G_ii S_i for the noise update               (gaussian.py:102-123, nodes_todo.py:43-62)
//   k_wresid_big         Wishart.update: qw = w0 + 1/2 (own + <M> G <M>^T + sum_i S_i G_ii) - H <M>^T     (:228-231)
//   k_syy_full_big       sum_t y_t y_t^T
// This is the straightforward form -- correct, not tuned: a column costs a full 128-pivot elimination by one workgroup
// (~0.4 M cycles), a replicate 2 D of them per iteration.  What it buys is that such a graph stays on the fused sweeps and
// statistics (k_big.hip) instead of falling to the node-by-node interpreter.  Known entries of A / C and outputs with NaN are
// served with Wishart noise by the 64-wide kernels only (pyvb_lds_create / the recogniser send larger graphs of that kind to
// the node-by-node plan).  Same three deviations from the unfinished reference class as k_wishart.hip.
#include "params.h"
#include "gj.h"

struct WBArgs {
    double *Q_w, *R_w, *Qbar, *Rbar, *lnd, *QA, *RC, *trA, *trC, *A_cov, *C_cov, *RQ, *RR, *SyyF, *SG, *scratch;
    const double *Q_a, *R_a;
    double *A_mean, *A_var, *C_mean, *C_var, *qld_A, *qld_C;
    const double *mom, *X, *Sigma, *Y;
    Priors pri;
    int* status;
    int N, T, D, K, DP;
    int which0, c0, c1, update, use_sg;
};

static WBArgs make_wbargs(pyvb_lds* h) {
    WBArgs a;
    a.Q_w = h->Q_w; a.R_w = h->R_w; a.Qbar = h->Qbar; a.Rbar = h->Rbar; a.lnd = h->lnd; a.QA = h->QA; a.RC = h->RC;
    a.trA = h->trA; a.trC = h->trC; a.A_cov = h->A_cov; a.C_cov = h->C_cov; a.RQ = h->RQ; a.RR = h->RR; a.SyyF = h->SyyF;
    a.SG = h->SG; a.scratch = h->scratch;
    a.Q_a = h->Q_a; a.R_a = h->R_a;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var; a.qld_A = h->qld_A; a.qld_C = h->qld_C;
    a.mom = h->mom; a.X = h->X[h->cur]; a.Sigma = h->Sigma; a.Y = h->Y;
    a.pri = h->pri; a.status = h->status;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.DP = h->L.DP;
    a.which0 = 0; a.c0 = 0; a.c1 = h->D; a.update = 0; a.use_sg = 0;
    return a;
}

// position of element (k, l) in the stored upper 8 x 8 tiles of a symmetric rows x rows matrix (k_wishart.hip: cov_pos)
__device__ __forceinline__ size_t covb_pos(int rows, int k, int l) {
    const int RT = (rows + 7) >> 3;
    const int lo = k < l ? k : l, hi = k < l ? l : k, A = lo >> 3, B = hi >> 3;
    const int t = A * RT - (A * (A - 1)) / 2 + (B - A);
    return (size_t)t * 64 + (A == B ? (k & 7) * 8 + (l & 7) : (lo & 7) * 8 + (hi & 7));
}

// ---- E[Q], E[R], ln det: the two symmetrised qw inverted one after the other
__global__ void __launch_bounds__(256) k_wexpect_big(WBArgs a) {
    __shared__ double rc[2 * GJB_BUF], pivs[128];
    const int n = blockIdx.x, tid = threadIdx.x, ta = tid >> 4, tb = tid & 15;
    for (int c = 0; c < 2; ++c) {
        const int dim = c == 0 ? a.D : a.K;
        const double* W = (c == 0 ? a.Q_w : a.R_w) + (size_t)n * dim * dim;
        double* out = (c == 0 ? a.Qbar : a.Rbar) + (size_t)n * dim * dim;
        const double qv = (c == 0 ? a.Q_a : a.R_a)[(size_t)n * dim];
        double v[8][8];
#pragma unroll
        for (int ra = 0; ra < 8; ++ra)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int i = 8 * ta + ra, j = 8 * tb + cb;
                v[ra][cb] = (i < dim && j < dim) ? 0.5 * (W[(size_t)i * dim + j] + W[(size_t)j * dim + i]) : (i == j ? 1.0 : 0.0);
            }
        __syncthreads();
        gj_wg128(v, dim, tid, rc, pivs);
        if (tid < 64) {
            double lp = 0.0;
            for (int k = tid; k < dim; k += 64) {
                const double piv = pivs[k];
                if (!(piv > 0.0)) atomicOr(a.status, 1);
                lp += log(piv);
            }
            lp = wave_sum(lp);
            if (tid == 0) {
                a.lnd[(size_t)n * 4 + 2 + c] = lp;                         // ln det sym(qw)
                a.lnd[(size_t)n * 4 + c] = dim * log(qv) - lp;             // ln det E[Lambda]
            }
        }
#pragma unroll
        for (int ra = 0; ra < 8; ++ra)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int i = 8 * ta + ra, j = 8 * tb + cb;
                if (i < dim && j < dim) out[(size_t)i * dim + j] = qv * v[ra][cb];
            }
        __syncthreads();
    }
}

// ---- QA = E[Q]<A>, RC = E[R]<C>, trA[i] = tr(S_i E[Q]), trC[i] = tr(S'_i E[R])
__global__ void __launch_bounds__(256) k_dense_pre_big(WBArgs a) {
    const int WHICH = blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* out = (WHICH == 0 ? a.QA : a.RC) + (size_t)n * rows * D;
    for (int idx = tid; idx < rows * D; idx += 256) {
        const int k = idx / D, j = idx % D;
        double s = 0.0;
        for (int l = 0; l < rows; ++l) s += Lbar[(size_t)k * rows + l] * M[(size_t)l * D + j];
        out[idx] = s;
    }
    // traces: a wavefront per column, a lane per element of an 8 x 8 tile; off-diagonal tiles count twice (both symmetric)
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    const int wv = tid >> 6, lane = tid & 63, ra = lane >> 3, cb = lane & 7;
    const int RT = (rows + 7) >> 3;
    for (int i = wv; i < D; i += 4) {
        const double* Si = cov + (size_t)i * cov_stride(rows);
        double s = 0.0;
        int t = 0;
        for (int A = 0; A < RT; ++A)
            for (int B = A; B < RT; ++B, ++t) {
                const int k = 8 * A + ra, l = 8 * B + cb;
                if (k < rows && l < rows) s += (A == B ? 1.0 : 2.0) * Si[(size_t)t * 64 + lane] * Lbar[(size_t)k * rows + l];
            }
        s = wave_sum(s);
        if (lane == 0) (WHICH == 0 ? a.trA : a.trC)[(size_t)n * D + i] = s;
    }
}

// ---- [a.update() for a in As[c0:c1]] (or Cs): one workgroup per (replicate, matrix), the columns in order.
//   precision_i = prior_prec_i + G[i,i] E[Lambda]                      (hstack.pass_up_m1_m2 m1, nodes_todo.py:56; gaussian.py:117)
//   r = H[:,i] - sum_{j != i} <m_j> G[i,j],  w = prior_prec_i prior_mean_i + E[Lambda] r,  qmu_i = qcov_i w    (:59-61, :122-123)
// <M> is updated in place in global memory (column i sees the new columns before it); sum_i G[i,i] S_i is accumulated in
// a.SG, every thread adding to the 64 elements of its own tile.
__global__ void __launch_bounds__(256) k_cols_wishart_big(WBArgs a) {
    __shared__ double rc[2 * GJB_BUF], pivs[128], rv[128], wv[128];
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int ta = tid >> 4, tb = tid & 15;
    const int rows = WHICH == 0 ? D : K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    double* var = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    double* SG = a.SG + ((size_t)n * 2 + WHICH) * 128 * 128;
    const int RT = (rows + 7) >> 3;
    const bool whole = a.c0 == 0 && a.c1 == D;      // the sum is complete (and used) only when every column goes through
    if (whole)
        for (int idx = tid; idx < rows * rows; idx += 256) SG[idx] = 0.0;
    __syncthreads();
    for (int i = a.c0; i < a.c1; ++i) {
        const double g = G[(size_t)i * D + i];
        double v[8][8];
#pragma unroll
        for (int ra = 0; ra < 8; ++ra) {
            const int k = 8 * ta + ra;
            const double pk = k < rows ? pp[(size_t)i * rows + k] : 0.0;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int l = 8 * tb + cb;
                v[ra][cb] = (k < rows && l < rows) ? __builtin_fma(g, Lbar[(size_t)l * rows + k], (k == l) ? pk : 0.0) : ((k == l) ? 1.0 : 0.0);
            }
        }
        gj_wg128(v, rows, tid, rc, pivs);
        if (tid < 64) {
            double lp = 0.0;
            for (int k = tid; k < rows; k += 64) {
                const double piv = pivs[k];
                if (!(piv > 0.0)) atomicOr(a.status, 1);
                lp += log(piv);
            }
            lp = wave_sum(lp);
            if (tid == 0) qld[i] = 0.5 / (0.5 * lp);            // q_ln_det, gaussian.py:120 (quirk Q1)
        }
        // the covariance: upper tiles (cov_pos), the diagonal for the lower bound, and its share of sum_i G_ii S_i
        if (ta <= tb && tb < RT) {
            double* ci = cov + (size_t)i * cov_stride(rows) + (size_t)(ta * RT - (ta * (ta - 1)) / 2 + (tb - ta)) * 64;
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) ci[8 * ra + cb] = v[ra][cb];
        }
        if (ta == tb) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) if (8 * ta + ra < rows) var[(size_t)i * rows + 8 * ta + ra] = v[ra][ra];
        }
        if (whole) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int k = 8 * ta + ra, l = 8 * tb + cb;
                    if (k < rows && l < rows) SG[(size_t)k * rows + l] = __builtin_fma(g, v[ra][cb], SG[(size_t)k * rows + l]);
                }
        }
        // the mean: r, w (a thread per row), then qmu = qcov w out of the tiles
        if (tid < 128) {
            double r = 0.0;
            if (tid < rows) {
                r = H[(size_t)tid * D + i];
                for (int j = 0; j < D; ++j) r = __builtin_fma(-M[(size_t)tid * D + j], (j != i) ? G[(size_t)i * D + j] : 0.0, r);
            }
            rv[tid] = r;
        }
        __syncthreads();
        if (tid < 128) {
            double w = 0.0;
            if (tid < rows) {
                w = pp[(size_t)i * rows + tid] * pm[(size_t)tid * D + i];
                for (int l = 0; l < rows; ++l) w = __builtin_fma(Lbar[(size_t)tid * rows + l], rv[l], w);
            }
            wv[tid] = w;
        }
        __syncthreads();
        double ws[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) ws[cb] = wv[8 * tb + cb];
#pragma unroll
        for (int ra = 0; ra < 8; ++ra) {
            double part = 0.0;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) part = __builtin_fma(v[ra][cb], ws[cb], part);
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 8, 64);
            if (tb == 0 && 8 * ta + ra < rows) M[(size_t)(8 * ta + ra) * D + i] = part;
        }
        __threadfence_block();
        __syncthreads();            // the new column is in memory before the next one reads the means
    }
}

// ---- Wishart.update for Q (which 0) or R (1): Rm = 1/2 (own + <M> G <M>^T + sum_i S_i G[i,i]) - H <M>^T, qw = w0 + Rm
__global__ void __launch_bounds__(256) k_wresid_big(WBArgs a) {
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K, T = a.T, DP = a.DP;
    const int rows = WHICH == 0 ? D : K;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    double* Rm = (WHICH == 0 ? a.RQ : a.RR) + (size_t)n * rows * rows;
    double* T1 = a.scratch + ((size_t)n * 2 + WHICH) * 128 * 128;       // <M> G  [rows][D] (k_prep_big's scratch: not in use now)
    const double* SG = a.SG + ((size_t)n * 2 + WHICH) * 128 * 128;
    for (int idx = tid; idx < rows * D; idx += 256) {
        const int k = idx / D, j = idx % D;
        double s = 0.0;
        for (int i = 0; i < D; ++i) s = __builtin_fma(M[(size_t)k * D + i], G[(size_t)i * D + j], s);
        T1[idx] = s;
    }
    __threadfence_block();
    __syncthreads();
    const double* x0 = a.X + (size_t)n * T * DP;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* GC = mo + MOM_GC(D, K);
    for (int idx = tid; idx < rows * rows; idx += 256) {
        const int k = idx / rows, l = idx % rows;
        double e = 0.0, hm = 0.0;
        for (int j = 0; j < D; ++j) {
            const double ml = M[(size_t)l * D + j];
            e = __builtin_fma(T1[(size_t)k * D + j], ml, e);
            hm = __builtin_fma(H[(size_t)k * D + j], ml, hm);
        }
        if (a.use_sg) e += SG[idx];
        else for (int i = 0; i < D; ++i) e += cov[(size_t)i * cov_stride(rows) + covb_pos(rows, k, l)] * G[(size_t)i * D + i];
        double own;
        if (WHICH == 0) own = GC[idx] - x0[xpos(k)] * x0[xpos(l)] - S0[idx];          // sum_{t >= 1} <x x^T>
        else own = a.SyyF[(size_t)n * K * K + idx];
        const double rr = 0.5 * (own + e) - hm;
        Rm[idx] = rr;
        if (a.update) {
            const double* w0 = WHICH == 0 ? a.pri.Q_w0 : a.pri.R_w0;
            (WHICH == 0 ? a.Q_w : a.R_w)[(size_t)n * rows * rows + idx] = w0[idx] + rr;
        }
    }
}

// ---- sum_t y_t y_t^T, once per set_observations
__global__ void __launch_bounds__(256) k_syy_full_big(WBArgs a) {
    const int n = blockIdx.x, tid = threadIdx.x, K = a.K, T = a.T;
    const double* Y = a.Y + (size_t)n * T * K;
    for (int idx = tid; idx < K * K; idx += 256) {
        const int k = idx / K, l = idx % K;
        double s = 0.0;
        for (int t = 0; t < T; ++t) s = __builtin_fma(Y[(size_t)t * K + k], Y[(size_t)t * K + l], s);
        a.SyyF[(size_t)n * K * K + idx] = s;
    }
}

int launch_wexpect_big(pyvb_lds* h) {
    WBArgs a = make_wbargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_wexpect_big, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_dense_pre_big(pyvb_lds* h) {
    WBArgs a = make_wbargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_dense_pre_big, dim3(h->N, 2), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_cols_dense_big(pyvb_lds* h, int which, int c0, int c1) {
    WBArgs a = make_wbargs(h);
    a.which0 = which == 1 ? 1 : 0; a.c0 = c0; a.c1 = c1;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_cols_wishart_big, dim3(h->N, which == 2 ? 2 : 1), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    const bool all = c0 == 0 && c1 == h->D;
    if (which == 0 || which == 2) h->sg_valid[0] = all;
    if (which == 1 || which == 2) h->sg_valid[1] = all;
    return PYVB_OK;
}

int launch_wresid_big(pyvb_lds* h, int which, int update) {
    WBArgs a = make_wbargs(h);
    a.which0 = which == 1 ? 1 : 0; a.update = update;
    const bool need0 = which != 1, need1 = which != 0;
    a.use_sg = ((!need0 || h->sg_valid[0]) && (!need1 || h->sg_valid[1])) ? 1 : 0;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_wresid_big, dim3(h->N, which == 2 ? 2 : 1), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_syy_full_big(pyvb_lds* h) {
    WBArgs a = make_wbargs(h);
    hipLaunchKernelGGL(k_syy_full_big, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
