// Wishart noise precisions for the LDS graph (Linear_Dynamic_System.py:55-56; nodes_todo.py:205-234):
//   Q = Wishart(D, v0, w0), R = Wishart(K, v0, w0)   E[Lambda] = qv * inv(qw)   (pass_down_Ex, :233-234)
//   qv = v0 + 1/2 per child (:224-227)
//   qw = w0 + sum over children of 1/2<x x^T> + 1/2<mu mu^T> - <x><mu>^T   (update, :228-231)
// The expected precisions are dense, so the column Gaussians of A and C get dense posterior covariances
// (D inversions of a D x D matrix per replicate and matrix) and everything that was a row scaling becomes a product.
//
// Where this departs from the reference (its Wishart is unfinished; the parity target is the state
// after the first full iteration, tests/golden/lds_wishart_d3k4_t40.npz):
//   * update() there starts from `self.qw = self.w0` and adds in place, so the PRIOR grows with every call (Q7); here
//     w0 is a constant.
//   * -<x><mu>^T is not symmetrised there, so qw is not symmetric.  qw is stored as the reference computes it; its
//     expectation uses the symmetric part, E[Lambda] = qv * inv((qw + qw^T)/2).  (Identical for the first update.)
//   * it has no pass_down_lndet / log_lower_bound (Q8).  Here: ln det E[Lambda] like the Gamma nodes (quirk Q2), and
//     the lower bound of a Wishart in the (a, B) form that reduces to Gamma.log_lower_bound (:149-157) for D = 1:
//       E ln p - E ln q,  ln p = (a0 - (D+1)/2) ln|L| - ln Gamma_D(a0) + a0 ln|B0| - tr(B0 L)   (parity unpinned).
#include "params.h"
#include "gj.h"

struct WArgs {
    double *Q_w, *R_w, *Qbar, *Rbar, *lnd, *QA, *RC, *trA, *trC, *A_cov, *C_cov, *RQ, *RR, *SyyF;
    const double *Q_a, *R_a;
    double *A_mean, *A_var, *C_mean, *C_var, *qld_A, *qld_C;
    const double *mom, *X, *Sigma, *Y, *qld_x;
    double* elbo;
    Priors pri;
    int* status;
    int N, T, D, K, DP;
    int which0, c0, c1, update;
};

static WArgs make_wargs(pyvb_lds* h) {
    WArgs a;
    a.Q_w = h->Q_w; a.R_w = h->R_w; a.Qbar = h->Qbar; a.Rbar = h->Rbar; a.lnd = h->lnd; a.QA = h->QA; a.RC = h->RC;
    a.trA = h->trA; a.trC = h->trC; a.A_cov = h->A_cov; a.C_cov = h->C_cov; a.RQ = h->RQ; a.RR = h->RR; a.SyyF = h->SyyF;
    a.Q_a = h->Q_a; a.R_a = h->R_a;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var; a.qld_A = h->qld_A; a.qld_C = h->qld_C;
    a.mom = h->mom; a.X = h->X[h->cur]; a.Sigma = h->Sigma; a.Y = h->Y; a.qld_x = h->qld_x; a.elbo = h->elbo;
    a.pri = h->pri; a.status = h->status;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.DP = h->L.DP;
    a.which0 = 0; a.c0 = 0; a.c1 = h->D; a.update = 0;
    return a;
}

// ---- E[Q], E[R] and their log-determinants: one workgroup per replicate inverts both symmetrised qw at once
__global__ void __launch_bounds__(256) k_wexpect(WArgs a) {
    __shared__ double gjbuf[2 * 2 * GJ_BUF + 128];
    const int n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const double* Qw = a.Q_w + (size_t)n * D * D;
    const double* Rw = a.R_w + (size_t)n * K * K;
    const int ta = tid >> 4, tb = tid & 15;
    double v[2][16];
#pragma unroll
    for (int ra = 0; ra < 4; ++ra)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int i = 4 * ta + ra, j = 4 * tb + cb, u = 4 * ra + cb;
            const double pad = (i == j) ? 1.0 : 0.0;
            v[0][u] = (i < D && j < D) ? 0.5 * (Qw[i * D + j] + Qw[j * D + i]) : pad;
            v[1][u] = (i < K && j < K) ? 0.5 * (Rw[i * K + j] + Rw[j * K + i]) : pad;
        }
    __syncthreads();
    gj_inverse<2>(v, D > K ? D : K, tid, gjbuf, gjbuf + 2 * 2 * GJ_BUF);
    const double qv = a.Q_a[(size_t)n * D], rv = a.R_a[(size_t)n * K];
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int dim = c == 0 ? D : K;
            double lp = 0.0;
            if (tid < dim) {
                const double piv = gjbuf[2 * 2 * GJ_BUF + c * 64 + tid];
                if (!(piv > 0.0)) atomicOr(a.status, 1);
                lp = log(piv);
            }
            lp = wave_sum(lp);
            if (tid == 0) {
                a.lnd[(size_t)n * 4 + 2 + c] = lp;                                     // ln det sym(qw)
                a.lnd[(size_t)n * 4 + c] = dim * log(c == 0 ? qv : rv) - lp;           // ln det E[Lambda]
            }
        }
    }
#pragma unroll
    for (int ra = 0; ra < 4; ++ra)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int i = 4 * ta + ra, j = 4 * tb + cb, u = 4 * ra + cb;
            if (i < D && j < D) a.Qbar[(size_t)n * D * D + i * D + j] = qv * v[0][u];
            if (i < K && j < K) a.Rbar[(size_t)n * K * K + i * K + j] = rv * v[1][u];
        }
}

// ---- QA = E[Q]<A>, RC = E[R]<C>, trA[i] = tr(S_i E[Q]), trC[i] = tr(S'_i E[R])      (the Wishart versions of the
// row scalings and of the diagonal of node.py:223-227 in k_prep)
#define WLD 65
__global__ void __launch_bounds__(256) k_dense_pre(WArgs a) {
    __shared__ double Lb[64 * WLD], Mb[64 * WLD];
    const int WHICH = blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* out = (WHICH == 0 ? a.QA : a.RC) + (size_t)n * rows * D;
    for (int idx = tid; idx < rows * rows; idx += 256) Lb[(idx / rows) * WLD + idx % rows] = Lbar[idx];
    for (int idx = tid; idx < rows * D; idx += 256) Mb[(idx / D) * WLD + idx % D] = M[idx];
    __syncthreads();
    for (int idx = tid; idx < rows * D; idx += 256) {
        const int k = idx / D, j = idx % D;
        double s = 0.0;
        for (int l = 0; l < rows; ++l) s += Lb[k * WLD + l] * Mb[l * WLD + j];
        out[idx] = s;
    }
    // traces: four threads per column, a quarter of the rows each
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * rows * rows;
    const int i = tid >> 2, part = tid & 3;
    double s = 0.0;
    if (i < D) {
        const double* Si = cov + (size_t)i * rows * rows;
        for (int k = part; k < rows; k += 4) {
#pragma unroll 16
            for (int l = 0; l < rows; ++l) s += Si[k * rows + l] * Lb[k * WLD + l];      // both symmetric
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (i < D && part == 0) (WHICH == 0 ? a.trA : a.trC)[(size_t)n * D + i] = s;
}

// ---- posterior covariances of the columns: qprec_i = diag(prior) + E[Lambda] * sum_t <x x^T>[i,i]   (gaussian.py:117 with
// m1 of hstack.pass_up_m1_m2, nodes_todo.py:56), qcov = inverse, q_ln_det (quirk Q1).  The precisions do not depend on
// the other columns: four columns per workgroup, all of them in parallel.
__global__ void __launch_bounds__(256) k_colcov(WArgs a) {
    __shared__ double gjbuf[4 * (2 * GJW_BUF + 64)];
    const int WHICH = a.which0 + blockIdx.z, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int wv = tid >> 6, lane = tid & 63, i = 4 * blockIdx.y + wv;      // one wavefront per column (gj_wave)
    const int rows = WHICH == 0 ? D : K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;        // [col][row]
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * rows * rows;
    double* var = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    if (!(i < D && i >= a.c0 && i < a.c1)) return;                  // wave-uniform; no workgroup barrier below
    double* rc = gjbuf + wv * (2 * GJW_BUF + 64);
    double* pivs = rc + 2 * GJW_BUF;
    const int ta = lane >> 3, tb = lane & 7;
    const double g = G[(size_t)i * D + i];
    double v[8][8];
    // loads are unconditional (indices clamped into the matrix, the select afterwards): 64 guarded loads would each wait
    // for the one before
#pragma unroll
    for (int ra = 0; ra < 8; ++ra) {
        const int k = 8 * ta + ra, kc = k < rows ? k : rows - 1;
        const double pk = pp[(size_t)i * rows + kc];
        double x[8];
        if (rows == 64) {
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) {
                const d2 t = *reinterpret_cast<const d2*>(Lbar + k * 64 + 8 * tb + 2 * h2);
                x[2 * h2] = t[0]; x[2 * h2 + 1] = t[1];
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) { const int l = 8 * tb + cb; x[cb] = Lbar[kc * rows + (l < rows ? l : rows - 1)]; }
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const int l = 8 * tb + cb;
            const double inside = __builtin_fma(g, x[cb], (k == l) ? pk : 0.0);
            v[ra][cb] = (k < rows && l < rows) ? inside : ((k == l) ? 1.0 : 0.0);
        }
    }
    gj_wave(v, rows, lane, rc, pivs);
    double lp = 0.0;
    if (lane < rows) {
        const double piv = pivs[lane];
        if (!(piv > 0.0)) atomicOr(a.status, 1);
        lp = log(piv);
    }
    lp = wave_sum(lp);
    if (lane == 0) qld[i] = 0.5 / (0.5 * lp);
    double* ci_ = cov + (size_t)i * rows * rows;
#pragma unroll
    for (int ra = 0; ra < 8; ++ra) {
        const int k = 8 * ta + ra;
        if (rows == 64) {
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) *reinterpret_cast<d2*>(ci_ + k * 64 + 8 * tb + 2 * h2) = d2{v[ra][2 * h2], v[ra][2 * h2 + 1]};
        } else if (k < rows) {
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) { const int l = 8 * tb + cb; if (l < rows) ci_[k * rows + l] = v[ra][cb]; }
        }
        if (ta == tb && k < rows) var[(size_t)i * rows + k] = v[ra][ra];
    }
}

// ---- posterior means of the columns, Gauss-Seidel over the columns (column i sees the new columns 0..i-1):
//   qmu_i = qcov_i ( prior_prec_i prior_mean_i + E[Lambda] ( H[:,i] - sum_{j != i} <m_j> G[i,j] ) )
// (gaussian.py:122-123 with m2 of hstack.pass_up_m1_m2, nodes_todo.py:59-61).  Lane = row.
__global__ void __launch_bounds__(64) k_colmean(WArgs a) {
    __shared__ double Mb[64 * WLD], Lb[64 * WLD], gv[64], rv[64], wv[64];
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, lane = threadIdx.x, D = a.D, K = a.K;
    const int rows = WHICH == 0 ? D : K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * rows * rows;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    const bool live = lane < rows;
    for (int idx = lane; idx < rows * D; idx += 64) Mb[(idx % D) * WLD + idx / D] = M[idx];             // Mb[col][row]
    for (int idx = lane; idx < rows * rows; idx += 64) Lb[(idx / rows) * WLD + idx % rows] = Lbar[idx];
    __syncthreads();
    for (int i = a.c0; i < a.c1; ++i) {
        if (lane < D) gv[lane] = G[(size_t)i * D + lane];
        __syncthreads();
        double r = 0.0;
        if (live) {
            r = H[(size_t)lane * D + i];
#pragma unroll 8
            for (int j = 0; j < D; ++j) r -= (j != i) ? Mb[j * WLD + lane] * gv[j] : 0.0;
        }
        rv[lane] = r;
        __syncthreads();
        double w = 0.0;
        if (live) {
            w = pp[(size_t)i * rows + lane] * pm[(size_t)lane * D + i];
#pragma unroll 8
            for (int l = 0; l < rows; ++l) w += Lb[lane * WLD + l] * rv[l];
        }
        wv[lane] = w;
        __syncthreads();
        if (live) {
            const double* Si = cov + (size_t)i * rows * rows;
            double mu = 0.0;
#pragma unroll 16
            for (int l = 0; l < rows; ++l) mu += Si[(size_t)l * rows + lane] * wv[l];      // symmetric: read down a column (16 loads in flight)
            Mb[i * WLD + lane] = mu;
        }
        __syncthreads();
    }
    for (int idx = lane; idx < rows * D; idx += 64) {
        const int i = idx % D;
        if (i >= a.c0 && i < a.c1) M[idx] = Mb[i * WLD + idx / D];
    }
}

// ---- Wishart.update (nodes_todo.py:228-231) for Q (children X_1.., mean parents Mult(A, X_{t-1})) or R (children Y_t):
//   Rm = 1/2 ( own + <M> G <M>^T + sum_i S_i G[i,i] ) - H <M>^T        (Multiplication.pass_down_ExxT node.py:260-271)
//   qw = w0 + Rm
// own = sum_t <x x^T> over the children (Q: t >= 1) or sum_t y y^T (R).  Rm is kept for the lower bound.
__global__ void __launch_bounds__(256) k_wresid(WArgs a) {
    __shared__ double Mb[64 * WLD], T1[64 * WLD];
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K, T = a.T, DP = a.DP;
    const int rows = WHICH == 0 ? D : K;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * rows * rows;
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    double* Rm = (WHICH == 0 ? a.RQ : a.RR) + (size_t)n * rows * rows;
    for (int idx = tid; idx < rows * D; idx += 256) Mb[(idx / D) * WLD + idx % D] = M[idx];
    __syncthreads();
    for (int idx = tid; idx < rows * D; idx += 256) {       // T1 = <M> G
        const int k = idx / D, j = idx % D;
        double s = 0.0;
#pragma unroll 16
        for (int i = 0; i < D; ++i) s += Mb[k * WLD + i] * G[(size_t)i * D + j];
        T1[k * WLD + j] = s;
    }
    __syncthreads();
    const double* x0 = a.X + (size_t)n * T * DP;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* GC = mo + MOM_GC(D, K);
    for (int idx = tid; idx < rows * rows; idx += 256) {
        const int k = idx / rows, l = idx % rows;
        double e = 0.0, hm = 0.0;
#pragma unroll 16
        for (int j = 0; j < D; ++j) { e += T1[k * WLD + j] * Mb[l * WLD + j]; hm += H[(size_t)k * D + j] * Mb[l * WLD + j]; }
#pragma unroll 16
        for (int i = 0; i < D; ++i) e += cov[(size_t)i * rows * rows + idx] * G[(size_t)i * D + i];
        double own;
        if (WHICH == 0) own = GC[idx] - x0[xpos(k)] * x0[xpos(l)] - S0[idx];          // sum_{t >= 1} <x x^T>
        else own = a.SyyF[(size_t)n * K * K + idx];
        const double r = 0.5 * (own + e) - hm;
        Rm[idx] = r;
        if (a.update) {
            const double* w0 = WHICH == 0 ? a.pri.Q_w0 : a.pri.R_w0;
            (WHICH == 0 ? a.Q_w : a.R_w)[(size_t)n * rows * rows + idx] = w0[idx] + r;
        }
    }
}

// ---- sum_t y_t y_t^T, once per set_observations
__global__ void __launch_bounds__(256) k_syy_full(WArgs a) {
    __shared__ double Yb[64 * WLD];
    const int n = blockIdx.x, tid = threadIdx.x, K = a.K, T = a.T;
    const double* Y = a.Y + (size_t)n * T * K;
    double acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int nt = T - t0 < 64 ? T - t0 : 64;
        __syncthreads();
        for (int idx = tid; idx < nt * K; idx += 256) Yb[(idx / K) * WLD + idx % K] = Y[(size_t)t0 * K + idx];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + 256 * u;
            if (idx < K * K) {
                const int k = idx / K, l = idx % K;
                double s = 0.0;
                for (int t = 0; t < nt; ++t) s += Yb[t * WLD + k] * Yb[t * WLD + l];
                acc[u] += s;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = tid + 256 * u;
        if (idx < K * K) a.SyyF[(size_t)n * K * K + idx] = acc[u];
    }
}

// ---- initial column covariances: diagonal (gaussian.py:70-72 draws isotropic ones)
__global__ void __launch_bounds__(256) k_colvar_to_cov(WArgs a) {
    const int WHICH = blockIdx.z, n = blockIdx.x, i = blockIdx.y, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + ((size_t)n * D + i) * rows * rows;
    const double* var = (WHICH == 0 ? a.A_var : a.C_var) + ((size_t)n * D + i) * rows;
    for (int idx = threadIdx.x; idx < rows * rows; idx += 256) cov[idx] = (idx / rows == idx % rows) ? var[idx / rows] : 0.0;
}

// ... and back: the diagonals (what the lower bound reads) of column covariances the caller supplied
__global__ void __launch_bounds__(64) k_cov_to_colvar(WArgs a) {
    const int WHICH = blockIdx.z, n = blockIdx.x, i = blockIdx.y, D = a.D, k = threadIdx.x;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + ((size_t)n * D + i) * rows * rows;
    double* var = (WHICH == 0 ? a.A_var : a.C_var) + ((size_t)n * D + i) * rows;
    if (k < rows) var[k] = cov[(size_t)k * rows + k];
}

__device__ static double psi_multi(double x, int D) {       // sum_{i<D} psi(x - i/2)
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += digamma_pos(x - 0.5 * i);
    return s;
}
__device__ static double lgamma_multi(double x, int D) {    // ln |Gamma_D(x)|
    double s = 0.25 * D * (D - 1) * 1.1447298858494002;     // ln pi
    for (int i = 0; i < D; ++i) s += lgamma(x - 0.5 * i);
    return s;
}

// ---- lower bound with Wishart noise: the six class sums as k_elbo (k_params.hip), traces against dense expectations
__global__ void __launch_bounds__(64) k_elbo_dense(WArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D, K = a.K, T = a.T;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* x0 = a.X + (size_t)n * T * a.DP;
    const double* qx = a.qld_x + (size_t)n * 3;
    const double* Qb = a.Qbar + (size_t)n * D * D;
    const double* Rb = a.Rbar + (size_t)n * K * K;
    const double* RQ = a.RQ + (size_t)n * D * D;
    const double* RR = a.RR + (size_t)n * K * K;
    const double* ln = a.lnd + (size_t)n * 4;
    double tq = 0.0, tr = 0.0, tq0 = 0.0, tr0 = 0.0;
    if (lane < D) for (int l = 0; l < D; ++l) { tq += Qb[lane * D + l] * RQ[l * D + lane]; tq0 += a.pri.Q_w0[lane * D + l] * Qb[l * D + lane]; }
    if (lane < K) for (int l = 0; l < K; ++l) { tr += Rb[lane * K + l] * RR[l * K + lane]; tr0 += a.pri.R_w0[lane * K + l] * Rb[l * K + lane]; }
    const double trQ = wave_sum(tq), trR = wave_sum(tr), trQ0 = wave_sum(tq0), trR0 = wave_sum(tr0);
    double e0 = 0.0;
    if (lane < D) {
        const int i = lane;
        for (int j = 0; j < D; ++j) {
            const double xj = x0[xpos(j)];
            const double ex = xj * x0[xpos(i)] + S0[j * D + i] + a.pri.x0_mean[j] * a.pri.x0_mean[i] - 2.0 * xj * a.pri.x0_mean[i];
            e0 += a.pri.x0_prec[i * D + j] * ex;
        }
    }
    e0 = wave_sum(e0);
    const double nint = (double)(T - 2);
    double LX = -0.5 * D * LN2PI + 0.5 * a.pri.x0_lndet - 0.5 * e0;
    LX += (double)(T - 1) * (-0.5 * D * LN2PI + 0.5 * ln[0]) - trQ;
    LX += (double)T * (0.5 * D * LN2PI + 0.5 * D) + 0.5 * (qx[0] + nint * qx[1] + qx[2]);
    const double LY = (double)T * (-0.5 * K * LN2PI + 0.5 * ln[1]) - trR;
    double la = 0.0, lc = 0.0;
    if (lane < D) {
        const int i = lane;
        auto column = [&](int rows, const double* pp, const double* pm, const double* M, const double* V, double qld, double lndet) {
            double trc = 0.0;
            for (int k = 0; k < rows; ++k) {
                const double m = M[(size_t)k * D + i], m0 = pm[(size_t)k * D + i];
                trc += pp[(size_t)i * rows + k] * (m * m + V[(size_t)i * rows + k] + m0 * m0 - 2.0 * m * m0);
            }
            return -0.5 * rows * LN2PI + 0.5 * lndet - 0.5 * trc + 0.5 * rows * LN2PI + 0.5 * qld + 0.5 * rows;
        };
        la = column(D, a.pri.A_pp, a.pri.A_pm, a.A_mean + (size_t)n * D * D, a.A_var + (size_t)n * D * D, a.qld_A[(size_t)n * D + i], a.pri.A_pld[i]);
        lc = column(K, a.pri.C_pp, a.pri.C_pm, a.C_mean + (size_t)n * K * D, a.C_var + (size_t)n * D * K, a.qld_C[(size_t)n * D + i], a.pri.C_pld[i]);
    }
    const double LA = wave_sum(la), LC = wave_sum(lc);
    if (lane == 0) {
        auto wishart_llb = [&](int dim, double a0, double qa, double lndw0, double lndw, double tr0) {
            const double Eln = psi_multi(qa, dim) - lndw;          // E ln det Lambda
            const double half = 0.5 * (dim + 1);
            double ret = (a0 - half) * Eln - lgamma_multi(a0, dim) + a0 * lndw0 - tr0;
            ret -= (qa - half) * Eln - lgamma_multi(qa, dim) + qa * lndw - qa * dim;
            return ret;
        };
        double* o = a.elbo + (size_t)n * 6;
        o[0] = LX; o[1] = LY; o[2] = LA; o[3] = LC;
        o[4] = wishart_llb(D, a.pri.Q_a0_host, a.Q_a[(size_t)n * D], a.pri.Q_w0_lndet, ln[2], trQ0);
        o[5] = wishart_llb(K, a.pri.R_a0_host, a.R_a[(size_t)n * K], a.pri.R_w0_lndet, ln[3], trR0);
    }
}

// ---- launchers
int launch_wexpect(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_wexpect, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_dense_pre(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_dense_pre, dim3(h->N, 2), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_cols_dense(pyvb_lds* h, int which, int c0, int c1) {
    WArgs a = make_wargs(h);
    a.which0 = which == 1 ? 1 : 0; a.c0 = c0; a.c1 = c1;
    const int nw = which == 2 ? 2 : 1;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_colcov, dim3(h->N, (h->D + 3) / 4, nw), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(k_colmean, dim3(h->N, nw), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_wresid(pyvb_lds* h, int which, int update) {
    WArgs a = make_wargs(h);
    a.which0 = which == 1 ? 1 : 0; a.update = update;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_wresid, dim3(h->N, which == 2 ? 2 : 1), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_syy_full(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    hipLaunchKernelGGL(k_syy_full, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_colvar_to_cov(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    hipLaunchKernelGGL(k_colvar_to_cov, dim3(h->N, h->D, 2), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_cov_to_colvar(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    hipLaunchKernelGGL(k_cov_to_colvar, dim3(h->N, h->D, 2), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo_dense(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    TimedLaunch tl(h, PYVB_K_ELBO);
    hipLaunchKernelGGL(k_elbo_dense, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
