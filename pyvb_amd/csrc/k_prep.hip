// k_prep: everything that depends only on the parameter posteriors and is shared by all
// X_t of one replicate: <A^T Q A>, <C^T R C>, the three posterior precisions of the states,
// their Cholesky factors / inverses / q_ln_det, the gain matrices, and the warm-up length
// of the segmented sweeps.
//
// Reference work replaced (per X_t.update(), 2T times per iteration in the reference):
//   Multiplication.pass_up_m1_m2, hstack branch    node.py:213-227   (<A^T Q A>, the D^4 tensor)
//   qprec = pprec + sum m1 ; cho_factor ; cho_solve gaussian.py:117-119
//   q_ln_det (quirk Q1)                             gaussian.py:120
// One workgroup per replicate; all D x D work stays in LDS.
#include "common.h"

struct PrepArgs {
    const double *A_mean, *A_var, *C_mean, *C_var, *Q_a, *Q_b, *R_a, *R_b, *x0_mean, *x0_prec;
    double *Sigma, *qld, *gains, *scratch;
    int *warm, *status;
    int N, T, D, K;
    Layout L;
};

#define PREP_THREADS 256

__device__ static void chol_lower(double* W, int D, int tid, int* status, int n) {
    for (int j = 0; j < D; ++j) {
        if (tid == 0) {
            double piv = W[j * D + j];
            if (!(piv > 0.0)) atomicOr(status, 1);
            W[j * D + j] = sqrt(piv);
        }
        __syncthreads();
        double d = W[j * D + j];
        for (int i = j + 1 + tid; i < D; i += PREP_THREADS) W[i * D + j] /= d;
        __syncthreads();
        int rem = D - j - 1;
        for (int idx = tid; idx < rem * rem; idx += PREP_THREADS) {
            int i = j + 1 + idx / rem, k = j + 1 + idx % rem;
            if (k <= i) W[i * D + k] -= W[i * D + j] * W[k * D + j];
        }
        __syncthreads();
    }
}

// Z = L^{-1} (lower triangular), one thread per column
__device__ static void tri_inverse(const double* Lw, double* Z, int D, int tid) {
    if (tid < D) {
        int j = tid;
        for (int i = 0; i < j; ++i) Z[i * D + j] = 0.0;
        Z[j * D + j] = 1.0 / Lw[j * D + j];
        for (int i = j + 1; i < D; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += Lw[i * D + k] * Z[k * D + j];
            Z[i * D + j] = -s / Lw[i * D + i];
        }
    }
    __syncthreads();
}

// position of matrix element (i, j) in an MFMA A-operand block with S k-steps
__device__ __forceinline__ size_t pos_nat(int i, int j, int S) {
    return ((size_t)((i >> 4) * S + (j >> 2)) * 64) + (j & 3) * 16 + (i & 15);
}
__device__ __forceinline__ size_t pos_perm(int i, int j, int S) {
    int s = 2 * (j >> 3) + (j & 1), q = (j & 7) >> 1;
    return ((size_t)((i >> 4) * S + s) * 64) + q * 16 + (i & 15);
}

// C = A * A for D x D matrices in LDS
__device__ static void mat_square(const double* A, double* C, int D, int tid) {
    for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
        int i = idx / D, j = idx % D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += A[i * D + k] * A[k * D + j];
        C[idx] = s;
    }
    __syncthreads();
}

__device__ static double inf_norm(const double* A, int D, int tid, double* red) {
    if (tid < D) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += fabs(A[tid * D + j]);
        red[tid] = s;
    }
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < D; ++i) m = fmax(m, red[i]);
    __syncthreads();
    return m;
}

// Number of recurrence steps after which the influence of the starting state of
// x_t = M x_{t-1} + c_t is below 1e-18 relative: ||M^J|| <= ||M^(2^k)||^(J/2^k).
// M is in W1 on entry; W1/W2 are clobbered.
__device__ static int warmup_length(double* W1, double* W2, int D, int tid, double* red) {
    const double lntol = -41.4465316738928;   // ln(1e-18)
    int best = 1 << 30;
    mat_square(W1, W2, D, tid);   // M^2
    mat_square(W2, W1, D, tid);   // M^4
    double* src = W1; double* dst = W2;
    for (int k = 3; k <= 5; ++k) {
        mat_square(src, dst, D, tid);         // M^(2^k)
        double nrm = inf_norm(dst, D, tid, red);
        if (nrm < 1.0) {
            double steps = (nrm > 0.0) ? ceil(lntol / log(nrm)) : 1.0;
            double J = (double)(1 << k) * steps;
            if (J < (double)best) best = (int)J;
        }
        double* t = src; src = dst; dst = t;
    }
    return best;
}

template <int DMAX>
__global__ void __launch_bounds__(PREP_THREADS) k_prep(PrepArgs a) {
    __shared__ double sm[4 * DMAX * DMAX + 256];   // static: up to 133 KB of the CU's 160 KB at DMAX = 64
    const int n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const Layout& L = a.L;
    double* sA = sm;                 // [D][D]   <A> row-major
    double* W1 = sA + D * D;
    double* W2 = W1 + D * D;
    double* W3 = W2 + D * D;
    double* qbar = W3 + D * D;       // [64]
    double* rbar = qbar + 64;        // [64]
    double* red = rbar + 64;         // [64]
    double* vec = red + 64;          // [64]
    const double* Am = a.A_mean + (size_t)n * D * D;
    const double* Av = a.A_var + (size_t)n * D * D;
    const double* Cm = a.C_mean + (size_t)n * K * D;
    const double* Cv = a.C_var + (size_t)n * D * K;
    double* g = a.gains + (size_t)n * L.gains_total;
    double* sc = a.scratch + (size_t)n * 2 * D * D;

    if (tid < D) qbar[tid] = a.Q_a[(size_t)n * D + tid] / a.Q_b[(size_t)n * D + tid];
    if (tid < K) rbar[tid] = a.R_a[(size_t)n * K + tid] / a.R_b[(size_t)n * K + tid];
    for (int idx = tid; idx < D * D; idx += PREP_THREADS) sA[idx] = Am[idx];
    __syncthreads();

    // <C^T R C> and <A^T Q A> (node.py:213-227): mean part + trace of the column covariances on the diagonal
    for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
        int i = idx / D, j = idx % D;
        double mc = 0.0, ma = 0.0;
        for (int k = 0; k < K; ++k) mc += Cm[k * D + i] * rbar[k] * Cm[k * D + j];
        for (int k = 0; k < D; ++k) ma += sA[k * D + i] * qbar[k] * sA[k * D + j];
        if (i == j) {
            double tc = 0.0, ta = 0.0;
            for (int k = 0; k < K; ++k) tc += Cv[i * K + k] * rbar[k];
            for (int k = 0; k < D; ++k) ta += Av[i * D + k] * qbar[k];
            mc += tc; ma += ta;
        }
        sc[idx] = mc;
        sc[D * D + idx] = mc + ma;
    }
    __syncthreads();

    const int order[3] = {0, 2, 1};
    for (int oi = 0; oi < 3; ++oi) {
        const int cls = order[oi];
        // qprec = pprec + (m1 from Mult(C,.) + m1 from Mult(A,.))   gaussian.py:117
        for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
            int i = idx / D, j = idx % D;
            double base = (cls == 2) ? sc[idx] : sc[D * D + idx];
            double prior = (cls == 0) ? a.x0_prec[idx] : (i == j ? qbar[i] : 0.0);
            W1[idx] = prior + base;
        }
        __syncthreads();
        chol_lower(W1, D, tid, a.status, n);
        if (tid == 0) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += log(W1[j * D + j]);
            a.qld[(size_t)n * 3 + cls] = 0.5 / s;              // gaussian.py:120 (quirk Q1)
        }
        tri_inverse(W1, W2, D, tid);
        for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
            int i = idx / D, j = idx % D;
            int k0 = i > j ? i : j;
            double s = 0.0;
            for (int k = k0; k < D; ++k) s += W2[k * D + i] * W2[k * D + j];
            W3[idx] = s;                                        // qcov = qprec^{-1}   gaussian.py:119
            a.Sigma[((size_t)n * 3 + cls) * D * D + idx] = s;
        }
        __syncthreads();

        // gains of this class
        double* FT = g + (cls == 1 ? L.oFT : L.oFLT);
        double* BT = g + (cls == 1 ? L.oBT : L.oB0T);
        double* GT = g + (cls == 1 ? L.oGT : (cls == 0 ? L.oG0T : L.oGLT));
        for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
            int i = idx / D, j = idx % D;
            if (cls != 0) {         // Sigma <Q><A>: multiplies the mean of X_{t-1}
                double s = 0.0;
                for (int k = 0; k < D; ++k) s += W3[i * D + k] * (qbar[k] * sA[k * D + j]);
                FT[(size_t)j * L.DP + i] = s;
                if (cls == 1) { g[L.oFn + pos_nat(i, j, L.DS)] = s; g[L.oFp + pos_perm(i, j, L.DS)] = s; }
            }
            if (cls != 2) {         // Sigma <A>^T<Q>: multiplies the mean of X_{t+1}
                double s = 0.0;
                for (int k = 0; k < D; ++k) s += W3[i * D + k] * sA[j * D + k];
                s *= qbar[j];
                BT[(size_t)j * L.DP + i] = s;
                if (cls == 1) { g[L.oBn + pos_nat(i, j, L.DS)] = s; g[L.oBp + pos_perm(i, j, L.DS)] = s; }
            }
        }
        for (int idx = tid; idx < D * K; idx += PREP_THREADS) {
            int i = idx / K, l = idx % K;   // Sigma <C>^T<R>: multiplies y_t
            double s = 0.0;
            for (int k = 0; k < D; ++k) s += W3[i * D + k] * Cm[l * D + k];
            s *= rbar[l];
            GT[(size_t)l * L.DP + i] = s;
            if (cls == 1) g[L.oGp + pos_perm(i, l, L.KS)] = s;
        }
        if (cls == 0) {             // h0 = Sigma_0 (L0 m0): the Constant mean parent of X_0
            if (tid < D) {
                double s = 0.0;
                for (int j = 0; j < D; ++j) s += a.x0_prec[tid * D + j] * a.x0_mean[j];
                vec[tid] = s;
            }
            __syncthreads();
            if (tid < D) {
                double s = 0.0;
                for (int k = 0; k < D; ++k) s += W3[tid * D + k] * vec[k];
                g[L.oh0 + tid] = s;
            }
        }
        __syncthreads();
    }

    // warm-up lengths of the segmented sweeps: forward recurrence matrix F, backward B (W3 = Sigma_1)
    for (int pass = 0; pass < 2; ++pass) {
        for (int idx = tid; idx < D * D; idx += PREP_THREADS) {
            int i = idx / D, j = idx % D;
            double s = 0.0;
            if (pass == 0) { for (int k = 0; k < D; ++k) s += W3[i * D + k] * (qbar[k] * sA[k * D + j]); }
            else { for (int k = 0; k < D; ++k) s += W3[i * D + k] * sA[j * D + k]; s *= qbar[j]; }
            W1[idx] = s;
        }
        __syncthreads();
        int J = warmup_length(W1, W2, D, tid, red);
        if (tid == 0) a.warm[n * 2 + pass] = J;
        __syncthreads();
    }
}

int launch_prep(pyvb_lds* h) {
    PrepArgs a;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var;
    a.Q_a = h->Q_a; a.Q_b = h->Q_b; a.R_a = h->R_a; a.R_b = h->R_b;
    a.x0_mean = h->pri.x0_mean; a.x0_prec = h->pri.x0_prec;
    a.Sigma = h->Sigma_new; a.qld = h->qld_x_new; a.gains = h->gains; a.scratch = h->scratch;
    a.warm = h->warm; a.status = h->status;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.L = h->L;
    TimedLaunch tl(h, PYVB_K_PREP);
    if (h->D <= 16) hipLaunchKernelGGL(k_prep<16>, dim3(h->N), dim3(PREP_THREADS), 0, h->stream, a);
    else if (h->D <= 32) hipLaunchKernelGGL(k_prep<32>, dim3(h->N), dim3(PREP_THREADS), 0, h->stream, a);
    else hipLaunchKernelGGL(k_prep<64>, dim3(h->N), dim3(PREP_THREADS), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
