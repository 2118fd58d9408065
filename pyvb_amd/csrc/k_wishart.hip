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

// ---- storage of the column covariances (symmetric, rows x rows, D of them per matrix and replicate): the upper 8 x 8 TILES
// only -- tile (A <= B) of the matrix padded to 8 RT rows is number A RT - A (A - 1) / 2 + (B - A), its 64 elements row-major;
// diagonal tiles hold their whole (symmetric) block.  This is the order in which a lane of gj_wave holds its part of the
// inverse, so the column kernel stores 512 contiguous bytes per lane, and it is 56 % of the dense size at 64 rows: the
// covariances are the largest object of the Wishart path (2.4 GB instead of 4.3 GB at N = 1024, D = K = 64).
// (cov_tiles / cov_stride: common.h)
__device__ __forceinline__ size_t cov_pos(int rows, int k, int l) {
    const int RT = (rows + 7) >> 3;
    const int lo = k < l ? k : l, hi = k < l ? l : k, A = lo >> 3, B = hi >> 3;
    const int t = A * RT - (A * (A - 1)) / 2 + (B - A);
    return (size_t)t * 64 + (A == B ? (k & 7) * 8 + (l & 7) : (lo & 7) * 8 + (hi & 7));
}
// (A, B) of tile t
__device__ __forceinline__ void cov_tile_ab(int rows, int t, int& A, int& B) {
    const int RT = (rows + 7) >> 3;
    A = 0;
    while (t >= RT - A) { t -= RT - A; ++A; }
    B = A + t;
}

struct WArgs {
    double *Q_w, *R_w, *Qbar, *Rbar, *lnd, *QA, *RC, *trA, *trC, *A_cov, *C_cov, *RQ, *RR, *SyyF;
    const double* YcovS; const double* Yent;   // outputs with missing entries (k_missing.hip): sum_t qcov_t [N][K][K], entropy terms [N]; or null
    double* ldm;        // [N][2][D]: ln det of the covariance of the UNKNOWN entries of a column that has known ones (gaussian.py:150)
    double* SG;         // [N][2][64][64]-slots holding [rows][rows]: sum_i G[i,i] S_i of A's / C's columns (k_cols_wishart), or null: k_wresid sums the covariances itself
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
    a.which0 = 0; a.c0 = 0; a.c1 = h->D; a.update = 0; a.SG = nullptr; a.ldm = h->ldm;
    a.YcovS = h->has_missing ? h->YcovS : nullptr; a.Yent = h->has_missing ? h->Yent : nullptr;
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
    // traces: a wavefront per column (16 columns each), a lane per element of an 8 x 8 tile: every load instruction reads
    // one whole stored tile (512 contiguous bytes); off-diagonal tiles count twice (both matrices are symmetric)
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    const int wv = tid >> 6, lane = tid & 63, ra = lane >> 3, cb = lane & 7;
    const int RT = (rows + 7) >> 3;
    for (int i = wv; i < D; i += 4) {
        const double* Si = cov + (size_t)i * cov_stride(rows);
        double s = 0.0;
        int t = 0;
        for (int A = 0; A < RT; ++A) {
#pragma unroll 4
            for (int B = A; B < RT; ++B, ++t) {
                const int k = 8 * A + ra, l = 8 * B + cb;
                const double x = Si[(size_t)t * 64 + lane];
                const double y = Lb[(k < rows ? k : 0) * WLD + (l < rows ? l : 0)];
                s = __builtin_fma((k < rows && l < rows) ? (A == B ? 1.0 : 2.0) : 0.0, x * y, s);
            }
        }
        s = wave_sum(s);
        if (lane == 0) (WHICH == 0 ? a.trA : a.trC)[(size_t)n * D + i] = s;
    }
}

// ---- the column update of one matrix in ONE launch: [a.update() for a in As] (or Cs) under Wishart noise.
// One workgroup of eight wavefronts per (replicate, matrix).  The posterior covariances do not depend on each other, so the
// wavefronts invert eight columns' precisions side by side (gj_wave, one matrix per wavefront, in registers); the means are a
// Gauss-Seidel chain over the columns (column i sees the new columns before it), so after its inversion a wavefront waits for
// its column's turn (a counter in LDS), and then forms
//   r = H[:,i] - sum_{j != i} <m_j> G[i,j],   w = prior_prec_i prior_mean_i + E[Lambda] r,   qmu_i = qcov_i w
// with <M> and E[Lambda] in LDS and qcov_i still in its registers -- no covariance is read back from memory -- and adds
// G[i,i] qcov_i to the running sum the noise update needs (Multiplication.pass_down_ExxT, node.py:260-271: sum_i S_i G_ii),
// kept in LDS as the 36 upper 8 x 8 tiles; the turns are in column order, so the sum is formed in a fixed order.
// Known entries of a column (As[i].observe with NaN, examples/LDS_knowns_in_A.py:73-74; gaussian.py:125-134): conditioning
// N(qmu, qcov) on them leaves, for the unknown entries u and the known ones o, qcov_uu = inv(P_uu) and
// qmu_u = inv(P_uu) (w_u - P_uo value_o) with P the precision of the unconditioned update and w its weighted mean.  After the
// full inversion (whose pivots give q_ln_det, as in the reference) the elimination step is applied once more to the pivots
// in o (gj_wave_subset): that exchanges them back and leaves inv(P_uu) and inv(P_uu) P_uo in place; the known entries are
// pinned, their rows and columns of the covariance zero.
// Against round 2's three kernels (all inversions; then a one-wavefront chain per matrix re-reading every covariance; then the
// loop over the columns in k_wresid re-reading them again) this saves two passes over the column covariances
// (profiles/r03/wishart_*).
#define CW_WAVES 8
__global__ void __launch_bounds__(64 * CW_WAVES) k_cols_wishart(WArgs a) {
    __shared__ double Lb[64 * 64];          // E[Lambda], [l][k] (symmetric)
    __shared__ double Mb[64 * 64];          // <M>, [col][row]
    __shared__ double SGs[64 * 36];         // sum_i G[i,i] S_i: element e = 8 ra + cb of upper tile t at [e][t]
    __shared__ double gjbuf[CW_WAVES * (2 * GJW_BUF + 64)];
    __shared__ double gv[64], rvv[64], wvv[64];
    __shared__ int turn, turn2;                 // chain steps done / sums done: the column (counted from c0) whose turn it is
    __shared__ double pre[CW_WAVES][4][64];     // per wavefront: G[i,:], H[:,i], prior_prec_i prior_mean_i, known values (NaN: unknown) of its column, fetched ahead of the chain
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int wv = tid >> 6, lane = tid & 63;
    const int rows = WHICH == 0 ? D : K;
    const double* Lbar = (WHICH == 0 ? a.Qbar : a.Rbar) + (size_t)n * rows * rows;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs; // [row][col], NaN = not known
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    double* var = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    double* ldm = a.ldm + ((size_t)n * 2 + WHICH) * D;
    const int RT = (rows + 7) >> 3;
    for (int idx = tid; idx < 64 * 64; idx += 64 * CW_WAVES) {
        const int k = idx & 63, l = idx >> 6;
        Lb[idx] = (k < rows && l < rows) ? Lbar[l * rows + k] : 0.0;
        Mb[idx] = (k < rows && l < D) ? M[(size_t)k * D + l] : 0.0;         // Mb[col l][row k]
    }
    for (int idx = tid; idx < 64 * 36; idx += 64 * CW_WAVES) SGs[idx] = 0.0;
    if (tid == 0) { turn = 0; turn2 = 0; }
    __syncthreads();
    double* rc = gjbuf + wv * (2 * GJW_BUF + 64);
    double* pivs = rc + 2 * GJW_BUF;
    const int ta = lane >> 3, tb = lane & 7;
    const int tile = ta * 8 - (ta * (ta - 1)) / 2 + (tb - ta);      // index of upper tile (ta <= tb)
    const int nrounds = (a.c1 - a.c0 + CW_WAVES - 1) / CW_WAVES;
    for (int round = 0; round < nrounds; ++round) {
        const int i = a.c0 + round * CW_WAVES + wv;
        const bool active = i < a.c1;                   // wave-uniform
        double v[8][8];
        double g = 0.0;
        if (active) {
            // what the chain step will need from memory, fetched before the elimination so that its latency is not in the chain
            g = G[(size_t)i * D + i];
            pre[wv][0][lane] = lane < D ? G[(size_t)i * D + lane] : 0.0;
            pre[wv][1][lane] = lane < rows ? H[(size_t)lane * D + i] : 0.0;
            pre[wv][2][lane] = lane < rows ? pp[(size_t)i * rows + lane] * pm[(size_t)lane * D + i] : 0.0;
            const double ob = lane < rows ? obs[(size_t)lane * D + i] : __builtin_nan("");
            pre[wv][3][lane] = ob;
            const unsigned long long kmask = __ballot(ob == ob);        // bit k: entry k of the column is known (wave-uniform)
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                const int k = 8 * ta + ra;
                const double pk = pp[(size_t)i * rows + (k < rows ? k : rows - 1)];
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int l = 8 * tb + cb;
                    const double inside = __builtin_fma(g, Lb[l * 64 + k], (k == l) ? pk : 0.0);
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
            const unsigned long long allrows = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
            if (lane == 0 && kmask != allrows) qld[i] = 0.5 / (0.5 * lp);      // q_ln_det, gaussian.py:120 (quirk Q1): of the whole precision
            bool rk[8], ck[8];              // this lane's rows / columns that are known entries
#pragma unroll
            for (int u = 0; u < 8; ++u) { rk[u] = (kmask >> (8 * ta + u)) & 1ull; ck[u] = (kmask >> (8 * tb + u)) & 1ull; }
            if (kmask != 0) {               // wave-uniform
                gj_wave_subset(v, kmask, lane, rc, pivs);
                double l2 = (lane < rows && ((kmask >> lane) & 1ull)) ? log(pivs[lane]) : 0.0;
                l2 = wave_sum(l2);
                if (lane == 0) ldm[i] = -(lp + l2);         // ln det qcov_uu = -ln det P_uu: what the bound of such a column reads (:150)
            }
            if (ta <= tb && tb < RT) {          // the upper tiles, 64 contiguous doubles per lane (cov_pos)
                double* ci_ = cov + (size_t)i * cov_stride(rows) + (size_t)(ta * RT - (ta * (ta - 1)) / 2 + (tb - ta)) * 64;
#pragma unroll
                for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                    for (int h2 = 0; h2 < 4; ++h2)
                        *reinterpret_cast<d2*>(ci_ + 8 * ra + 2 * h2) = d2{(rk[ra] || ck[2 * h2]) ? 0.0 : v[ra][2 * h2],
                                                                          (rk[ra] || ck[2 * h2 + 1]) ? 0.0 : v[ra][2 * h2 + 1]};
            }
            if (ta == tb) {
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) if (8 * ta + ra < rows) var[(size_t)i * rows + 8 * ta + ra] = rk[ra] ? 0.0 : v[ra][ra];
            }
            // ---- the chain step of column i, when the columns before it have had theirs.  No workgroup barrier: a
            // wavefront that is done goes on with its next elimination while the others take their turns, so in steady
            // state the turns are staggered and nobody waits.  (All eight wavefronts are resident: the wait cannot deadlock.)
            const int rel = round * CW_WAVES + wv;
            while (__hip_atomic_load(&turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != rel) __builtin_amdgcn_s_sleep(4);
            gv[lane] = pre[wv][0][lane];
            gjw_sync();
            double r4[4] = {pre[wv][1][lane], 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int j = 0; j < 64; j += 4)
#pragma unroll
                for (int u = 0; u < 4; ++u) r4[u] = __builtin_fma(-Mb[(j + u) * 64 + lane], (j + u != i) ? gv[j + u] : 0.0, r4[u]);
            rvv[lane] = (r4[0] + r4[1]) + (r4[2] + r4[3]);
            gjw_sync();
            double w4[4] = {pre[wv][2][lane], 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int l = 0; l < 64; l += 4)
#pragma unroll
                for (int u = 0; u < 4; ++u) w4[u] = __builtin_fma(Lb[(l + u) * 64 + lane], rvv[l + u], w4[u]);
            const double wl = (w4[0] + w4[1]) + (w4[2] + w4[3]);
            const double obl = pre[wv][3][lane];
            // qmu_u = [u,u] w_u - [u,o] value_o (gj_wave_subset): the known entries enter with their negated values
            wvv[lane] = lane < rows ? ((obl == obl) ? -obl : wl) : 0.0;
            gjw_sync();
            double ws[8];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) ws[cb] = wvv[8 * tb + cb];
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                double p0 = v[ra][0] * ws[0], p1 = v[ra][1] * ws[1];
#pragma unroll
                for (int cb = 2; cb < 8; cb += 2) { p0 = __builtin_fma(v[ra][cb], ws[cb], p0); p1 = __builtin_fma(v[ra][cb + 1], ws[cb + 1], p1); }
                double part = p0 + p1;
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64);
                if (tb == 0) {
                    const double known = pre[wv][3][8 * ta + ra];
                    Mb[i * 64 + 8 * ta + ra] = (8 * ta + ra < rows) ? ((known == known) ? known : part) : 0.0;
                }
            }
            __hip_atomic_store(&turn, rel + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ---- the running sum, in column order too, but as a stage of its own behind the chain step: the next column's
            // chain step does not wait for it
            while (__hip_atomic_load(&turn2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != rel) __builtin_amdgcn_s_sleep(2);
            if (ta <= tb) {
                // eight elements at a time, loads first: written as one statement per element the compiler orders every
                // load behind the store before it (they could alias for all it knows)
#pragma unroll
                for (int ra = 0; ra < 8; ++ra) {
                    double t8[8];
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) t8[cb] = SGs[(8 * ra + cb) * 36 + tile];
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) t8[cb] = __builtin_fma(g, (rk[ra] || ck[cb]) ? 0.0 : v[ra][cb], t8[cb]);
#pragma unroll
                    for (int cb = 0; cb < 8; ++cb) SGs[(8 * ra + cb) * 36 + tile] = t8[cb];
                }
            }
            __hip_atomic_store(&turn2, rel + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < rows * D; idx += 64 * CW_WAVES) {
        const int k = idx / D, col = idx % D;
        if (col >= a.c0 && col < a.c1) M[idx] = Mb[col * 64 + k];
    }
    if (a.SG) {
        double* SG = a.SG + ((size_t)n * 2 + WHICH) * 64 * 64;
        for (int idx = tid; idx < rows * rows; idx += 64 * CW_WAVES) {
            const int k = idx / rows, l = idx % rows;
            const int lo = k < l ? k : l, hi2 = k < l ? l : k;          // element (lo, hi2) of the upper triangle
            const int A8 = lo >> 3, B8 = hi2 >> 3;
            int e;
            if (A8 == B8) { const int r0 = k & 7, c0 = l & 7; e = 8 * (r0 < c0 ? r0 : c0) + (r0 < c0 ? c0 : r0); }    // diagonal tile: its own upper part
            else e = 8 * (lo & 7) + (hi2 & 7);
            SG[(size_t)k * rows + l] = SGs[e * 36 + (A8 * 8 - (A8 * (A8 - 1)) / 2 + (B8 - A8))];
        }
    }
}

// ---- Wishart.update (nodes_todo.py:228-231) for Q (children X_1.., mean parents Mult(A, X_{t-1})) or R (children Y_t):
//   Rm = 1/2 ( own + <M> G <M>^T + sum_i S_i G[i,i] ) - H <M>^T        (Multiplication.pass_down_ExxT node.py:260-271)
//   qw = w0 + Rm
// own = sum_t <x x^T> over the children (Q: t >= 1) or sum_t y y^T (R).  Rm is kept for the lower bound.
__global__ void __launch_bounds__(256) k_wresid(WArgs a) {
    // Three 64 x 64 x 64 products per (replicate, matrix) on the matrix cores: T1 = <M> G, E = T1 <M>^T, HM = H <M>^T (they were
    // 0.54 ms of scalar FMAs with strided operand walks).  Wavefront w owns row tile w; operands zero padded to 64 in LDS
    // (stride WLD: an A operand's 16 rows of a k-step then fall into different banks).
    __shared__ double Mb[64 * WLD], T1[64 * WLD], Hb[64 * WLD], Gb[64 * WLD];
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K, T = a.T, DP = a.DP;
    const int w = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int rows = WHICH == 0 ? D : K;
    const double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + (size_t)n * D * cov_stride(rows);
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    double* Rm = (WHICH == 0 ? a.RQ : a.RR) + (size_t)n * rows * rows;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        Mb[i * WLD + j] = (i < rows && j < D) ? M[(size_t)i * D + j] : 0.0;
        Hb[i * WLD + j] = (i < rows && j < D) ? H[(size_t)i * D + j] : 0.0;
        Gb[i * WLD + j] = (i < D && j < D) ? G[(size_t)i * D + j] : 0.0;
    }
    __syncthreads();
    // T1 = <M> G: row tile w; A operand (row 16 w + r, k = 4 s + q) of Mb, B operand (k, column 16 nn + r) of Gb
    {
        d4 acc[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) acc[nn] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const double av = Mb[(16 * w + r) * WLD + 4 * s4 + q];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) acc[nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Gb[(4 * s4 + q) * WLD + 16 * nn + r], acc[nn], 0, 0, 0);
        }
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int e = 0; e < 4; ++e) T1[(16 * w + 4 * e + q) * WLD + 16 * nn + r] = acc[nn][e];
    }
    __syncthreads();
    // E = T1 <M>^T and HM = H <M>^T: B operand (k = j, column l) = Mb[l][j]
    d4 accE[4], accH[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) { accE[nn] = d4{0.0, 0.0, 0.0, 0.0}; accH[nn] = accE[nn]; }
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
        const double at = T1[(16 * w + r) * WLD + 4 * s4 + q], ah = Hb[(16 * w + r) * WLD + 4 * s4 + q];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            const double bm = Mb[(16 * nn + r) * WLD + 4 * s4 + q];
            accE[nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, bm, accE[nn], 0, 0, 0);
            accH[nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ah, bm, accH[nn], 0, 0, 0);
        }
    }
    const double* x0 = a.X + (size_t)n * T * DP;
    const double* S0 = a.Sigma + (size_t)n * 3 * D * D;
    const double* GC = mo + MOM_GC(D, K);
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int k = 16 * w + 4 * e4 + q, l = 16 * nn + r;         // accumulator element (row k, column l)
            if (k >= rows || l >= rows) continue;
            const int idx = k * rows + l;
            double e = accE[nn][e4];
            const double hm = accH[nn][e4];
            if (a.SG) e += a.SG[((size_t)n * 2 + WHICH) * 64 * 64 + idx];       // sum_i S_i G[i,i], formed by k_cols_wishart
            else {
#pragma unroll 16
                for (int i = 0; i < D; ++i) e += cov[(size_t)i * cov_stride(rows) + cov_pos(rows, k, l)] * G[(size_t)i * D + i];
            }
            double own;
            if (WHICH == 0) own = GC[idx] - x0[xpos(k)] * x0[xpos(l)] - S0[idx];          // sum_{t >= 1} <x x^T>
            else own = a.SyyF[(size_t)n * K * K + idx] + (a.YcovS ? a.YcovS[(size_t)n * K * K + idx] : 0.0);        // <y y^T> = qmu qmu^T + qcov
            const double rr = 0.5 * (own + e) - hm;
            Rm[idx] = rr;
            if (a.update) {
                const double* w0 = WHICH == 0 ? a.pri.Q_w0 : a.pri.R_w0;
                (WHICH == 0 ? a.Q_w : a.R_w)[(size_t)n * rows * rows + idx] = w0[idx] + rr;
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
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + ((size_t)n * D + i) * cov_stride(rows);
    const double* var = (WHICH == 0 ? a.A_var : a.C_var) + ((size_t)n * D + i) * rows;
    for (int idx = threadIdx.x; idx < (int)cov_stride(rows); idx += 256) cov[idx] = 0.0;
    __syncthreads();
    for (int k = threadIdx.x; k < rows; k += 256) cov[cov_pos(rows, k, k)] = var[k];
}

// ... and back: the diagonals (what the lower bound reads) of column covariances the caller supplied
__global__ void __launch_bounds__(128) k_cov_to_colvar(WArgs a) {
    const int WHICH = blockIdx.z, n = blockIdx.x, i = blockIdx.y, D = a.D, k = threadIdx.x;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + ((size_t)n * D + i) * cov_stride(rows);
    double* var = (WHICH == 0 ? a.A_var : a.C_var) + ((size_t)n * D + i) * rows;
    if (k < rows) var[k] = cov[cov_pos(rows, k, k)];
}

// ---- the C ABI exchanges dense [rows][rows] covariances (pyvb_lds_get/set_column_cov): dense <-> upper tiles, in slices of
// replicates so that the dense staging buffer stays small.  to_packed reads the upper triangle of what the caller supplied.
struct CovConvArgs { double* packed; double* dense; int n0, D, rows, to_packed; };
__global__ void __launch_bounds__(256) k_cov_convert(CovConvArgs c) {
    const int n = blockIdx.x, i = blockIdx.y, rows = c.rows;
    double* P = c.packed + ((size_t)(c.n0 + n) * c.D + i) * cov_stride(rows);
    double* Dn = c.dense + ((size_t)n * c.D + i) * rows * rows;
    for (int idx = threadIdx.x; idx < rows * rows; idx += 256) {
        const int k = idx / rows, l = idx % rows;
        if (c.to_packed) { if (k <= l || (k >> 3) == (l >> 3)) P[cov_pos(rows, k, l)] = Dn[idx]; }
        else Dn[idx] = P[cov_pos(rows, k, l)];
    }
}

// Gaussian.observe on fully known columns (gaussian.py:97-100) under Wishart noise: their dense covariance is zero too
// (k_observe, k_params.hip, has set the means and the diagonal)
__global__ void __launch_bounds__(256) k_cov_observe(WArgs a) {
    const int WHICH = blockIdx.z, n = blockIdx.x, i = blockIdx.y, D = a.D;
    const int rows = WHICH == 0 ? a.D : a.K;
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs;
    for (int k = 0; k < rows; ++k) { const double ob = obs[(size_t)k * D + i]; if (!(ob == ob)) return; }      // block-uniform
    double* cov = (WHICH == 0 ? a.A_cov : a.C_cov) + ((size_t)n * D + i) * cov_stride(rows);
    for (int idx = threadIdx.x; idx < (int)cov_stride(rows); idx += 256) cov[idx] = 0.0;
}

int launch_cov_observe(pyvb_lds* h) {
    WArgs a = make_wargs(h);
    hipLaunchKernelGGL(k_cov_observe, dim3(h->N, h->D, 2), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_cov_convert(pyvb_lds* h, int which, double* dense, int n0, int count, int to_packed) {
    CovConvArgs c; c.packed = which == 0 ? h->A_cov : h->C_cov; c.dense = dense; c.n0 = n0; c.D = h->D;
    c.rows = which == 0 ? h->D : h->K; c.to_packed = to_packed;
    hipLaunchKernelGGL(k_cov_convert, dim3(count, h->D), dim3(256), 0, h->stream, c);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

__device__ __forceinline__ double psi_multi(double x, int D) {       // sum_{i<D} psi(x - i/2)
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += digamma_pos(x - 0.5 * i);
    return s;
}
__device__ __forceinline__ double lgamma_multi(double x, int D) {    // ln |Gamma_D(x)|
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
    // (a lane takes the entries lane, lane + 64: D, K <= 128)
    double tq = 0.0, tr = 0.0, tq0 = 0.0, tr0 = 0.0;
    for (int i = lane; i < D; i += 64) for (int l = 0; l < D; ++l) { tq += Qb[i * D + l] * RQ[l * D + i]; tq0 += a.pri.Q_w0[i * D + l] * Qb[l * D + i]; }
    for (int i = lane; i < K; i += 64) for (int l = 0; l < K; ++l) { tr += Rb[i * K + l] * RR[l * K + i]; tr0 += a.pri.R_w0[i * K + l] * Rb[l * K + i]; }
    const double trQ = wave_sum(tq), trR = wave_sum(tr), trQ0 = wave_sum(tq0), trR0 = wave_sum(tr0);
    double e0 = 0.0;
    for (int i = lane; i < D; i += 64) {
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
    double LY = (double)T * (-0.5 * K * LN2PI + 0.5 * ln[1]) - trR;
    if (a.Yent) LY -= a.Yent[n];
    double la = 0.0, lc = 0.0;
    for (int i = lane; i < D; i += 64) {
        // the last term depends on how much of the column is known (gaussian.py:145-150, as k_elbo in k_params.hip): nothing ->
        // the q_ln_det form, some entries -> ln det of the covariance of the rest (ldm, from k_cols_wishart), all -> no term
        auto column = [&](int rows, const double* pp, const double* pm, const double* M, const double* V, double qld, double lndet,
                          const double* obs, double ldmv) {
            double trc = 0.0;
            int missing = 0;
            for (int k = 0; k < rows; ++k) {
                const double m = M[(size_t)k * D + i], m0 = pm[(size_t)k * D + i], ob = obs[(size_t)k * D + i];
                trc += pp[(size_t)i * rows + k] * (m * m + V[(size_t)i * rows + k] + m0 * m0 - 2.0 * m * m0);
                if (!(ob == ob)) ++missing;
            }
            double r = -0.5 * rows * LN2PI + 0.5 * lndet - 0.5 * trc;
            if (missing == rows) r += 0.5 * rows * LN2PI + 0.5 * qld + 0.5 * rows;
            else if (missing > 0) r -= 0.5 * missing * LN2PI - 0.5 * ldmv - 0.5 * missing;
            return r;
        };
        la += column(D, a.pri.A_pp, a.pri.A_pm, a.A_mean + (size_t)n * D * D, a.A_var + (size_t)n * D * D, a.qld_A[(size_t)n * D + i], a.pri.A_pld[i],
                     a.pri.A_obs, a.ldm[((size_t)n * 2 + 0) * D + i]);
        lc += column(K, a.pri.C_pp, a.pri.C_pm, a.C_mean + (size_t)n * K * D, a.C_var + (size_t)n * D * K, a.qld_C[(size_t)n * D + i], a.pri.C_pld[i],
                     a.pri.C_obs, a.ldm[((size_t)n * 2 + 1) * D + i]);
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
    if (h->big) return launch_wexpect_big(h);
    WArgs a = make_wargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_wexpect, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_dense_pre(pyvb_lds* h) {
    if (h->big) return launch_dense_pre_big(h);
    WArgs a = make_wargs(h);
    TimedLaunch tl(h, PYVB_K_PREP);
    hipLaunchKernelGGL(k_dense_pre, dim3(h->N, 2), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_cols_dense(pyvb_lds* h, int which, int c0, int c1) {
    if (h->big) return launch_cols_dense_big(h, which, c0, c1);
    WArgs a = make_wargs(h);
    a.which0 = which == 1 ? 1 : 0; a.c0 = c0; a.c1 = c1;
    const int nw = which == 2 ? 2 : 1;
    a.SG = h->SG;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_cols_wishart, dim3(h->N, nw), dim3(64 * CW_WAVES), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    // the sums hold every column's covariance as it is now only when all columns went through the launch (and the
    // statistics they were weighted with stay the current ones: states_changed() drops them)
    const bool all = c0 == 0 && c1 == h->D;
    if (which == 0 || which == 2) h->sg_valid[0] = all;
    if (which == 1 || which == 2) h->sg_valid[1] = all;
    return PYVB_OK;
}

int launch_wresid(pyvb_lds* h, int which, int update) {
    if (h->big) return launch_wresid_big(h, which, update);
    WArgs a = make_wargs(h);
    a.which0 = which == 1 ? 1 : 0; a.update = update;
    const bool need0 = which != 1, need1 = which != 0;
    a.SG = ((!need0 || h->sg_valid[0]) && (!need1 || h->sg_valid[1])) ? h->SG : nullptr;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_wresid, dim3(h->N, which == 2 ? 2 : 1), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_syy_full(pyvb_lds* h) {
    if (h->big) return launch_syy_full_big(h);
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
    hipLaunchKernelGGL(k_cov_to_colvar, dim3(h->N, h->D, 2), dim3(128), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_elbo_dense(pyvb_lds* h, hipStream_t stream) {
    WArgs a = make_wargs(h);
    hipStream_t s = stream ? stream : h->stream;
    TimedLaunch tl(h, PYVB_K_ELBO, s);
    hipLaunchKernelGGL(k_elbo_dense, dim3(h->N), dim3(64), 0, s, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
