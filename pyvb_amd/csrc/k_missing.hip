// Outputs with missing entries: Y_t.observe(v) with NaN in v (gaussian.py:74-100) leaves the node partially observed,
// or -- all NaN -- not observed at all.  Such a Y_t is a variational node of its own: it messages its posterior mean to
// X_t like an observed one (gaussian.py:179-183: (<R>, <R> qmu)), so the sweeps only need the current means in the Y
// array; what changes is
//   k_impute       [y.update() for y in Ys if not y.observed]: Gaussian.update gaussian.py:102-134 with parents only --
//                  qprec = <R> (diagonal), qmu = <C> mu_t, then the known entries are conditioned on (:125-134), which for
//                  a diagonal covariance pins them (variance 0) and leaves the others at <C> mu_t with variance 1 / <R>_k
//   k_syy_missing  sum_t <y y^T> diagonal = sum_t (qmu^2 + variance), and what Gaussian.log_lower_bound subtracts for the
//                  rows that are not fully observed (:145-150)
// Until its first update() a partially observed row keeps the constructor's posterior in ALL entries (observe() only
// records the known values): k_missing_init writes that state.
#include "params.h"
#include "gj.h"

struct MissArgs {
    double* Y; const double* Yobs; double* Yvar; double* Yqld;
    // Wishart noise (dense <R>): E[R] [N][K][K], [N][4] log-determinants (k_wexpect), the symmetrised qw and qv; per row the
    // ln det of the covariance of its missing entries (NaN: not updated yet, the diagonal initial state), and sum_t qcov_t
    const double *Rbar, *lnd, *R_w; double* Yld; double* YcovS;
    int diag_cov;       // k_missing_ent_dense: 1 = the rows still carry their diagonal initial covariances: form YcovS from Yvar
    const double* X; const double* C_mean; const double *R_a, *R_b;
    double* Syy; double* Yent;
    const double* Yq0; const double* Yrowvar0;
    int N, T, K, D, DP;
};

// One wavefront per row; a lane handles entries lane and lane + 64 (K <= 128: the second shape class, k_big.hip).
__global__ void __launch_bounds__(256) k_missing_init(MissArgs a) {
    const int n = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, K = a.K;
    if (t >= a.T) return;
    const size_t row = ((size_t)n * a.T + t) * K;
    double ob[2];
    bool miss = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        ob[h] = k < K ? a.Yobs[row + k] : 0.0;
        miss = miss || (k < K && !(ob[h] == ob[h]));
    }
    const bool any = __ballot(miss) != 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k >= K) continue;
        a.Y[row + k] = any ? (a.Yq0 ? a.Yq0[row + k] : 0.0) : ob[h];
        a.Yvar[row + k] = any ? (a.Yrowvar0 ? a.Yrowvar0[(size_t)n * a.T + t] : 1.0) : 0.0;
    }
    if (lane == 0) { a.Yqld[(size_t)n * a.T + t] = nan(""); if (a.Yld) a.Yld[(size_t)n * a.T + t] = nan(""); }
}

__global__ void __launch_bounds__(256) k_impute(MissArgs a) {
    const int n = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, K = a.K, D = a.D;
    if (t >= a.T) return;
    const size_t row = ((size_t)n * a.T + t) * K;
    double ob[2], rbar[2];
    bool miss[2];
    double lr = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        ob[h] = k < K ? a.Yobs[row + k] : 0.0;
        miss[h] = k < K && !(ob[h] == ob[h]);
        rbar[h] = k < K ? a.R_a[(size_t)n * K + k] / a.R_b[(size_t)n * K + k] : 1.0;
        lr += k < K ? 0.5 * log(rbar[h]) : 0.0;         // sum log diag chol(<R>)
    }
    if (__ballot(miss[0] || miss[1]) == 0) return;      // fully observed: never updates (gaussian.py:109-110)
    lr = wave_sum(lr);
    if (lane == 0) a.Yqld[(size_t)n * a.T + t] = 0.5 / lr;    // gaussian.py:120 (quirk Q1)
    const double* x = a.X + ((size_t)n * a.T + t) * a.DP;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k >= K) continue;
        if (miss[h]) {
            const double* c = a.C_mean + ((size_t)n * K + k) * D;
            double m = 0.0;
            for (int j = 0; j < D; ++j) m += c[j] * x[xpos(j)];
            a.Y[row + k] = m;
            a.Yvar[row + k] = 1.0 / rbar[h];
        } else {
            a.Y[row + k] = ob[h];
            a.Yvar[row + k] = 0.0;
        }
    }
}

__global__ void __launch_bounds__(256) k_syy_missing(MissArgs a) {
    __shared__ double red[4][130];
    const int n = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, K = a.K, T = a.T;
    const double* Y = a.Y + (size_t)n * T * K;
    const double* Yo = a.Yobs + (size_t)n * T * K;
    const double* Yv = a.Yvar + (size_t)n * T * K;
    double s[2] = {0.0, 0.0}, ent = 0.0;
    for (int t = w; t < T; t += 4) {
        bool miss[2];
        double lv = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = lane + 64 * h;
            const bool live = k < K;
            const double y = live ? Y[(size_t)t * K + k] : 0.0, v = live ? Yv[(size_t)t * K + k] : 0.0, ob = live ? Yo[(size_t)t * K + k] : 0.0;
            miss[h] = live && !(ob == ob);
            s[h] += y * y + v;
            lv += miss[h] ? log(v) : 0.0;
        }
        const int nm = __popcll(__ballot(miss[0])) + __popcll(__ballot(miss[1]));
        if (nm == 0) continue;                          // wave-uniform
        lv = wave_sum(lv);
        if (nm == K) ent += -0.5 * K * LN2PI - 0.5 * a.Yqld[(size_t)n * T + t] - 0.5 * K;        // gaussian.py:145-147
        else ent += 0.5 * nm * LN2PI - 0.5 * lv - 0.5 * nm;                                        // gaussian.py:148-150
    }
    red[w][lane] = s[0]; red[w][64 + lane] = s[1];
    if (lane == 0) red[w][128] = ent;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = lane + 64 * h;
            if (k < K) a.Syy[(size_t)n * K + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        }
        if (lane == 0) a.Yent[n] = red[0][128] + red[1][128] + red[2][128] + red[3][128];
    }
}

// ---- Wishart noise: <R> is dense, so a row's missing entries u are correlated with each other and regress on the known ones o
// (Gaussian.update gaussian.py:102-134 with parents only: qprec = <R>, qmu = <C> mu_t, then the conditioning :125-134):
//   qcov_uu = inv(<R>_uu),   qmu_u = (<C> mu_t)_u - inv(<R>_uu) <R>_uo (y_o - (<C> mu_t)_o),   known entries pinned.
// Applying the Gauss-Jordan step of gj.h to the pivots u of <R> itself is the exchange that leaves exactly inv(<R>_uu) and
// inv(<R>_uu) <R>_uo in place (gj_wave_subset), |u| steps per row; a row without any known entry is the unconditioned
// posterior N(<C> mu_t, inv <R>) = qw / qv.  One workgroup per replicate, its four wavefronts take the rows with missing
// entries in turn; the sum of the rows' covariances, which the Wishart update of R needs (<y y^T> = qmu qmu^T + qcov,
// nodes_todo.py:228-231), is kept in registers per wavefront and reduced in a fixed order at the end.
__global__ void __launch_bounds__(256) k_impute_dense(MissArgs a) {
    __shared__ double Rb[64 * 64];          // E[R], [l][k] (symmetric)
    __shared__ double Cb[64 * 64];          // <C>^T: Cb[j][k] = C[k][j]
    __shared__ double red[64 * 64];
    __shared__ double gjbuf[4 * (GJW_BUF + 64)];
    __shared__ double vec[4][3][64];
    const int n = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, K = a.K, D = a.D, T = a.T;
    const double* Rbar = a.Rbar + (size_t)n * K * K;
    const double* Rw = a.R_w + (size_t)n * K * K;
    const double qv = a.R_a[(size_t)n * K];
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int k = idx & 63, l = idx >> 6;
        Rb[idx] = (k < K && l < K) ? Rbar[l * K + k] : ((k == l) ? 1.0 : 0.0);
        Cb[idx] = (k < K && l < D) ? a.C_mean[((size_t)n * K + k) * D + l] : 0.0;       // Cb[j = l][k]
    }
    __syncthreads();
    double* rc = gjbuf + wv * (GJW_BUF + 64);
    double* pivs = rc + GJW_BUF;
    const int ta = lane >> 3, tb = lane & 7;
    const double qld_full = 0.5 / (0.5 * a.lnd[(size_t)n * 4 + 1]);        // gaussian.py:120 (quirk Q1) of qprec = <R>
    double acc[8][8];
#pragma unroll
    for (int ra = 0; ra < 8; ++ra)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[ra][cb] = 0.0;
    for (int t = wv; t < T; t += 4) {
        const size_t row = ((size_t)n * T + t) * K;
        const double ob = lane < K ? a.Yobs[row + lane] : 0.0;
        const unsigned long long umask = __ballot(lane < K && !(ob == ob));
        if (umask == 0) continue;                       // fully observed: never updates (wave-uniform)
        const int nm = __popcll(umask);
        const double* x = a.X + ((size_t)n * T + t) * a.DP;
        double pmu = 0.0;
        for (int j = 0; j < D; ++j) pmu = __builtin_fma(Cb[j * 64 + lane], x[xpos(j)], pmu);       // (<C> mu_t)[lane]
        double v[8][8];
        double mean = pmu, var = 0.0;
        if (nm == K) {
            // nothing known: qcov = inv <R> = sym(qw) / qv
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int k = 8 * ta + ra, l = 8 * tb + cb;
                    v[ra][cb] = (k < K && l < K) ? 0.5 * (Rw[k * K + l] + Rw[l * K + k]) / qv : 0.0;
                }
            if (lane < K) var = 0.5 * (Rw[lane * K + lane] + Rw[lane * K + lane]) / qv;
        } else {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) v[ra][cb] = Rb[(8 * tb + cb) * 64 + 8 * ta + ra];
            gj_wave_subset(v, umask, lane, rc, pivs);
            double lp = ((umask >> lane) & 1ull) ? log(pivs[lane]) : 0.0;
            if (((umask >> lane) & 1ull) && !(pivs[lane] > 0.0)) lp = nan("");
            lp = wave_sum(lp);
            if (lane == 0) a.Yld[(size_t)n * T + t] = -lp;          // ln det qcov_uu = -ln det <R>_uu
            // qmu_u = pmu_u - [u,o] (y_o - pmu_o)
            gjw_sync();
            vec[wv][0][lane] = ((umask >> lane) & 1ull) || lane >= K ? 0.0 : ob - pmu;
            gjw_sync();
            double ws[8];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) ws[cb] = vec[wv][0][8 * tb + cb];
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                double part = 0.0;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) part = __builtin_fma(v[ra][cb], ws[cb], part);
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64);
                if (tb == 0) vec[wv][1][8 * ta + ra] = part;
                if (ta == tb && ((umask >> (8 * ta + ra)) & 1ull)) vec[wv][2][8 * ta + ra] = v[ra][ra];     // its variance
            }
            gjw_sync();
            const bool mine = (umask >> lane) & 1ull;
            mean = mine ? pmu - vec[wv][1][lane] : ob;
            var = mine ? vec[wv][2][lane] : 0.0;
            // only the block of the missing entries is this row's covariance
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const bool in = ((umask >> (8 * ta + ra)) & 1ull) && ((umask >> (8 * tb + cb)) & 1ull);
                    v[ra][cb] = in ? v[ra][cb] : 0.0;
                }
        }
        if (lane < K) { a.Y[row + lane] = mean; a.Yvar[row + lane] = var; }
        if (lane == 0) a.Yqld[(size_t)n * T + t] = qld_full;
#pragma unroll
        for (int ra = 0; ra < 8; ++ra)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[ra][cb] += v[ra][cb];
    }
    // sum over the four wavefronts, in order
    for (int turn = 0; turn < 4; ++turn) {
        __syncthreads();
        if (wv == turn) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra)
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    double* r = red + (8 * ta + ra) * 64 + 8 * tb + cb;
                    *r = (turn == 0 ? 0.0 : *r) + acc[ra][cb];
                }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < K * K; idx += 256) a.YcovS[(size_t)n * K * K + idx] = red[(idx / K) * 64 + idx % K];
}

// what Gaussian.log_lower_bound subtracts for the rows that are not fully observed (gaussian.py:145-150), and -- while the rows
// still carry their diagonal initial covariances (diag_cov) -- the sum of those
__global__ void __launch_bounds__(256) k_missing_ent_dense(MissArgs a) {
    __shared__ double red[4][65];
    const int n = blockIdx.x, w = threadIdx.x >> 6, k = threadIdx.x & 63, K = a.K, T = a.T;
    const double* Yo = a.Yobs + (size_t)n * T * K;
    const double* Yv = a.Yvar + (size_t)n * T * K;
    double s = 0.0, ent = 0.0;
    for (int t = w; t < T; t += 4) {
        const bool live = k < K;
        const double v = live ? Yv[(size_t)t * K + k] : 0.0, ob = live ? Yo[(size_t)t * K + k] : 0.0;
        const bool miss = live && !(ob == ob);
        s += v;
        const int nm = __popcll(__ballot(miss));
        if (nm == 0) continue;                          // wave-uniform
        const double ld = a.Yld[(size_t)n * T + t];
        const double lv = (ld == ld) ? ld : wave_sum(miss ? log(v) : 0.0);     // not updated yet: the diagonal initial state
        if (nm == K) ent += -0.5 * K * LN2PI - 0.5 * a.Yqld[(size_t)n * T + t] - 0.5 * K;        // gaussian.py:145-147
        else ent += 0.5 * nm * LN2PI - 0.5 * lv - 0.5 * nm;                                        // gaussian.py:148-150
    }
    red[w][k] = s;
    if (k == 0) red[w][64] = ent;
    __syncthreads();
    if (w == 0) {
        if (k == 0) a.Yent[n] = red[0][64] + red[1][64] + red[2][64] + red[3][64];
        if (a.diag_cov) {
            const double d = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            for (int l = 0; l < K; ++l) if (k < K) a.YcovS[(size_t)n * K * K + (size_t)k * K + l] = (k == l) ? d : 0.0;
        }
    }
}

static MissArgs make_margs(pyvb_lds* h) {
    MissArgs a;
    a.Y = h->Y; a.Yobs = h->Yobs; a.Yvar = h->Yvar; a.Yqld = h->Yqld; a.X = h->X[h->cur]; a.C_mean = h->C_mean;
    a.R_a = h->R_a; a.R_b = h->R_b; a.Syy = h->Syy; a.Yent = h->Yent; a.Yq0 = nullptr; a.Yrowvar0 = nullptr;
    a.N = h->N; a.T = h->T; a.K = h->K; a.D = h->D; a.DP = h->L.DP;
    a.Rbar = h->Rbar; a.lnd = h->lnd; a.R_w = h->R_w; a.Yld = h->dense ? h->Yld : nullptr; a.YcovS = h->YcovS; a.diag_cov = 0;
    return a;
}

int launch_impute_dense(pyvb_lds* h) {
    MissArgs a = make_margs(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_impute_dense, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_missing_ent_dense(pyvb_lds* h, int diag_cov) {
    MissArgs a = make_margs(h);
    a.diag_cov = diag_cov;
    hipLaunchKernelGGL(k_missing_ent_dense, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_missing_init(pyvb_lds* h, const double* Yq0, const double* Yrowvar0) {
    MissArgs a = make_margs(h);
    a.Yq0 = Yq0; a.Yrowvar0 = Yrowvar0;
    hipLaunchKernelGGL(k_missing_init, dim3((h->T + 3) / 4, h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_impute(pyvb_lds* h) {
    MissArgs a = make_margs(h);
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_impute, dim3((h->T + 3) / 4, h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_syy_missing(pyvb_lds* h) {
    MissArgs a = make_margs(h);
    hipLaunchKernelGGL(k_syy_missing, dim3(h->N), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
