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

struct MissArgs {
    double* Y; const double* Yobs; double* Yvar; double* Yqld;
    const double* X; const double* C_mean; const double *R_a, *R_b;
    double* Syy; double* Yent;
    const double* Yq0; const double* Yrowvar0;
    int N, T, K, D, DP;
};

__global__ void __launch_bounds__(256) k_missing_init(MissArgs a) {
    const int n = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63, K = a.K;
    if (t >= a.T) return;
    const size_t row = ((size_t)n * a.T + t) * K;
    const double ob = k < K ? a.Yobs[row + k] : 0.0;
    const bool any = __ballot(k < K && !(ob == ob)) != 0;
    if (k >= K) return;
    a.Y[row + k] = any ? (a.Yq0 ? a.Yq0[row + k] : 0.0) : ob;
    a.Yvar[row + k] = any ? (a.Yrowvar0 ? a.Yrowvar0[(size_t)n * a.T + t] : 1.0) : 0.0;
    if (k == 0) a.Yqld[(size_t)n * a.T + t] = nan("");
}

__global__ void __launch_bounds__(256) k_impute(MissArgs a) {
    const int n = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63, K = a.K, D = a.D;
    if (t >= a.T) return;
    const size_t row = ((size_t)n * a.T + t) * K;
    const double ob = k < K ? a.Yobs[row + k] : 0.0;
    const bool miss = k < K && !(ob == ob);
    if (__ballot(miss) == 0) return;                    // fully observed: never updates (gaussian.py:109-110)
    const double rbar = k < K ? a.R_a[(size_t)n * K + k] / a.R_b[(size_t)n * K + k] : 1.0;
    double lr = k < K ? 0.5 * log(rbar) : 0.0;          // sum log diag chol(<R>)
    lr = wave_sum(lr);
    if (k == 0) a.Yqld[(size_t)n * a.T + t] = 0.5 / lr;    // gaussian.py:120 (quirk Q1)
    if (k >= K) return;
    if (miss) {
        const double* x = a.X + ((size_t)n * a.T + t) * a.DP;
        const double* c = a.C_mean + ((size_t)n * K + k) * D;
        double m = 0.0;
        for (int j = 0; j < D; ++j) m += c[j] * x[xpos(j)];
        a.Y[row + k] = m;
        a.Yvar[row + k] = 1.0 / rbar;
    } else {
        a.Y[row + k] = ob;
        a.Yvar[row + k] = 0.0;
    }
}

__global__ void __launch_bounds__(256) k_syy_missing(MissArgs a) {
    __shared__ double red[4][65];
    const int n = blockIdx.x, w = threadIdx.x >> 6, k = threadIdx.x & 63, K = a.K, T = a.T;
    const double* Y = a.Y + (size_t)n * T * K;
    const double* Yo = a.Yobs + (size_t)n * T * K;
    const double* Yv = a.Yvar + (size_t)n * T * K;
    double s = 0.0, ent = 0.0;
    for (int t = w; t < T; t += 4) {
        const bool live = k < K;
        const double y = live ? Y[(size_t)t * K + k] : 0.0, v = live ? Yv[(size_t)t * K + k] : 0.0, ob = live ? Yo[(size_t)t * K + k] : 0.0;
        const bool miss = live && !(ob == ob);
        s += y * y + v;
        const int nm = __popcll(__ballot(miss));
        if (nm == 0) continue;                          // wave-uniform
        const double lv = wave_sum(miss ? log(v) : 0.0);
        if (nm == K) ent += -0.5 * K * LN2PI - 0.5 * a.Yqld[(size_t)n * T + t] - 0.5 * K;        // gaussian.py:145-147
        else ent += 0.5 * nm * LN2PI - 0.5 * lv - 0.5 * nm;                                        // gaussian.py:148-150
    }
    red[w][k] = s;
    if (k == 0) red[w][64] = ent;
    __syncthreads();
    if (w == 0) {
        if (k < K) a.Syy[(size_t)n * K + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        if (k == 0) a.Yent[n] = red[0][64] + red[1][64] + red[2][64] + red[3][64];
    }
}

static MissArgs make_margs(pyvb_lds* h) {
    MissArgs a;
    a.Y = h->Y; a.Yobs = h->Yobs; a.Yvar = h->Yvar; a.Yqld = h->Yqld; a.X = h->X[h->cur]; a.C_mean = h->C_mean;
    a.R_a = h->R_a; a.R_b = h->R_b; a.Syy = h->Syy; a.Yent = h->Yent; a.Yq0 = nullptr; a.Yrowvar0 = nullptr;
    a.N = h->N; a.T = h->T; a.K = h->K; a.D = h->D; a.DP = h->L.DP;
    return a;
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
