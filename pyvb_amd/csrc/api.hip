// Host side of libpyvb_hip.so: handle lifetime, host<->device copies, the dependency tracking that
// decides which kernels a node-level call needs, timing, and the RCCL all-reduce of the lower bound.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <mutex>
#include <dlfcn.h>

static thread_local char g_err[512] = "";

void pyvb_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pyvb_hip_fail(hipError_t e, const char* what, const char* file, int line) {
    pyvb_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return PYVB_E_HIP;
}

TimedLaunch::TimedLaunch(pyvb_lds* h_, int k_, hipStream_t s_) : h(h_), slot(-1), s(s_ ? s_ : h_->stream) {
    if (!h->timing) return;
    if (h->pool_used == PYVB_EVENT_POOL) pyvb_timing_resolve(h);     // pool exhausted: drain (synchronises)
    slot = h->pool_used++;
    h->pool[slot].kernel = k_;
    if (hipEventRecord(h->pool[slot].e0, s) != hipSuccess) { h->timing_errors += 1; h->pool[slot].kernel = -1; }
}
TimedLaunch::~TimedLaunch() {
    if (slot >= 0 && hipEventRecord(h->pool[slot].e1, s) != hipSuccess) { h->timing_errors += 1; h->pool[slot].kernel = -1; }
}

void pyvb_timing_resolve(pyvb_lds* h) {
    for (int i = 0; i < h->pool_used; ++i) {
        float ms = 0;
        if (h->pool[i].kernel < 0) continue;                          // an event of this pair was never recorded
        if (hipEventSynchronize(h->pool[i].e1) == hipSuccess && hipEventElapsedTime(&ms, h->pool[i].e0, h->pool[i].e1) == hipSuccess) {
            h->timers[h->pool[i].kernel].total_ms += ms;
            h->timers[h->pool[i].kernel].launches += 1;
        } else {
            h->timing_errors += 1;
        }
    }
    h->pool_used = 0;
}

#define ARGCHK(cond, msg) do { if (!(cond)) { pyvb_set_error("%s", msg); return PYVB_E_ARG; } } while (0)

extern "C" {

const char* pyvb_last_error(void) { return g_err; }
int pyvb_version(void) { return 100; }

int pyvb_device_count(int* count) {
    ARGCHK(count, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return PYVB_OK;
}

static int dev_alloc(double** p, size_t n) {
    HIPCHK(hipMalloc((void**)p, n * sizeof(double)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(double)));
    return PYVB_OK;
}

int pyvb_lds_create(pyvb_lds** out, int device, int N, int T, int D, int K, int noise_kind) {
    ARGCHK(out, "out is NULL");
    ARGCHK(N >= 1, "N must be >= 1");
    ARGCHK(T >= 2, "T must be >= 2 (a chain needs X_0 and X_{T-1})");
    ARGCHK(D >= 1 && D <= 128, "latent dimension D must be in 1..128");
    ARGCHK(K >= 1 && K <= 128, "observed dimension K must be in 1..128");
    ARGCHK(noise_kind == PYVB_NOISE_DIAGONAL_GAMMA || noise_kind == PYVB_NOISE_GAMMA || noise_kind == PYVB_NOISE_WISHART, "unknown noise kind");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    ARGCHK(device >= 0 && device < ndev, "no such device");
    HIPCHK(hipSetDevice(device));
    pyvb_lds* h = new pyvb_lds();
    memset(h, 0, sizeof(*h));
    h->device = device; h->N = N; h->T = T; h->D = D; h->K = K; h->noise = noise_kind;
    h->L = make_layout(D, K);
    h->big = D > 64 || K > 64;           // the workgroup-per-replicate kernels of k_big.hip
    h->dense = noise_kind == PYVB_NOISE_WISHART;
    const Layout& L = h->L;
    int rc = PYVB_OK;
#define TRY(x) do { rc = (x); if (rc != PYVB_OK) { pyvb_lds_destroy(h); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { rc = pyvb_hip_fail(_e, #x, __FILE__, __LINE__); pyvb_lds_destroy(h); return rc; } } while (0)
    TRYHIP(hipStreamCreate(&h->stream));
    TRYHIP(hipStreamCreate(&h->side));
    TRYHIP(hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
    TRYHIP(hipEventCreateWithFlags(&h->ev_elbo, hipEventDisableTiming));
    h->pool = (EventPair*)calloc(PYVB_EVENT_POOL, sizeof(EventPair));
    for (int i = 0; i < PYVB_EVENT_POOL; ++i) { TRYHIP(hipEventCreate(&h->pool[i].e0)); TRYHIP(hipEventCreate(&h->pool[i].e1)); }
    const size_t n = (size_t)N;
    TRY(dev_alloc(&h->Y, n * T * K));
    TRY(dev_alloc(&h->Syy, n * K));
    TRY(dev_alloc(&h->X[0], n * T * L.DP));
    TRY(dev_alloc(&h->X[1], n * T * L.DP));
    TRY(dev_alloc(&h->A_mean, n * D * D));
    TRY(dev_alloc(&h->A_var, n * D * D));
    TRY(dev_alloc(&h->C_mean, n * K * D));
    TRY(dev_alloc(&h->C_var, n * D * K));
    TRY(dev_alloc(&h->Q_a, n * D)); TRY(dev_alloc(&h->Q_b, n * D));
    TRY(dev_alloc(&h->R_a, n * K)); TRY(dev_alloc(&h->R_b, n * K));
    TRY(dev_alloc(&h->qld_A, n * D)); TRY(dev_alloc(&h->qld_C, n * D));
    TRY(dev_alloc(&h->Sigma, n * 3 * D * D)); TRY(dev_alloc(&h->Sigma_new, n * 3 * D * D));
    TRY(dev_alloc(&h->qld_x, n * 3)); TRY(dev_alloc(&h->qld_x_new, n * 3));
    TRY(dev_alloc(&h->gains, n * L.gains_total));
    TRY(dev_alloc(&h->scratch, h->big ? n * 2 * L.DP * L.DP : n * 2 * D * D));
    TRY(dev_alloc(&h->zeros, 128));
    TRY(dev_alloc(&h->U, n * T * L.DP));                        // the c_t cache between a forward sweep and the backward one behind it
    TRYHIP(hipMalloc((void**)&h->warm, n * 2 * sizeof(int)));
    TRYHIP(hipMemset(h->warm, 0, n * 2 * sizeof(int)));
    // time chunks of the statistics kernel: enough wavefronts to fill the chip when N is small
    int nchunk = (1024 + N - 1) / N;
    if (nchunk > 32) nchunk = 32;
    if (nchunk > (T + 15) / 16) nchunk = (T + 15) / 16;
    if (nchunk < 1) nchunk = 1;
    int clen = (T + nchunk - 1) / nchunk;
    clen = (clen + 3) & ~3;
    nchunk = (T + clen - 1) / clen;
    h->nchunk = nchunk; h->chunk_len = clen;
    TRY(dev_alloc(&h->stats, n * nchunk * L.stats_total));
    TRY(dev_alloc(&h->mom, n * ((size_t)3 * D * D + (size_t)K * D + D)));
    // The sweeps may split the time axis over W wavefronts per replicate (k_sweep.hip).  A wavefront takes
    // ceil(part/16) + J steps (J ~ 32 warm-up steps), the chip runs 1024 of them at a time: W minimises
    // rounds x steps, with parts of at least 64 nodes.  At N >= 1024 that is W = 1.
    // (The 128-wide class: a part is a workgroup of four wavefronts, one per CU at a time.)
    h->W = 1;
    {
        const long Tint = T - 2, slots = h->big ? 256 : 1024;
        double best = 1e300;
        for (int W = 1; W <= 128 && (W == 1 || Tint / W >= 64); ++W) {
            const long part = (((Tint + W - 1) / W) + 15) & ~15L;
            const double cost = (double)(((long)N * W + slots - 1) / slots) * (double)((part + 15) / 16 + 32);
            if (cost < best * 0.97) { best = cost; h->W = W; }      // prefer fewer wavefronts unless clearly better
        }
    }
    TRY(dev_alloc(&h->sxx, h->big ? 8 : n * (size_t)h->W * L.DP * L.DP));        // (the 128-wide sweeps fuse no statistics)
    if (h->big && h->W > 1) TRY(dev_alloc(&h->U2, n * T * L.DP));
    TRY(dev_alloc(&h->trash, n * (size_t)(h->big ? h->W : 1) * 512));      // 128-wide class: per (replicate, part of the time axis)
    TRY(dev_alloc(&h->resQ, n * D)); TRY(dev_alloc(&h->resR, n * K));
    TRY(dev_alloc(&h->elbo, n * 6)); TRY(dev_alloc(&h->elbo_sum, 8));
    TRY(dev_alloc(&h->elbo_hist, (size_t)PYVB_ELBO_HISTORY * 8));
    TRYHIP(hipMalloc((void**)&h->status, sizeof(int)));
    TRYHIP(hipMemset(h->status, 0, sizeof(int)));
    // priors block: x0_mean D, x0_prec D*D, A_pm D*D, A_pp D*D, C_pm K*D, C_pp D*K, Q_a0 D, Q_b0 D, R_a0 K, R_b0 K
    size_t pn = (size_t)D + 3 * (size_t)D * D + 2 * (size_t)K * D + 2 * (size_t)D + 2 * (size_t)K + (size_t)D * D + (size_t)K * D + 2 * (size_t)D
                + (size_t)D * D + (size_t)K * K;
    TRY(dev_alloc(&h->pri_block, pn));
    double* p = h->pri_block;
    h->pri.x0_mean = p; p += D; h->pri.x0_prec = p; p += D * D;
    h->pri.A_pm = p; p += D * D; h->pri.A_pp = p; p += D * D;
    h->pri.C_pm = p; p += K * D; h->pri.C_pp = p; p += D * K;
    h->pri.Q_a0 = p; p += D; h->pri.Q_b0 = p; p += D; h->pri.R_a0 = p; p += K; h->pri.R_b0 = p; p += K;
    h->pri.A_obs = p; p += D * D; h->pri.C_obs = p; p += K * D;
    h->pri.A_pld = p; p += D; h->pri.C_pld = p; p += D;
    h->pri.Q_w0 = p; p += D * D; h->pri.R_w0 = p; p += K * K;
    if (h->dense) {
        TRY(dev_alloc(&h->Q_w, n * D * D)); TRY(dev_alloc(&h->R_w, n * K * K));
        TRY(dev_alloc(&h->Qbar, n * D * D)); TRY(dev_alloc(&h->Rbar, n * K * K));
        TRY(dev_alloc(&h->lnd, n * 4));
        TRY(dev_alloc(&h->QA, n * D * D)); TRY(dev_alloc(&h->RC, n * K * D));
        TRY(dev_alloc(&h->trA, n * D)); TRY(dev_alloc(&h->trC, n * D));
        TRY(dev_alloc(&h->A_cov, n * D * cov_stride(D))); TRY(dev_alloc(&h->C_cov, n * D * cov_stride(K)));
        TRY(dev_alloc(&h->SyyF, n * K * K));
        TRY(dev_alloc(&h->RQ, n * D * D)); TRY(dev_alloc(&h->RR, n * K * K));
        TRY(dev_alloc(&h->SG, h->big ? n * 2 * 128 * 128 : n * 2 * 64 * 64));
        TRY(dev_alloc(&h->ldm, n * 2 * D));
    }
    TRYHIP(hipMemset(h->pri.A_obs, 0xFF, ((size_t)D * D + (size_t)K * D) * sizeof(double)));     // all-ones bytes = NaN = nothing observed
    h->fresh = (unsigned char*)calloc(T, 1);
    h->world = 1;
    // q_ln_det is undefined until a node has been updated (the reference raises AttributeError)
    {
        std::vector<double> nanv(n * (size_t)(D > 3 ? D : 3), NAN);
        TRYHIP(hipMemcpy(h->qld_x, nanv.data(), n * 3 * sizeof(double), hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(h->qld_A, nanv.data(), n * D * sizeof(double), hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(h->qld_C, nanv.data(), n * D * sizeof(double), hipMemcpyHostToDevice));
    }
#undef TRY
#undef TRYHIP
    *out = h;
    return PYVB_OK;
}

int pyvb_lds_destroy(pyvb_lds* h) {
    if (!h) return PYVB_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side) (void)hipStreamSynchronize(h->side);
    pyvb_lds_comm_destroy(h);
    if (h->elbo_hist) (void)hipFree(h->elbo_hist);
    if (h->ev_params) (void)hipEventDestroy(h->ev_params);
    if (h->ev_elbo) (void)hipEventDestroy(h->ev_elbo);
    if (h->side) (void)hipStreamDestroy(h->side);
    double* bufs[] = {h->Y, h->Syy, h->X[0], h->X[1], h->A_mean, h->A_var, h->C_mean, h->C_var, h->Q_a, h->Q_b, h->R_a, h->R_b,
                      h->qld_A, h->qld_C, h->Sigma, h->Sigma_new, h->qld_x, h->qld_x_new, h->gains, h->scratch, h->stats,
                      h->resQ, h->resR, h->elbo, h->elbo_sum, h->pri_block, h->trash, h->zeros, h->mom, h->sxx, h->U,
                      h->Q_w, h->R_w, h->Qbar, h->Rbar, h->lnd, h->QA, h->RC, h->trA, h->trC, h->A_cov, h->C_cov, h->SyyF, h->RQ, h->RR, h->SG, h->ldm,
                      h->Yobs, h->Yvar, h->Yqld, h->Yent, h->Yld, h->YcovS, h->U2};
    for (double* b : bufs) if (b) (void)hipFree(b);
    if (h->warm) (void)hipFree(h->warm);
    if (h->status) (void)hipFree(h->status);
    if (h->pool) {
        for (int i = 0; i < PYVB_EVENT_POOL; ++i) { if (h->pool[i].e0) (void)hipEventDestroy(h->pool[i].e0); if (h->pool[i].e1) (void)hipEventDestroy(h->pool[i].e1); }
        free(h->pool);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    free(h->fresh);
    delete h;
    return PYVB_OK;
}

static void states_changed(pyvb_lds* h);
static int join_elbo(pyvb_lds* h);
// Every entry point but pyvb_lds_iterate first lets the main stream wait for a lower-bound evaluation that the last
// pyvb_lds_iterate may have left in flight on the side stream (it reads states and parameters).
#define ENTER_RAW(h) do { ARGCHK(h, "handle is NULL"); HIPCHK(hipSetDevice((h)->device)); } while (0)
#define ENTER(h) do { ENTER_RAW(h); if ((h)->elbo_in_flight) { int _rc = join_elbo(h); if (_rc) return _rc; } } while (0)

static int h2d(pyvb_lds* h, double* dst, const double* src, size_t n) {
    if (!src) return PYVB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return PYVB_OK;
}
static int d2h(pyvb_lds* h, double* dst, const double* src, size_t n) {
    if (!dst) return PYVB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return PYVB_OK;
}

static void params_changed(pyvb_lds* h) {
    if (h->gains_valid && h->fresh_count != 0 && h->fresh_count != h->T) h->mixed_cov = true;
    h->gains_valid = false;
    h->u_valid = false;
}
static void states_changed(pyvb_lds* h) { h->stats_valid = false; h->resQ_valid = false; h->resR_valid = false; h->sg_valid[0] = h->sg_valid[1] = false; }

// ln det of a symmetric positive definite matrix (Constant.lndet, node.py:301-302)
static int host_lndet(const double* Ain, int D, double* out) {
    std::vector<double> A(Ain, Ain + (size_t)D * D);
    double s = 0.0;
    for (int j = 0; j < D; ++j) {
        double piv = A[j * D + j];
        for (int k = 0; k < j; ++k) piv -= A[j * D + k] * A[j * D + k];
        if (!(piv > 0.0)) return PYVB_E_LINALG;
        double d = sqrt(piv);
        A[j * D + j] = d; s += log(d);
        for (int i = j + 1; i < D; ++i) {
            double v = A[i * D + j];
            for (int k = 0; k < j; ++k) v -= A[i * D + k] * A[j * D + k];
            A[i * D + j] = v / d;
        }
    }
    *out = 2.0 * s;
    return PYVB_OK;
}

int pyvb_lds_set_priors(pyvb_lds* h, const double* x0_mean, const double* x0_prec,
                        const double* A_pm, const double* A_pp, const double* C_pm, const double* C_pp,
                        const double* Q_a0, const double* Q_b0, const double* R_a0, const double* R_b0) {
    ENTER(h);
    ARGCHK(x0_mean && x0_prec && A_pm && A_pp && C_pm && C_pp, "the priors of X_0 and of the columns are required");
    ARGCHK(h->dense || (Q_a0 && Q_b0 && R_a0 && R_b0), "the Gamma priors of Q and R are required");
    const int D = h->D, K = h->K, T = h->T, N = h->N;
    for (int i = 0; i < D * D; ++i) ARGCHK(A_pp[i] > 0.0, "A_prior_prec must be positive");
    for (int i = 0; i < D * K; ++i) ARGCHK(C_pp[i] > 0.0, "C_prior_prec must be positive");
    if (host_lndet(x0_prec, D, &h->pri.x0_lndet) != PYVB_OK) { pyvb_set_error("x0_prec is not positive definite"); return PYVB_E_LINALG; }
    int rc;
    if ((rc = h2d(h, h->pri.x0_mean, x0_mean, D))) return rc;
    if ((rc = h2d(h, h->pri.x0_prec, x0_prec, (size_t)D * D))) return rc;
    if ((rc = h2d(h, h->pri.A_pm, A_pm, (size_t)D * D))) return rc;
    if ((rc = h2d(h, h->pri.A_pp, A_pp, (size_t)D * D))) return rc;
    if ((rc = h2d(h, h->pri.C_pm, C_pm, (size_t)K * D))) return rc;
    if ((rc = h2d(h, h->pri.C_pp, C_pp, (size_t)D * K))) return rc;
    {   // ln det of the diagonal prior precision of every column (Constant.lndet of the parents, node.py:301-302)
        std::vector<double> ld(2 * (size_t)D, 0.0);
        for (int i = 0; i < D; ++i) {
            for (int k = 0; k < D; ++k) ld[i] += log(A_pp[(size_t)i * D + k]);
            for (int k = 0; k < K; ++k) ld[D + i] += log(C_pp[(size_t)i * K + k]);
        }
        HIPCHK(hipMemcpyAsync(h->pri.A_pld, ld.data(), 2 * (size_t)D * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (h->dense) {         // the Wishart priors come through pyvb_lds_set_wishart_priors
        HIPCHK(hipStreamSynchronize(h->stream));
        params_changed(h);
        h->resQ_valid = h->resR_valid = false;
        return PYVB_OK;
    }
    if ((rc = h2d(h, h->pri.Q_a0, Q_a0, D))) return rc;
    if ((rc = h2d(h, h->pri.Q_b0, Q_b0, D))) return rc;
    if ((rc = h2d(h, h->pri.R_a0, R_a0, K))) return rc;
    if ((rc = h2d(h, h->pri.R_b0, R_b0, K))) return rc;
    // qa is fixed by the graph: update_a, nodes_todo.py:125-128 (Gamma: +0.5*child.shape[0] per child)
    // and :183-186 (DiagonalGamma: +0.5 per child); Q has T-1 children X_1.., R has T children Y_t
    std::vector<double> qa((size_t)N * D), ra((size_t)N * K);
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < D; ++k)
            qa[(size_t)n * D + k] = (h->noise == PYVB_NOISE_GAMMA) ? Q_a0[0] + 0.5 * D * (T - 1) : Q_a0[k] + 0.5 * (T - 1);
        for (int k = 0; k < K; ++k)
            ra[(size_t)n * K + k] = (h->noise == PYVB_NOISE_GAMMA) ? R_a0[0] + 0.5 * K * T : R_a0[k] + 0.5 * T;
    }
    if ((rc = h2d(h, h->Q_a, qa.data(), qa.size()))) return rc;
    if ((rc = h2d(h, h->R_a, ra.data(), ra.size()))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    params_changed(h);
    h->resQ_valid = h->resR_valid = false;
    h->sg_valid[0] = h->sg_valid[1] = false;
    return PYVB_OK;
}

int pyvb_lds_set_wishart_priors(pyvb_lds* h, double Q_v0, const double* Q_w0, double R_v0, const double* R_w0) {
    ENTER(h);
    ARGCHK(h->dense, "the handle was not created with PYVB_NOISE_WISHART");
    ARGCHK(Q_w0 && R_w0, "Q_w0 and R_w0 are required");
    const int D = h->D, K = h->K, T = h->T, N = h->N;
    if (host_lndet(Q_w0, D, &h->pri.Q_w0_lndet) != PYVB_OK || host_lndet(R_w0, K, &h->pri.R_w0_lndet) != PYVB_OK) {
        pyvb_set_error("a Wishart prior w0 is not symmetric positive definite");
        return PYVB_E_LINALG;
    }
    h->pri.Q_a0_host = Q_v0; h->pri.R_a0_host = R_v0;
    int rc;
    if ((rc = h2d(h, h->pri.Q_w0, Q_w0, (size_t)D * D))) return rc;
    if ((rc = h2d(h, h->pri.R_w0, R_w0, (size_t)K * K))) return rc;
    // qv is fixed by the graph: update_v, nodes_todo.py:224-227 (+0.5 per child); Q has T-1 children, R has T
    std::vector<double> qa((size_t)N * D, Q_v0 + 0.5 * (T - 1)), ra((size_t)N * K, R_v0 + 0.5 * T);
    if ((rc = h2d(h, h->Q_a, qa.data(), qa.size()))) return rc;
    if ((rc = h2d(h, h->R_a, ra.data(), ra.size()))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->expect_valid = false;
    params_changed(h);
    h->resQ_valid = h->resR_valid = false;
    return PYVB_OK;
}

int pyvb_lds_set_wishart_state(pyvb_lds* h, const double* Q_w, const double* R_w) {
    ENTER(h);
    ARGCHK(h->dense, "the handle was not created with PYVB_NOISE_WISHART");
    int rc;
    if ((rc = h2d(h, h->Q_w, Q_w, (size_t)h->N * h->D * h->D))) return rc;
    if ((rc = h2d(h, h->R_w, R_w, (size_t)h->N * h->K * h->K))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->expect_valid = false;
    params_changed(h);
    return PYVB_OK;
}

int pyvb_lds_get_wishart_state(pyvb_lds* h, double* Q_v, double* Q_w, double* R_v, double* R_w) {
    ENTER(h);
    ARGCHK(h->dense, "the handle was not created with PYVB_NOISE_WISHART");
    const size_t N = h->N, D = h->D, K = h->K;
    int rc;
    std::vector<double> qa(Q_v ? N * D : 0), ra(R_v ? N * K : 0);       // qv is stored once per dimension, like the Gamma kinds' qa
    if (Q_v && (rc = d2h(h, qa.data(), h->Q_a, N * D))) return rc;
    if (R_v && (rc = d2h(h, ra.data(), h->R_a, N * K))) return rc;
    if ((rc = d2h(h, Q_w, h->Q_w, N * D * D))) return rc;
    if ((rc = d2h(h, R_w, h->R_w, N * K * K))) return rc;
    if ((rc = pyvb_lds_sync(h))) return rc;
    for (size_t n = 0; Q_v && n < N; ++n) Q_v[n] = qa[n * D];
    for (size_t n = 0; R_v && n < N; ++n) R_v[n] = ra[n * K];
    return PYVB_OK;
}

// dense [N][D][rows][rows] on the host <-> the tiled device storage, through a device staging buffer of at most 64 MB
static int column_cov_io(pyvb_lds* h, int which, double* host, bool to_device) {
    if (!host) return PYVB_OK;
    const size_t D = h->D, rows = which == 0 ? h->D : h->K, per = D * rows * rows;
    size_t slice = ((size_t)8 << 20) / per;                 // replicates per slice
    if (slice < 1) slice = 1;
    if (slice > (size_t)h->N) slice = h->N;
    double* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, slice * per * sizeof(double)));
    int rc = PYVB_OK;
    for (size_t n0 = 0; n0 < (size_t)h->N && rc == PYVB_OK; n0 += slice) {
        const size_t cnt = (n0 + slice <= (size_t)h->N) ? slice : (size_t)h->N - n0;
        hipError_t e = hipSuccess;
        if (to_device) {
            e = hipMemcpyAsync(stage, host + n0 * per, cnt * per * sizeof(double), hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) rc = launch_cov_convert(h, which, stage, (int)n0, (int)cnt, 1);
        } else {
            rc = launch_cov_convert(h, which, stage, (int)n0, (int)cnt, 0);
            if (rc == PYVB_OK) e = hipMemcpyAsync(host + n0 * per, stage, cnt * per * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);       // the staging buffer is reused by the next slice
        if (e != hipSuccess && rc == PYVB_OK) rc = pyvb_hip_fail(e, "column covariance transfer", __FILE__, __LINE__);
    }
    (void)hipFree(stage);
    return rc;
}

int pyvb_lds_set_column_cov(pyvb_lds* h, const double* A_cov, const double* C_cov) {
    ENTER(h);
    ARGCHK(h->dense, "dense column covariances exist with PYVB_NOISE_WISHART only");
    int rc;
    if ((rc = column_cov_io(h, 0, const_cast<double*>(A_cov), true))) return rc;
    if ((rc = column_cov_io(h, 1, const_cast<double*>(C_cov), true))) return rc;
    if ((A_cov || C_cov) && (rc = launch_cov_to_colvar(h))) return rc;      // the diagonals (what the lower bound reads) follow
    HIPCHK(hipStreamSynchronize(h->stream));
    params_changed(h);
    h->resQ_valid = h->resR_valid = false;
    h->sg_valid[0] = h->sg_valid[1] = false;
    return PYVB_OK;
}

int pyvb_lds_get_column_cov(pyvb_lds* h, double* A_cov, double* C_cov) {
    ENTER(h);
    ARGCHK(h->dense, "dense column covariances exist with PYVB_NOISE_WISHART only");
    int rc;
    if ((rc = column_cov_io(h, 0, A_cov, false))) return rc;
    if ((rc = column_cov_io(h, 1, C_cov, false))) return rc;
    return pyvb_lds_sync(h);
}

int pyvb_lds_set_column_observations(pyvb_lds* h, const double* A_obs, const double* C_obs) {
    ENTER(h);
    if (h->dense && h->big) {
        pyvb_set_error("with Wishart noise, known entries of A / C are served for D, K <= 64 only (k_wishart_big.hip)");
        return PYVB_E_UNSUPPORTED;
    }
    int rc;
    if ((rc = h2d(h, h->pri.A_obs, A_obs, (size_t)h->D * h->D))) return rc;
    if ((rc = h2d(h, h->pri.C_obs, C_obs, (size_t)h->K * h->D))) return rc;
    if ((rc = launch_observe(h))) return rc;
    if (h->dense) {
        if ((rc = launch_cov_observe(h))) return rc;
        h->sg_valid[0] = h->sg_valid[1] = false;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    params_changed(h);
    h->resQ_valid = h->resR_valid = false;
    return PYVB_OK;
}

static int ensure_expect(pyvb_lds* h);
// Wishart noise, after the means of the outputs changed: sum_t qmu qmu^T, and the entropy terms of the rows that are not
// fully observed (diag_cov: they still carry their diagonal initial covariances, whose sum is formed here too)
static int outputs_changed_dense(pyvb_lds* h, int diag_cov) {
    int rc;
    if ((rc = launch_syy_full(h))) return rc;
    return launch_missing_ent_dense(h, diag_cov);
}

int pyvb_lds_set_observations(pyvb_lds* h, const double* Y) {
    ENTER(h);
    ARGCHK(Y, "Y is NULL");
    const size_t n = (size_t)h->N * h->T * h->K;
    bool missing = false;
    for (size_t i = 0; i < n && !missing; ++i) missing = Y[i] != Y[i];
    int rc;
    if (missing && h->dense && h->big) {
        pyvb_set_error("with Wishart noise, outputs that hold NaN are served for D, K <= 64 only (k_wishart_big.hip)");
        return PYVB_E_UNSUPPORTED;
    }
    if (missing) {
        if (!h->Yobs) {
            if ((rc = dev_alloc(&h->Yobs, n))) return rc;
            if ((rc = dev_alloc(&h->Yvar, n))) return rc;
            if ((rc = dev_alloc(&h->Yqld, (size_t)h->N * h->T))) return rc;
            if ((rc = dev_alloc(&h->Yent, h->N))) return rc;
            if (h->dense) {
                if ((rc = dev_alloc(&h->Yld, (size_t)h->N * h->T))) return rc;
                if ((rc = dev_alloc(&h->YcovS, (size_t)h->N * h->K * h->K))) return rc;
            }
        }
        if ((rc = h2d(h, h->Yobs, Y, n))) return rc;
        h->has_missing = true;
        // rows with NaN start as N(0, I) until pyvb_lds_set_output_state says otherwise
        if ((rc = launch_missing_init(h, nullptr, nullptr))) return rc;
        if ((rc = h->dense ? outputs_changed_dense(h, 1) : launch_syy_missing(h))) return rc;
    } else {
        h->has_missing = false;
        if ((rc = h2d(h, h->Y, Y, n))) return rc;
        if ((rc = launch_syy(h))) return rc;
        if (h->dense && (rc = launch_syy_full(h))) return rc;
    }
    h->u_valid = false;
    HIPCHK(hipStreamSynchronize(h->stream));
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_set_output_state(pyvb_lds* h, const double* Yq, const double* Yrowvar) {
    ENTER(h);
    ARGCHK(h->has_missing, "the observations hold no NaN: every output is observed");
    ARGCHK(Yq && Yrowvar, "Yq and Yrowvar are required");
    const size_t n = (size_t)h->N * h->T * h->K;
    // staged through buffers that are free until the next sweep / statistics pass
    double* dq = h->U;                  // [N][T][DP] >= [N][T][K]?  not for K > DP: use a temporary then
    double* tmp = nullptr;
    if ((size_t)h->K > (size_t)h->L.DP) { HIPCHK(hipMalloc((void**)&tmp, n * sizeof(double))); dq = tmp; }
    double* dv = h->X[1 - h->cur];      // [N][T][DP] >= [N][T]
    int rc;
    if ((rc = h2d(h, dq, Yq, n)) || (rc = h2d(h, dv, Yrowvar, (size_t)h->N * h->T)) ||
        (rc = launch_missing_init(h, dq, dv)) || (rc = h->dense ? outputs_changed_dense(h, 1) : launch_syy_missing(h))) {
        if (tmp) { (void)hipStreamSynchronize(h->stream); (void)hipFree(tmp); }
        return rc;
    }
    const hipError_t se = hipStreamSynchronize(h->stream);
    if (tmp) (void)hipFree(tmp);
    HIPCHK(se);
    h->u_valid = false;
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_get_outputs(pyvb_lds* h, double* Yq, double* Yvar, double* Yqld) {
    ENTER(h);
    const size_t n = (size_t)h->N * h->T * h->K;
    int rc;
    if ((rc = d2h(h, Yq, h->Y, n))) return rc;
    if (Yvar) {
        if (h->has_missing) { if ((rc = d2h(h, Yvar, h->Yvar, n))) return rc; }
        else memset(Yvar, 0, n * sizeof(double));
    }
    if (Yqld) {
        if (h->has_missing) { if ((rc = d2h(h, Yqld, h->Yqld, (size_t)h->N * h->T))) return rc; }
        else for (size_t i = 0; i < (size_t)h->N * h->T; ++i) Yqld[i] = NAN;
    }
    return pyvb_lds_sync(h);
}

int pyvb_lds_update_Y(pyvb_lds* h) {
    ENTER(h);
    if (!h->has_missing) return PYVB_OK;            // observed nodes never update (gaussian.py:109-110)
    int rc;
    if (h->dense) {
        if ((rc = ensure_expect(h))) return rc;     // E[R] and its log-determinant
        if ((rc = launch_impute_dense(h))) return rc;
        if ((rc = outputs_changed_dense(h, 0))) return rc;
    } else {
        if ((rc = launch_impute(h))) return rc;
        if ((rc = launch_syy_missing(h))) return rc;
    }
    h->u_valid = false;                             // c_t = F mu + G y_t was formed with the old y_t
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_set_state(pyvb_lds* h, const double* X, const double* A_mean, const double* A_colvar,
                       const double* C_mean, const double* C_colvar, const double* Q_b, const double* R_b) {
    ENTER(h);
    const size_t N = h->N, T = h->T, D = h->D, K = h->K;
    int rc;
    if (X) {    // the other buffer is free between sweeps: stage the API layout there, then permute
        if ((rc = h2d(h, h->X[1 - h->cur], X, N * T * D))) return rc;
        if ((rc = launch_permute(h, h->X[1 - h->cur], h->X[h->cur], 1))) return rc;
    }
    if ((rc = h2d(h, h->A_mean, A_mean, N * D * D))) return rc;
    if ((rc = h2d(h, h->A_var, A_colvar, N * D * D))) return rc;
    if ((rc = h2d(h, h->C_mean, C_mean, N * K * D))) return rc;
    if ((rc = h2d(h, h->C_var, C_colvar, N * D * K))) return rc;
    if ((rc = h2d(h, h->Q_b, Q_b, N * D))) return rc;
    if ((rc = h2d(h, h->R_b, R_b, N * K))) return rc;
    if (h->dense && (A_colvar || C_colvar)) {       // diagonal initial covariances
        if ((rc = launch_colvar_to_cov(h))) return rc;
        h->sg_valid[0] = h->sg_valid[1] = false;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (X) { states_changed(h); h->u_valid = false; h->sxx_valid = false; }
    if (A_mean || A_colvar || C_mean || C_colvar || Q_b || R_b) { params_changed(h); h->resQ_valid = h->resR_valid = false; }
    return PYVB_OK;
}

int pyvb_lds_get_state(pyvb_lds* h, double* X, double* A_mean, double* A_colvar, double* C_mean, double* C_colvar,
                       double* Q_a, double* Q_b, double* R_a, double* R_b) {
    ENTER(h);
    const size_t N = h->N, T = h->T, D = h->D, K = h->K;
    int rc;
    if (X) {
        if ((rc = launch_permute(h, h->X[h->cur], h->X[1 - h->cur], 0))) return rc;
        if ((rc = d2h(h, X, h->X[1 - h->cur], N * T * D))) return rc;
    }
    if ((rc = d2h(h, A_mean, h->A_mean, N * D * D))) return rc;
    if ((rc = d2h(h, A_colvar, h->A_var, N * D * D))) return rc;
    if ((rc = d2h(h, C_mean, h->C_mean, N * K * D))) return rc;
    if ((rc = d2h(h, C_colvar, h->C_var, N * D * K))) return rc;
    if ((rc = d2h(h, Q_a, h->Q_a, N * D))) return rc;
    if ((rc = d2h(h, Q_b, h->Q_b, N * D))) return rc;
    if ((rc = d2h(h, R_a, h->R_a, N * K))) return rc;
    if ((rc = d2h(h, R_b, h->R_b, N * K))) return rc;
    return pyvb_lds_sync(h);
}

int pyvb_lds_get_posterior_classes(pyvb_lds* h, double* Sigma, double* qld_x) {
    ENTER(h);
    const size_t N = h->N, D = h->D;
    int rc;
    if ((rc = d2h(h, Sigma, h->Sigma, N * 3 * D * D))) return rc;
    if ((rc = d2h(h, qld_x, h->qld_x, N * 3))) return rc;
    return pyvb_lds_sync(h);
}

int pyvb_lds_set_posterior_classes(pyvb_lds* h, const double* Sigma, const double* qld_x) {
    ENTER(h);
    ARGCHK(Sigma, "Sigma is NULL");
    const size_t N = h->N, D = h->D;
    int rc;
    if ((rc = h2d(h, h->Sigma, Sigma, N * 3 * D * D))) return rc;
    if ((rc = h2d(h, h->qld_x, qld_x, N * 3))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->classes_valid = true;
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_get_column_qld(pyvb_lds* h, double* qld_A, double* qld_C) {
    ENTER(h);
    int rc;
    if ((rc = d2h(h, qld_A, h->qld_A, (size_t)h->N * h->D))) return rc;
    if ((rc = d2h(h, qld_C, h->qld_C, (size_t)h->N * h->D))) return rc;
    return pyvb_lds_sync(h);
}

int pyvb_lds_get_time_split(pyvb_lds* h, int* W) {
    ENTER(h);
    ARGCHK(W, "W is NULL");
    *W = h->W;
    return PYVB_OK;
}

int pyvb_lds_set_time_split(pyvb_lds* h, int W) {
    ENTER(h);
    ARGCHK(W >= 1 && W <= 128, "W must be in 1..128");
    ARGCHK(W == 1 || (h->T - 2) / W >= 16, "parts of fewer than 16 nodes");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (W != h->W) {
        double* p = nullptr;
        int rc = dev_alloc(&p, h->big ? 8 : (size_t)h->N * W * h->L.DP * h->L.DP);
        if (rc) return rc;
        if (h->big && W > 1 && !h->U2 && (rc = dev_alloc(&h->U2, (size_t)h->N * h->T * h->L.DP))) { (void)hipFree(p); return rc; }
        if (h->big) {
            double* tr = nullptr;
            if ((rc = dev_alloc(&tr, (size_t)h->N * W * 512))) { (void)hipFree(p); return rc; }
            (void)hipFree(h->trash);
            h->trash = tr;
        }
        (void)hipFree(h->sxx);
        h->sxx = p; h->W = W;
        h->sxx_valid = false; h->u_valid = false;
        states_changed(h);
    }
    return PYVB_OK;
}

int pyvb_lds_get_warmup(pyvb_lds* h, int* warm) {
    ENTER(h);
    ARGCHK(warm, "warm is NULL");
    HIPCHK(hipMemcpyAsync(warm, h->warm, (size_t)h->N * 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return pyvb_lds_sync(h);
}

// ---- dependency tracking -------------------------------------------------------------------
// gains (k_prep) depend on the parameter posteriors.  The posterior covariance classes that the
// statistics use are those of the X_t's LAST update; they switch to the freshly prepared ones
// once every X_t has been updated under the current parameters.
static int ensure_expect(pyvb_lds* h) {        // E[Q], E[R] of the current Wishart posteriors
    if (h->expect_valid) return PYVB_OK;
    int rc = launch_wexpect(h);
    if (rc) return rc;
    h->expect_valid = true;
    return PYVB_OK;
}

static int ensure_gains(pyvb_lds* h) {
    if (h->gains_valid) return PYVB_OK;
    int rc;
    if (h->dense) {
        if ((rc = ensure_expect(h))) return rc;
        if ((rc = launch_dense_pre(h))) return rc;
    }
    if ((rc = launch_prep(h))) return rc;
    h->gains_valid = true;
    memset(h->fresh, 0, h->T);
    h->fresh_count = 0;
    return PYVB_OK;
}

static void adopt_classes(pyvb_lds* h) {
    double* t = h->Sigma; h->Sigma = h->Sigma_new; h->Sigma_new = t;
    t = h->qld_x; h->qld_x = h->qld_x_new; h->qld_x_new = t;
    h->classes_valid = true;
}

static void mark_all_fresh(pyvb_lds* h) {
    if (h->fresh_count < h->T) {
        // Sigma_new holds the classes of the current parameters exactly when some node is not fresh yet
        adopt_classes(h);
        memset(h->fresh, 1, h->T);
        h->fresh_count = h->T;
    }
    h->mixed_cov = false;
}

static int ensure_stats(pyvb_lds* h) {
    if (h->stats_valid) return PYVB_OK;
    if (!h->classes_valid) {
        // the reference's X_t start with individual random covariances (gaussian.py:70-72); the three-class form the
        // statistics use exists once every X_t has been updated, or once the caller has supplied the classes
        pyvb_set_error("the posterior covariances of the X_t are undefined before the first complete sweep "
                       "(run pyvb_lds_sweep, or give them with pyvb_lds_set_posterior_classes)");
        return PYVB_E_STALE;
    }
    if (h->mixed_cov || (h->gains_valid && h->fresh_count != 0 && h->fresh_count != h->T)) {
        pyvb_set_error("%d of %d X_t were updated since the parameters changed: their covariances differ; sweep all states first",
                       h->fresh_count, h->T);
        return PYVB_E_STALE;
    }
    int rc = launch_stats(h, !h->sxx_valid);
    if (rc) return rc;
    if ((rc = launch_moments(h, h->sxx_valid))) return rc;
    h->stats_valid = true;
    return PYVB_OK;
}

static int ensure_resid(pyvb_lds* h, int which) {
    bool& valid = which == 0 ? h->resQ_valid : h->resR_valid;
    if (valid) return PYVB_OK;
    int rc = ensure_stats(h);
    if (rc) return rc;
    if ((rc = h->dense ? launch_wresid(h, which, 0) : launch_resid(h, which))) return rc;
    valid = true;
    return PYVB_OK;
}

static int sweep(pyvb_lds* h, int direction, bool keep_x) {
    int rc = ensure_gains(h);
    if (rc) return rc;
    if ((rc = launch_sweep(h, direction, keep_x))) return rc;
    mark_all_fresh(h);
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_sweep(pyvb_lds* h, int direction) {
    ENTER(h);
    ARGCHK(direction == PYVB_FORWARD || direction == PYVB_BACKWARD, "direction must be PYVB_FORWARD or PYVB_BACKWARD");
    return sweep(h, direction, true);
}

int pyvb_lds_update_x(pyvb_lds* h, int t) {
    ENTER(h);
    ARGCHK(t >= 0 && t < h->T, "t out of range");
    int rc = ensure_gains(h);
    if (rc) return rc;
    if ((rc = launch_step(h, t))) return rc;
    h->u_valid = false;
    h->sxx_valid = false;
    if (!h->fresh[t]) {
        h->fresh[t] = 1;
        if (++h->fresh_count == h->T) { adopt_classes(h); h->mixed_cov = false; }
    }
    states_changed(h);
    return PYVB_OK;
}

int pyvb_lds_update_columns(pyvb_lds* h, int which, int col_begin, int col_end) {
    ENTER(h);
    ARGCHK(which == 0 || which == 1, "which must be 0 (A) or 1 (C)");
    ARGCHK(col_begin >= 0 && col_begin < col_end && col_end <= h->D, "bad column range");
    int rc = ensure_stats(h);
    if (rc) return rc;
    if (h->dense) {
        if ((rc = ensure_expect(h))) return rc;
        if ((rc = launch_cols_dense(h, which, col_begin, col_end))) return rc;
    } else if ((rc = launch_cols(h, which, col_begin, col_end))) return rc;
    params_changed(h);
    if (which == 0) h->resQ_valid = false; else h->resR_valid = false;
    return PYVB_OK;
}

int pyvb_lds_update_A(pyvb_lds* h) { ARGCHK(h, "handle is NULL"); return pyvb_lds_update_columns(h, 0, 0, h->D); }
int pyvb_lds_update_C(pyvb_lds* h) { ARGCHK(h, "handle is NULL"); return pyvb_lds_update_columns(h, 1, 0, h->D); }

static int update_noise(pyvb_lds* h, int which) {
    int rc;
    if (h->dense) {     // Wishart.update: the residual matrix and qw = w0 + it in one launch
        if ((rc = ensure_stats(h))) return rc;
        if ((rc = launch_wresid(h, which, 1))) return rc;
        (which == 0 ? h->resQ_valid : h->resR_valid) = true;
        h->expect_valid = false;
    } else {
        if ((rc = ensure_resid(h, which))) return rc;
        if ((rc = launch_noise(h, which))) return rc;
    }
    params_changed(h);
    return PYVB_OK;
}

int pyvb_lds_update_Q(pyvb_lds* h) { ENTER(h); return update_noise(h, 0); }
int pyvb_lds_update_R(pyvb_lds* h) { ENTER(h); return update_noise(h, 1); }

int pyvb_lds_elbo(pyvb_lds* h) {
    ENTER(h);
    int rc;
    if ((rc = ensure_resid(h, 0))) return rc;
    if ((rc = ensure_resid(h, 1))) return rc;
    if (h->dense) {
        if ((rc = ensure_expect(h))) return rc;
        return launch_elbo_dense(h);
    }
    return launch_elbo(h);
}

int pyvb_lds_get_elbo(pyvb_lds* h, double* parts) {
    ENTER(h);
    ARGCHK(parts, "parts is NULL");
    int rc = d2h(h, parts, h->elbo, (size_t)h->N * 6);
    if (rc) return rc;
    return pyvb_lds_sync(h);
}

// The main stream may not overwrite what a lower-bound evaluation still in flight on the side stream reads
// (the states: next backward sweep; the parameters: next column update).
static int join_elbo(pyvb_lds* h) {
    if (h->elbo_in_flight) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_elbo, 0));
        h->elbo_in_flight = false;
    }
    return PYVB_OK;
}

int pyvb_lds_iterate(pyvb_lds* h, int niters) {
    ENTER_RAW(h);
    ARGCHK(niters >= 0, "niters must be >= 0");
    int rc;
    for (int it = 0; it < niters; ++it) {
        // the backward sweep follows at once and reads c_t, not the forward states: those are not written out
        if ((rc = sweep(h, PYVB_FORWARD, false))) return rc;
        if ((rc = join_elbo(h))) return rc;
        if ((rc = sweep(h, PYVB_BACKWARD, true))) return rc;
        // A and C are independent given the statistics, and so are Q and R given A and C: the pairs
        // share launches here (same arithmetic as update_A, update_C, update_Q, update_R in turn)
        if ((rc = ensure_stats(h))) return rc;
        if (h->dense) {
            if ((rc = ensure_expect(h))) return rc;
            if ((rc = launch_cols_dense(h, 2, 0, h->D))) return rc;
            params_changed(h);
            if ((rc = launch_wresid(h, 2, 1))) return rc;          // both residual matrices and both qw = w0 + residual
            h->expect_valid = false;
            params_changed(h);
            if ((rc = ensure_expect(h))) return rc;                // E[Q], E[R] of the new posteriors: the bound and the next k_prep read them
        } else {
            if ((rc = launch_cols(h, 2, 0, h->D, 3))) return rc;       // columns, residuals and noise update in one launch
            params_changed(h);
        }
        h->resQ_valid = h->resR_valid = true;
        // The lower bound (network.py:49) feeds nothing in the next iteration: it is evaluated on the side stream while
        // the main one goes on with k_prep and the forward sweep.  Its per-iteration totals (summed over the replicates,
        // and over the ranks when a communicator is attached) go into a history ring (pyvb_lds_get_elbo_history).
        HIPCHK(hipEventRecord(h->ev_params, h->stream));
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_params, 0));
        if ((rc = h->dense ? launch_elbo_dense(h, h->side) : launch_elbo(h, h->side))) return rc;
        double* slot = h->elbo_hist + (size_t)(h->hist_count % PYVB_ELBO_HISTORY) * 8;
        if ((rc = launch_elbo_sum(h, slot, h->side))) return rc;
        if (h->comm && (rc = pyvb_allreduce_f64(h->comm, slot, 6, h->side))) return rc;
        h->hist_count += 1;
        HIPCHK(hipEventRecord(h->ev_elbo, h->side));
        h->elbo_in_flight = true;
    }
    // the last lower bound may still be in flight: the next pyvb_lds_iterate overlaps it with its k_prep and forward
    // sweep, any other entry point joins it first (ENTER), pyvb_lds_sync waits for both streams
    return PYVB_OK;
}

int pyvb_lds_get_elbo_history(pyvb_lds* h, double* out, int max_count, int* count) {
    ENTER(h);
    ARGCHK(out && count && max_count >= 0, "bad arguments");
    int n = h->hist_count < PYVB_ELBO_HISTORY ? h->hist_count : PYVB_ELBO_HISTORY;
    if (n > max_count) n = max_count;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->side));
    std::vector<double> tmp((size_t)n * 8);
    for (int i = 0; i < n; ++i) {            // the most recent n rows, oldest first
        const size_t row = (size_t)((h->hist_count - n + i) % PYVB_ELBO_HISTORY);
        HIPCHK(hipMemcpy(tmp.data() + (size_t)i * 8, h->elbo_hist + row * 8, 6 * sizeof(double), hipMemcpyDeviceToHost));
        for (int p = 0; p < 6; ++p) out[(size_t)i * 6 + p] = tmp[(size_t)i * 8 + p];
    }
    *count = n;
    return pyvb_lds_sync(h);
}

int pyvb_lds_reset_elbo_history(pyvb_lds* h) { ENTER(h); HIPCHK(hipStreamSynchronize(h->side)); h->hist_count = 0; return PYVB_OK; }

int pyvb_lds_sync(pyvb_lds* h) {
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->side));
    int st = 0;
    HIPCHK(hipMemcpy(&st, h->status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        pyvb_set_error("a posterior precision was not positive definite (numpy.linalg.LinAlgError in the reference)");
        HIPCHK(hipMemset(h->status, 0, sizeof(int)));
        return PYVB_E_LINALG;
    }
    return PYVB_OK;
}

int pyvb_lds_timing_enable(pyvb_lds* h, int on) { ENTER(h); h->timing = on != 0; return PYVB_OK; }
int pyvb_lds_timing_reset(pyvb_lds* h) { ENTER(h); pyvb_timing_resolve(h); memset(h->timers, 0, sizeof(h->timers)); return PYVB_OK; }
int pyvb_lds_timing_get(pyvb_lds* h, int kernel, double* total_ms, int* launches) {
    ENTER(h);
    ARGCHK(kernel >= 0 && kernel < PYVB_K_COUNT, "no such kernel id");
    pyvb_timing_resolve(h);
    if (h->timing_errors) { pyvb_set_error("%d timing events could not be recorded or resolved", h->timing_errors); h->timing_errors = 0; return PYVB_E_HIP; }
    if (total_ms) *total_ms = h->timers[kernel].total_ms;
    if (launches) *launches = h->timers[kernel].launches;
    return PYVB_OK;
}

// ---- RCCL (loaded on demand so that single-GPU use does not depend on it) -------------------
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_getuid)(nccl_uid*);
typedef int (*fn_initrank)(void**, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
static struct { void* lib; fn_getuid getuid; fn_initrank initrank; fn_allreduce allreduce; fn_destroy destroy; fn_errstr errstr; } g_nccl;

// The only process-wide state of the library: the dlopen'ed RCCL entry points, set once under a lock.
static std::mutex g_nccl_lock;
static int load_rccl() {
    std::lock_guard<std::mutex> guard(g_nccl_lock);
    if (g_nccl.lib) return PYVB_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { pyvb_set_error("cannot load librccl: %s", dlerror()); return PYVB_E_RCCL; }
    g_nccl.getuid = (fn_getuid)dlsym(lib, "ncclGetUniqueId");
    g_nccl.initrank = (fn_initrank)dlsym(lib, "ncclCommInitRank");
    g_nccl.allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    g_nccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_nccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    if (!g_nccl.getuid || !g_nccl.initrank || !g_nccl.allreduce || !g_nccl.destroy) { pyvb_set_error("librccl lacks a required symbol"); return PYVB_E_RCCL; }
    g_nccl.lib = lib;
    return PYVB_OK;
}
#define NCCLCHK(x) do { int _r = (x); if (_r != 0) { pyvb_set_error("RCCL error %d (%s) in %s", _r, g_nccl.errstr ? g_nccl.errstr(_r) : "?", #x); return PYVB_E_RCCL; } } while (0)

int pyvb_comm_unique_id(char id[128]) {
    ARGCHK(id, "id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid u;
    NCCLCHK(g_nccl.getuid(&u));
    memcpy(id, u.internal, 128);
    return PYVB_OK;
}

}  // extern "C"

// shared with the PCA path (api_pca.hip)
int pyvb_comm_create(pyvb_comm** comm, const char id[128], int rank, int world) {
    ARGCHK(comm && id && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    ARGCHK(!*comm, "a communicator is attached already");
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid u;
    memcpy(u.internal, id, 128);
    void* nc = nullptr;
    NCCLCHK(g_nccl.initrank(&nc, world, u, rank));
    pyvb_comm* c = new pyvb_comm();
    c->nccl = nc; c->fn = nullptr; c->user = nullptr; c->host = nullptr; c->cap = 0;
    *comm = c;
    return PYVB_OK;
}
int pyvb_comm_create_host(pyvb_comm** comm, pyvb_host_allreduce_fn fn, void* user) {
    ARGCHK(comm && fn, "bad communicator arguments");
    pyvb_comm* c = new pyvb_comm();
    c->nccl = nullptr; c->fn = fn; c->user = user; c->host = nullptr; c->cap = 0;
    *comm = c;
    return PYVB_OK;
}
void pyvb_comm_free(pyvb_comm* c) {
    if (!c) return;
    if (c->nccl && g_nccl.destroy) g_nccl.destroy(c->nccl);
    if (c->host) (void)hipHostFree(c->host);
    delete c;
}
int pyvb_allreduce_f64(pyvb_comm* c, double* buf, size_t count, hipStream_t stream) {
    if (c->nccl) {
        NCCLCHK(g_nccl.allreduce(buf, buf, count, 8, 0, c->nccl, stream));     // ncclDouble = 8, ncclSum = 0
        return PYVB_OK;
    }
    // host transport: device -> pinned host, the caller's all-reduce (blocking, in place), back.  The stream waits for it.
    if (c->cap < count) {
        if (c->host) (void)hipHostFree(c->host);
        c->host = nullptr; c->cap = 0;
        HIPCHK(hipHostMalloc((void**)&c->host, count * sizeof(double), hipHostMallocDefault));
        c->cap = count;
    }
    HIPCHK(hipMemcpyAsync(c->host, buf, count * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const int r = c->fn(c->host, count, c->user);
    if (r != 0) { pyvb_set_error("the host all-reduce callback failed (%d)", r); return PYVB_E_RCCL; }
    HIPCHK(hipMemcpyAsync(buf, c->host, count * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));          // the staging buffer is reused by the next call
    return PYVB_OK;
}

extern "C" {

int pyvb_lds_comm_init(pyvb_lds* h, const char id[128], int rank, int world) {
    ENTER(h);
    int rc = pyvb_comm_create(&h->comm, id, rank, world);
    if (rc) return rc;
    h->rank = rank; h->world = world;
    return PYVB_OK;
}

int pyvb_lds_comm_init_host(pyvb_lds* h, pyvb_host_allreduce_fn fn, void* user, int rank, int world) {
    ENTER(h);
    ARGCHK(fn && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    ARGCHK(!h->comm, "a communicator is attached already");
    int rc = pyvb_comm_create_host(&h->comm, fn, user);
    if (rc) return rc;
    h->rank = rank; h->world = world;
    return PYVB_OK;
}

int pyvb_lds_comm_destroy(pyvb_lds* h) {
    if (h && h->comm) { pyvb_comm_free(h->comm); h->comm = nullptr; h->world = 1; }
    return PYVB_OK;
}

int pyvb_lds_elbo_total(pyvb_lds* h, double out[6]) {
    ENTER(h);
    ARGCHK(out, "out is NULL");
    int rc = launch_elbo_sum(h);
    if (rc) return rc;
    if (h->comm && (rc = pyvb_allreduce_f64(h->comm, h->elbo_sum, 6, h->stream))) return rc;
    if ((rc = d2h(h, out, h->elbo_sum, 6))) return rc;
    return pyvb_lds_sync(h);
}

}  // extern "C"
