// Host side of the VB-PCA path: handle, copies, dependency tracking, the all-reduce of the statistics.
#include "pca.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ARGCHK(cond, msg) do { if (!(cond)) { pyvb_set_error("%s", msg); return PYVB_E_ARG; } } while (0)
#define ENTER(h) do { ARGCHK(h, "handle is NULL"); HIPCHK(hipSetDevice((h)->device)); } while (0)


// digamma for x > 0 (recurrence up to 10, then the asymptotic series), as the device code had it
static double digamma_host(double x) {
    if (!(x > 0.0)) return NAN;
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double ser = f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
    return r + std::log(x) - 0.5 / x - ser;
}

static int alloc_d(double** p, size_t n) {
    HIPCHK(hipMalloc((void**)p, n * sizeof(double)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(double)));
    return PYVB_OK;
}

extern "C" {

int pyvb_pca_destroy(pyvb_pca* h) {
    if (!h) return PYVB_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) pyvb_comm_free(h->comm);
    void* bufs[] = {h->X, h->M, h->xvar, h->nmiss, h->Z, h->W_mean, h->W_var, h->Mu_mean, h->Mu_var, h->Z_cov, h->qld_W, h->W_pm, h->W_pp,
                    h->Mu_pm, h->Mu_pp, h->scal, h->Gz, h->g0, h->part, h->stats, h->aux, h->elbo, h->status, h->red2, h->Xdata, h->pinned, h->sx_local,
                    h->W_x, h->Mu_x};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PYVB_OK;
}

int pyvb_pca_create(pyvb_pca** out, int device, long N, int d, int q, long N_total, long row_offset) {
    ARGCHK(out, "out is NULL");
    ARGCHK(N >= 1 && N_total >= N && row_offset >= 0, "bad row counts");
    ARGCHK(d >= 1 && d <= 256, "observed dimension d must be in 1..256");
    ARGCHK(q >= 1 && q <= 32, "latent dimension q must be in 1..32");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    ARGCHK(device >= 0 && device < ndev, "no such device");
    HIPCHK(hipSetDevice(device));
    pyvb_pca* h = new pyvb_pca();
    memset(h, 0, sizeof(*h));
    h->device = device; h->N = N; h->N_total = N_total; h->row_offset = row_offset; h->d = d; h->q = q;
    h->DP = (d + 15) & ~15; h->QP = (q + 15) & ~15; h->DT = h->DP / 16; h->QT = h->QP / 16;
    h->SL = pca_stats_layout(h->DP, h->QP);
    h->world = 1;
    long nchunk = (16384 + h->DT - 1) / h->DT;      // pass 2 runs nchunk workgroups of ceil(DT / 2) wavefronts, pass 1 nchunk x 4
    if (h->DT >= 13) {
        // the sweep's workgroup (seven or eight wavefronts, 120 KB of LDS) has a CU to itself: ONE chunk per CU, so that every
        // workgroup pays its prologue (operands, first tiles) and its partial sums once -- measured at N = 10^6 x 256:
        // 1024 chunks (four rounds) 1.040 ms per iteration, 768: 1.028, 512: 1.009, 256: 0.990 (profiles/r04/pca_chunks.txt)
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) nchunk = ncu;
    }
    { const char* e = getenv("PYVB_PCA_CHUNKS"); if (e && atol(e) > 0) nchunk = atol(e); }      // (experiments)
    const long ntile = (N + 15) / 16;
    if (nchunk > ntile) nchunk = ntile;
    if (nchunk < 1) nchunk = 1;
    long rows = (N + nchunk - 1) / nchunk;
    rows = (rows + 15) & ~15L;
    nchunk = (N + rows - 1) / rows;
    h->nchunk = (int)nchunk; h->chunk_rows = rows;
    int rc = PYVB_OK;
#define TRY(x) do { rc = (x); if (rc != PYVB_OK) { pyvb_pca_destroy(h); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { rc = pyvb_hip_fail(_e, #x, __FILE__, __LINE__); pyvb_pca_destroy(h); return rc; } } while (0)
    TRYHIP(hipStreamCreate(&h->stream));
    const size_t n = (size_t)N, DP = h->DP, QP = h->QP;
    TRY(alloc_d(&h->X, n * DP));
    TRYHIP(hipMalloc((void**)&h->M, n * DP)); TRYHIP(hipMemset(h->M, 1, n * DP));
    TRY(alloc_d(&h->xvar, n));
    TRYHIP(hipMalloc((void**)&h->nmiss, n * sizeof(int))); TRYHIP(hipMemset(h->nmiss, 0, n * sizeof(int)));
    TRY(alloc_d(&h->Z, n * QP));
    TRY(alloc_d(&h->W_mean, (size_t)d * q)); TRY(alloc_d(&h->W_var, (size_t)q * d));
    TRY(alloc_d(&h->Mu_mean, d)); TRY(alloc_d(&h->Mu_var, d)); TRY(alloc_d(&h->Z_cov, (size_t)q * q)); TRY(alloc_d(&h->qld_W, q));
    TRY(alloc_d(&h->W_pm, (size_t)d * q)); TRY(alloc_d(&h->W_pp, (size_t)q * d)); TRY(alloc_d(&h->Mu_pm, d)); TRY(alloc_d(&h->Mu_pp, d));
    TRY(alloc_d(&h->W_x, (size_t)d * q)); TRY(alloc_d(&h->Mu_x, d));
    { const char* e = getenv("PYVB_PCA_WRITEBACK"); h->lazy_ok = !(e && e[0] == '1'); }
    {   // k_pca_rows: one workgroup of four wavefronts per CU, the rows dealt out in multiples of 16
        int ncu = 0;
        TRYHIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        if (ncu < 1) ncu = 1;
        long rowsB = (N + ncu - 1) / ncu;
        rowsB = (rowsB + 15) & ~15L;
        if (rowsB < 64) rowsB = 64;
        long ncB = (N + rowsB - 1) / rowsB;
        if (ncB > nchunk) { ncB = nchunk; rowsB = ((N + ncB - 1) / ncB + 15) & ~15L; ncB = (N + rowsB - 1) / rowsB; }
        h->nchunkB = (int)ncB; h->chunk_rowsB = rowsB;
        // which kernel the lazy sweep is (k_pca.hip): PYVB_PCA_SWEEP = columns (k_pca_pass12<.., LAZY>) / rows (k_pca_rows) / pairs
        // (k_pca_pairs); unset: pairs where a CU's share is long enough to stream (measured at 10^6 x 256: 0.95 against 0.97 ms per
        // iteration), columns otherwise
        const char* e = getenv("PYVB_PCA_SWEEP");
        if (e && e[0] == 'r') h->rows_ok = 1;
        else if (e && e[0] == 'p') h->rows_ok = 2;
        else if (e && e[0] == 'c') h->rows_ok = 0;
        else h->rows_ok = (h->DT >= 13 && N >= 512L * ncu) ? 2 : 0;
    }
    TRY(alloc_d(&h->scal, PS_COUNT));
    TRY(alloc_d(&h->Gz, (size_t)h->QT * (DP / 4) * 64)); TRY(alloc_d(&h->g0, QP));
    TRY(alloc_d(&h->part, (size_t)h->nchunk * (h->SL.total + h->DT)));
    TRY(alloc_d(&h->stats, h->SL.total));
    TRY(alloc_d(&h->sx_local, DP));
    TRY(alloc_d(&h->red2, (size_t)PCA_RED * h->SL.total));
    TRY(alloc_d(&h->aux, (size_t)4 * h->nchunk * QP + QP + DP));
    TRY(alloc_d(&h->elbo, 8));
    TRYHIP(hipMalloc((void**)&h->status, sizeof(int))); TRYHIP(hipMemset(h->status, 0, sizeof(int)));
#undef TRY
#undef TRYHIP
    *out = h;
    return PYVB_OK;
}

static int up(pyvb_pca* h, double* dst, const double* src, size_t n) {
    if (!src) return PYVB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return PYVB_OK;
}
static int down(pyvb_pca* h, double* dst, const double* src, size_t n) {
    if (!dst) return PYVB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return PYVB_OK;
}

// A requested Z update whose rows have not been written (see pyvb_pca_update_Z) is carried out: pass 1 on its own.  Called by
// everything that reads or replaces Z, X or the parameters outside the fused sweep.
static int resolve_z(pyvb_pca* h) {
    int rc = pca_materialize_x(h);          // the same callers want X as it stands, and Z is about to change or be replaced
    if (rc) return rc;
    if (!h->z_pending) return PYVB_OK;
    rc = pca_launch_pass1(h);
    h->z_pending = false; h->z0_done = false;
    return rc;
}

int pyvb_pca_set_priors(pyvb_pca* h, const double* W_pm, const double* W_pp, const double* Mu_pm, const double* Mu_pp,
                        double beta_a0, double beta_b0) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    ARGCHK(W_pm && W_pp && Mu_pm && Mu_pp, "all prior arrays are required");
    int rc;
    if ((rc = up(h, h->W_pm, W_pm, (size_t)h->d * h->q))) return rc;
    if ((rc = up(h, h->W_pp, W_pp, (size_t)h->q * h->d))) return rc;
    if ((rc = up(h, h->Mu_pm, Mu_pm, h->d))) return rc;
    if ((rc = up(h, h->Mu_pp, Mu_pp, h->d))) return rc;
    double sc[PS_COUNT];
    HIPCHK(hipMemcpy(sc, h->scal, sizeof(sc), hipMemcpyDeviceToHost));
    sc[PS_BETA_A0] = beta_a0; sc[PS_BETA_B0] = beta_b0;
    sc[PS_BETA_A] = beta_a0 + 0.5 * (double)h->d * (double)h->N_total;      // Gamma.update_a, nodes_todo.py:125-128
    sc[PS_LGAMMA_A0] = std::lgamma(beta_a0); sc[PS_LGAMMA_A] = std::lgamma(sc[PS_BETA_A]); sc[PS_DIGAMMA_A] = digamma_host(sc[PS_BETA_A]);
    sc[PS_QLD_Z] = sc[PS_QLD_X] = sc[PS_QLD_MU] = NAN;
    HIPCHK(hipMemcpy(h->scal, sc, sizeof(sc), hipMemcpyHostToDevice));
    std::vector<double> nanq((size_t)h->q, (double)NAN);        // no column has been updated on this handle yet
    HIPCHK(hipMemcpy(h->qld_W, nanq.data(), nanq.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PYVB_OK;
}

// rows [N][d] on the host -> padded device rows [N][DP]
static int upload_rows(pyvb_pca* h, double* dst, const double* src, int cols, int CP) {
    std::vector<double> buf((size_t)h->N * CP, 0.0);
    for (long n = 0; n < h->N; ++n) memcpy(&buf[(size_t)n * CP], src + (size_t)n * cols, cols * sizeof(double));
    HIPCHK(hipMemcpy(dst, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
    return PYVB_OK;
}
static int download_rows(pyvb_pca* h, double* dst, const double* src, int cols, int CP) {
    std::vector<double> buf((size_t)h->N * CP);
    HIPCHK(hipMemcpy(buf.data(), src, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (long n = 0; n < h->N; ++n) memcpy(dst + (size_t)n * cols, &buf[(size_t)n * CP], cols * sizeof(double));
    return PYVB_OK;
}

int pyvb_pca_set_data(pyvb_pca* h, const double* X) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    ARGCHK(X, "X is NULL");
    const long N = h->N; const int d = h->d, DP = h->DP;
    std::vector<double> xb((size_t)N * DP, 0.0);
    std::vector<unsigned char> mb((size_t)N * DP, 1);
    std::vector<int> nm(N, 0);
    double counts[3] = {0, 0, 0};          // missing entries in partially observed rows, rows without observations, partial rows
    for (long n = 0; n < N; ++n) {
        int miss = 0;
        for (int k = 0; k < d; ++k) {
            const double v = X[(size_t)n * d + k];
            if (v != v) { mb[(size_t)n * DP + k] = 0; ++miss; } else xb[(size_t)n * DP + k] = v;
        }
        nm[n] = miss;
        if (miss == d) counts[1] += 1; else if (miss > 0) { counts[0] += miss; counts[2] += 1; }
    }
    HIPCHK(hipMemcpy(h->X, xb.data(), xb.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->M, mb.data(), mb.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->nmiss, nm.data(), nm.size() * sizeof(int), hipMemcpyHostToDevice));
    if (h->comm) {       // global counts
        HIPCHK(hipMemcpy(h->elbo, counts, sizeof(counts), hipMemcpyHostToDevice));
        int rc = pyvb_allreduce_f64(h->comm, h->elbo, 3, h->stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(counts, h->elbo, sizeof(counts), hipMemcpyDeviceToHost));
    }
    h->n_part_missing = (long)counts[0]; h->n_none_rows = (long)counts[1]; h->n_part_rows = (long)counts[2];
    h->full_valid = h->lin_valid = false; h->res_valid = false;
    return PYVB_OK;
}

int pyvb_pca_set_state(pyvb_pca* h, const double* X_missing, const double* W_mean, const double* Z, const double* Z_cov,
                       const double* Mu_mean, const double* beta_b) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    const long N = h->N; const int d = h->d, q = h->q, DP = h->DP;
    int rc;
    if (X_missing) {        // posterior means of the missing entries; observed positions of the argument are ignored
        std::vector<double> xb((size_t)N * DP);
        std::vector<unsigned char> mb((size_t)N * DP);
        HIPCHK(hipMemcpy(xb.data(), h->X, xb.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(mb.data(), h->M, mb.size(), hipMemcpyDeviceToHost));
        for (long n = 0; n < N; ++n)
            for (int k = 0; k < d; ++k)
                if (!mb[(size_t)n * DP + k]) xb[(size_t)n * DP + k] = X_missing[(size_t)n * d + k];
        HIPCHK(hipMemcpy(h->X, xb.data(), xb.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if ((rc = up(h, h->W_mean, W_mean, (size_t)d * q))) return rc;
    if (Z && (rc = upload_rows(h, h->Z, Z, q, h->QP))) return rc;
    if ((rc = up(h, h->Z_cov, Z_cov, (size_t)q * q))) return rc;
    if ((rc = up(h, h->Mu_mean, Mu_mean, d))) return rc;
    if (beta_b) HIPCHK(hipMemcpyAsync(h->scal + PS_BETA_B, beta_b, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->full_valid = h->lin_valid = false; h->res_valid = false;
    return PYVB_OK;
}

int pyvb_pca_set_initial_variances(pyvb_pca* h, const double* W_var, const double* Mu_var) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    int rc;
    if ((rc = up(h, h->W_var, W_var, (size_t)h->q * h->d))) return rc;
    if ((rc = up(h, h->Mu_var, Mu_var, h->d))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->res_valid = false;
    return PYVB_OK;
}

int pyvb_pca_set_unpinned_rows(pyvb_pca* h, const double* X_full, const double* row_var) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    ARGCHK(X_full && row_var, "X_full and row_var are required");
    const long N = h->N; const int d = h->d, DP = h->DP;
    std::vector<double> xb((size_t)N * DP), xv(N);
    std::vector<int> nm(N);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(xb.data(), h->X, xb.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(xv.data(), h->xvar, N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nm.data(), h->nmiss, N * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> data(xb);           // the observations (and whatever the missing entries held)
    std::vector<unsigned char> pin(N, 1);
    long unpinned = 0;
    for (long n = 0; n < N; ++n) {
        if (nm[n] == 0) continue;           // fully observed: qmu is the observation, qcov zero (gaussian.py:97-100)
        ARGCHK(row_var[n] > 0.0, "row_var must be positive for rows with missing entries");
        xv[n] = row_var[n];
        for (int k = 0; k < d; ++k) xb[(size_t)n * DP + k] = X_full[(size_t)n * d + k];
        if (nm[n] < d) { pin[n] = 0; ++unpinned; }      // some entries observed: they are pinned by the row's first update
    }
    if (unpinned > 0) {
        if (!h->Xdata) {
            HIPCHK(hipMalloc((void**)&h->Xdata, xb.size() * sizeof(double)));
            HIPCHK(hipMalloc((void**)&h->pinned, (size_t)N));
        }
        HIPCHK(hipMemcpy(h->Xdata, data.data(), data.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->pinned, pin.data(), (size_t)N, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(h->X, xb.data(), xb.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->xvar, xv.data(), N * sizeof(double), hipMemcpyHostToDevice));
    h->full_valid = h->lin_valid = false; h->res_valid = false;
    return PYVB_OK;
}

int pyvb_pca_sync(pyvb_pca* h) {
    ENTER(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    int st = 0;
    HIPCHK(hipMemcpy(&st, h->status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        pyvb_set_error("a posterior precision was not positive definite (numpy.linalg.LinAlgError in the reference)");
        HIPCHK(hipMemset(h->status, 0, sizeof(int)));
        return PYVB_E_LINALG;
    }
    return PYVB_OK;
}

int pyvb_pca_get_state(pyvb_pca* h, double* X, double* X_rowvar, double* W_mean, double* W_var, double* Z, double* Z_cov,
                       double* Mu_mean, double* Mu_var, double* beta_ab) {
    ENTER(h);
    { int rz = resolve_z(h); if (rz) return rz; }
    const int d = h->d, q = h->q;
    int rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (X && (rc = download_rows(h, X, h->X, d, h->DP))) return rc;
    if (Z && (rc = download_rows(h, Z, h->Z, q, h->QP))) return rc;
    if ((rc = down(h, X_rowvar, h->xvar, h->N))) return rc;
    if ((rc = down(h, W_mean, h->W_mean, (size_t)d * q))) return rc;
    if ((rc = down(h, W_var, h->W_var, (size_t)q * d))) return rc;
    if ((rc = down(h, Z_cov, h->Z_cov, (size_t)q * q))) return rc;
    if ((rc = down(h, Mu_mean, h->Mu_mean, d))) return rc;
    if ((rc = down(h, Mu_var, h->Mu_var, d))) return rc;
    if ((rc = down(h, beta_ab, h->scal + PS_BETA_A, 2))) return rc;
    return pyvb_pca_sync(h);
}

int pyvb_pca_get_qld(pyvb_pca* h, double* qld_W, double* qld_Z, double* qld_Mu, double* qld_X) {
    ENTER(h);
    int rc;
    double sc[PS_COUNT];
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((rc = down(h, qld_W, h->qld_W, (size_t)h->q))) return rc;
    HIPCHK(hipMemcpyAsync(sc, h->scal, sizeof(sc), hipMemcpyDeviceToHost, h->stream));
    double* tmp = nullptr;
    if (qld_X) {
        HIPCHK(hipMalloc((void**)&tmp, (size_t)h->N * sizeof(double)));
        if ((rc = pca_launch_rowqld(h, tmp)) || (rc = down(h, qld_X, tmp, (size_t)h->N))) {
            (void)hipStreamSynchronize(h->stream); (void)hipFree(tmp);
            return rc;
        }
    }
    const hipError_t se = hipStreamSynchronize(h->stream);
    if (tmp) (void)hipFree(tmp);
    HIPCHK(se);
    if (qld_Z) *qld_Z = sc[PS_QLD_Z];
    if (qld_Mu) *qld_Mu = sc[PS_QLD_MU];
    return PYVB_OK;
}

// ---- dependency tracking: "full" = every sum current, "lin" = at least sum x and sum z ----
static int full_stats(pyvb_pca* h, long lo_upd, long hi_upd) {
    int rc;
    if (h->z_pending) {          // the Z update rides along: one sweep over X instead of two
        rc = pca_launch_pass12(h, lo_upd, hi_upd);
        h->z_pending = false; h->z0_done = false;
        if (rc) return rc;
    } else if ((rc = pca_launch_pass2(h, lo_upd, hi_upd))) return rc;
    if ((rc = pca_launch_reduce(h, 0))) return rc;
    if (h->comm && (rc = pyvb_allreduce_f64(h->comm, h->stats, h->SL.total, h->stream))) return rc;
    h->full_valid = h->lin_valid = true;
    if (hi_upd > lo_upd) h->res_valid = false;          // rows changed: the cached residual of the last Beta update is stale
    return PYVB_OK;
}
static int ensure_full(pyvb_pca* h) { return h->full_valid ? PYVB_OK : full_stats(h, 0, 0); }

int pyvb_pca_update_W(pyvb_pca* h) {
    ENTER(h);
    int rc = ensure_full(h);
    if (rc) return rc;
    h->res_valid = false;
    return pca_launch_small(h, PCA_W);
}

// exchange [new sum z | delta of sum x] and fold it into the statistics
static int exchange_lin(pyvb_pca* h) {
    int rc;
    double* v = h->aux + (size_t)4 * h->nchunk * h->QP;
    if (h->comm && (rc = pyvb_allreduce_f64(h->comm, v, (size_t)h->QP + h->DP, h->stream))) return rc;
    return pca_launch_small(h, PCA_APPLY);
}

int pyvb_pca_update_Z(pyvb_pca* h) {
    ENTER(h);
    int rc;
    // Deferred: the posterior of the Z_n (Sigma_z, Gz, g0) and the one sum the next nodes in the crawl order read, sum z, are
    // formed now (sum z = Gz sum x - N g0 over this rank's rows, from the sum of x kept from the last sweep; all-reduced like
    // the sum a pass over the rows would give); the rows of Z are written by the next sweep over X -- normally the X update
    // that follows (k_pca_pass12) -- or by resolve_z() if something asks for them first.
    if (!h->lin_valid && (rc = ensure_full(h))) return rc;      // may carry out an earlier pending request
    h->z_pending = true; h->z0_done = false;
    if ((rc = pca_launch_small(h, PCA_PREPZ))) { h->z_pending = false; return rc; }
    if ((rc = exchange_lin(h))) return rc;
    h->full_valid = false; h->res_valid = false;                // lin_valid stays: sum x is unchanged, sum z is the new one
    return PYVB_OK;
}

// Xs[0].update() alone (the crawl order of fetch_network puts it before Mu): sum x is kept current without a pass over
// all rows.  With a communicator this is a collective step for every rank: the owner of global row 0 (row_offset == 0)
// updates it, everyone exchanges [sum z | delta of sum x]; the sum of z travels once, from the owner.
static int x0_step(pyvb_pca* h) {
    int rc;
    if (!h->lin_valid && (rc = ensure_full(h))) return rc;
    if (h->xlazy && h->row_offset == 0 && h->vlo == 0 && (rc = pca_materialize_x(h))) return rc;     // row 0 itself is read here
    double* v = h->aux + (size_t)4 * h->nchunk * h->QP;
    HIPCHK(hipMemcpyAsync(v, h->stats + h->SL.osz, h->QP * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->comm && h->row_offset != 0)
        HIPCHK(hipMemsetAsync(v, 0, h->QP * sizeof(double), h->stream));
    if ((rc = pca_launch_small(h, PCA_X0))) return rc;      // a no-op for the data of the other ranks (k_pca.hip checks row_offset)
    if (h->z_pending && h->row_offset == 0) h->z0_done = true;     // the kernel has stored z_0 from the row as it was
    if ((rc = exchange_lin(h))) return rc;
    h->full_valid = false; h->res_valid = false;
    return PYVB_OK;
}

// Xs[lo:hi] updates + every sum over n; collective (the statistics are all-reduced) even for an empty range
static int x_rows(pyvb_pca* h, long lo, long hi) {
    if (lo == hi && !h->comm) return PYVB_OK;
    return full_stats(h, lo, hi);
}

int pyvb_pca_update_X(pyvb_pca* h, long lo, long hi) {
    ENTER(h);
    ARGCHK(lo >= 0 && lo <= hi && hi <= h->N, "bad row range");
    // the single-row step of global row 0 is a different collective (x0_step all-reduces QP + DP doubles, x_rows the whole
    // statistics vector): with a communicator attached every rank must make the same call, so the shortcut is taken only
    // without one -- sharded callers reach it through pyvb_pca_update_X0 on every rank
    if (lo == 0 && hi == 1 && h->row_offset == 0 && !h->comm) return x0_step(h);
    return x_rows(h, lo, hi);
}

int pyvb_pca_update_X0(pyvb_pca* h) {
    ENTER(h);
    return x0_step(h);
}

int pyvb_pca_update_Mu(pyvb_pca* h) {
    ENTER(h);
    int rc;
    if (!h->lin_valid && (rc = ensure_full(h))) return rc;
    h->res_valid = false;
    return pca_launch_small(h, PCA_MU);
}

int pyvb_pca_update_Beta(pyvb_pca* h) {
    ENTER(h);
    int rc = ensure_full(h);
    if (rc) return rc;
    if ((rc = pca_launch_small(h, PCA_BETA))) return rc;
    h->res_valid = true;        // the residual does not depend on Beta: the lower bound can reuse it
    return PYVB_OK;
}

int pyvb_pca_elbo(pyvb_pca* h, double parts[5]) {
    ENTER(h);
    int rc = ensure_full(h);
    if (rc) return rc;
    if ((rc = pca_launch_small(h, PCA_ELBO))) return rc;
    if (parts) {
        HIPCHK(hipMemcpyAsync(parts, h->elbo, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        return pyvb_pca_sync(h);
    }
    return PYVB_OK;
}

// niters passes of Network.learn's body over the fetched PCA network: crawl order W, Z, X_0, Mu, X_1.., Beta, bound
int pyvb_pca_iterate(pyvb_pca* h, int niters) {
    ENTER(h);
    ARGCHK(niters >= 0, "niters must be >= 0");
    int rc;
    for (int it = 0; it < niters; ++it) {
        if (!h->comm) {
            // Without a communicator nothing sits between the small steps of the crawl order but their own data: they run as three
            // launches (W, Z-prepare | X_0, Mu | Beta, bound) around the one sweep over the rows; same code, same order, same flags
            // as the calls of the general path below.
            if ((rc = ensure_full(h))) return rc;                       // update_W
            h->z_pending = true; h->z0_done = false;                    // update_Z, deferred (lin_valid holds after ensure_full)
            if ((rc = pca_launch_small(h, PCA_RUN_HEAD))) { h->z_pending = false; return rc; }
            h->full_valid = false; h->res_valid = false;
            if ((rc = pca_launch_small(h, PCA_RUN_MID))) return rc;     // x0_step, update_Mu
            h->z0_done = h->row_offset == 0;                            // the kernel stored z_0 if this handle holds global row 0
            if ((rc = x_rows(h, h->row_offset == 0 ? 1 : 0, h->N))) return rc;
            if ((rc = ensure_full(h))) return rc;                       // N == 1: no row left for x_rows, the sweep still has to run
            if ((rc = pca_launch_small(h, PCA_RUN_TAIL))) return rc;    // update_Beta, elbo
            h->res_valid = true;
            continue;
        }
        if ((rc = pyvb_pca_update_W(h))) return rc;
        if ((rc = pyvb_pca_update_Z(h))) return rc;
        const long first = h->row_offset == 0 ? 1 : 0;         // global row 0 lives on the rank with offset 0
        if ((rc = x0_step(h))) return rc;                       // every rank: the exchange is collective
        if ((rc = pyvb_pca_update_Mu(h))) return rc;
        if ((rc = x_rows(h, first, h->N))) return rc;           // every rank, even with no rows left (N == 1 on the owner)
        if ((rc = pyvb_pca_update_Beta(h))) return rc;
        if ((rc = pyvb_pca_elbo(h, nullptr))) return rc;
    }
    return PYVB_OK;
}

int pyvb_pca_comm_init(pyvb_pca* h, const char id[128], int rank, int world) {
    ENTER(h);
    ARGCHK(world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    ARGCHK((rank == 0) == (h->row_offset == 0), "rows shard contiguously in rank order: rank 0, and only rank 0, holds global row 0");
    int rc = pyvb_comm_create(&h->comm, id, rank, world);
    if (rc) return rc;
    h->rank = rank; h->world = world;
    return PYVB_OK;
}

int pyvb_pca_comm_init_host(pyvb_pca* h, pyvb_host_allreduce_fn fn, void* user, int rank, int world) {
    ENTER(h);
    ARGCHK(fn && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    ARGCHK((rank == 0) == (h->row_offset == 0), "rows shard contiguously in rank order: rank 0, and only rank 0, holds global row 0");
    ARGCHK(!h->comm, "a communicator is attached already");
    int rc = pyvb_comm_create_host(&h->comm, fn, user);
    if (rc) return rc;
    h->rank = rank; h->world = world;
    return PYVB_OK;
}

}  // extern "C"
