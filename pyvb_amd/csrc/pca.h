// Internal definitions of the VB-PCA-with-missing-data path (examples/PCA_missing_data.py of the reference).
#pragma once
#include "common.h"

// Reduced statistics vector (doubles), all sums over the rows n of this rank (then over ranks):
//   Szz [QP][QP]  sum z_n z_n^T (means only; the shared covariance is added by the consumers)
//   Sxz [DP][QP]  sum x_n z_n^T           sx [DP]  sum x_n           sz [QP]  sum z_n
//   sxx           sum ||x_n||^2           sxv      sum_n (#missing_n * var_n)
//   slv           sum over partially observed rows of #missing_n * log(var_n)
//   sql           sum over rows without any observation of their q_ln_det = 0.5 / (d/2 log(1 / var_n))   (quirk Q1; each row
//                 keeps the <beta> of ITS last update)
struct PcaStatsLayout {
    int DP, QP;
    size_t oSzz, oSxz, osx, osz, osxx, osxv, oslv, osql, total;
};
static inline PcaStatsLayout pca_stats_layout(int DP, int QP) {
    PcaStatsLayout L; L.DP = DP; L.QP = QP;
    size_t o = 0;
    L.oSzz = o; o += (size_t)QP * QP; L.oSxz = o; o += (size_t)DP * QP; L.osx = o; o += DP; L.osz = o; o += QP;
    L.osxx = o++; L.osxv = o++; L.oslv = o++; L.osql = o++;
    L.total = (o + 7) & ~(size_t)7;
    return L;
}

#define PCA_RED 128     // slices of the chunk partials summed in parallel (stage 0 of k_pca_reduce)

// device scalars
enum { PS_BETA_A = 0, PS_BETA_B, PS_QLD_Z, PS_QLD_X /* unused: latent rows keep their own, statistics slot sql */, PS_QLD_MU, PS_BETA_A0, PS_BETA_B0, PS_RES,
       PS_LGAMMA_A0, PS_LGAMMA_A, PS_DIGAMMA_A /* of the two shape parameters, which never change after set_priors: formed on the host */, PS_COUNT = 16 };   // PS_RES: the residual of the last Beta update (see res_valid)

struct pyvb_pca {
    int device; long N, N_total, row_offset; int d, q, DP, QP, DT, QT;
    hipStream_t stream;
    double *X; unsigned char* M;         // [N][DP] posterior means of the X_n (data / imputed); 1 = observed
    double *xvar;                        // [N] variance of the missing entries of row n
    double *Xdata; unsigned char* pinned; // only after pyvb_pca_set_unpinned_rows: the observations [N][DP] of rows that still
                                         // carry their initial mean at ALL entries (pinned[n] == 0) until their first update
    int *nmiss;                          // [N]
    double *Z;                           // [N][QP]
    double *W_mean, *W_var, *Mu_mean, *Mu_var, *Z_cov, *qld_W;   // [d][q], [q][d], [d], [d], [q][q], [q]
    double *W_pm, *W_pp, *Mu_pm, *Mu_pp; // priors: [d][q], [q][d], [d], [d]
    double *scal;                        // [PS_COUNT]
    double *Gz, *g0;                     // Z-pass operands: Gz^T as MFMA B operands [QT][DP/4][64], g0 [QP]
    double *part; int nchunk; long chunk_rows;   // [nchunk][DT+1][stats.total] partial statistics
    double *stats;                       // [stats.total] reduced (global after the all-reduce)
    double *sx_local;                    // [DP] sum of x over THIS rank's rows (what stats holds before the all-reduce), kept current by
                                         // the X_0 step: the deferred Z update forms its sum of z from it
    double *aux;                         // [nchunk][QP] pass-1 partials, then [QP + DP]: new sum z | delta of sum x
    double *red2;                        // [PCA_RED][stats.total] second-stage partials of the reductions
    double *elbo;                        // [5]
    int *status;
    PcaStatsLayout SL;
    long n_part_missing, n_none_rows, n_part_rows;   // global counts (from the mask)
    bool full_valid, lin_valid;          // all statistics current / at least sum x and sum z current
    bool res_valid;                      // scal[PS_RES] is the residual of the current W, Z, X, Mu (nothing but Beta updated since)
    bool z_pending;                      // [z.update() for z in Zs] has been requested and its operands (Gz, g0, sum z) are set, but the
                                         // rows of Z are not written yet: the next pass over X does it on its way (k_pca_pass12)
    bool z0_done;                        // while z_pending: Xs[0].update() has run and stored z_0 itself
    double *W_x, *Mu_x;                  // [d][q], [d]: the parameters the last lazy sweep imputed with
    bool xlazy; long vlo, vhi;           // the missing entries of rows [vlo, vhi) are not in X: they stand for <W>_x z_n + <Mu>_x
                                         // (k_pca_pass12<.., LAZY>); pca_materialize_x puts them there
    bool lazy_ok;                        // PYVB_PCA_WRITEBACK=1 in the environment at creation turns the lazy sweep off (A/B measurements)
    int rows_ok;                         // the lazy sweep is k_pca_pass12<.., LAZY> (0), k_pca_rows (1) or k_pca_pairs (2): chosen at creation (api_pca.hip, PYVB_PCA_SWEEP)
    int nchunkB; long chunk_rowsB;       // k_pca_rows' partition of the rows: a workgroup per CU
    int part_chunks;                     // chunks of the partial statistics in `part` now
    bool rows_attr_set;
    pyvb_comm* comm; int rank, world;
};

int pca_launch_small(pyvb_pca* h, int mode);
int pca_launch_pass1(pyvb_pca* h);
int pca_launch_pass2(pyvb_pca* h, long lo_upd, long hi_upd);
int pca_launch_pass12(pyvb_pca* h, long lo_upd, long hi_upd);
int pca_materialize_x(pyvb_pca* h);
int pca_launch_reduce(pyvb_pca* h, int what);
int pca_launch_rowqld(pyvb_pca* h, double* out);      // out: device [N]
enum { PCA_W = 0, PCA_PREPZ = 1, PCA_MU = 2, PCA_BETA = 3, PCA_ELBO = 4, PCA_X0 = 5, PCA_APPLY = 6,
       PCA_RUN_HEAD = 16, PCA_RUN_MID = 17, PCA_RUN_TAIL = 18 };     // runs of steps in one launch (k_pca.hip: pca_launch_small)
