// Internal definitions shared by the translation units of libpyvb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "../../include/pyvb_hip.h"

struct pyvb_comm;
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Per-replicate block of "gains": every matrix the state updates multiply by, written by
// k_prep once per parameter change and read by the sweep / step kernels.
//   Sigma_c = posterior covariance of class c (0: X_0, 1: interior, 2: X_{T-1})
//   F  = Sigma_1 <Q><A>      B  = Sigma_1 <A>^T<Q>     G  = Sigma_1 <C>^T<R>
// The two boundary nodes are single matrix-vector chains, mu = Sigma_c (sum of the messages), so for them
// the block holds the pieces instead of products: S0 = Sigma_0, S2 = Sigma_2 ([row][DP]), qr = <Q> diagonal
// (QR = 64 entries, 128 in the second shape class) then <R> diagonal (QR), w0 = L0 m0 (the Constant parents of X_0).
// "n"/"p" blocks are laid out as v_mfma_f64_16x16x4 A-operands: element [(m*S+s)*64+lane]
// = M[16m + (lane&15)][kidx(s, lane>>4)], kidx natural = 4s+q (F, B: they meet states, which come in
// accumulator order), permuted = 8(s>>1)+2q+(s&1) (G: it meets rows of Y read 16 bytes per lane).
// (pos_nat / pos_perm below give the position of element (i, j) in such a block.)
struct Layout {
    int D, K, DT, KT, DP, KP, DS, KS;
    int QR;                 // length of each of the two noise diagonals at oqr: 64, or 128 in the second shape class
    size_t oFn, oBn, oGp;
    size_t oS0, oS2, oqr, ow0;
    size_t gains_total;     // doubles per replicate
    size_t stats_total;     // doubles per replicate per chunk: Sxx[DP][DP], Sx1x[DP][DP], Syx[KP][DP]
    size_t oSxx, oSx1x, oSyx;
};

// Internal layout of a state row (X buffers, stride DP): "accumulator order".  Dimension
// d = 16m + 4r + q sits at 16m + 4q + r, which is where lane group q keeps register r of tile m of a
// v_mfma_f64_16x16x4 accumulator -- and also its B operand of k-step 4m + r.  A lane therefore moves
// four contiguous doubles per tile, for the state stores as well as for the neighbour loads.
__host__ __device__ static inline int xpos(int d) { return (d & ~15) | ((d & 3) << 2) | ((d >> 2) & 3); }

// position of matrix element (i, j) in an MFMA A-operand block with S k-steps
__host__ __device__ static inline size_t pos_nat(int i, int j, int S) {
    return ((size_t)((i >> 4) * S + (j >> 2)) * 64) + (j & 3) * 16 + (i & 15);
}
__host__ __device__ static inline size_t pos_perm(int i, int j, int S) {
    int s = 2 * (j >> 3) + (j & 1), q = (j & 7) >> 1;
    return ((size_t)((i >> 4) * S + s) * 64) + q * 16 + (i & 15);
}

// column covariances under Wishart noise are stored as the upper 8 x 8 tiles of the matrix (k_wishart.hip: cov_pos)
__host__ __device__ static inline int cov_tiles(int rows) { const int RT = (rows + 7) >> 3; return RT * (RT + 1) / 2; }
__host__ __device__ static inline size_t cov_stride(int rows) { return (size_t)cov_tiles(rows) * 64; }

static inline int tiles16(int d) { return d <= 16 ? 1 : (d <= 32 ? 2 : (d <= 64 ? 4 : 8)); }

static inline Layout make_layout(int D, int K) {
    Layout L;
    L.D = D; L.K = K;
    L.DT = tiles16(D); L.KT = tiles16(K);
    if (L.DT > 4 || L.KT > 4) L.DT = L.KT = 8;      // the second shape class (k_big.hip): both dimensions padded to 128
    L.QR = L.DT > 4 ? 128 : 64;
    L.DP = 16 * L.DT; L.KP = 16 * L.KT;
    L.DS = 4 * L.DT; L.KS = 4 * L.KT;
    size_t o = 0;
    size_t dd = (size_t)L.DT * L.DS * 64, dk = (size_t)L.DT * L.KS * 64;
    L.oFn = o; o += dd; L.oBn = o; o += dd; L.oGp = o; o += dk;
    size_t tdd = (size_t)L.DP * L.DP, tkd = (size_t)L.KP * L.DP;
    L.oS0 = o; o += tdd; L.oS2 = o; o += tdd; L.oqr = o; o += 2 * (size_t)L.QR; L.ow0 = o; o += L.DP;
    L.gains_total = o;
    L.oSxx = 0; L.oSx1x = tdd; L.oSyx = 2 * tdd;
    L.stats_total = 2 * tdd + tkd;
    return L;
}

struct Priors {          // device pointers, shared by all replicates
    double *x0_mean, *x0_prec, *A_pm, *A_pp, *C_pm, *C_pp, *Q_a0, *Q_b0, *R_a0, *R_b0;
    double *A_obs, *C_obs;   // observed entries of the matrices, [row][col], NaN = not observed
    double *Q_w0, *R_w0;     // Wishart priors w0: [D][D], [K][K] (v0 in Q_a0[0], R_a0[0]); nodes_todo.py:207-211
    double *A_pld, *C_pld;   // [D] ln det of each column's diagonal prior precision
    double x0_lndet;     // ln det of x0_prec (Constant.lndet, node.py:301-302)
    double Q_a0_host, R_a0_host, Q_w0_lndet, R_w0_lndet;    // Wishart: v0 and ln det w0 of the two priors
};

struct EventPair;
struct KernelTimer {
    double total_ms; int launches;
};

struct pyvb_lds {
    int device, N, T, D, K, noise;
    bool big;                       // 64 < max(D, K) <= 128: the workgroup-per-replicate kernels of k_big.hip
    bool big_attr_prep, big_attr_cols;      // their dynamic-LDS limits have been raised on this handle's device
    double* U2;                     // 128-wide class with the time axis split (W > 1): the c_t of a forward sweep (k_big.hip: BigSweepArgs.Uc)
    Layout L;
    hipStream_t stream;
    struct EventPair* pool; int pool_used;
    // state
    double *Y, *Syy;                // [N][T][K], [N][K] (sum_t y^2)
    double *X[2]; int cur;          // ping-pong [N][T][DP], rows in accumulator order (xpos)
    double *A_mean, *A_var, *C_mean, *C_var;
    double *Q_a, *Q_b, *R_a, *R_b;
    double *qld_A, *qld_C;          // [N][D]
    Priors pri;
    double *pri_block;
    // derived
    double *Sigma, *qld_x;          // as of the last X update: [N][3][D][D], [N][3]
    double *Sigma_new, *qld_x_new;  // written by k_prep for the current parameters
    double *gains;                  // [N][L.gains_total]
    double *scratch;                // [N][2][DP][DP] (M_C, M_A + M_C)
    int *warm;                      // [N][2]
    double *zeros;                  // [64] zeros: the row that k_stats reads where there is no row
    double *trash;                  // [N][256] dump rows for masked-out stores of the sweep
    double *U; bool u_valid;        // [N][T][DP] c_t = F mu_{t-1} + G y_t written by the forward sweep for the backward one that follows it
    double *stats; int nchunk, chunk_len;   // [N][nchunk][L.stats_total]
    int W;                          // wavefronts per replicate in the sweeps (time split; 1 unless N is small)
    double *sxx; bool sxx_valid;    // [N][W][DP][DP] interior sum of mu mu^T from the backward sweep (valid while X is that sweep's result)
    double *mom;                    // [N][3 D^2 + K D + D] second moments (k_moments)
    double *resQ, *resR;            // [N][D], [N][K]
    double *elbo, *elbo_sum;        // [N][6], [6]
    int *status;                    // device flag: nonzero if a Cholesky failed
    // host-side validity tracking
    bool gains_valid, stats_valid, resQ_valid, resR_valid;
    int fresh_count; unsigned char* fresh;  // X_t updated since the parameters last changed
    bool mixed_cov;                         // the X_t hold covariances of different parameter generations
    bool classes_valid;                     // Sigma / qld_x describe the X_t (after the first complete sweep, or set by the caller)
    bool timing; KernelTimer timers[PYVB_K_COUNT]; int timing_errors;
    pyvb_comm* comm; int rank, world;
    // ---- Wishart noise precisions (nodes_todo.py:205-234): dense expectations, dense column covariances
    bool dense;                     // noise == PYVB_NOISE_WISHART
    double *Q_w, *R_w;              // [N][D][D], [N][K][K] posterior qw as the reference stores it (qv is in Q_a / R_a)
    double *Qbar, *Rbar;            // [N][D][D], [N][K][K] E[Lambda] = qv inv(sym qw)
    double *lnd;                    // [N][4]: ln det E[Q], ln det E[R], ln det sym(Q_w), ln det sym(R_w)
    double *QA, *RC;                // [N][D][D] E[Q]<A>, [N][K][D] E[R]<C>
    double *trA, *trC;              // [N][D] tr(S_i E[Q]), tr(S'_i E[R])
    double *A_cov, *C_cov;          // [N][D][cov_stride(D)], [N][D][cov_stride(K)] column covariances, upper 8 x 8 tiles (k_wishart.hip)
    double *SyyF;                   // [N][K][K] sum_t y y^T
    double *RQ, *RR;                // [N][D][D], [N][K][K]: sum over children of 1/2<xx^T> + 1/2<mu mu^T> - <x><mu>^T
    bool expect_valid;              // Qbar, Rbar, lnd belong to the current Q_w, R_w
    double *ldm;                    // [N][2][D] ln det of the covariance of the unknown entries of partially known columns of A / C
    double *SG; bool sg_valid[2];   // [N][2][64][64] sum_i G[i,i] S_i over the columns of A / C (k_cols_wishart); valid while neither the
                                    // covariances nor the statistics have changed since
    // ---- outputs with missing entries (k_missing.hip); allocated when set_observations sees NaN
    bool has_missing;
    double *Yobs, *Yvar, *Yqld, *Yent;      // [N][T][K] observations (NaN = missing), [N][T][K] variances, [N][T], [N]
    double *Yld, *YcovS;                    // Wishart noise: [N][T] ln det of each row's covariance of missing entries, [N][K][K] sum_t qcov_t
    // ---- the lower bound does not feed the next iteration: inside pyvb_lds_iterate it runs on a side stream
    hipStream_t side;
    hipEvent_t ev_params, ev_elbo;  // parameters of this iteration complete (main) / lower bound of it read them (side)
    bool elbo_in_flight;            // ev_elbo has been recorded and not yet waited for by the main stream
    double* elbo_hist; int hist_count;      // [PYVB_ELBO_HISTORY][8]: parts summed over replicates (and ranks), one row per iteration
};
#define PYVB_ELBO_HISTORY 4096

// ---- launchers implemented in the kernel translation units ----
int launch_prep(pyvb_lds* h);
int launch_sweep(pyvb_lds* h, int direction, bool keep_x = true);
int launch_step(pyvb_lds* h, int t);
int launch_syy(pyvb_lds* h);
int launch_permute(pyvb_lds* h, const double* src, double* dst, int to_internal);
int launch_stats(pyvb_lds* h, bool with_sxx);   // with_sxx = false: Sxx comes from the backward sweep (h->sxx)
int launch_moments(pyvb_lds* h, bool sxx_from_sweep);
int launch_observe(pyvb_lds* h);
int launch_cols(pyvb_lds* h, int which, int c0, int c1, int fuse = 0);   // which: 0 = A, 1 = C, 2 = both; columns [c0, c1); fuse: see k_cols.hip
int launch_resid(pyvb_lds* h, int which);     // 0 = Q, 1 = R
int launch_noise(pyvb_lds* h, int which);
int launch_elbo(pyvb_lds* h, hipStream_t stream = nullptr);                           // stream: the handle's main one unless given
int launch_elbo_sum(pyvb_lds* h, double* out = nullptr, hipStream_t stream = nullptr);    // out: h->elbo_sum unless given
// k_big.hip
int launch_prep_big(pyvb_lds* h);
int launch_sweep_big(pyvb_lds* h, int direction);
int launch_step_big(pyvb_lds* h, int t);
int launch_stats_big(pyvb_lds* h);
int launch_cols_big(pyvb_lds* h, int which, int c0, int c1, int fuse);
// k_wishart.hip
int launch_wexpect(pyvb_lds* h);                    // Qbar, Rbar, lnd from Q_w, R_w
int launch_dense_pre(pyvb_lds* h);                  // QA, RC, trA, trC
int launch_cols_dense(pyvb_lds* h, int which, int c0, int c1);
int launch_wresid(pyvb_lds* h, int which, int update);
int launch_syy_full(pyvb_lds* h);
int launch_elbo_dense(pyvb_lds* h, hipStream_t stream = nullptr);
int launch_colvar_to_cov(pyvb_lds* h);              // A_var/C_var (diagonals) -> A_cov/C_cov
int launch_cov_to_colvar(pyvb_lds* h);              // and back
int launch_cov_observe(pyvb_lds* h);                // zero covariance of the fully known columns
int launch_cov_convert(pyvb_lds* h, int which, double* dense, int n0, int count, int to_packed);   // dense [count][D][rows][rows] (device) <-> the tiles of replicates n0..
// k_wishart_big.hip: the same on the second shape class (64 < max(D, K) <= 128)
int launch_wexpect_big(pyvb_lds* h);
int launch_dense_pre_big(pyvb_lds* h);
int launch_cols_dense_big(pyvb_lds* h, int which, int c0, int c1);
int launch_wresid_big(pyvb_lds* h, int which, int update);
int launch_syy_full_big(pyvb_lds* h);
// k_missing.hip
int launch_missing_init(pyvb_lds* h, const double* Yq0, const double* Yrowvar0);     // device pointers or null
int launch_impute(pyvb_lds* h);
int launch_syy_missing(pyvb_lds* h);
int launch_impute_dense(pyvb_lds* h);                 // Wishart noise: [y.update() for y in Ys if not y.observed]
int launch_missing_ent_dense(pyvb_lds* h, int diag_cov);

// communicators, shared by the LDS and the PCA path (api.hip): RCCL, or the caller's own host-side all-reduce
// (pyvb_*_comm_init_host: the sums travel through host memory -- a rehearsal transport for boxes where RCCL cannot run)
struct pyvb_comm {
    void* nccl;                                    // ncclComm_t, or null
    pyvb_host_allreduce_fn fn; void* user;         // host transport
    double* host; size_t cap;                      // its staging buffer
};
int pyvb_comm_create(pyvb_comm** comm, const char id[128], int rank, int world);
int pyvb_comm_create_host(pyvb_comm** comm, pyvb_host_allreduce_fn fn, void* user);
void pyvb_comm_free(pyvb_comm* comm);
int pyvb_allreduce_f64(pyvb_comm* comm, double* buf, size_t count, hipStream_t stream);

void pyvb_set_error(const char* fmt, ...);
int pyvb_hip_fail(hipError_t e, const char* what, const char* file, int line);
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return pyvb_hip_fail(_e, #x, __FILE__, __LINE__); } while (0)

// Timed launch bracket.  When timing is on, an event pair from the handle's pool is recorded
// around the launch on the handle's stream; nothing synchronises until pyvb_lds_timing_get(),
// so timing can stay on inside a measured region.
#define PYVB_EVENT_POOL 2048
struct EventPair { hipEvent_t e0, e1; int kernel; };
struct TimedLaunch {
    pyvb_lds* h; int slot; hipStream_t s;
    TimedLaunch(pyvb_lds* h_, int k_, hipStream_t s_ = nullptr);
    ~TimedLaunch();
};
void pyvb_timing_resolve(pyvb_lds* h);
