/*
 * pyvb_hip.h -- C ABI of libpyvb_hip.so: the MI355X (gfx950) implementation of pyvb's
 * variational update loop for the linear-dynamical-system graph, batched over N
 * independent replicates.
 *
 * The reference (jameshensman/pyvb) is pure Python and has no FFI layer; the boundary it
 * exposes is the node/network Python API (the files under src/pyvb/nodes and src/pyvb/network.py).
 * Each entry point below therefore names the reference METHODS whose work it performs for
 * every node of one class at once.  pyvb_amd/ binds these symbols with ctypes
 * (pyvb_amd/_capi.py) and re-exposes the node API on top of them; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, opaque handle, one handle per GPU; global state: the thread-local error string and the RCCL entry
 *     points (dlopen'ed once, under a lock, when a communicator is first asked for); thread-compatible (one host
 *     thread per handle).
 *   - every function returns 0 on success or a pyvb_status; pyvb_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - all arrays are caller-owned HOST buffers of float64, C-contiguous, leading axis N
 *     (replicate); a NULL array pointer means "skip this one".  Device memory is owned
 *     by the handle.
 *   - all work is queued on the handle's HIP stream; getters synchronise.
 *
 * Array shapes (D latent dim, K observed dim, T time steps, N replicates)
 *   Y[N][T][K]  X[N][T][D]
 *   A_mean[N][D][D]  (row, col)        A_colvar[N][D][D]  (column i, entry k): diagonal of
 *   C_mean[N][K][D]  (row, col)        C_colvar[N][D][K]   the covariance of column i
 *   Q_a,Q_b[N][D]   R_a,R_b[N][K]      (PYVB_NOISE_GAMMA: all entries of a row are equal)
 *   Sigma[N][3][D][D], qld_x[N][3]     posterior covariance / q_ln_det of X_0, X_interior, X_{T-1}
 *   elbo parts [N][6] = L_X, L_Y, L_A, L_C, L_Q, L_R
 */
#ifndef PYVB_HIP_H
#define PYVB_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pyvb_lds pyvb_lds;

typedef enum {
    PYVB_OK = 0,
    PYVB_E_ARG = 1,          /* bad argument (shape limits: 2 <= T, 1 <= D,K <= 128) */
    PYVB_E_HIP = 2,          /* HIP runtime error */
    PYVB_E_LINALG = 3,       /* a posterior precision was not positive definite (numpy LinAlgError in the reference) */
    PYVB_E_STALE = 4,        /* statistics requested while the X_t were updated under different parameters */
    PYVB_E_RCCL = 5,         /* RCCL error */
    PYVB_E_UNSUPPORTED = 6
} pyvb_status;

enum { PYVB_NOISE_DIAGONAL_GAMMA = 0,   /* nodes_todo.py:159-204 DiagonalGamma */
       PYVB_NOISE_GAMMA = 1,            /* nodes_todo.py:88-157  Gamma (isotropic) */
       PYVB_NOISE_WISHART = 2 };        /* nodes_todo.py:205-234 Wishart (dense precisions; see pyvb_lds_set_wishart_priors) */
enum { PYVB_FORWARD = 0, PYVB_BACKWARD = 1 };

/* which kernels pyvb_lds_timing_get() reports on */
enum { PYVB_K_PREP = 0, PYVB_K_SWEEP_FWD = 1, PYVB_K_STATS = 2, PYVB_K_PARAMS = 3, PYVB_K_STEP = 4,
       PYVB_K_SWEEP_BWD = 5, PYVB_K_ELBO = 6, PYVB_K_GY = 7 /* 128-wide class: G y_t ahead of a sweep */, PYVB_K_COUNT = 8 };

const char* pyvb_last_error(void);
int pyvb_version(void);
/* first 32 hex digits of the sha256 over the library's source files (pyvb_amd/csrc/Makefile: BUILD_ID): lets a caller check
 * that a prebuilt libpyvb_hip.so is the build of the sources beside it */
const char* pyvb_build_id(void);
int pyvb_device_count(int* count);

/* Graph construction: Linear_Dynamic_System.py:46-66 for N replicates
 * (2T Gaussian, 2 hstack of D Gaussian columns, 2 noise nodes per replicate). */
int pyvb_lds_create(pyvb_lds** out, int device, int N, int T, int D, int K, int noise_kind);
int pyvb_lds_destroy(pyvb_lds* h);

/* Constant parents (node.py:279-311), shared by all replicates:
 *   x0_mean[D], x0_prec[D][D]                     Gaussian(q, pmu, pprec) for X_0  (:58)
 *   A_prior_mean[D][D] (row,col), A_prior_prec[D][D] (column i, diagonal entry k)   (:47)
 *   C_prior_mean[K][D], C_prior_prec[D][K]                                          (:49)
 *   Q_a0,Q_b0[D], R_a0,R_b0[K]                    DiagonalGamma / Gamma priors      (:51-54)
 * Dense prior precisions for the columns are not supported (PYVB_E_UNSUPPORTED is the caller's check). */
int pyvb_lds_set_priors(pyvb_lds* h, const double* x0_mean, const double* x0_prec,
                        const double* A_prior_mean, const double* A_prior_prec,
                        const double* C_prior_mean, const double* C_prior_prec,
                        const double* Q_a0, const double* Q_b0, const double* R_a0, const double* R_b0);

/* Wishart noise precisions (Linear_Dynamic_System.py:55-56; nodes_todo.py:205-234), for handles created with
 * PYVB_NOISE_WISHART: Q = Wishart(D, Q_v0, Q_w0[D][D]), R = Wishart(K, R_v0, R_w0[K][K]); call after pyvb_lds_set_priors
 * (whose Gamma arguments may then be NULL).  State: qv (fixed by the graph, update_v :224-227) and qw ([N][D][D] /
 * [N][K][K], update :228-231); E[Lambda] = qv * inv(qw) (:233-234).  The columns of A and C then have dense posterior
 * covariances A_cov[N][D][D][D] (column i: [D][D]) and C_cov[N][D][K][K]; pyvb_lds_set_state's A_colvar / C_colvar give
 * diagonal initial ones.  Above 64 dimensions (k_wishart_big.hip) Wishart noise does not combine with known entries of A / C or
 * with outputs that hold NaN: pyvb_lds_set_column_observations / pyvb_lds_set_observations return PYVB_E_UNSUPPORTED there.
 * Deviations from the unfinished reference class (prior not mutated by update(), symmetric part
 * of qw in the expectation, a defined lower bound) are listed at the top of pyvb_amd/csrc/k_wishart.hip. */
int pyvb_lds_set_wishart_priors(pyvb_lds* h, double Q_v0, const double* Q_w0, double R_v0, const double* R_w0);
int pyvb_lds_set_wishart_state(pyvb_lds* h, const double* Q_w, const double* R_w);
int pyvb_lds_get_wishart_state(pyvb_lds* h, double* Q_v, double* Q_w, double* R_v, double* R_w);
int pyvb_lds_set_column_cov(pyvb_lds* h, const double* A_cov, const double* C_cov);
int pyvb_lds_get_column_cov(pyvb_lds* h, double* A_cov, double* C_cov);

/* Gaussian.observe for every Y_t (gaussian.py:74-100).  NaN = missing: a row with some NaN is partially observed, a row of
 * NaN is not observed at all; such Y_t are variational nodes of their own (DiagonalGamma / Gamma noise only):
 *   pyvb_lds_set_output_state  their initial posterior (mean Yq[N][T][K], isotropic variance Yrowvar[N][T]; rows without NaN
 *                              ignored) -- until its first update a partially observed node keeps the constructor's draw in
 *                              ALL its entries (observe() only records the known values); default N(0, I)
 *   pyvb_lds_update_Y          [y.update() for y in Ys if not y.observed]  (gaussian.py:102-134, parents only)
 *   pyvb_lds_get_outputs       their posterior means and variances ([N][T][K] each; fully observed rows: value, 0) and the
 *                              q_ln_det of the rows updated so far ([N][T], NaN otherwise); NULL = skip
 * The example's loop (pyvb_lds_iterate) never updates the outputs; call pyvb_lds_update_Y where the script does. */
int pyvb_lds_set_observations(pyvb_lds* h, const double* Y);
int pyvb_lds_set_output_state(pyvb_lds* h, const double* Yq, const double* Yrowvar);
int pyvb_lds_update_Y(pyvb_lds* h);
int pyvb_lds_get_outputs(pyvb_lds* h, double* Yq, double* Yvar, double* Yqld);

/* As[i].observe(v) / Cs[i].observe(v) (gaussian.py:74-100; examples/LDS_knowns_in_A.py:73-74): known entries of the
 * transition / observation matrices, A_obs[D][D] and C_obs[K][D] as (row, col), NaN = not observed; NULL = leave.
 * Call after set_state: fully known columns take their value at once, partially known ones at their next update. */
int pyvb_lds_set_column_observations(pyvb_lds* h, const double* A_obs, const double* C_obs);

/* Explicit posterior state instead of the constructors' random initialisation
 * (gaussian.py:70-72, nodes_todo.py:119,177). */
int pyvb_lds_set_state(pyvb_lds* h, const double* X, const double* A_mean, const double* A_colvar,
                       const double* C_mean, const double* C_colvar, const double* Q_b, const double* R_b);
int pyvb_lds_get_state(pyvb_lds* h, double* X, double* A_mean, double* A_colvar, double* C_mean, double* C_colvar,
                       double* Q_a, double* Q_b, double* R_a, double* R_b);
/* qcov / q_ln_det of the X_t as of their last update (gaussian.py:119-120); qld of the columns. */
int pyvb_lds_get_posterior_classes(pyvb_lds* h, double* Sigma, double* qld_x);
/* Initial covariances of the X_t, one per class (the reference draws an individual random one per node, gaussian.py:70-72;
 * they are overwritten by the first sweep).  Without this call, statistics / parameter updates / the lower bound return
 * PYVB_E_STALE until a complete sweep has run.  qld_x may be NULL. */
int pyvb_lds_set_posterior_classes(pyvb_lds* h, const double* Sigma, const double* qld_x);
int pyvb_lds_get_column_qld(pyvb_lds* h, double* qld_A, double* qld_C);

/* [x.update() for x in Xs] in forward or reversed order (Linear_Dynamic_System.py:70-73):
 * Gaussian.update gaussian.py:102-123 with Multiplication.pass_up_m1_m2 node.py:182-232 and
 * pass_down_Ex :235-242 for all T nodes of all replicates in one launch. */
int pyvb_lds_sweep(pyvb_lds* h, int direction);
/* Xs[t].update() alone. */
int pyvb_lds_update_x(pyvb_lds* h, int t);
/* [a.update() for a in As] / Cs: Gaussian.update + hstack.pass_up_m1_m2 nodes_todo.py:43-62 */
int pyvb_lds_update_A(pyvb_lds* h);
int pyvb_lds_update_C(pyvb_lds* h);
/* As[i].update() for i in [col_begin, col_end) in order (which = 0), or the same for Cs (which = 1) */
int pyvb_lds_update_columns(pyvb_lds* h, int which, int col_begin, int col_end);
/* Q.update() / R.update(): nodes_todo.py:130-138, :187-190 with Multiplication.pass_down_ExxT node.py:244-276 */
int pyvb_lds_update_Q(pyvb_lds* h);
int pyvb_lds_update_R(pyvb_lds* h);
/* sum of log_lower_bound() per node class (network.py:49; gaussian.py:136-151; nodes_todo.py:149-157,:199-204),
 * reference mode (quirks Q1, Q2 of SURVEY.md reproduced). Leaves the parts on the device. */
int pyvb_lds_elbo(pyvb_lds* h);
int pyvb_lds_get_elbo(pyvb_lds* h, double* parts);
/* parts summed over this handle's replicates, then over all ranks if a communicator is attached
 * (one ncclAllReduce of 6 doubles). */
int pyvb_lds_elbo_total(pyvb_lds* h, double out[6]);
/* niters passes of: forward sweep, backward sweep, A, C, Q, R, ELBO (the example's loop body plus network.py:49). */
int pyvb_lds_iterate(pyvb_lds* h, int niters);
/* pyvb_lds_iterate evaluates the lower bound of every iteration on a side stream (it feeds nothing in the next one) and
 * keeps the six parts, summed over the replicates and -- with a communicator attached -- over all ranks, in a ring of the
 * last 4096 iterations.  out[count][6], oldest first; synchronises. */
int pyvb_lds_get_elbo_history(pyvb_lds* h, double* out, int max_count, int* count);
int pyvb_lds_reset_elbo_history(pyvb_lds* h);
int pyvb_lds_sync(pyvb_lds* h);

/* HIP-event timing of the kernels on the handle's stream (for bench.py's roofline figures). */
int pyvb_lds_timing_enable(pyvb_lds* h, int on);
int pyvb_lds_timing_reset(pyvb_lds* h);
int pyvb_lds_timing_get(pyvb_lds* h, int kernel, double* total_ms, int* launches);
/* device-side diagnostics: warm-up lengths chosen for the segmented sweeps [N][2] */
int pyvb_lds_get_warmup(pyvb_lds* h, int* warm);
/* Wavefronts per replicate in the sweeps (the time axis is dealt out to W of them when there are few replicates;
 * pyvb_lds_create picks W, = 1 from about 1024 replicates on).  The setter lets a test run the W = 1 code path
 * (the one the headline workload uses) on a handful of replicates; results do not depend on W beyond rounding. */
int pyvb_lds_get_time_split(pyvb_lds* h, int* W);
int pyvb_lds_set_time_split(pyvb_lds* h, int W);

/* Multi-GPU: replicates are sharded over ranks; the only exchange is the ELBO all-reduce. */
int pyvb_comm_unique_id(char id[128]);
int pyvb_lds_comm_init(pyvb_lds* h, const char id[128], int rank, int world);
int pyvb_lds_comm_destroy(pyvb_lds* h);
/* The same exchange through the caller's own transport instead of RCCL: fn sums buf[0..count) over all ranks in place
 * (host memory, blocking, every rank calls it in the same order) and returns 0.  A rehearsal transport -- several ranks
 * on ONE GPU, where RCCL refuses duplicate devices, or hosts without a working RCCL: every collective then costs a
 * device-host round trip and a stream synchronisation, so it is never what a scaling number should be measured with. */
typedef int (*pyvb_host_allreduce_fn)(double* buf, size_t count, void* user);
int pyvb_lds_comm_init_host(pyvb_lds* h, pyvb_host_allreduce_fn fn, void* user, int rank, int world);

/* ------------------------------------------------------------------------------------------------
 * VB-PCA with missing data: the graph of examples/PCA_missing_data.py:31-42 --
 *   W = hstack(q Gaussian columns of dim d), Mu Gaussian(d), Beta = Gamma(d), Z_n ~ N(0, I),
 *   X_n ~ N(W * Z_n + Mu, Beta), X_n.observe(row with NaN for missing entries), n = 0..N-1.
 * One handle holds N rows of one model; with several GPUs the rows are sharded (N_total rows in all,
 * this handle's first row has global index row_offset) and the sums over n are all-reduced over RCCL.
 * Arrays: X[N][d], Z[N][q], W_mean[d][q] (row, col), W_var[q][d] (column i, entry k), Z_cov[q][q],
 * Mu_mean[d], Mu_var[d], X_rowvar[N] (variance of the missing entries of row n), beta_ab[2] = (qa, qb).
 * Limits: d <= 256, q <= 32. */
typedef struct pyvb_pca pyvb_pca;
int pyvb_pca_create(pyvb_pca** out, int device, long N, int d, int q, long N_total, long row_offset);
int pyvb_pca_destroy(pyvb_pca* h);
/* Constant parents of the W columns and of Mu (diagonal precisions), Gamma hyper-parameters (PCA_missing_data.py:31-34) */
int pyvb_pca_set_priors(pyvb_pca* h, const double* W_prior_mean, const double* W_prior_prec,
                        const double* Mu_prior_mean, const double* Mu_prior_prec, double beta_a0, double beta_b0);
/* [x.observe(row) for x in Xs] (gaussian.py:74-100): NaN = missing; rows may be fully, partially or not observed */
int pyvb_pca_set_data(pyvb_pca* h, const double* X);
/* explicit posterior state; X_missing[N][d] supplies the posterior means of the missing entries only */
int pyvb_pca_set_state(pyvb_pca* h, const double* X_missing, const double* W_mean, const double* Z, const double* Z_cov,
                       const double* Mu_mean, const double* beta_b);
/* The X_n as Gaussian.__init__ leaves them (gaussian.py:70-72; observe() with NaN changes neither qmu nor qcov, :90-96):
 * a row that is not fully observed carries the posterior mean X_full[n] at ALL its d entries and the covariance
 * row_var[n] * I, messages with them, and is conditioned on its observed entries by its first update (:125-134).
 * Call after pyvb_pca_set_data; rows without missing entries are ignored.  Without this call the observed entries of a
 * partially observed row count as pinned from the start (what pyvb_pca_set_state describes). */
int pyvb_pca_set_unpinned_rows(pyvb_pca* h, const double* X_full, const double* row_var);
/* diagonals of the initial covariances of the W columns [q][d] and of Mu [d] (either may be NULL): read only by updates that
 * come before the first update of the node itself -- Z or Beta before W, Beta before Mu; the crawl order never does that */
int pyvb_pca_set_initial_variances(pyvb_pca* h, const double* W_var, const double* Mu_var);
int pyvb_pca_get_state(pyvb_pca* h, double* X, double* X_rowvar, double* W_mean, double* W_var, double* Z, double* Z_cov,
                       double* Mu_mean, double* Mu_var, double* beta_ab);
/* q_ln_det of the nodes (gaussian.py:120, the quantity of quirk Q1 that log_lower_bound reads, gaussian.py:147), as their
 * updates on THIS handle left it; NaN for a node that has not been updated on it (the reference, too, sets it only in
 * update()).  qld_W [q]; qld_Z, qld_Mu one double each (all Z_n share one); qld_X [N]: rows without any observed entry,
 * NaN for the others.  Any pointer may be NULL. */
int pyvb_pca_get_qld(pyvb_pca* h, double* qld_W, double* qld_Z, double* qld_Mu, double* qld_X);
/* [w.update() for w in Ws]; [z.update() for z in Zs]; Xs[lo:hi] updates; Mu.update(); Beta.update()
 * (gaussian.py:102-134, nodes_todo.py:130-138).  Results are what the reference's order of node updates gives, call by call; the work
 * behind pyvb_pca_update_Z is scheduled lazily: the call forms the shared posterior covariance of the Z_n, the gains and the sum of
 * the new means (which is linear in the sum of x), and the rows of Z are written by the next pass over the rows -- normally the
 * pyvb_pca_update_X that follows, which then reads X once for both updates -- or by the first call that reads or replaces Z, X or
 * the parameters (pyvb_pca_get_state, the setters). */
int pyvb_pca_update_W(pyvb_pca* h);
int pyvb_pca_update_Z(pyvb_pca* h);
int pyvb_pca_update_X(pyvb_pca* h, long lo, long hi);
/* Xs[0].update() of the GLOBAL row 0 (the crawl order updates it alone, before Mu).  With a communicator attached every update
 * call is a collective: all ranks issue the same calls in the same order, pyvb_pca_update_X with their local part of the range
 * (possibly empty).  This one is the single-row step for every rank: the owner of global row 0 updates it, all ranks exchange
 * the change of sum x.  (Without a communicator pyvb_pca_update_X(h, 0, 1) takes this shortcut by itself; with one it does not
 * -- the two are different collectives -- and runs the general row-range update, which every rank matches by calling
 * pyvb_pca_update_X with its own, possibly empty, part.) */
int pyvb_pca_update_X0(pyvb_pca* h);
int pyvb_pca_update_Mu(pyvb_pca* h);
int pyvb_pca_update_Beta(pyvb_pca* h);
/* sum of log_lower_bound() per node class: parts[5] = W columns, Z_n, X_n, Mu, Beta (NULL: leave on the device) */
int pyvb_pca_elbo(pyvb_pca* h, double parts[5]);
/* niters passes of Network.learn's loop body (network.py:46-49) in the crawl order of fetch_network:
 * W columns, Z_0.., X_0, Mu, X_1.., Beta, lower bound */
int pyvb_pca_iterate(pyvb_pca* h, int niters);
int pyvb_pca_sync(pyvb_pca* h);
int pyvb_pca_comm_init(pyvb_pca* h, const char id[128], int rank, int world);
int pyvb_pca_comm_init_host(pyvb_pca* h, pyvb_host_allreduce_fn fn, void* user, int rank, int world);

/* ------------------------------------------------------------------------------------------------
 * Generic graphs, node by node (network.py:40-56 over arbitrary node lists; src/tests.py:9-202): every posterior,
 * constant, message and temporary of one graph lives in a device arena of doubles; the work of one reference method --
 *   Gaussian.update gaussian.py:102-134, Gaussian.log_lower_bound :136-151, Gaussian.pass_up_m1_m2 :179-183,
 *   Addition.pass_up_m1_m2 / pass_down_* node.py:95-129, Multiplication.* node.py:182-276, hstack.* nodes_todo.py:33-62,
 *   Gamma.* :125-157, DiagonalGamma.* :183-204, Wishart.* :224-234, Constant.* node.py:304-311
 * -- is a TAPE of small dense operations on arena offsets (records of 8 int32: opcode, dst, a, b, m, n, p, flags; the
 * opcodes are documented in pyvb_amd/csrc/k_tape.hip and mirrored by pyvb_amd/generic.py), interpreted by one workgroup
 * per launch.  Tapes are uploaded once and replayed (the graph is static).  pyvb_graph_tape_create checks every extent a
 * record touches against the arena (PYVB_E_ARG names the record); gather / scatter indices are data: the kernel skips one
 * that points outside and the next pyvb_graph_sync / pyvb_graph_read returns PYVB_E_ARG. */
typedef struct pyvb_graph pyvb_graph;
int pyvb_graph_create(pyvb_graph** out, int device, size_t arena_doubles);
int pyvb_graph_destroy(pyvb_graph* g);
int pyvb_graph_write(pyvb_graph* g, size_t offset, const double* src, size_t n);
int pyvb_graph_read(pyvb_graph* g, size_t offset, double* dst, size_t n);
int pyvb_graph_tape_create(pyvb_graph* g, const int* ops, int nops, int* tape_id);
/* Optional: how pyvb_graph_tape_run issues the tape -- launches[nl][2] = (first block, number of blocks), in order, and
 * blocks[nb][2] = (first record, number of records); the blocks of one launch run side by side, one workgroup each (the
 * updates of nodes none of which reads what another writes: [z.update() for z in Zs] of a PCA-like graph).  The caller
 * guarantees that independence; blocks and launches must tile the tape in order (checked).  Without a program one workgroup
 * interprets the whole tape.  What a block "writes" includes the GAPS of a strided destination: the extent of a COPY2D or FILL
 * with ld > cols is the whole span (rows - 1) * ld + cols, which a block may hold in LDS from its start and write back at its
 * end -- that span, gaps included, must be disjoint from everything any other block of the same launch writes. */
int pyvb_graph_tape_set_program(pyvb_graph* g, int tape_id, const int* blocks, int nblocks, const int* launches, int nlaunches);
int pyvb_graph_tape_run(pyvb_graph* g, int tape_id);
int pyvb_graph_tape_destroy(pyvb_graph* g, int tape_id);
int pyvb_graph_sync(pyvb_graph* g);

#ifdef __cplusplus
}
#endif
#endif
