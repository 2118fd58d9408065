// Generic per-node path: graphs the recognisers (pyvb_amd/_recognise.py) have no fused plan for run node by node.
// Every quantity of such a graph -- posterior parameters, constants, messages, temporaries -- lives in one device
// arena of doubles; what a node's update(), pass_up_m1_m2(), pass_down_*() or log_lower_bound() computes in the reference
// (gaussian.py:102-183, node.py:95-129,182-276, nodes_todo.py:33-62,125-157,183-204,224-234) is emitted by the host
// (pyvb_amd/generic.py) as a TAPE of small dense operations on arena offsets, and one workgroup interprets the tape:
// one launch per node update, or per whole Network.learn iteration, instead of one launch per numpy call.
// The matrices of this path are tiny (dimensions of single nodes); the interpreter favours generality over speed.
#include "common.h"
#include <vector>

enum {
    T_NOP = 0,
    T_COPY2D = 1,     // dst[i*p0 + j] = a[i*p1 + j]                         i < m, j < n       (p0, p1: leading dimensions)
    T_FILL = 2,       // dst[i*p0 + j] = (flags & 1) ? (i == j) : 0
    T_AXPBY = 3,      // dst = alpha a + beta b (m x n, contiguous); alpha = arena[p0], beta = arena[p1]; b < 0: dst = alpha a
    T_GEMM = 4,       // dst[m x n] (+)= op(a)[m x k] op(b)[k x n]; flags 1: a^T, 2: b^T, 4: accumulate, 8: subtract
    T_SCALE = 5,      // dst = a * s (flags 0) or a / s (flags 1), s = arena[b]; m x n
    T_TRACE = 6,      // dst[0] (+)= tr(a[m x m]) (flags 4: accumulate)
    T_DIAG = 7,       // flags 0: dst[m] = diag(a[m x m]); flags 1: dst[m x m] = diag(a[m])
    T_CHOLINV = 8,    // dst[m x m] = inverse of the s.p.d. a[m x m]; arena[b] = 0.5 / sum log diag chol (quirk Q1), arena[b+1] = sum log diag chol; p0 = scratch (2 m^2)
    T_DOT = 9,        // dst[0] (+)= sum_ij a_ij b_ij (m x n); flags 4: accumulate
    T_UNARY = 10,     // dst = f(a) elementwise, m x n; flags: 0 log, 1 digamma, 2 lgamma, 3 reciprocal, 4 negate, 5 exp
    T_GATHER = 11,    // dst[i*n + j] = a[r_i * p + c_j], r = (int)arena[b + i], c = (int)arena[flags + j]
    T_SCATTER = 12,   // dst[r_i * p + c_j] (+)= a[i*n + j], c = (int)arena[(flags & ~T_ACC) + j]; flags & T_ACC: accumulate
    T_MUL = 13,       // dst = a .* b elementwise, m x n
};

#define T_ACC 0x40000000
// record layout: o[0] opcode, o[1] dst, o[2] a, o[3] b (or a leading dimension / scalar offset), o[4] m, o[5] n, o[6] p, o[7] flags

struct TapeArgs { double* arena; size_t arena_n; const int* ops; int nops; int* status; };

#define TAPE_THREADS 256

__device__ static double tape_digamma(double x) {
    // the recurrence below takes 10 - x steps: bounded here, so that no argument (a degenerate qv, -inf, a NaN from bad
    // state) can keep a workgroup spinning.  Not finite: NaN (+inf: +inf); below -64: the reflection formula.
    if (!(x - x == 0.0) || x < -4.5e15) return x > 0 ? x : __builtin_nan("");      // below -2^52 every double is an integer: a pole
    double r = 0.0;
    if (x < -64.0) { r = -M_PI / tan(M_PI * x); x = 1.0 - x; }
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double ser = f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
    return r + log(x) - 0.5 / x - ser;
}

// records [first, first + count) of the tape, interpreted by the calling workgroup
__device__ static void tape_exec(const TapeArgs& t, int first, int count, double* red) {
    double* A = t.arena;
    const int tid = threadIdx.x;
    for (int pc = first; pc < first + count; ++pc) {
        const int* o = t.ops + 8 * pc;
        const int op = o[0], m = o[4], n = o[5], flags = o[7];
        double* dst = A + o[1];
        const double* a = A + o[2];
        const double* b = A + (o[3] < 0 ? 0 : o[3]);
        switch (op) {
        case T_COPY2D:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[(idx / n) * o[3] + idx % n] = a[(idx / n) * o[6] + idx % n];
            break;
        case T_FILL:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[(idx / n) * o[3] + idx % n] = ((flags & 1) && idx / n == idx % n) ? 1.0 : 0.0;
            break;
        case T_AXPBY: {
            const double al = A[o[6]];
            if (o[3] < 0) { for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = al * a[idx]; }
            else { const double be = A[flags]; for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = al * a[idx] + be * b[idx]; }
            break; }
        case T_GEMM: {
            const int k = o[6];
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) {
                const int i = idx / n, j = idx % n;
                double s = 0.0;
                for (int l = 0; l < k; ++l) s += ((flags & 1) ? a[l * m + i] : a[i * k + l]) * ((flags & 2) ? b[j * k + l] : b[l * n + j]);
                if (flags & 8) s = -s;
                dst[idx] = (flags & 4) ? dst[idx] + s : s;
            }
            break; }
        case T_SCALE: {
            const double s = A[o[3]];
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = (flags & 1) ? a[idx] / s : a[idx] * s;
            break; }
        case T_TRACE: case T_DOT: {
            double s = 0.0;
            if (op == T_TRACE) { for (int i = tid; i < m; i += TAPE_THREADS) s += a[i * m + i]; }
            else { for (int idx = tid; idx < m * n; idx += TAPE_THREADS) s += a[idx] * b[idx]; }
            red[tid] = s;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int i = 0; i < TAPE_THREADS; ++i) tot += red[i];
                dst[0] = (flags & 4) ? dst[0] + tot : tot;
            }
            break; }
        case T_DIAG:
            if (flags & 1) { for (int idx = tid; idx < m * m; idx += TAPE_THREADS) dst[idx] = (idx / m == idx % m) ? a[idx / m] : 0.0; }
            else { for (int i = tid; i < m; i += TAPE_THREADS) dst[i] = a[i * m + i]; }
            break;
        case T_CHOLINV: {
            // L (lower, row major) in scratch, column by column; then X = L^{-1} by forward substitution, one thread per
            // column of the identity; then dst = X^T X.  (cho_factor / cho_solve of gaussian.py:118-119.)
            double* L = A + o[6];
            double* X = L + m * m;
            for (int idx = tid; idx < m * m; idx += TAPE_THREADS) L[idx] = a[idx];
            __syncthreads();
            bool bad = false;
            for (int j = 0; j < m; ++j) {
                if (tid == 0) {
                    const double piv = L[j * m + j];
                    if (!(piv > 0.0)) { atomicOr(t.status, 1); L[j * m + j] = nan(""); }
                    else L[j * m + j] = sqrt(piv);
                }
                __syncthreads();
                const double d = L[j * m + j];
                for (int i = j + 1 + tid; i < m; i += TAPE_THREADS) L[i * m + j] /= d;
                __syncthreads();
                const int rem = m - j - 1;                      // trailing update, lower triangle
                for (int idx = tid; idx < rem * rem; idx += TAPE_THREADS) {
                    const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
                    if (c <= i) L[i * m + c] -= L[i * m + j] * L[c * m + j];
                }
                __syncthreads();
            }
            (void)bad;
            if (tid == 0) {
                double s = 0.0;
                for (int j = 0; j < m; ++j) s += log(L[j * m + j]);
                A[o[3]] = 0.5 / s;          // gaussian.py:120: .5 / np.log(np.prod(np.diag(chol)))
                A[o[3] + 1] = s;
            }
            for (int c = tid; c < m; c += TAPE_THREADS) {       // X[:, c] = L^{-1} e_c
                for (int i = 0; i < m; ++i) {
                    double s = (i == c) ? 1.0 : 0.0;
                    for (int l = c; l < i; ++l) s -= L[i * m + l] * X[l * m + c];
                    X[i * m + c] = (i < c) ? 0.0 : s / L[i * m + i];
                }
            }
            __syncthreads();
            for (int idx = tid; idx < m * m; idx += TAPE_THREADS) {
                const int i = idx / m, j = idx % m;
                double s = 0.0;
                for (int l = (i > j ? i : j); l < m; ++l) s += X[l * m + i] * X[l * m + j];
                dst[idx] = s;
            }
            break; }
        case T_UNARY:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) {
                const double x = a[idx];
                double y;
                switch (flags) {
                    case 0: y = log(x); break;
                    case 1: y = tape_digamma(x); break;
                    case 2: y = lgamma(x); break;
                    case 3: y = 1.0 / x; break;
                    case 4: y = -x; break;
                    default: y = exp(x); break;
                }
                dst[idx] = y;
            }
            break;
        case T_GATHER:          // the indices are data: checked here (status bit 1), everything else in pyvb_graph_tape_create
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) {
                const long pos = (long)o[2] + (long)A[o[3] + idx / n] * o[6] + (long)A[flags + idx % n];
                if (pos < 0 || (size_t)pos >= t.arena_n) { atomicOr(t.status, 2); continue; }
                dst[idx] = A[pos];
            }
            break;
        case T_SCATTER:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) {
                const long pos = (long)o[1] + (long)A[o[3] + idx / n] * o[6] + (long)A[(flags & ~T_ACC) + idx % n];
                if (pos < 0 || (size_t)pos >= t.arena_n) { atomicOr(t.status, 2); continue; }
                A[pos] = (flags & T_ACC) ? A[pos] + a[idx] : a[idx];
            }
            break;
        case T_MUL:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = a[idx] * b[idx];
            break;
        default: break;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TAPE_THREADS) k_tape(TapeArgs t) {
    __shared__ double red[TAPE_THREADS];
    tape_exec(t, 0, t.nops, red);
}

// A PROGRAM over a tape: launches in order, each of them a set of record ranges ("blocks") that touch disjoint state and so
// run side by side, one workgroup per block (the updates of nodes none of which reads what another writes: the Z_n of a
// PCA-like graph, its X_n).  blocks: [first record, count] per workgroup of this launch.
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_blocks(TapeArgs t, const int* blocks) {
    __shared__ double red[TAPE_THREADS];
    tape_exec(t, blocks[2 * blockIdx.x], blocks[2 * blockIdx.x + 1], red);
}

struct pyvb_graph {
    int device;
    hipStream_t stream;
    double* arena; size_t arena_n;
    int* status;
    std::vector<int*> tapes; std::vector<int> tape_len;
    std::vector<int*> prog_blocks;                   // per tape: device table [nblocks][2], or null
    std::vector<std::vector<int>> prog_launches;     // per tape: (first block, number of blocks) per launch
};

#define ARGCHK(cond, msg) do { if (!(cond)) { pyvb_set_error("%s", msg); return PYVB_E_ARG; } } while (0)

extern "C" {

int pyvb_graph_create(pyvb_graph** out, int device, size_t arena_doubles) {
    ARGCHK(out && arena_doubles > 0 && arena_doubles < ((size_t)1 << 31), "bad arguments (the arena is addressed with 32-bit offsets)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    ARGCHK(device >= 0 && device < ndev, "no such device");
    HIPCHK(hipSetDevice(device));
    pyvb_graph* g = new pyvb_graph();
    g->device = device; g->arena_n = arena_doubles; g->arena = nullptr; g->status = nullptr; g->stream = nullptr;
    hipError_t e = hipStreamCreate(&g->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&g->arena, arena_doubles * sizeof(double));
    if (e == hipSuccess) e = hipMemset(g->arena, 0, arena_doubles * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&g->status, sizeof(int));
    if (e == hipSuccess) e = hipMemset(g->status, 0, sizeof(int));
    if (e != hipSuccess) { pyvb_graph_destroy(g); return pyvb_hip_fail(e, "pyvb_graph_create", __FILE__, __LINE__); }
    *out = g;
    return PYVB_OK;
}

int pyvb_graph_destroy(pyvb_graph* g) {
    if (!g) return PYVB_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    for (int* t : g->tapes) if (t) (void)hipFree(t);
    for (int* t : g->prog_blocks) if (t) (void)hipFree(t);
    if (g->arena) (void)hipFree(g->arena);
    if (g->status) (void)hipFree(g->status);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return PYVB_OK;
}

int pyvb_graph_write(pyvb_graph* g, size_t offset, const double* src, size_t n) {
    ARGCHK(g && src && n <= g->arena_n && offset <= g->arena_n - n, "write outside the arena");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipMemcpyAsync(g->arena + offset, src, n * sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));        // src is the caller's buffer
    return PYVB_OK;
}

int pyvb_graph_sync(pyvb_graph* g) {
    ARGCHK(g, "handle is NULL");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->stream));
    int st = 0;
    HIPCHK(hipMemcpy(&st, g->status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        HIPCHK(hipMemset(g->status, 0, sizeof(int)));
        if (st & 2) { pyvb_set_error("a gather / scatter index of a tape pointed outside the arena (skipped)"); return PYVB_E_ARG; }
        pyvb_set_error("a posterior precision was not positive definite (numpy.linalg.LinAlgError in the reference)");
        return PYVB_E_LINALG;
    }
    return PYVB_OK;
}

int pyvb_graph_read(pyvb_graph* g, size_t offset, double* dst, size_t n) {
    ARGCHK(g && dst && n <= g->arena_n && offset <= g->arena_n - n, "read outside the arena");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipMemcpyAsync(dst, g->arena + offset, n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    return pyvb_graph_sync(g);
}

// does [off, off + len) lie inside the arena (len == 0: nothing is touched)
static bool tape_fits(const pyvb_graph* g, long off, size_t len) { return len == 0 || (off >= 0 && (size_t)off + len <= g->arena_n); }
static size_t tape_span(int rows, int cols, int ld) { return rows > 0 && cols > 0 ? (size_t)(rows - 1) * (size_t)ld + cols : 0; }

// every extent a record touches, per opcode (the layouts of the enum at the top); gather / scatter indices are data and are
// checked by the kernel
static bool tape_record_ok(const pyvb_graph* g, const int* o) {
    const int op = o[0], m = o[4], n = o[5], p = o[6], flags = o[7];
    const size_t mn = (size_t)m * n, mm = (size_t)m * m;
    if (m < 0 || n < 0) return false;
    switch (op) {
    case T_NOP: return true;
    case T_COPY2D: return o[3] >= n && p >= n && tape_fits(g, o[1], tape_span(m, n, o[3])) && tape_fits(g, o[2], tape_span(m, n, p));
    case T_FILL: return o[3] >= n && tape_fits(g, o[1], tape_span(m, n, o[3]));
    case T_AXPBY: return tape_fits(g, o[1], mn) && tape_fits(g, o[2], mn) && tape_fits(g, p, 1) && (o[3] < 0 || (tape_fits(g, o[3], mn) && tape_fits(g, flags, 1)));
    case T_GEMM: return p >= 0 && tape_fits(g, o[1], mn) && tape_fits(g, o[2], (size_t)m * p) && tape_fits(g, o[3], (size_t)p * n);
    case T_SCALE: return tape_fits(g, o[1], mn) && tape_fits(g, o[2], mn) && tape_fits(g, o[3], 1);
    case T_TRACE: return tape_fits(g, o[1], 1) && tape_fits(g, o[2], mm);
    case T_DOT: return tape_fits(g, o[1], 1) && tape_fits(g, o[2], mn) && tape_fits(g, o[3], mn);
    case T_DIAG: return (flags & 1) ? (tape_fits(g, o[1], mm) && tape_fits(g, o[2], m)) : (tape_fits(g, o[1], m) && tape_fits(g, o[2], mm));
    case T_CHOLINV: return tape_fits(g, o[1], mm) && tape_fits(g, o[2], mm) && tape_fits(g, o[3], 2) && tape_fits(g, p, 2 * mm);
    case T_UNARY: return flags >= 0 && flags <= 5 && tape_fits(g, o[1], mn) && tape_fits(g, o[2], mn);
    case T_GATHER: return p >= 0 && tape_fits(g, o[1], mn) && tape_fits(g, o[2], 1) && tape_fits(g, o[3], m) && tape_fits(g, flags, n);
    case T_SCATTER: return p >= 0 && tape_fits(g, o[1], 1) && tape_fits(g, o[2], mn) && tape_fits(g, o[3], m) && tape_fits(g, flags & ~T_ACC, n);
    case T_MUL: return tape_fits(g, o[1], mn) && tape_fits(g, o[2], mn) && tape_fits(g, o[3], mn);
    default: return false;
    }
}

/* A tape: nops records of 8 int32 (opcode, dst, a, b, m, n, p, flags; see the enum at the top of k_tape.hip).  Every extent a
 * record touches is checked against the arena here, once, so the kernel does not have to. */
int pyvb_graph_tape_create(pyvb_graph* g, const int* ops, int nops, int* tape_id) {
    ARGCHK(g && ops && nops > 0 && tape_id, "bad arguments");
    HIPCHK(hipSetDevice(g->device));
    for (int i = 0; i < nops; ++i) {
        if (!tape_record_ok(g, ops + 8 * i)) {
            pyvb_set_error("tape record %d (opcode %d) is malformed or touches memory outside the arena", i, ops[8 * i]);
            return PYVB_E_ARG;
        }
    }
    int* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)nops * 8 * sizeof(int)));
    HIPCHK(hipMemcpyAsync(d, ops, (size_t)nops * 8 * sizeof(int), hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    g->tapes.push_back(d); g->tape_len.push_back(nops);
    g->prog_blocks.push_back(nullptr); g->prog_launches.emplace_back();
    *tape_id = (int)g->tapes.size() - 1;
    return PYVB_OK;
}

/* How pyvb_graph_tape_run issues the tape: launches[nl][2] = (first block, number of blocks), in order; blocks[nb][2] =
 * (first record, number of records).  The blocks of one launch run side by side: the caller guarantees that none of them
 * reads or writes what another writes.  Every record must belong to exactly one block and the launches must cover the
 * blocks in order (checked). */
int pyvb_graph_tape_set_program(pyvb_graph* g, int tape_id, const int* blocks, int nblocks, const int* launches, int nlaunches) {
    ARGCHK(g && tape_id >= 0 && tape_id < (int)g->tapes.size() && g->tapes[tape_id], "no such tape");
    ARGCHK(blocks && launches && nblocks > 0 && nlaunches > 0, "bad arguments");
    HIPCHK(hipSetDevice(g->device));
    int rec = 0;
    for (int b = 0; b < nblocks; ++b) {
        ARGCHK(blocks[2 * b] == rec && blocks[2 * b + 1] > 0, "the blocks of a program must tile the tape in order");
        rec += blocks[2 * b + 1];
    }
    ARGCHK(rec == g->tape_len[tape_id], "the blocks of a program must cover every record of the tape");
    int blk = 0;
    for (int l = 0; l < nlaunches; ++l) {
        ARGCHK(launches[2 * l] == blk && launches[2 * l + 1] > 0, "the launches of a program must tile the blocks in order");
        blk += launches[2 * l + 1];
    }
    ARGCHK(blk == nblocks, "the launches of a program must cover every block");
    HIPCHK(hipStreamSynchronize(g->stream));
    if (g->prog_blocks[tape_id]) { (void)hipFree(g->prog_blocks[tape_id]); g->prog_blocks[tape_id] = nullptr; }
    int* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)nblocks * 2 * sizeof(int)));
    HIPCHK(hipMemcpy(d, blocks, (size_t)nblocks * 2 * sizeof(int), hipMemcpyHostToDevice));
    g->prog_blocks[tape_id] = d;
    g->prog_launches[tape_id].assign(launches, launches + 2 * nlaunches);
    return PYVB_OK;
}

int pyvb_graph_tape_run(pyvb_graph* g, int tape_id) {
    ARGCHK(g && tape_id >= 0 && tape_id < (int)g->tapes.size() && g->tapes[tape_id], "no such tape");
    HIPCHK(hipSetDevice(g->device));
    TapeArgs t; t.arena = g->arena; t.arena_n = g->arena_n; t.ops = g->tapes[tape_id]; t.nops = g->tape_len[tape_id]; t.status = g->status;
    if (g->prog_blocks[tape_id]) {
        const std::vector<int>& L = g->prog_launches[tape_id];
        for (size_t l = 0; l + 1 < L.size(); l += 2)
            hipLaunchKernelGGL(k_tape_blocks, dim3(L[l + 1]), dim3(TAPE_THREADS), 0, g->stream, t, g->prog_blocks[tape_id] + 2 * L[l]);
    } else
        hipLaunchKernelGGL(k_tape, dim3(1), dim3(TAPE_THREADS), 0, g->stream, t);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int pyvb_graph_tape_destroy(pyvb_graph* g, int tape_id) {
    ARGCHK(g && tape_id >= 0 && tape_id < (int)g->tapes.size(), "no such tape");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->stream));
    if (g->tapes[tape_id]) { (void)hipFree(g->tapes[tape_id]); g->tapes[tape_id] = nullptr; }
    if (g->prog_blocks[tape_id]) { (void)hipFree(g->prog_blocks[tape_id]); g->prog_blocks[tape_id] = nullptr; }
    return PYVB_OK;
}

}  // extern "C"
