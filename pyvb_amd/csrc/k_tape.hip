// Generic per-node path: graphs the recognisers (pyvb_amd/_recognise.py) have no fused plan for run node by node.
// Every quantity of such a graph -- posterior parameters, constants, messages, temporaries -- lives in one device
// arena of doubles; what a node's update(), pass_up_m1_m2(), pass_down_*() or log_lower_bound() computes in the reference
// (gaussian.py:102-183, node.py:95-129,182-276, nodes_todo.py:33-62,125-157,183-204,224-234) is emitted by the host
// (pyvb_amd/generic.py) as a TAPE of small dense operations on arena offsets, and one workgroup interprets the tape:
// one launch per node update, or per whole Network.learn iteration, instead of one launch per numpy call.
// The matrices of this path are tiny (dimensions of single nodes); the interpreter favours generality over speed.
#include "common.h"
#include <algorithm>
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
//
// Operands in LDS.  A record of the plain tape addresses the arena in global memory, and a record costs three dependent
// round trips to it (the record itself, its operands, the drain of its stores before the barrier): about 3 us whatever
// the arithmetic.  When a tape is uploaded the host therefore works out, per block of records that one workgroup
// interprets, which arena extents the block touches (the same per-opcode table that validates the records), merges them
// into segments and, if all of them fit the LDS budget, has the workgroup load them once, run the records out of LDS -- the
// offsets are rewritten to LDS positions and carry T_LDS; the interpreter is instantiated with LDS pointers for such a block
// -- and write the segments it has written back at the end.  (A block whose working set does not fit stays on global memory.)  Blocks of one launch touch disjoint state (the program's
// contract), so the write-back cannot collide; blocks with gather / scatter records, whose addresses are data, are left
// on global memory.  The records of a block are staged in LDS too, in chunks.
#define T_LDS 0x40000000        // in an offset field of a resolved record: position in the workgroup's LDS window, not in the arena
#define TAPE_CHUNK 512          // records staged at a time
#define TAPE_LDS_CAP 12288      // doubles of arena a block may keep in LDS (96 KB)
#define TAPE_MAX_SEGS 4096

struct TapeArgs { double* arena; size_t arena_n; const int* ops; int nops; int* status; };

#ifndef TAPE_THREADS
#define TAPE_THREADS 256    // (one wavefront, 64, measured the same on the node-sized matrices of this path: the barrier per record is not what a record costs)
#endif

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

// `count` records starting at `recs` (global memory, or a chunk staged in LDS), interpreted by the calling workgroup.
// WIN: every operand of every record lies in the workgroup's LDS window `win` (the offsets are window positions, tagged
// T_LDS); the pointers are then LDS pointers and the accesses ds_read / ds_write (a few dozen cycles) instead of flat ones.
typedef __attribute__((address_space(3))) double lds_double;
template <bool WIN> struct TapePtr { typedef double* type; };
template <> struct TapePtr<true> { typedef lds_double* type; };

template <bool WIN>
__device__ static void tape_exec(const TapeArgs& t, const int* recs, int count, double* red, typename TapePtr<WIN>::type win) {
    typedef typename TapePtr<WIN>::type P;
    double* A = t.arena;
    const int tid = threadIdx.x;
    auto at = [&](int off) -> P { if constexpr (WIN) return win + (off & ~T_LDS); else return A + off; };
    for (int pc = 0; pc < count; ++pc) {
        const int* o = recs + 8 * pc;
        const int op = o[0], m = o[4], n = o[5], flags = o[7];
        P dst = at(o[1]);
        const P a = at(o[2]);
        const P b = o[3] < 0 ? at(0) : at(o[3]);
        switch (op) {
        case T_COPY2D:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[(idx / n) * o[3] + idx % n] = a[(idx / n) * o[6] + idx % n];
            break;
        case T_FILL:
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[(idx / n) * o[3] + idx % n] = ((flags & 1) && idx / n == idx % n) ? 1.0 : 0.0;
            break;
        case T_AXPBY: {
            const double al = *at(o[6]);
            if (o[3] < 0) { for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = al * a[idx]; }
            else { const double be = *at(flags); for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = al * a[idx] + be * b[idx]; }
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
            const double s = *b;
            for (int idx = tid; idx < m * n; idx += TAPE_THREADS) dst[idx] = (flags & 1) ? a[idx] / s : a[idx] * s;
            break; }
        case T_TRACE: case T_DOT: {
            double s = 0.0;
            if (op == T_TRACE) { for (int i = tid; i < m; i += TAPE_THREADS) s += a[i * m + i]; }
            else { for (int idx = tid; idx < m * n; idx += TAPE_THREADS) s += a[idx] * b[idx]; }
            // fixed tree: shuffles inside a wavefront, then the four wavefronts in order (one thread adding up 256 values was
            // 7 us per record)
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
#pragma unroll
                for (int w = 0; w < TAPE_THREADS / 64; ++w) tot += red[w];
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
            P L = at(o[6]);
            P X = L + m * m;
            P out2 = at(o[3]);
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
                out2[0] = 0.5 / s;          // gaussian.py:120: .5 / np.log(np.prod(np.diag(chol)))
                out2[1] = s;
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
    tape_exec<false>(t, t.ops, t.nops, red, nullptr);
}

// A PROGRAM over a tape: launches in order, each of them a set of record ranges ("blocks") that touch disjoint state and so
// run side by side, one workgroup per block (the updates of nodes none of which reads what another writes: the Z_n of a
// PCA-like graph, its X_n).  blocks: [first record, count] per workgroup of this launch.
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_blocks(TapeArgs t, const int* blocks) {
    __shared__ double red[TAPE_THREADS];
    tape_exec<false>(t, t.ops + 8 * (size_t)blocks[2 * blockIdx.x], blocks[2 * blockIdx.x + 1], red, nullptr);
}

// The same with the block's working set in LDS.  meta[block][8] = {first record, count, first segment, number of segments, window
// doubles, ...}; segs[seg][4] = {arena offset, length, window offset, written}; t.ops holds the RESOLVED records (cached
// offsets rewritten and tagged T_LDS).  Dynamic LDS: the window, then TAPE_CHUNK staged records.
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_cached(TapeArgs t, const int* meta, const int* segs) {
    extern __shared__ double win[];
    __shared__ double red[TAPE_THREADS];
    const int* me = meta + 8 * blockIdx.x;
    const int first = me[0], count = me[1], s0 = me[2], ns = me[3], wd = me[4];
    const int tid = threadIdx.x;
    // long segments by all threads together, short ones (most: a node's mean, a scalar) one per thread
    for (int s = 0; s < ns; ++s) {
        const int* sg = segs + 4 * (s0 + s);
        if (sg[1] < 64) continue;
        const double* src = t.arena + sg[0];
        double* dst = win + sg[2];
        for (int idx = tid; idx < sg[1]; idx += TAPE_THREADS) dst[idx] = src[idx];
    }
    for (int s = tid; s < ns; s += TAPE_THREADS) {
        const int* sg = segs + 4 * (s0 + s);
        if (sg[1] >= 64) continue;
        const double* src = t.arena + sg[0];
        double* dst = win + sg[2];
        for (int idx = 0; idx < sg[1]; ++idx) dst[idx] = src[idx];
    }
    int* staged = reinterpret_cast<int*>(win + wd);
    for (int c0 = 0; c0 < count; c0 += TAPE_CHUNK) {
        const int nc = count - c0 < TAPE_CHUNK ? count - c0 : TAPE_CHUNK;
        __syncthreads();
        const int* src = t.ops + 8 * (size_t)(first + c0);
        for (int idx = tid; idx < 8 * nc; idx += TAPE_THREADS) staged[idx] = src[idx];
        __syncthreads();
        if (ns > 0) tape_exec<true>(t, staged, nc, red, (lds_double*)win);
        else tape_exec<false>(t, staged, nc, red, nullptr);
    }
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int* sg = segs + 4 * (s0 + s);
        if (!sg[3] || sg[1] < 64) continue;
        double* dst = t.arena + sg[0];
        const double* src = win + sg[2];
        for (int idx = tid; idx < sg[1]; idx += TAPE_THREADS) dst[idx] = src[idx];
    }
    for (int s = tid; s < ns; s += TAPE_THREADS) {
        const int* sg = segs + 4 * (s0 + s);
        if (!sg[3] || sg[1] >= 64) continue;
        double* dst = t.arena + sg[0];
        const double* src = win + sg[2];
        for (int idx = 0; idx < sg[1]; ++idx) dst[idx] = src[idx];
    }
}

struct pyvb_graph {
    int device;
    hipStream_t stream;
    double* arena; size_t arena_n;
    int* status;
    std::vector<int*> tapes; std::vector<int> tape_len;
    std::vector<int*> prog_blocks;                   // per tape: device table [nblocks][2], or null
    std::vector<std::vector<int>> prog_launches;     // per tape: (first block, number of blocks) per launch
    // the LDS-window form of a tape (see the top of the file): resolved records, block table, segment table, LDS bytes per launch
    std::vector<std::vector<int>> host_ops;
    std::vector<int*> c_ops, c_meta, c_segs;
    std::vector<std::vector<size_t>> c_lds;
};

#define ARGCHK(cond, msg) do { if (!(cond)) { pyvb_set_error("%s", msg); return PYVB_E_ARG; } } while (0)
static void tape_cache_free(pyvb_graph* g, int id);

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
    for (size_t i = 0; i < g->c_ops.size(); ++i) tape_cache_free(g, (int)i);
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

// ---- the LDS-window form of a tape (see the top of the file)
struct TapeExtent { long off; size_t len; bool write; int field; };     // field: index of the record field that holds `off`

// the extents record o touches, field by field (the table of tape_record_ok); false: addresses that are data (gather / scatter)
static bool tape_extents(const int* o, std::vector<TapeExtent>& out) {
    const int op = o[0], m = o[4], n = o[5], p = o[6], flags = o[7];
    const size_t mn = (size_t)m * n, mm = (size_t)m * m;
    auto add = [&](int field, size_t len, bool write) { if (len) out.push_back(TapeExtent{(long)o[field], len, write, field}); };
    switch (op) {
    case T_NOP: return true;
    case T_COPY2D: add(1, tape_span(m, n, o[3]), true); add(2, tape_span(m, n, p), false); return true;
    case T_FILL: add(1, tape_span(m, n, o[3]), true); return true;
    case T_AXPBY: add(1, mn, true); add(2, mn, false); add(6, 1, false); if (o[3] >= 0) { add(3, mn, false); add(7, 1, false); } return true;
    case T_GEMM: add(1, mn, true); add(2, (size_t)m * p, false); add(3, (size_t)p * n, false); return true;
    case T_SCALE: add(1, mn, true); add(2, mn, false); add(3, 1, false); return true;
    case T_TRACE: add(1, 1, true); add(2, mm, false); return true;
    case T_DOT: add(1, 1, true); add(2, mn, false); add(3, mn, false); return true;
    case T_DIAG: if (flags & 1) { add(1, mm, true); add(2, m, false); } else { add(1, m, true); add(2, mm, false); } return true;
    case T_CHOLINV: add(1, mm, true); add(2, mm, false); add(3, 2, true); add(6, 2 * mm, true); return true;
    case T_UNARY: add(1, mn, true); add(2, mn, false); return true;
    case T_MUL: add(1, mn, true); add(2, mn, false); add(3, mn, false); return true;
    default: return false;      // gather, scatter
    }
}

static void tape_cache_free(pyvb_graph* g, int id) {
    if (g->c_ops[id]) { (void)hipFree(g->c_ops[id]); g->c_ops[id] = nullptr; }
    if (g->c_meta[id]) { (void)hipFree(g->c_meta[id]); g->c_meta[id] = nullptr; }
    if (g->c_segs[id]) { (void)hipFree(g->c_segs[id]); g->c_segs[id] = nullptr; }
    g->c_lds[id].clear();
}

// Build the window form of tape `id` for the given blocks ([first, count] each) and launches ([first block, number]).
// Nothing is cached when the arena does not leave bit 30 of an offset free.
static int tape_cache_build(pyvb_graph* g, int id, const std::vector<int>& blocks, const std::vector<int>& launches) {
    tape_cache_free(g, id);
    if (g->arena_n >= (size_t)T_LDS) return PYVB_OK;
    std::vector<int> ops = g->host_ops[id];
    const int nb = (int)blocks.size() / 2;
    std::vector<int> meta((size_t)nb * 8, 0), segs;
    std::vector<TapeExtent> ext;
    struct Seg { long off, end; int touches; bool write; int lds; };
    for (int b = 0; b < nb; ++b) {
        const int first = blocks[2 * b], count = blocks[2 * b + 1];
        meta[8 * b] = first; meta[8 * b + 1] = count; meta[8 * b + 2] = (int)segs.size() / 4;
        ext.clear();
        bool ok = count >= 3;                               // a window costs a load and a write-back: not for a record or two
        for (int r = first; ok && r < first + count; ++r) ok = tape_extents(&ops[8 * (size_t)r], ext);
        if (!ok) continue;
        // merge the extents into segments
        std::vector<Seg> sg;
        {
            std::vector<TapeExtent> e2 = ext;
            std::sort(e2.begin(), e2.end(), [](const TapeExtent& x, const TapeExtent& y) { return x.off < y.off; });
            for (const TapeExtent& e : e2) {
                const long end = e.off + (long)e.len;
                if (!sg.empty() && e.off <= sg.back().end) {      // overlapping or adjacent only: a gap may be another block's state
                    if (end > sg.back().end) sg.back().end = end;
                    sg.back().touches += 1; sg.back().write = sg.back().write || e.write;
                } else sg.push_back(Seg{e.off, end, 1, e.write, -1});
            }
        }
        // the most used ones first, while they fit
        std::vector<int> order(sg.size());
        for (size_t i = 0; i < sg.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return sg[x].touches != sg[y].touches ? sg[x].touches > sg[y].touches : sg[x].off < sg[y].off; });
        long used = 0; int nsel = 0;
        for (int i : order) {
            const long len = sg[i].end - sg[i].off;
            if (nsel >= TAPE_MAX_SEGS || used + len > TAPE_LDS_CAP) { nsel = -1; break; }
            sg[i].lds = (int)used; used += (len + 1) & ~1L; ++nsel;
        }
        if (nsel <= 0) continue;            // all of the block's working set, or nothing: the interpreter then knows its pointers
        for (const Seg& q : sg)
            if (q.lds >= 0) { segs.push_back((int)q.off); segs.push_back((int)(q.end - q.off)); segs.push_back(q.lds); segs.push_back(q.write ? 1 : 0); }
        meta[8 * b + 3] = nsel; meta[8 * b + 4] = (int)used;
        // rewrite the offsets that fall into a cached segment
        for (int r = first; r < first + count; ++r) {
            ext.clear();
            tape_extents(&ops[8 * (size_t)r], ext);
            for (const TapeExtent& e : ext) {
                size_t lo = 0, hi = sg.size();              // the segment that holds e.off: last one starting at or before it
                while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (sg[mid].off <= e.off) lo = mid; else hi = mid; }
                if (sg[lo].lds >= 0) ops[8 * (size_t)r + e.field] = T_LDS | (sg[lo].lds + (int)(e.off - sg[lo].off));
            }
        }
    }
    bool any = false;
    for (int b = 0; b < nb; ++b) any = any || meta[8 * b + 3] > 0;
    if (!any) return PYVB_OK;
    if (segs.empty()) segs.assign(4, 0);
    HIPCHK(hipMalloc((void**)&g->c_ops[id], ops.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&g->c_meta[id], meta.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&g->c_segs[id], segs.size() * sizeof(int)));
    HIPCHK(hipMemcpy(g->c_ops[id], ops.data(), ops.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->c_meta[id], meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->c_segs[id], segs.data(), segs.size() * sizeof(int), hipMemcpyHostToDevice));
    for (size_t l = 0; l + 1 < launches.size(); l += 2) {
        size_t need = 0;
        for (int b = launches[l]; b < launches[l] + launches[l + 1]; ++b) need = std::max(need, (size_t)meta[8 * b + 4]);
        g->c_lds[id].push_back((need + (size_t)TAPE_CHUNK * 4) * sizeof(double));
    }
    return PYVB_OK;
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
    g->host_ops.emplace_back(ops, ops + (size_t)nops * 8);
    g->c_ops.push_back(nullptr); g->c_meta.push_back(nullptr); g->c_segs.push_back(nullptr); g->c_lds.emplace_back();
    *tape_id = (int)g->tapes.size() - 1;
    return tape_cache_build(g, *tape_id, std::vector<int>{0, nops}, std::vector<int>{0, 1});     // one block, one launch
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
    return tape_cache_build(g, tape_id, std::vector<int>(blocks, blocks + 2 * nblocks), g->prog_launches[tape_id]);
}

int pyvb_graph_tape_run(pyvb_graph* g, int tape_id) {
    ARGCHK(g && tape_id >= 0 && tape_id < (int)g->tapes.size() && g->tapes[tape_id], "no such tape");
    HIPCHK(hipSetDevice(g->device));
    TapeArgs t; t.arena = g->arena; t.arena_n = g->arena_n; t.ops = g->tapes[tape_id]; t.nops = g->tape_len[tape_id]; t.status = g->status;
    if (g->c_ops[tape_id]) {
        // the window form: one workgroup per block, its working set in LDS
        static bool attr_set = false;
        if (!attr_set) {
            HIPCHK(hipFuncSetAttribute((const void*)k_tape_cached, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((TAPE_LDS_CAP + TAPE_CHUNK * 4) * sizeof(double))));
            attr_set = true;
        }
        t.ops = g->c_ops[tape_id];
        static const std::vector<int> single{0, 1};
        const std::vector<int>& L = g->prog_blocks[tape_id] ? g->prog_launches[tape_id] : single;
        for (size_t l = 0; l + 1 < L.size(); l += 2)
            hipLaunchKernelGGL(k_tape_cached, dim3(L[l + 1]), dim3(TAPE_THREADS), g->c_lds[tape_id][l / 2], g->stream, t,
                               g->c_meta[tape_id] + 8 * L[l], g->c_segs[tape_id]);
    } else if (g->prog_blocks[tape_id]) {
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
    tape_cache_free(g, tape_id);
    return PYVB_OK;
}

}  // extern "C"
