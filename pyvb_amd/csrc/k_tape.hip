// Generic per-node path: graphs the recognisers (pyvb_amd/_recognise.py) have no fused plan for run node by node.
// Every quantity of such a graph -- posterior parameters, constants, messages, temporaries -- lives in one device
// arena of doubles; what a node's update(), pass_up_m1_m2(), pass_down_*() or log_lower_bound() computes in the reference
// (gaussian.py:102-183, node.py:95-129,182-276, nodes_todo.py:33-62,125-157,183-204,224-234) is emitted by the host
// (pyvb_amd/generic.py) as a TAPE of small dense operations on arena offsets, and one workgroup interprets the tape:
// one launch per node update, or per whole Network.learn iteration, instead of one launch per numpy call.
// The matrices of this path are tiny (dimensions of single nodes); the interpreter favours generality over speed.
#include "common.h"
#include <algorithm>
#include <cstdio>
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
// the arithmetic.  When a tape is uploaded the host therefore works out which arena extents the records touch (the same
// per-opcode table that validates them) and cuts every block of records that one workgroup interprets into WINDOWS: the longest
// runs of consecutive records whose extents, merged into segments, fit the LDS budget.  The workgroup takes a block's windows
// in order: load the segments, run the records out of LDS -- their offsets are rewritten to LDS positions and carry T_LDS; the
// interpreter is instantiated with LDS pointers for them --, write the segments it has written back.  Blocks of one launch touch
// disjoint state (the program's contract), so the write-backs cannot collide; records with gather / scatter, whose addresses are
// data, and runs too short to pay for a window stay on the arena.  The records of a window are staged in LDS too, in chunks.
// Inside a window whose records are all node-sized the host also SCHEDULES them: bundles of TAPE_BUNDLE mutually independent
// records, a wavefront each (tape_bundle, k_tape_cached).
#define T_LDS 0x40000000        // in an offset field of a resolved record: position in the workgroup's LDS window, not in the arena
#define TAPE_CHUNK 512          // records staged at a time
#define TAPE_LDS_CAP 12288      // doubles of arena a block may keep in LDS (96 KB)
#define TAPE_MAX_SEGS 4096
#ifndef TAPE_BUNDLE
#define TAPE_BUNDLE 8            // records per bundle = wavefronts of a k_tape_cached workgroup
#endif
#define TAPE_CTHREADS (64 * TAPE_BUNDLE)

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

// NT: threads of the workgroup.  NT == 64 (one wavefront; the host picks it for launches whose records all have at most 64
// elements, WIN only) needs no s_barrier between records: the LDS operations of a wavefront execute in order, a fence keeps the
// compiler from moving them.  What a record of a node-sized graph costs is its chain of dependent steps -- fetch the record,
// form the addresses, fetch the operands, store -- so: the record is fetched as two 16-byte words one record ahead, and the
// row / column of an element comes from a float reciprocal with a fix-up instead of two integer divisions (about 80 instructions).
typedef int __attribute__((ext_vector_type(4))) tape_i4;
__device__ __forceinline__ void tape_divmod(int idx, int n, float rn, int& i, int& j) {
    i = (int)(((float)idx + 0.5f) * rn);
    j = idx - i * n;
    if (j < 0) { --i; j += n; } else if (j >= n) { ++i; j -= n; }
}

template <bool WIN, int NT>
__device__ static void tape_exec(const TapeArgs& t, const int* recs, int count, double* red, typename TapePtr<WIN>::type win, const int tid) {
    typedef typename TapePtr<WIN>::type P;
    double* A = t.arena;
    auto at = [&](int off) -> P { if constexpr (WIN) return win + (off & ~T_LDS); else return A + off; };
    auto sync = [&]() {
        if constexpr (NT == 64 && WIN) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else __syncthreads();
    };
    const tape_i4* r4 = reinterpret_cast<const tape_i4*>(recs);
    tape_i4 nlo = r4[0], nhi = r4[1];
    for (int pc = 0; pc < count; ++pc) {
        const tape_i4 lo = nlo, hi = nhi;
        if (pc + 1 < count) { nlo = r4[2 * pc + 2]; nhi = r4[2 * pc + 3]; }
        const int o[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int op = o[0], m = o[4], n = o[5], flags = o[7];
        const float rn = __builtin_amdgcn_rcpf((float)(n > 0 ? n : 1));
        P dst = at(o[1]);
        const P a = at(o[2]);
        const P b = o[3] < 0 ? at(0) : at(o[3]);
        switch (op) {
        case T_COPY2D:
            for (int idx = tid; idx < m * n; idx += NT) { int i, j; tape_divmod(idx, n, rn, i, j); dst[i * o[3] + j] = a[i * o[6] + j]; }
            break;
        case T_FILL:
            for (int idx = tid; idx < m * n; idx += NT) { int i, j; tape_divmod(idx, n, rn, i, j); dst[i * o[3] + j] = ((flags & 1) && i == j) ? 1.0 : 0.0; }
            break;
        case T_AXPBY: {
            const double al = *at(o[6]);
            if (o[3] < 0) { for (int idx = tid; idx < m * n; idx += NT) dst[idx] = al * a[idx]; }
            else { const double be = *at(flags); for (int idx = tid; idx < m * n; idx += NT) dst[idx] = al * a[idx] + be * b[idx]; }
            break; }
        case T_GEMM: {
            const int k = o[6];
            // strides instead of a choice per term, and eight terms fetched at a time: the sum over the children of a node
            // with thousands of them is one such product with a row of ones, a load latency per term otherwise
            const int as = (flags & 1) ? m : 1, bs = (flags & 2) ? 1 : n;
            auto part_sum = [&](int i, int j, int l0, int step) {      // terms l0, l0 + step, ... of element (i, j), in that order
                const P ap = a + ((flags & 1) ? i : i * k), bp = b + ((flags & 2) ? j * k : j);
                double s = 0.0;
                int l = l0;
                for (; l + 7 * step < k; l += 8 * step) {
                    double av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { av[u] = ap[(l + u * step) * as]; bv[u] = bp[(l + u * step) * bs]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += av[u] * bv[u];
                }
                for (; l < k; l += step) s += ap[l * as] * bp[l * bs];
                return s;
            };
            if (k >= 64 && 2 * m * n <= NT) {
                // a long sum into few elements (the messages of thousands of children, the terms of a bound): every element's
                // terms are split over PL neighbouring lanes of a wavefront -- lane p takes terms p, p + PL, ... -- and the PL
                // partial sums are added in a fixed tree (shuffles): deterministic, but not the order of a chain of additions
                int PL = 64;
                while (PL * m * n > NT) PL >>= 1;
                {
                    const int e = tid / PL, p0 = tid % PL;
                    const bool live = e < m * n;
                    int i = 0, j = 0;
                    if (live) tape_divmod(e, n, rn, i, j);
                    double s = live ? part_sum(i, j, p0, PL) : 0.0;
                    for (int sh = PL >> 1; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
                    if (live && p0 == 0) {
                        if (flags & 8) s = -s;
                        dst[e] = (flags & 4) ? dst[e] + s : s;
                    }
                }
                break;
            }
            for (int idx = tid; idx < m * n; idx += NT) {
                int i, j; tape_divmod(idx, n, rn, i, j);
                double s = part_sum(i, j, 0, 1);
                if (flags & 8) s = -s;
                dst[idx] = (flags & 4) ? dst[idx] + s : s;
            }
            break; }
        case T_SCALE: {
            const double s = *b;
            for (int idx = tid; idx < m * n; idx += NT) dst[idx] = (flags & 1) ? a[idx] / s : a[idx] * s;
            break; }
        case T_TRACE: case T_DOT: {
            double s = 0.0;
            if (op == T_TRACE) { for (int i = tid; i < m; i += NT) s += a[i * m + i]; }
            else { for (int idx = tid; idx < m * n; idx += NT) s += a[idx] * b[idx]; }
            // fixed tree: shuffles inside a wavefront, then the wavefronts in order (one thread adding up 256 values was
            // 7 us per record)
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
            if constexpr (NT == 64) {
                if (tid == 0) dst[0] = (flags & 4) ? dst[0] + s : s;
            } else {
                if ((tid & 63) == 0) red[tid >> 6] = s;
                __syncthreads();
                if (tid == 0) {
                    double tot = 0.0;
#pragma unroll
                    for (int w = 0; w < NT / 64; ++w) tot += red[w];
                    dst[0] = (flags & 4) ? dst[0] + tot : tot;
                }
            }
            break; }
        case T_DIAG:
            if (flags & 1) { const float rm = __builtin_amdgcn_rcpf((float)(m > 0 ? m : 1)); for (int idx = tid; idx < m * m; idx += NT) { int i, j; tape_divmod(idx, m, rm, i, j); dst[idx] = (i == j) ? a[i] : 0.0; } }
            else { for (int i = tid; i < m; i += NT) dst[i] = a[i * m + i]; }
            break;
        case T_CHOLINV: {
            // L (lower, row major) in scratch, column by column; then X = L^{-1} by forward substitution, one thread per
            // column of the identity; then dst = X^T X.  (cho_factor / cho_solve of gaussian.py:118-119.)
            P L = at(o[6]);
            P X = L + m * m;
            P out2 = at(o[3]);
            for (int idx = tid; idx < m * m; idx += NT) L[idx] = a[idx];
            sync();
            bool bad = false;
            for (int j = 0; j < m; ++j) {
                if (tid == 0) {
                    const double piv = L[j * m + j];
                    if (!(piv > 0.0)) { atomicOr(t.status, 1); L[j * m + j] = nan(""); }
                    else L[j * m + j] = sqrt(piv);
                }
                sync();
                const double d = L[j * m + j];
                for (int i = j + 1 + tid; i < m; i += NT) L[i * m + j] /= d;
                sync();
                const int rem = m - j - 1;                      // trailing update, lower triangle
                const float rr = __builtin_amdgcn_rcpf((float)(rem > 0 ? rem : 1));
                for (int idx = tid; idx < rem * rem; idx += NT) {
                    int i, c; tape_divmod(idx, rem, rr, i, c); i += j + 1; c += j + 1;
                    if (c <= i) L[i * m + c] -= L[i * m + j] * L[c * m + j];
                }
                sync();
            }
            (void)bad;
            if (tid == 0) {
                double s = 0.0;
                for (int j = 0; j < m; ++j) s += log(L[j * m + j]);
                out2[0] = 0.5 / s;          // gaussian.py:120: .5 / np.log(np.prod(np.diag(chol)))
                out2[1] = s;
            }
            for (int c = tid; c < m; c += NT) {       // X[:, c] = L^{-1} e_c
                for (int i = 0; i < m; ++i) {
                    double s = (i == c) ? 1.0 : 0.0;
                    for (int l = c; l < i; ++l) s -= L[i * m + l] * X[l * m + c];
                    X[i * m + c] = (i < c) ? 0.0 : s / L[i * m + i];
                }
            }
            sync();
            const float rm = __builtin_amdgcn_rcpf((float)(m > 0 ? m : 1));
            for (int idx = tid; idx < m * m; idx += NT) {
                int i, j; tape_divmod(idx, m, rm, i, j);
                double s = 0.0;
                for (int l = (i > j ? i : j); l < m; ++l) s += X[l * m + i] * X[l * m + j];
                dst[idx] = s;
            }
            break; }
        case T_UNARY:
            for (int idx = tid; idx < m * n; idx += NT) {
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
            for (int idx = tid; idx < m * n; idx += NT) {
                const long pos = (long)o[2] + (long)A[o[3] + idx / n] * o[6] + (long)A[flags + idx % n];
                if (pos < 0 || (size_t)pos >= t.arena_n) { atomicOr(t.status, 2); continue; }
                dst[idx] = A[pos];
            }
            break;
        case T_SCATTER:
            for (int idx = tid; idx < m * n; idx += NT) {
                const long pos = (long)o[1] + (long)A[o[3] + idx / n] * o[6] + (long)A[(flags & ~T_ACC) + idx % n];
                if (pos < 0 || (size_t)pos >= t.arena_n) { atomicOr(t.status, 2); continue; }
                A[pos] = (flags & T_ACC) ? A[pos] + a[idx] : a[idx];
            }
            break;
        case T_MUL:
            for (int idx = tid; idx < m * n; idx += NT) dst[idx] = a[idx] * b[idx];
            break;
        default: break;
        }
        sync();
    }
}

__global__ void __launch_bounds__(TAPE_THREADS) k_tape(TapeArgs t) {
    __shared__ double red[TAPE_THREADS];
    tape_exec<false, TAPE_THREADS>(t, t.ops, t.nops, red, nullptr, threadIdx.x);
}

// A PROGRAM over a tape: launches in order, each of them a set of record ranges ("blocks") that touch disjoint state and so
// run side by side, one workgroup per block (the updates of nodes none of which reads what another writes: the Z_n of a
// PCA-like graph, its X_n).  blocks: [first record, count] per workgroup of this launch.
__global__ void __launch_bounds__(TAPE_THREADS) k_tape_blocks(TapeArgs t, const int* blocks) {
    __shared__ double red[TAPE_THREADS];
    tape_exec<false, TAPE_THREADS>(t, t.ops + 8 * (size_t)blocks[2 * blockIdx.x], blocks[2 * blockIdx.x + 1], red, nullptr, threadIdx.x);
}

// The same with the working set in LDS.  A block is cut into WINDOWS, consecutive runs of records whose extents fit the LDS budget
// together (a chain of node updates does not fit as a whole: the LDS example at T = 200 touches 20 000 doubles): the workgroup
// takes them in order -- load the window's segments, run its records out of LDS, write the written segments back -- so what one
// window leaves for the next travels through the arena.  bmeta[block][2] = {first window, number of windows};
// meta[window][8] = {first record, count, first segment, number of segments (0: the records address the arena), window doubles};
// segs[seg][4] = {arena offset, length, window offset, written}; t.ops holds the RESOLVED records (cached offsets rewritten and
// tagged T_LDS).  Dynamic LDS: the window, then TAPE_CHUNK staged records.
// meta[window][5] = 1: a BUNDLED window.  Its records all have at most 64 elements and have been scheduled by the host into
// bundles of four that do not touch each other's extents (slots without a record hold T_NOP): wavefront w of the workgroup
// interprets record 4 b + w of bundle b on its own (tape_exec<true, 64>: no barrier inside), then the workgroup meets.  A
// record of such a graph costs 300-700 ns whatever it does (profiles/tape_record_cost.py: a chain of LDS round trips and taken
// branches, about 70 instructions); independent records -- the messages of different children, the next node's messages
// while this node's covariance is inverted -- now overlap on the four SIMDs.
// BW: wavefronts of the workgroup = records of a bundle.  A launch of a few long blocks takes 8 (the LDS example's chain of 400 node
// updates), a launch of many short ones 4: such blocks rarely have eight independent records, and workgroups half as wide fit a CU twice
// as often (2000 blocks of a node update each: 170 -> about 120 us).
template <int BW>
__global__ void __launch_bounds__(64 * BW) k_tape_cached(TapeArgs t, const int* bmeta, const int* meta, const int* segs) {
    extern __shared__ double win[];
    constexpr int NT = 64 * BW;
    __shared__ double red[NT];
    const int tid = threadIdx.x;
    const int w0 = bmeta[2 * blockIdx.x], nwin = bmeta[2 * blockIdx.x + 1];
    for (int w = w0; w < w0 + nwin; ++w) {
        const int* me = meta + 8 * w;
        const int first = me[0], count = me[1], s0 = me[2], ns = me[3], wd = me[4], bundled = me[5];
        // long segments by all threads together, short ones (most: a node's mean, a scalar) one per thread
        for (int s = 0; s < ns; ++s) {
            const int* sg = segs + 4 * (s0 + s);
            if (sg[1] < 64) continue;
            const double* src = t.arena + sg[0];
            double* dst = win + sg[2];
            for (int idx = tid; idx < sg[1]; idx += NT) dst[idx] = src[idx];
        }
        for (int s = tid; s < ns; s += NT) {
            const int* sg = segs + 4 * (s0 + s);
            if (sg[1] >= 64) continue;
            const double* src = t.arena + sg[0];
            double* dst = win + sg[2];
            for (int idx = 0; idx < sg[1]; ++idx) dst[idx] = src[idx];
        }
        int* staged = reinterpret_cast<int*>(win + ((wd + 1) & ~1));         // 16-byte aligned: records are fetched as vectors
        for (int c0 = 0; c0 < count; c0 += TAPE_CHUNK) {
            const int nc = count - c0 < TAPE_CHUNK ? count - c0 : TAPE_CHUNK;
            __syncthreads();
            const int* src = t.ops + 8 * (size_t)(first + c0);
            for (int idx = tid; idx < 8 * nc; idx += NT) staged[idx] = src[idx];
            __syncthreads();
            if (bundled) {
                for (int b0 = 0; b0 < nc; b0 += BW) {          // TAPE_CHUNK is a multiple of BW: a bundle never straddles two chunks
                    tape_exec<true, 64>(t, staged + 8 * (b0 + (tid >> 6)), 1, red, (lds_double*)win, tid & 63);
                    __syncthreads();
                }
            } else if (ns > 0) tape_exec<true, NT>(t, staged, nc, red, (lds_double*)win, tid);
            else tape_exec<false, NT>(t, staged, nc, red, nullptr, tid);
        }
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const int* sg = segs + 4 * (s0 + s);
            if (!sg[3] || sg[1] < 64) continue;
            double* dst = t.arena + sg[0];
            const double* src = win + sg[2];
            for (int idx = tid; idx < sg[1]; idx += NT) dst[idx] = src[idx];
        }
        for (int s = tid; s < ns; s += NT) {
            const int* sg = segs + 4 * (s0 + s);
            if (!sg[3] || sg[1] >= 64) continue;
            double* dst = t.arena + sg[0];
            const double* src = win + sg[2];
            for (int idx = 0; idx < sg[1]; ++idx) dst[idx] = src[idx];
        }
        if (w + 1 < w0 + nwin) { __threadfence(); __syncthreads(); }        // the next window reads what this one has written back
    }
}

struct pyvb_graph {
    int device;
    bool lds_attr_set;                               // the dynamic-LDS limit of k_tape_cached is raised on this graph's device
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
    std::vector<std::vector<int>> c_bw;              // per tape and launch: bundle width (wavefronts per workgroup)
    std::vector<int> c_nblocks;                      // per tape: blocks in the window form (c_meta = block table [nb][2], then the window table)
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
    g->c_bw[id].clear();
}

// Build the window form of tape `id` for the given blocks ([first, count] each) and launches ([first block, number]).
// Nothing is cached when the arena does not leave bit 30 of an offset free.
struct TapeSeg { long off, end; bool write; int lds; };
// the extents of records [first, first + count) merged into segments (overlapping or adjacent only: a gap may be another
// block's state); false if a record's addresses are data (gather / scatter)
static bool tape_segments(const std::vector<int>& ops, int first, int count, std::vector<TapeSeg>& sg, long& total) {
    std::vector<TapeExtent> ext;
    for (int r = first; r < first + count; ++r)
        if (!tape_extents(&ops[8 * (size_t)r], ext)) return false;
    std::sort(ext.begin(), ext.end(), [](const TapeExtent& x, const TapeExtent& y) { return x.off < y.off; });
    sg.clear(); total = 0;
    for (const TapeExtent& e : ext) {
        const long end = e.off + (long)e.len;
        if (!sg.empty() && e.off <= sg.back().end) {
            if (end > sg.back().end) sg.back().end = end;
            sg.back().write = sg.back().write || e.write;
        } else sg.push_back(TapeSeg{e.off, end, e.write, -1});
    }
    for (const TapeSeg& q : sg) total += ((q.end - q.off) + 1) & ~1L;
    return true;
}

// Records [first, first + count) of `ops` (resolved: window offsets) scheduled into bundles of four mutually independent records,
// appended to `out` (T_NOP in the free slots).  A record depends on every earlier one that touches one of its extents unless both
// only read it; it goes into the first bundle after all of its dependencies that has a free slot (list scheduling: any order that
// respects the dependencies computes what the tape computes).  Extents are taken from the unresolved records `raw`.
#define TAPE_BUNDLE_MAX 2048        // records scheduled together (the dependency search is quadratic)
static void tape_bundle(const std::vector<int>& raw, const std::vector<int>& ops, int first, int count, std::vector<int>& out, int BW) {
    std::vector<std::vector<TapeExtent>> ext((size_t)count);
    for (int r = 0; r < count; ++r) tape_extents(&raw[8 * (size_t)(first + r)], ext[r]);
    std::vector<int> bundle((size_t)count, 0), fill;
    for (int r = 0; r < count; ++r) {
        int earliest = 0;
        for (int q = r - 1; q >= 0; --q) {
            if (bundle[q] < earliest) continue;             // cannot raise the bound
            bool hazard = false;
            for (const TapeExtent& x : ext[r]) {
                for (const TapeExtent& y : ext[q])
                    if ((x.write || y.write) && x.off < y.off + (long)y.len && y.off < x.off + (long)x.len) { hazard = true; break; }
                if (hazard) break;
            }
            if (hazard) earliest = bundle[q] + 1;
        }
        int b = earliest;
        while (b < (int)fill.size() && fill[b] >= BW) ++b;
        if (b >= (int)fill.size()) fill.resize((size_t)b + 1, 0);
        bundle[r] = b; ++fill[b];
    }
    const size_t base = out.size();
    out.resize(base + fill.size() * 8 * BW, 0);                 // T_NOP == 0
    std::vector<int> slot(fill.size(), 0);
    for (int r = 0; r < count; ++r) {
        int* dst = &out[base + ((size_t)bundle[r] * BW + slot[bundle[r]]++) * 8];
        for (int k = 0; k < 8; ++k) dst[k] = ops[8 * (size_t)(first + r) + k];
    }
}

static int tape_cache_build(pyvb_graph* g, int id, const std::vector<int>& blocks, const std::vector<int>& launches) {
    tape_cache_free(g, id);
    if (g->arena_n >= (size_t)T_LDS) return PYVB_OK;
    const std::vector<int>& raw = g->host_ops[id];
    std::vector<int> ops = raw, cops;                       // ops: resolved in place; cops: what the device gets, window by window
    const int nb = (int)blocks.size() / 2;
    std::vector<int> bmeta((size_t)nb * 2, 0), meta, segs;
    std::vector<size_t> block_lds((size_t)nb, 0);
    std::vector<TapeSeg> sg, best;
    std::vector<TapeExtent> ext;
    bool any = false;
    std::vector<int> bw((size_t)nb, TAPE_BUNDLE), lbw;          // bundle width per block: that of its launch
    for (size_t l = 0; l + 1 < launches.size(); l += 2) {
        const int w = launches[l + 1] >= 512 ? 4 : TAPE_BUNDLE;
        lbw.push_back(w);
        for (int b = launches[l]; b < launches[l] + launches[l + 1]; ++b) bw[b] = w;
    }
    for (int b = 0; b < nb; ++b) {
        const int first = blocks[2 * b], count = blocks[2 * b + 1];
        bmeta[2 * b] = (int)meta.size() / 8;
        int r = first;
        while (r < first + count) {
            // the longest run of records from r whose segments fit: grown geometrically, then bisected
            int len = 0; long used = 0;
            {
                int lo = 0, hi = 1;             // lo fits (0 = nothing tried), hi is the next candidate
                long tot = 0;
                const int left = first + count - r;
                while (true) {
                    const int c = hi < left ? hi : left;
                    if (tape_segments(raw, r, c, sg, tot) && tot <= TAPE_LDS_CAP && (int)sg.size() <= TAPE_MAX_SEGS) {
                        lo = c; best = sg; used = tot;
                        if (c == left) break;
                        hi = c * 2;
                    } else { hi = c; break; }
                }
                while (hi - lo > 1 && lo < left) {          // lo fits, hi does not
                    const int mid = (lo + hi) / 2;
                    if (tape_segments(raw, r, mid, sg, tot) && tot <= TAPE_LDS_CAP && (int)sg.size() <= TAPE_MAX_SEGS) { lo = mid; best = sg; used = tot; }
                    else hi = mid;
                }
                len = lo;
            }
            if (len < 3) {
                // a record whose addresses are data (gather / scatter), one that does not fit by itself, or a run too short to pay
                // for a load and a write-back: on the arena.  Runs of such records are kept together.
                const int c = len > 0 ? len : 1;
                const size_t last = meta.size() - 8;
                const int at = (int)(cops.size() / 8);
                cops.insert(cops.end(), raw.begin() + 8 * (size_t)r, raw.begin() + 8 * (size_t)(r + c));
                if ((int)meta.size() / 8 > bmeta[2 * b] && meta[last + 3] == 0 && meta[last] + meta[last + 1] == at) meta[last + 1] += c;
                else {
                    const size_t mw = meta.size();
                    meta.resize(mw + 8, 0);
                    meta[mw] = at; meta[mw + 1] = c; meta[mw + 2] = (int)segs.size() / 4;
                }
                r += c;
                continue;
            }
            const size_t mw = meta.size();
            meta.resize(mw + 8, 0);
            long pos = 0;
            for (TapeSeg& q : best) { q.lds = (int)pos; pos += ((q.end - q.off) + 1) & ~1L; }
            // rewrite the offsets: every extent of these records lies in one of the window's segments
            long widest = 0;
            for (int rr = r; rr < r + len; ++rr) {
                ext.clear();
                tape_extents(&raw[8 * (size_t)rr], ext);
                for (const TapeExtent& e : ext) {
                    size_t lo = 0, hi = best.size();        // the segment that holds e.off: last one starting at or before it
                    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (best[mid].off <= e.off) lo = mid; else hi = mid; }
                    ops[8 * (size_t)rr + e.field] = T_LDS | (best[lo].lds + (int)(e.off - best[lo].off));
                }
                const int* o = &raw[8 * (size_t)rr];
                const long mm = (long)o[4] * o[4], mn = (long)o[4] * o[5];
                widest = std::max(widest, (o[0] == T_CHOLINV || o[0] == T_DIAG || o[0] == T_TRACE) ? mm : mn);
            }
            const int at = (int)(cops.size() / 8);
            const bool bundled = widest <= 64;
            if (bundled) {                  // a long window is scheduled piece by piece (the search is quadratic in the piece)
                for (int p0 = 0; p0 < len; p0 += TAPE_BUNDLE_MAX) tape_bundle(raw, ops, r + p0, std::min(TAPE_BUNDLE_MAX, len - p0), cops, bw[b]);
            }
            else cops.insert(cops.end(), ops.begin() + 8 * (size_t)r, ops.begin() + 8 * (size_t)(r + len));
            meta[mw] = at; meta[mw + 1] = (int)(cops.size() / 8) - at; meta[mw + 2] = (int)segs.size() / 4; meta[mw + 3] = (int)best.size();
            meta[mw + 4] = (int)used; meta[mw + 5] = bundled ? 1 : 0;
            for (const TapeSeg& q : best) { segs.push_back((int)q.off); segs.push_back((int)(q.end - q.off)); segs.push_back(q.lds); segs.push_back(q.write ? 1 : 0); }
            block_lds[b] = std::max(block_lds[b], (size_t)used);
            any = true;
            r += len;
        }
        bmeta[2 * b + 1] = (int)meta.size() / 8 - bmeta[2 * b];
    }
    if (getenv("PYVB_TAPE_STATS")) {
        int nwin = (int)meta.size() / 8, nb_ = 0, nlds = 0; long recs = 0, slots = 0, wdoubles = 0;
        for (int w = 0; w < nwin; ++w) { if (meta[8 * w + 5]) { ++nb_; slots += meta[8 * w + 1]; } if (meta[8 * w + 3]) { ++nlds; wdoubles += meta[8 * w + 4]; } recs += meta[8 * w + 1]; }
        fprintf(stderr, "tape %d: %zu records in %d blocks -> %d windows (%d in LDS, %ld doubles; %d bundled: %ld slots = %ld bundles), %ld device records\n",
                id, raw.size() / 8, nb, nwin, nlds, wdoubles, nb_, slots, slots / (lbw.empty() ? TAPE_BUNDLE : lbw[0]), recs);
    }
    if (!any) return PYVB_OK;
    if (segs.empty()) segs.assign(4, 0);
    HIPCHK(hipMalloc((void**)&g->c_ops[id], cops.size() * sizeof(int)));
    HIPCHK(hipMalloc((void**)&g->c_meta[id], (bmeta.size() + meta.size()) * sizeof(int)));
    HIPCHK(hipMalloc((void**)&g->c_segs[id], segs.size() * sizeof(int)));
    HIPCHK(hipMemcpy(g->c_ops[id], cops.data(), cops.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->c_meta[id], bmeta.data(), bmeta.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->c_meta[id] + bmeta.size(), meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->c_segs[id], segs.data(), segs.size() * sizeof(int), hipMemcpyHostToDevice));
    g->c_nblocks[id] = nb;
    for (size_t l = 0; l + 1 < launches.size(); l += 2) {
        size_t need = 0;
        for (int b = launches[l]; b < launches[l] + launches[l + 1]; ++b) need = std::max(need, block_lds[b]);
        g->c_lds[id].push_back((need + 2 + (size_t)TAPE_CHUNK * 4) * sizeof(double));
        g->c_bw[id].push_back(lbw[l / 2]);
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
    g->c_ops.push_back(nullptr); g->c_meta.push_back(nullptr); g->c_segs.push_back(nullptr); g->c_lds.emplace_back(); g->c_bw.emplace_back(); g->c_nblocks.push_back(0);
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
        if (!g->lds_attr_set) {            // per graph (= per device the graph lives on; a handle is used by one host thread)
            const int cap = (int)((TAPE_LDS_CAP + 2 + TAPE_CHUNK * 4) * sizeof(double));
            HIPCHK(hipFuncSetAttribute((const void*)k_tape_cached<TAPE_BUNDLE>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            HIPCHK(hipFuncSetAttribute((const void*)k_tape_cached<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            g->lds_attr_set = true;
        }
        t.ops = g->c_ops[tape_id];
        static const std::vector<int> single{0, 1};
        const std::vector<int>& L = g->prog_blocks[tape_id] ? g->prog_launches[tape_id] : single;
        for (size_t l = 0; l + 1 < L.size(); l += 2) {
            const int* bm = g->c_meta[tape_id] + 2 * L[l];
            const int* wm = g->c_meta[tape_id] + 2 * g->c_nblocks[tape_id];
            if (g->c_bw[tape_id][l / 2] == 4)
                hipLaunchKernelGGL(k_tape_cached<4>, dim3(L[l + 1]), dim3(256), g->c_lds[tape_id][l / 2], g->stream, t, bm, wm, g->c_segs[tape_id]);
            else
                hipLaunchKernelGGL(k_tape_cached<TAPE_BUNDLE>, dim3(L[l + 1]), dim3(TAPE_CTHREADS), g->c_lds[tape_id][l / 2], g->stream, t, bm, wm, g->c_segs[tape_id]);
        }
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
