// The second shape class of the fused LDS path: 64 < max(D, K) <= 128 (both padded to 128 = eight 16-row tiles).
// The reference has no size limit (gaussian.py:43-46); the kernels of k_sweep.hip / k_prep.hip / k_stats.hip / k_cols.hip keep one
// 64 x 64 matrix per wavefront in registers and stop at 64.  Here a replicate is a WORKGROUP of four wavefronts:
//
//   k_gy_big      G y_t of every interior node as a batched product of its own (16 time steps = the 16 MFMA columns), into the
//                 c_t buffer: the sequential kernel is left with two products per step and no operand in LDS but the state.
//   k_sweep_big   wavefront w owns the row tiles 2w, 2w+1 (32 rows) of the two recurrence matrices as MFMA A operands in
//                 registers (2 x 64 doubles per lane each); the state of the 16 time segments (128 x 16, the MFMA B operand)
//                 lives in two LDS buffers: every step each wavefront writes the two accumulator tiles it has just formed --
//                 accumulator layout = B-operand layout, as in k_sweep.hip -- into one and reads all eight from the other.
//                 Same segmentation of the time axis, same warm-up from k_prep's contraction bound, same boundary nodes, same
//                 c_t cache between the forward sweep and the backward one behind it (no fused Sxx).
//   k_prep_big    one work matrix in LDS (128 x 130 doubles), intermediates in a per-replicate global scratch; products as chains
//                 of dependent MFMAs on staged operands; the three precisions inverted one after the other by the whole
//                 workgroup (8 x 8 tile per thread).
//   k_stats_big   Sxx, Sx1x, Syx in one pass per (replicate, chunk): a wavefront holds row tile w of all three, its column
//                 tiles in rotated order so that the symmetric Sxx is formed once per pair of tiles.
//   k_cols_big    a row of the matrix = two lanes with half of it each in registers; the Gauss-Seidel pass over the columns runs
//                 column by column without a barrier (the rows decouple under diagonal noise and diagonal column priors), G in LDS.
// Reference methods as in the small kernels: Gaussian.update gaussian.py:102-123, Multiplication.pass_up_m1_m2 node.py:182-232,
// hstack.pass_up_m1_m2 nodes_todo.py:43-62, Gamma / DiagonalGamma.update nodes_todo.py:130-138, :187-190.
// Diagonal-Gamma and Gamma noise, known entries of A / C, outputs with NaN (k_missing.hip, two entries per lane) and single X_t
// updates are served; Wishart noise too (k_wishart_big.hip: the dense expectations reach k_prep_big and the boundary nodes of the
// sweeps as the products <Q><A>, <R><C>), though not together with known entries or outputs that hold NaN.
#include "params.h"
#include "gj.h"
#include <cstdio>
#include <cstdlib>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define BDP 128     // padded dimension
#define BDT 8       // 16-row tiles
#define BDS 32      // k-steps of 4
#ifdef BLD_OVERRIDE
#define BLD BLD_OVERRIDE
#else
#define BLD 130     // row stride of the LDS work matrix (A-operand reads conflict free)
#endif
#ifndef CHAIN_RUN
#define CHAIN_RUN 32    // dependent MFMAs on one accumulator before the next accumulator takes its turn (k_gy_big, k_sweep_big);
                        // runs of 16 or 8 measured the same as whole chains (lds_d128: 49.9 / 49.9 / 49.7 ms)
#endif

// ======================================================================================================================
// sweep
// ======================================================================================================================
struct BigSweepArgs {
    const double* Xold; double* Xnew; const double* Y; const double* gains; const int* warm;
    const double *A_mean, *C_mean;
    const double *QA, *RC;      // Wishart noise: <Q><A> [D][D], <R><C> [K][D] per replicate (k_wishart_big.hip), else null
    double* U;          // [N][T][128]: c_t = R mu_{t-1} + G y_t of the interior nodes (MODE 1 writes, MODE 2 reads), accumulator order
    double* Uc;         // where a forward sweep stores c_t: U itself, or, when the time axis is split over several workgroups
                        // (W > 1), a buffer of its own -- a part's warm-up reads G y_t rows of the part before it, which that part's
                        // workgroup overwrites with c_t on its own schedule (within a workgroup a barrier orders the two)
    double* trash;      // [N][W][512]
    int N, T, D, K, dir, W;
    Layout L;
};

// Xs[0].update() / Xs[T-1].update() by the whole workgroup, thread = row (k_sweep.hip: boundary_update).
// nb(j): entry j of the one neighbour's mean.  vs: 128 doubles of LDS.  Every thread of the workgroup must call it.
template <class NB>
__device__ __forceinline__ double big_boundary(bool first, const double* g, const Layout& L, const double* Am, const double* Cm,
                                               int D, int K, int tid, NB nb, const double* y, double* vs,
                                               const double* QA = nullptr, const double* RC = nullptr) {
    const double* qb = g + L.oqr;
    const double* rb = qb + BDP;
    double v = 0.0;
    if (QA) {                   // Wishart noise: dense expectations, QA = <Q><A>, RC = <R><C> (k_sweep.hip: boundary_update)
        if (tid < D) {
            if (first) {
                v = g[L.ow0 + tid];
                for (int i = 0; i < D; ++i) v += QA[(size_t)i * D + tid] * nb(i);
            } else {
                for (int j = 0; j < D; ++j) v += QA[(size_t)tid * D + j] * nb(j);
            }
            for (int k = 0; k < K; ++k) v += RC[(size_t)k * D + tid] * y[k];
        }
    } else if (tid < D) {
        if (first) {
            v = g[L.ow0 + tid];
            for (int i = 0; i < D; ++i) v += Am[(size_t)i * D + tid] * (qb[i] * nb(i));
        } else {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += Am[(size_t)tid * D + j] * nb(j);
            v = qb[tid] * s;
        }
        for (int k = 0; k < K; ++k) v += Cm[(size_t)k * D + tid] * (rb[k] * y[k]);
    }
    if (tid < BDP) vs[tid] = v;
    __syncthreads();
    const double* S = g + (first ? L.oS0 : L.oS2);
    double s = 0.0;
    if (tid < D)
        for (int j = 0; j < D; ++j) s += S[(size_t)j * BDP + tid] * vs[j];      // Sigma is symmetric
    __syncthreads();
    return s;
}

// G y_t of every interior node as a batched product of its own, into U (accumulator order, as the sweep reads it): 16 time
// steps are the 16 MFMA columns, wavefront w forms the row tiles 2w, 2w+1 with its rows of G (A operands, permuted k order of
// k_prep) in registers; y rows are fetched a block ahead.  Taking this product out of the sweep leaves the sequential kernel
// two products per step and no operand in LDS except the shared state.
struct BigGyArgs {
    const double* Y; const double* gains; double* U; double* trash;
    int N, T, K, nblk;          // nblk: blocks of 16 time steps per workgroup
    Layout L;
};

// (Eight wavefronts with one row tile each, two per SIMD, were slower: 11.0 against 7.3 ms -- every wavefront fetches all of y.)
template <bool YVEC>     // YVEC: K even, a lane's two k of a step are one 16-byte load
__global__ void __launch_bounds__(256) k_gy_big(BigGyArgs a) {
    constexpr int NT = 2;       // row tiles per wavefront
    const int n = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int T = a.T, K = a.K;
    const double* g = a.gains + (size_t)n * a.L.gains_total + a.L.oGp;
    const double* Yn = a.Y + (size_t)n * T * K;
    double* Un = a.U + (size_t)n * T * BDP;
    double* const trash = a.trash + (size_t)n * 512 + 256;
    double gr[NT][BDS];
#pragma unroll
    for (int mm = 0; mm < NT; ++mm)
#pragma unroll
        for (int s = 0; s < BDS; ++s) gr[mm][s] = g[((size_t)(NT * w + mm) * BDS + s) * 64 + lane];
    auto load_y = [&](d2 (&yv)[BDS / 2], int t) {
        const double* yp = Yn + (size_t)(t <= T - 2 ? t : 1) * K;
#pragma unroll
        for (int i = 0; i < BDS / 2; ++i) {
            const int d0 = 8 * i + 2 * q;
            if constexpr (YVEC) yv[i] = *reinterpret_cast<const d2*>(yp + (d0 + 1 < K ? d0 : K - 2));      // padded k meet zero gains
            else { yv[i][0] = yp[d0 < K ? d0 : K - 1]; yv[i][1] = yp[d0 + 1 < K ? d0 + 1 : K - 1]; }
        }
    };
    auto block = [&](const d2 (&yv)[BDS / 2], int t) {
        d4 acc[NT];
#pragma unroll
        for (int mm = 0; mm < NT; ++mm) acc[mm] = d4{0.0, 0.0, 0.0, 0.0};
        // runs of CHAIN_RUN dependent MFMAs, the accumulators in turn (profiles/r01/microbench_f64.txt: at one wavefront per SIMD
        // runs of 16 over several accumulators sustain more than one chain to its end, single MFMAs in turn far less)
#pragma unroll
        for (int s0 = 0; s0 < BDS; s0 += CHAIN_RUN)
#pragma unroll
            for (int mm = 0; mm < NT; ++mm)
#pragma unroll
                for (int s = s0; s < s0 + CHAIN_RUN; ++s) acc[mm] = MFMA(gr[mm][s], yv[s >> 1][s & 1], acc[mm]);
        double* ur = (t <= T - 2) ? Un + (size_t)t * BDP : trash;
#pragma unroll
        for (int mm = 0; mm < NT; ++mm) *reinterpret_cast<d4*>(ur + ((NT * w + mm) * 4 + q) * 4) = acc[mm];
    };
    // two operand sets: the rows of a block are in flight while the block before it is multiplied (nblk is even)
    d2 ya[BDS / 2], yb[BDS / 2];
    const int t0 = 1 + blockIdx.x * a.nblk * 16 + c;
    load_y(ya, t0);
    for (int b = 0; b < a.nblk; b += 2) {
        const int t = t0 + b * 16;
        if (t - c > T - 2) break;
        load_y(yb, t + 16);
        __builtin_amdgcn_sched_barrier(0);
        block(ya, t);
        __builtin_amdgcn_sched_barrier(0);
        load_y(ya, t + 32);
        __builtin_amdgcn_sched_barrier(0);
        block(yb, t + 16);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Round 4: the same product with EIGHT wavefronts, one row tile each, two to a SIMD -- and the block of y shared through LDS.
// At one wavefront per SIMD a chain of dependent MFMAs sustains 70 % of the pipe (profiles/r01/microbench_f64.txt; k_gy_big: 69 %
// busy by the counters), with two it sustains 92 %; what stopped round 3's eight-wavefront form (11.0 against 7.3 ms) was that every
// wavefront fetched all of y from memory.  Here the workgroup's 512 threads fetch a block of 16 time steps once (a block ahead, into
// registers, then into the other of two LDS buffers: one barrier per block) and every wavefront reads its B operands from there.
// Measured at N = 1024, T = 10^4, D = K = 128: lds_d128 49.24 ms against 49.41 with k_gy_big on the same box -- the product moves
// 21 GB (Y in, G y_t out) in 7.5 ms, and reads and writes together do not pass 5 TB/s on this part (profiles/microbench/hbm_read.hip:
// a copy makes 2.4-2.8 TB/s each way): it is the traffic of the c_t buffer, not the matrix pipe, that the kernel waits for.  Kept
// behind PYVB_GY_BIG=8 with the tests of the class run through it once; k_gy_big stays the kernel in use.
#define GY_LDY 130      // row stride of a y block in LDS: the 16 lanes of a 16-byte operand read fall into 64 different banks
template <bool Y4>      // Y4: K a multiple of 4 -- a thread's four entries of a block are one 32-byte load
__global__ void __launch_bounds__(512) k_gy_big8(BigGyArgs a) {
    __shared__ double yl[2][16 * GY_LDY];
    const int n = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int T = a.T, K = a.K;
    const double* g = a.gains + (size_t)n * a.L.gains_total + a.L.oGp;
    const double* Yn = a.Y + (size_t)n * T * K;
    double* Un = a.U + (size_t)n * T * BDP;
    double* const trash = a.trash + (size_t)n * 512 + 256;
    double gr[BDS];
#pragma unroll
    for (int s = 0; s < BDS; ++s) gr[s] = g[((size_t)w * BDS + s) * 64 + lane];
    // staging: thread -> time step srow of the block, entries scol .. scol + 3
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const int tb0 = 1 + blockIdx.x * a.nblk * 16;
    d4 stage;
    auto fetch = [&](int tb) {
        const int t = tb + srow;
        const double* yp = Yn + (size_t)(t <= T - 2 ? t : 1) * K;
        if constexpr (Y4) {
            const d4 v = *reinterpret_cast<const d4*>(yp + (scol < K ? scol : 0));
            stage = scol < K ? v : d4{0.0, 0.0, 0.0, 0.0};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double v = yp[scol + e < K ? scol + e : K - 1]; stage[e] = scol + e < K ? v : 0.0; }
        }
    };
    auto put = [&](int buf) {
        double* d = yl[buf] + srow * GY_LDY + scol;
        *reinterpret_cast<d2*>(d) = d2{stage[0], stage[1]};
        *reinterpret_cast<d2*>(d + 2) = d2{stage[2], stage[3]};
    };
    fetch(tb0); put(0);
    __syncthreads();
    for (int b = 0; b < a.nblk; ++b) {
        const int tb = tb0 + b * 16;
        if (tb > T - 2) break;                                  // block-uniform
        const bool more = b + 1 < a.nblk && tb + 16 <= T - 2;
        if (more) fetch(tb + 16);
        const double* yb = yl[b & 1] + c * GY_LDY + 2 * q;      // B operand of step s = 2 i + j: y_t[8 i + 2 q + j], t = tb + c
        d2 yv[BDS / 2];
#pragma unroll
        for (int i = 0; i < BDS / 2; ++i) yv[i] = *reinterpret_cast<const d2*>(yb + 8 * i);
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < BDS; ++s) acc = MFMA(gr[s], yv[s >> 1][s & 1], acc);
        const int t = tb + c;
        double* ur = (t <= T - 2) ? Un + (size_t)t * BDP : trash;
        *reinterpret_cast<d4*>(ur + (w * 4 + q) * 4) = acc;
        if (more) put((b + 1) & 1);
        __syncthreads();
    }
}

// The sweep proper.  The parameters are frozen between the two sweeps of an iteration, and the backward update
//   mu_t <- B mu_{t+1}(new) + F mu_{t-1}(forward result) + G y_t
// contains c_t = F mu_{t-1} + G y_t, which the forward sweep has just formed (k_sweep.hip).
//   MODE 3  U holds G y_t (k_gy_big ran just before): R mu (new neighbour) and I mu (old neighbour) are added; a forward
//           sweep stores c_t back into U, in place.  In the warm-up a column reads rows of the segment before it: it does so at
//           loop indices j < 0, the owner overwrites them at j >= 0, and a barrier separates the steps.
//   MODE 2  the backward sweep directly behind a forward one: reads c_t and runs ONE product per step.
// The state of the 16 segments lives in LDS only, in two buffers (a step reads one and writes the other: one barrier per step).
// NTW: row tiles per wavefront -- 2: four wavefronts per replicate.  (1: eight, two per SIMD, one register set for the old neighbour --
// measured at N = 1024, D = K = 128: forward sweep 19.4 ms against 16.0; every wavefront reads the whole state for half as many
// products.  Only NTW = 2 is instantiated.)
template <int MODE, int NTW>
__global__ void __launch_bounds__(512 / NTW) k_sweep_big(BigSweepArgs a) {
    extern __shared__ double lds[];
    double* xb0 = lds;                              // 2 x [BDS][64]: the state of the 16 segments, B-operand order
    double* xs = lds + 2 * BDS * 64;                // [128] boundary state exchange
    double* vs = xs + BDP;                          // [128] boundary scratch
    const int n = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
#ifdef BIG_CLOCK        // (profiles/build_variant.sh k_big clock "-DBIG_CLOCK": the chip's clock while this kernel runs, from its two time counters)
    const unsigned long long ck_c0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
    struct ClockReport { unsigned long long c0, r0; int on, mode; __device__ ~ClockReport() {
        if (on) { const unsigned long long dc = __builtin_amdgcn_s_memtime() - c0, dr = __builtin_amdgcn_s_memrealtime() - r0;
                  printf("k_sweep_big<%d>: %.1f us, %.3f GHz\n", mode, dr / 100.0, dc / (dr * 10.0)); } } }
        ck_report{ck_c0, ck_r0, blockIdx.x == 300 && blockIdx.y == 0 && threadIdx.x == 0, MODE};
#endif
    const int part = blockIdx.y;            // the time axis is dealt out to a.W workgroups per replicate when there are few replicates (k_sweep.hip: SPLIT)
    const int T = a.T, D = a.D, K = a.K;
    const bool fwd = (a.dir == 0);
    const int sgn = fwd ? 1 : -1;
    const Layout& L = a.L;
    const double* g = a.gains + (size_t)n * L.gains_total;
    const double* Xo = a.Xold + (size_t)n * T * BDP;
    double* Xn = a.Xnew + (size_t)n * T * BDP;
    const double* Yn = a.Y + (size_t)n * T * K;

    // ---- this wavefront's rows of the recurrence matrices
    double rn[NTW][BDS], ip[NTW][BDS];
    {
        const double* Rn = g + (fwd ? L.oFn : L.oBn);
        const double* Ip = g + (fwd ? L.oBn : L.oFn);
#pragma unroll
        for (int mm = 0; mm < NTW; ++mm)
#pragma unroll
            for (int s = 0; s < BDS; ++s) {
                const size_t o = ((size_t)(NTW * w + mm) * BDS + s) * 64 + lane;
                rn[mm][s] = Rn[o];
                ip[mm][s] = MODE == 2 ? 0.0 : Ip[o];
            }
    }
    const int Tint = T - 2;
    const int J = a.warm[n * 2 + a.dir];
    // this workgroup's part of the interior: Lw nodes from interior node ow on (counted from the side the sweep starts at), again
    // cut into 16 segments; all columns but those that reach the chain's first node within J steps warm up from zero
    const int Lw = a.W > 1 ? ((((Tint + a.W - 1) / a.W) + 15) & ~15) : Tint;
    const int ow = part * Lw;
    const int Tw = (Tint - ow < Lw) ? Tint - ow : Lw;
    const int t_first = fwd ? 0 : T - 1, t_last = fwd ? T - 1 : 0;
    const double* Am = a.A_mean + (size_t)n * D * D;
    const double* Cm = a.C_mean + (size_t)n * K * D;
    const double* QAm = a.QA ? a.QA + (size_t)n * D * D : nullptr;
    const double* RCm = a.RC ? a.RC + (size_t)n * K * D : nullptr;

    // ---- first boundary node: only the old neighbour.  Every part whose warm-up can reach it computes it; the first one stores it.
    if (ow <= J) {               // block-uniform
        const double* xo = Xo + (size_t)(t_first + sgn) * BDP;
        const double s = big_boundary(fwd, g, L, Am, Cm, D, K, tid, [&](int j) { return xo[xpos(j)]; }, Yn + (size_t)t_first * K, vs, QAm, RCm);
        if (tid < BDP) {
            if (part == 0) Xn[(size_t)t_first * BDP + xpos(tid)] = (tid < D) ? s : 0.0;
            xs[tid] = (tid < D) ? s : 0.0;
        }
    }
    __syncthreads();

    if (Tw > 0) {
        const int Lseg = (Tw + 15) >> 4;
        const int cL = c * Lseg;
        const int before = ow + cL;                                          // interior nodes between the chain's start and this column
        const int jc = -(J < before ? J : before);                           // first loop index of this column
        const int jstart = -((J < ow + 15 * Lseg) ? J : ow + 15 * Lseg);     // of the workgroup
        // xb[4m + r][lane] = row 16m + 4r + q of column c: the registers go to the matrices and to the operands fetched a step ahead
        if (w == 0) {
#pragma unroll
            for (int m = 0; m < BDT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) xb0[(4 * m + r) * 64 + lane] = (jc == -before) ? xs[16 * m + 4 * r + q] : 0.0;
        }
        __syncthreads();
        const int tbase = fwd ? (1 + before) : (T - 2 - before);
        const int tsafe = fwd ? 1 : T - 2;
        auto active = [&](int j) { int tt = cL + j; return j >= jc && j < Lseg && tt < Tw; };
        double* const trash = a.trash + ((size_t)n * a.W + part) * 512;
        // Operands of a step from global memory.  (a) This wavefront's rows of U (G y_t, or c_t for the cached backward sweep): three
        // register sets, a row is requested two steps before it is used.  (b) The old neighbour mu_{t+dir}: the same 128 x 16 block
        // for all four wavefronts -- each fetches the two row tiles it also owns of the state (a quarter of the block) and puts
        // them into a ring of three LDS stages (B-operand order, like the state), requested three steps before the product that
        // reads them.  Loads are unconditional: an inactive column reads a valid row (tsafe) and its result is discarded by the
        // select.  (Every wavefront fetching the whole neighbour block into registers a step ahead, and U one step ahead: the
        // forward sweep lost a quarter of its time to late loads and to the barrier skew they cause -- 16.3 ms at N = 1024;
        // without the loads 14.6, without the barrier 14.6, without either and without the stores 12.3.)
        auto row_of = [&](int j) { return active(j) ? tbase + sgn * j : tsafe; };
        double* const ring = vs + BDP;                  // [3][BDS][64]
        d4 gq[NTW];                                     // this wavefront's share of a neighbour block on its way to LDS
        auto load_o = [&](int j) {
            const double* op = Xo + (size_t)(row_of(j) + sgn) * BDP;
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm) gq[mm] = *reinterpret_cast<const d4*>(op + ((NTW * w + mm) * 4 + q) * 4);
        };
        auto put_o = [&](int j) {                       // stage of step j: ((j - jstart) mod 3)
            double* st = ring + ((j - jstart) % 3) * (BDS * 64);
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(4 * (NTW * w + mm) + r) * 64 + lane] = gq[mm][r];
        };
        const double* const Un = (MODE == 2 ? a.Uc : a.U) + (size_t)n * T * BDP;     // what a step starts from: G y_t, or the cached c_t
        double* const Uw = a.Uc + (size_t)n * T * BDP;                               // where a forward sweep leaves c_t
        d4 cvA[NTW], cvB[NTW], cvC[NTW];
        auto load_c = [&](d4 (&cv)[NTW], int j) {
            const double* cp = Un + (size_t)row_of(j) * BDP;
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm) cv[mm] = *reinterpret_cast<const d4*>(cp + ((NTW * w + mm) * 4 + q) * 4);
        };
        // The product with the old neighbour, I mu_{t+dir}, does not depend on the state: the one of step j + 1 is formed at the
        // END of step j, between this wavefront's state writes and the barrier.
        d4 accI[NTW];
        auto iprod = [&](int j) {
            const double* st = ring + ((j - jstart) % 3) * (BDS * 64);
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm) accI[mm] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s0 = 0; s0 < BDS; s0 += CHAIN_RUN)
#pragma unroll
                for (int mm = 0; mm < NTW; ++mm)
#pragma unroll
                    for (int s = s0; s < s0 + CHAIN_RUN; ++s) accI[mm] = MFMA(ip[mm][s], st[s * 64 + lane], accI[mm]);
        };
        // cv: U rows of step j; cv_fill: the set that takes the rows of step j + 2
        auto step = [&](int j, const double* xr, double* xw, const d4 (&cv)[NTW], d4 (&cv_fill)[NTW]) {
            const bool act = active(j);
            d4 acc[NTW];
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm) acc[mm] = cv[mm];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE != 2) { put_o(j + 2); load_o(j + 3); }       // the share fetched a step ago goes to LDS, the next one is requested
            else load_c(cv_fill, j + 3);
            __builtin_amdgcn_sched_barrier(0);
            // R mu_{t-dir} (new): the segments' state
#pragma unroll
            for (int s0 = 0; s0 < BDS; s0 += CHAIN_RUN)
#pragma unroll
                for (int mm = 0; mm < NTW; ++mm)
#pragma unroll
                    for (int s = s0; s < s0 + CHAIN_RUN; ++s) acc[mm] = MFMA(rn[mm][s], xr[s * 64 + lane], acc[mm]);
            if constexpr (MODE != 2) {
                if (fwd) {                      // c_t for the backward sweep, this wavefront's rows
                    double* ur = (act && j >= 0) ? Uw + (size_t)(tbase + sgn * j) * BDP : trash + 256;
#pragma unroll
                    for (int mm = 0; mm < NTW; ++mm) *reinterpret_cast<d4*>(ur + ((NTW * w + mm) * 4 + q) * 4) = acc[mm];
                }
#pragma unroll
                for (int mm = 0; mm < NTW; ++mm) acc[mm] += accI[mm];           // I mu_{t+dir} (old), formed at the end of the step before
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE != 2) load_c(cv_fill, j + 3);             // after the store above: a column in its warm-up may read a row another column owns, never the
            __builtin_amdgcn_sched_barrier(0);  // other way round in the same step (see MODE 3 at the top)
            // this wavefront's rows of the new state: kept where the column is active, then shared
            double* out = (act && j >= 0) ? Xn + (size_t)(tbase + sgn * j) * BDP : trash;
#pragma unroll
            for (int mm = 0; mm < NTW; ++mm) {
                const int m = NTW * w + mm;
                d4 nx;
#pragma unroll
                for (int r = 0; r < 4; ++r) nx[r] = act ? acc[mm][r] : xr[(4 * m + r) * 64 + lane];
                *reinterpret_cast<d4*>(out + (m * 4 + q) * 4) = nx;
#pragma unroll
                for (int r = 0; r < 4; ++r) xw[(4 * m + r) * 64 + lane] = nx[r];
            }
            if constexpr (MODE != 2) {
                __builtin_amdgcn_sched_barrier(0);
                iprod(j + 1);
            }
            // the new state is complete; the buffer just read is free for the next step's writes.  An LDS-only barrier: the
            // wavefronts talk through LDS alone; nothing here should wait for the rows requested three steps ahead or the rows
            // just stored (the full __syncthreads() measured the same on this target).
            // (The one global hand-over, a warm-up column reading a U row its neighbour overwrites ~Lseg steps later, is ordered by
            // the use of the loaded value before a barrier that precedes the store.)
            if constexpr (MODE == 2) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // prologue: U rows of the first three steps; neighbour blocks of the first two steps in LDS, the third on its way
        load_c(cvA, jstart); load_c(cvB, jstart + 1); load_c(cvC, jstart + 2);
        if constexpr (MODE != 2) {
            load_o(jstart); put_o(jstart); load_o(jstart + 1); put_o(jstart + 1); load_o(jstart + 2);
            __syncthreads();
            iprod(jstart);
        }
        double* const xb1 = xb0 + BDS * 64;
        const double* xfin = xb0;
        // the U sets rotate with period 3, the state buffers with period 2
        // (six steps per turn: both rotations come round, every buffer and register set is a compile-time choice)
        int j = jstart;
        for (; j < Lseg; j += 6) {
            step(j, xb0, xb1, cvA, cvA); xfin = xb1;
            if (j + 1 >= Lseg) break;
            step(j + 1, xb1, xb0, cvB, cvB); xfin = xb0;
            if (j + 2 >= Lseg) break;
            step(j + 2, xb0, xb1, cvC, cvC); xfin = xb1;
            if (j + 3 >= Lseg) break;
            step(j + 3, xb1, xb0, cvA, cvA); xfin = xb0;
            if (j + 4 >= Lseg) break;
            step(j + 4, xb0, xb1, cvB, cvB); xfin = xb1;
            if (j + 5 >= Lseg) break;
            step(j + 5, xb1, xb0, cvC, cvC); xfin = xb0;
        }
        // the column that holds the last interior node hands its state to the closing boundary step
        const int clast = (Tw - 1) / Lseg;
        if (w == 0 && c == clast) {
#pragma unroll
            for (int m = 0; m < BDT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[16 * m + 4 * r + q] = xfin[(4 * m + r) * 64 + lane];
        }
        __syncthreads();
    }
    // ---- closing boundary node: only the new neighbour; done by the part that owns the last interior node
    if (part == ((Tint > 0) ? (Tint - 1) / Lw : 0)) {
        const double s = big_boundary(!fwd, g, L, Am, Cm, D, K, tid, [&](int j) { return xs[j]; }, Yn + (size_t)t_last * K, vs, QAm, RCm);
        if (tid < BDP) Xn[(size_t)t_last * BDP + xpos(tid)] = (tid < D) ? s : 0.0;
    }
}

int launch_sweep_big(pyvb_lds* h, int direction) {
    BigSweepArgs a;
    a.Xold = h->X[h->cur]; a.Xnew = h->X[1 - h->cur]; a.Y = h->Y; a.gains = h->gains; a.warm = h->warm;
    a.A_mean = h->A_mean; a.C_mean = h->C_mean; a.trash = h->trash; a.U = h->U;
    a.QA = h->dense ? h->QA : nullptr; a.RC = h->dense ? h->RC : nullptr;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.dir = direction; a.L = h->L;
    a.W = h->W; a.Uc = h->W > 1 ? h->U2 : h->U;
    // a backward sweep right behind a forward one (h->u_valid) reads c_t; anything else starts from G y_t
    const bool cached = direction == PYVB_BACKWARD && h->u_valid;
    const size_t lds = ((size_t)2 * BDS * 64 + 2 * BDP + (cached ? 0 : 3 * BDS * 64)) * sizeof(double);        // MODE 3: + the neighbour ring
    if (!cached && h->T > 2) {
        BigGyArgs ga;
        ga.Y = h->Y; ga.gains = h->gains; ga.U = h->U; ga.trash = h->trash;
        ga.N = h->N; ga.T = h->T; ga.K = h->K; ga.L = h->L;
        const int blocks = (h->T - 2 + 15) / 16;
        ga.nblk = blocks < 32 ? ((blocks + 1) & ~1) : 32;
        TimedLaunch tl(h, PYVB_K_GY);
        const dim3 grid((blocks + ga.nblk - 1) / ga.nblk, h->N);
        static const bool four = [] { const char* e = getenv("PYVB_GY_BIG"); return !(e && e[0] == '8'); }();     // PYVB_GY_BIG=8: k_gy_big8 (measured the same: see there)
        if (four) {
            if ((h->K & 1) == 0) hipLaunchKernelGGL(k_gy_big<true>, grid, dim3(256), 0, h->stream, ga);
            else hipLaunchKernelGGL(k_gy_big<false>, grid, dim3(256), 0, h->stream, ga);
        }
        else if ((h->K & 3) == 0) hipLaunchKernelGGL(k_gy_big8<true>, grid, dim3(512), 0, h->stream, ga);
        else hipLaunchKernelGGL(k_gy_big8<false>, grid, dim3(512), 0, h->stream, ga);
    }
    {
        TimedLaunch tl(h, direction == PYVB_FORWARD ? PYVB_K_SWEEP_FWD : PYVB_K_SWEEP_BWD);
        if (cached) hipLaunchKernelGGL((k_sweep_big<2, 2>), dim3(h->N, h->W), dim3(256), lds, h->stream, a);
        else hipLaunchKernelGGL((k_sweep_big<3, 2>), dim3(h->N, h->W), dim3(256), lds, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

// Xs[t].update() alone, in place in the current buffer (k_sweep.hip: k_step), thread = row
struct BigStepArgs {
    double* X; const double* Y; const double* gains; const double *A_mean, *C_mean, *QA, *RC;
    int N, T, D, K, t;
    Layout L;
};

__global__ void __launch_bounds__(128) k_step_big(BigStepArgs a) {
    __shared__ double vs[BDP];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int T = a.T, D = a.D, K = a.K, t = a.t;
    const Layout& L = a.L;
    const double* g = a.gains + (size_t)n * L.gains_total;
    double* X = a.X + (size_t)n * T * BDP;
    const double* y = a.Y + ((size_t)n * T + t) * K;
    const int cls = (t == 0) ? 0 : (t == T - 1 ? 2 : 1);
    if (cls != 1) {
        const double* nbr = X + (size_t)(cls == 0 ? 1 : T - 2) * BDP;
        const double s = big_boundary(cls == 0, g, L, a.A_mean + (size_t)n * D * D, a.C_mean + (size_t)n * K * D, D, K, tid,
                                      [&](int j) { return nbr[xpos(j)]; }, y, vs,
                                      a.QA ? a.QA + (size_t)n * D * D : nullptr, a.RC ? a.RC + (size_t)n * K * D : nullptr);
        X[(size_t)t * BDP + xpos(tid)] = (tid < D) ? s : 0.0;
        return;
    }
    // interior node: rows of F, B, G picked out of the MFMA operand blocks (the rare path)
    double s = 0.0;
    if (tid < D) {
        for (int j = 0; j < D; ++j) s += g[L.oFn + pos_nat(tid, j, BDS)] * X[(size_t)(t - 1) * BDP + xpos(j)];
        for (int j = 0; j < D; ++j) s += g[L.oBn + pos_nat(tid, j, BDS)] * X[(size_t)(t + 1) * BDP + xpos(j)];
        for (int k = 0; k < K; ++k) s += g[L.oGp + pos_perm(tid, k, BDS)] * y[k];
    }
    __syncthreads();            // every thread has read its neighbours' rows (t - 1, t + 1 are other rows: no hazard; kept for clarity)
    X[(size_t)t * BDP + xpos(tid)] = (tid < D) ? s : 0.0;
}

int launch_step_big(pyvb_lds* h, int t) {
    BigStepArgs a;
    a.X = h->X[h->cur]; a.Y = h->Y; a.gains = h->gains; a.A_mean = h->A_mean; a.C_mean = h->C_mean;
    a.QA = h->dense ? h->QA : nullptr; a.RC = h->dense ? h->RC : nullptr;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.t = t; a.L = h->L;
    TimedLaunch tl(h, PYVB_K_STEP);
    hipLaunchKernelGGL(k_step_big, dim3(h->N), dim3(128), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

// ======================================================================================================================
// statistics
// ======================================================================================================================
struct BigStatsArgs {
    const double* X; const double* Y; double* part; const double* zeros;     // zeros: 128 doubles
    int N, T, D, K, nchunk, chunk_len;
    Layout L;
};

// Sxx = sum_t mu_t mu_t^T,  Sx1x = sum_t mu_{t+1} mu_t^T,  Syx = sum_t y_t mu_t^T  in one pass: T is the MFMA K dimension
// (k_stats.hip).  Eight wavefronts per workgroup; wavefront w forms row tile w of all three sums against the eight column
// tiles of mu_t, so the eight B operands of a k-step (the expensive part: 4 KB per wavefront) serve 24 MFMAs, and the A
// operand of Sxx is one of them.  96 accumulator registers per lane; operands are fetched one k-step ahead; the eight
// wavefronts read the same rows, seven of them from L1.
__global__ void __launch_bounds__(512) k_stats_big(BigStatsArgs a) {
    const int ch = blockIdx.x, n = blockIdx.y;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int T = a.T, K = a.K;
    const double* X = a.X + (size_t)n * T * BDP;
    const double* Y = a.Y + (size_t)n * T * K;
    const double* Z = a.zeros;
    const int t0 = ch * a.chunk_len;
    const int t1 = (t0 + a.chunk_len < T) ? t0 + a.chunk_len : T;
    double* P = a.part + ((size_t)n * a.nchunk + ch) * a.L.stats_total;
    // Column tiles in ROTATED order: operand j of this wavefront is tile (w + j) mod 8.  Operand 0 is then its own row tile -- the A
    // operand of Sxx -- and the symmetric Sxx needs only j = 0..3, plus j = 4 on the wavefronts 0..3: every unordered pair of tiles
    // {a, b} is formed exactly once, by the wavefront from which the other tile is at most 4 (3 for w >= 4) steps ahead, and written
    // to both places.  20-21 MFMAs per k-step instead of 24.
    int xoff[BDT];
#pragma unroll
    for (int j = 0; j < BDT; ++j) xoff[j] = xpos(16 * ((w + j) & 7) + r);
    const int ydim = 16 * w + r;
    d4 sxx[5], sx1[BDT], syx[BDT];
#pragma unroll
    for (int k = 0; k < BDT; ++k) { sx1[k] = d4{0.0, 0.0, 0.0, 0.0}; syx[k] = sx1[k]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) sxx[k] = d4{0.0, 0.0, 0.0, 0.0};
    double bv[3][BDT], x1v[3], yv[3];     // three operand sets: a k-step's rows are requested two k-steps (about 2.4 us) before their MFMAs
    auto fetch = [&](int tb, int h) {
        const int t = tb + q;
        const double* xb = t < t1 ? X + (size_t)t * BDP : Z;                          // rows beyond the chunk read as zeros
        const double* x1 = (t < t1 && t + 1 < T) ? X + (size_t)(t + 1) * BDP : Z;
        const double* yb = t < t1 ? Y + (size_t)t * K : Z;
#pragma unroll
        for (int k = 0; k < BDT; ++k) bv[h][k] = xb[xoff[k]];
        x1v[h] = x1[xoff[0]];
        yv[h] = ydim < K ? yb[ydim] : 0.0;
    };
    auto step = [&](int h) {
        const double xa = bv[h][0];
#pragma unroll
        for (int k = 0; k < BDT; ++k) {
            if (k < 4) sxx[k] = MFMA(xa, bv[h][k], sxx[k]);
            sx1[k] = MFMA(x1v[h], bv[h][k], sx1[k]);
            syx[k] = MFMA(yv[h], bv[h][k], syx[k]);
        }
        if (w < 4) sxx[4] = MFMA(xa, bv[h][4], sxx[4]);
    };
    fetch(t0, 0); fetch(t0 + 4, 1);
    for (int tb = t0; tb < t1; tb += 12) {
        fetch(tb + 8, 2);
        step(0);
        fetch(tb + 12, 0);
        step(1);
        fetch(tb + 16, 1);
        step(2);
    }
#pragma unroll
    for (int k = 0; k < BDT; ++k) {
        const int ct = (w + k) & 7;             // the column tile operand k stands for
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t pos = (size_t)(16 * w + 4 * e + q) * BDP + 16 * ct + r;
            P[a.L.oSx1x + pos] = sx1[k][e];
            P[a.L.oSyx + pos] = syx[k][e];
            if (k < 4 || (k == 4 && w < 4)) {
                const double v = sxx[k < 5 ? k : 0][e];
                P[a.L.oSxx + pos] = v;
                if (k > 0) P[a.L.oSxx + (size_t)(16 * ct + r) * BDP + 16 * w + 4 * e + q] = v;      // the mirrored tile
            }
        }
    }
}

int launch_stats_big(pyvb_lds* h) {
    BigStatsArgs a;
    a.X = h->X[h->cur]; a.Y = h->Y; a.part = h->stats; a.zeros = h->zeros;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.nchunk = h->nchunk; a.chunk_len = h->chunk_len; a.L = h->L;
    {
        TimedLaunch tl(h, PYVB_K_STATS);
        hipLaunchKernelGGL(k_stats_big, dim3(h->nchunk, h->N), dim3(512), 0, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

// ======================================================================================================================
// prep
// ======================================================================================================================
struct BigPrepArgs {
    const double *A_mean, *A_var, *C_mean, *C_var, *Q_a, *Q_b, *R_a, *R_b, *x0_mean, *x0_prec;
    double *Sigma, *qld, *gains, *scratch;      // scratch: [N][2][128][128]
    // Wishart noise (dense): E[Q] [D][D], E[Q]<A> [D][D], E[R]<C> [K][D], tr(S_i E[Q]) [D], tr(S'_i E[R]) [D] per replicate
    const double *Qbar, *QA, *RC, *trA, *trC;
    int *warm, *status;
    int N, T, D, K, noise, dense;
    Layout L;
};

// C = A * B, 128 x 128 x 128 on the matrix cores; wavefront w owns row tiles w and w + 4; a_at(i, k), b_at(k, j) fetch
// operand elements, store(i, j, v) consumes results.  A row tile's 32 A operands are fetched at once; a column tile is ONE chain
// of 32 dependent MFMAs into one accumulator (at one wavefront per SIMD a chain runs at 70 cycles per MFMA, eight accumulators
// taken in turn at twice that), its 32 B operands fetched while the chain before it runs.  (Operands fetched where they were
// used, eight accumulators in turn: 620 cycles per MFMA with operands in global memory, 170 with both in LDS -- cycle stamps.)
// b_at should walk memory along j for neighbouring lanes (row-major B): its loads are 16 lanes x 8 contiguous bytes.
#ifdef PREP_STAMP
#define MMSTAMP(i) do { if (mst) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); mst[i] += t_ - mt; mt = t_; } } while (0)
#else
#define MMSTAMP(i) do { } while (0)
#endif
template <class FA, class FB, class FS>
__device__ __forceinline__ void mm128(int wave, int lane, FA a_at, FB b_at, FS store, unsigned long long* mst = nullptr) {
    const int r = lane & 15, q = lane >> 4;
#ifdef PREP_STAMP
    unsigned long long mt = __builtin_amdgcn_s_memtime();
#endif
    for (int m = wave; m < BDT; m += 4) {
        double av[BDS], bA[BDS], bB[BDS];
#pragma unroll
        for (int s = 0; s < BDS; ++s) av[s] = a_at(16 * m + r, 4 * s + q);
        MMSTAMP(0);
        auto fetch = [&](double (&bv)[BDS], int nn) {
            const int nc = nn < BDT ? nn : BDT - 1;
#pragma unroll
            for (int s = 0; s < BDS; ++s) bv[s] = b_at(4 * s + q, 16 * nc + r);
        };
        auto chain = [&](const double (&bv)[BDS], int nn) {
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            MMSTAMP(1);
#if defined(MM_WAIT)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(MM_FENCE)
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int s = 0; s < BDS; ++s) acc = MFMA(av[s], bv[s], acc);
#ifdef PREP_STAMP
            if (mst) { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
#endif
            MMSTAMP(2);
#pragma unroll
            for (int e = 0; e < 4; ++e) store(16 * m + 4 * e + q, 16 * nn + r, acc[e]);
            MMSTAMP(3);
        };
        fetch(bA, 0);
#pragma unroll 1
        for (int nn = 0; nn < BDT; nn += 2) {
            fetch(bB, nn + 1);
            __builtin_amdgcn_sched_barrier(0);
            chain(bA, nn);
            __builtin_amdgcn_sched_barrier(0);
            fetch(bA, nn + 2);
            __builtin_amdgcn_sched_barrier(0);
            chain(bB, nn + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// W <- W * W in place (LDS, stride BLD): both row tiles of every wavefront into registers, a barrier, then out.  Chains as in mm128:
// a column tile is one chain of 32 dependent MFMAs whose 32 B operands were requested while the chain before it ran (read inside
// the chain, each MFMA waited for its own LDS round trip: 170 cycles per MFMA instead of 70 -- the ten squarings of the two warm-up
// bounds were 0.9 M of k_prep_big's 2.9 M cycles per workgroup).
__device__ __forceinline__ void square128(double* W, int wave, int lane) {
    const int r = lane & 15, q = lane >> 4;
    d4 acc[2][BDT];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int m = wave + 4 * h2;
        double av[BDS], bA[BDS], bB[BDS];
#pragma unroll
        for (int s = 0; s < BDS; ++s) av[s] = W[(16 * m + r) * BLD + 4 * s + q];
        auto fetch = [&](double (&bv)[BDS], int nn) {
            const int nc = nn < BDT ? nn : BDT - 1;
#pragma unroll
            for (int s = 0; s < BDS; ++s) bv[s] = W[(4 * s + q) * BLD + 16 * nc + r];
        };
        auto chain = [&](const double (&bv)[BDS], d4& c) {
            c = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < BDS; ++s) c = MFMA(av[s], bv[s], c);
        };
        fetch(bA, 0);
#pragma unroll
        for (int nn = 0; nn < BDT; nn += 2) {
            fetch(bB, nn + 1);
            __builtin_amdgcn_sched_barrier(0);
            chain(bA, acc[h2][nn]);
            __builtin_amdgcn_sched_barrier(0);
            fetch(bA, nn + 2);
            __builtin_amdgcn_sched_barrier(0);
            chain(bB, acc[h2][nn + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int nn = 0; nn < BDT; ++nn)
#pragma unroll
            for (int e = 0; e < 4; ++e) W[(16 * (wave + 4 * h2) + 4 * e + q) * BLD + 16 * nn + r] = acc[h2][nn][e];
    __syncthreads();
}

// max_i sum_j |W_ij| over the 128 x 128 matrix: two threads per row
__device__ __forceinline__ double inf_norm128(const double* W, int tid, double* red) {
    const int row = tid >> 1, part = tid & 1;
    double s = 0.0;
#pragma unroll 8
    for (int u = 0; u < 64; ++u) s += fabs(W[row * BLD + 64 * part + u]);
    s += __shfl_xor(s, 1, 64);
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) s = fmax(s, __shfl_xor(s, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    return m;
}

// k_prep.hip: warmup_length, for the 128 x 128 recurrence matrix in W (clobbered)
// (forced inline: as a called function its squarings ran with ~1000 scratch accesses each -- the calling convention's saved registers
// and the spills they cause -- and took 2.7x the matrix pipe's time)
__device__ __forceinline__ int warmup128(double* W, int tid, double* red) {
    const double lntol = -41.4465316738928;   // ln(1e-18)
    const int wave = tid >> 6, lane = tid & 63;
    double l2 = 1.0, l3 = 1.0, l4 = 1.0, l5 = 1.0;
    for (int k = 1; k <= 5; ++k) {
        square128(W, wave, lane);
        if (k >= 2) {
            const double nrm = inf_norm128(W, tid, red);
            const double l = nrm < 1.0 ? ((nrm > 0.0) ? log(nrm) : -1e300) : 1.0;
            if (k == 2) l2 = l; else if (k == 3) l3 = l; else if (k == 4) l4 = l; else l5 = l;
        }
    }
    int best = 1 << 30;
    if (tid == 0) {
        for (int d = 0; d <= 16; ++d)
            for (int abc = 0; abc < 8; ++abc) {
                const int a = abc & 1, b = (abc >> 1) & 1, c = abc >> 2;
                if ((a && l2 > 0.0) || (b && l3 > 0.0) || (c && l4 > 0.0) || (d && l5 > 0.0)) continue;
                const double bound = (a ? l2 : 0.0) + (b ? l3 : 0.0) + (c ? l4 : 0.0) + (d ? d * l5 : 0.0);
                const int J = 4 * a + 8 * b + 16 * c + 32 * d;
                if (J > 0 && bound <= lntol && J < best) best = J;
            }
    }
    return best;
}

// a pass over the 128 x 128 elements by the workgroup's 256 threads, eight loads in flight per thread (rolled, with one: ~60 000
// cycles a pass, and the kernel makes eight of them)
#define STAGE_LOOP _Pragma("unroll 8") for (int u_ = 0, idx = threadIdx.x; u_ < BDP * BDP / 256; ++u_, idx += 256)
__global__ void __launch_bounds__(256) k_prep_big(BigPrepArgs a) {
    extern __shared__ double lds[];
#ifdef PREP_STAMP       // (profiles/build_variant.sh k_big pstamp "-DPREP_STAMP": where a workgroup's time goes, shader-clock ticks)
    unsigned long long ps_t = __builtin_amdgcn_s_memtime(), ps_acc[6] = {0, 0, 0, 0, 0, 0}, ps_mm[4] = {0, 0, 0, 0}, ps_t0 = 0;
#define PSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ps_acc[i] += t_ - ps_t; ps_t = t_; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    double* Pm = lds;                       // [128][BLD]
    double* qbar = Pm + BDP * BLD;          // [128]
    double* rbar = qbar + BDP;
    double* rowp = rbar + BDP;              // tr(S'_i <R>), later scratch of the norms
    double* colp = rowp + BDP;              // tr(S_i <Q>)
    double* gjrc = colp + BDP;              // [2][GJB_BUF]
    double* pivs = gjrc + 2 * GJB_BUF;      // [128]
    const int n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int wave = tid >> 6, lane = tid & 63;
    const Layout& L = a.L;
    const double* Am = a.A_mean + (size_t)n * D * D;
    const double* Av = a.A_var + (size_t)n * D * D;
    const double* Cm = a.C_mean + (size_t)n * K * D;
    const double* Cv = a.C_var + (size_t)n * D * K;
    double* g = a.gains + (size_t)n * L.gains_total;
    double* S1 = a.scratch + (size_t)n * 2 * BDP * BDP;     // M_C, then F (zero padded)
    const bool dense = a.dense != 0;            // Wishart noise: the products with the noise expectations come from k_dense_pre_big
    if (tid < BDP) {
        qbar[tid] = (!dense && tid < D) ? a.Q_a[(size_t)n * D + tid] / a.Q_b[(size_t)n * D + tid] : 0.0;
        rbar[tid] = (!dense && tid < K) ? a.R_a[(size_t)n * K + tid] / a.R_b[(size_t)n * K + tid] : 0.0;
    }
    const double* Qd = dense ? a.Qbar + (size_t)n * D * D : nullptr;
    const double* QAd = dense ? a.QA + (size_t)n * D * D : nullptr;
    const double* RCd = dense ? a.RC + (size_t)n * K * D : nullptr;
    auto QA_at = [&](int k, int j) { const double v = QAd[(size_t)(k < D ? k : D - 1) * D + (j < D ? j : D - 1)]; return (k < D && j < D) ? v : 0.0; };
    auto RC_at = [&](int k, int j) { const double v = RCd[(size_t)(k < K ? k : K - 1) * D + (j < D ? j : D - 1)]; return (k < K && j < D) ? v : 0.0; };
    auto A_at = [&](int i, int j) { const double v = Am[(size_t)(i < D ? i : D - 1) * D + (j < D ? j : D - 1)]; return (i < D && j < D) ? v : 0.0; };
    auto C_at = [&](int k, int j) { const double v = Cm[(size_t)(k < K ? k : K - 1) * D + (j < D ? j : D - 1)]; return (k < K && j < D) ? v : 0.0; };
    __syncthreads();
    if (tid < BDP) {        // traces of the column covariances against the noise expectations (diagonal of node.py:223-227)
        double tc = 0.0, ta = 0.0;
        if (tid < D) {
            if (dense) { tc = a.trC[(size_t)n * D + tid]; ta = a.trA[(size_t)n * D + tid]; }
            else {
                for (int k = 0; k < K; ++k) tc += Cv[(size_t)tid * K + k] * rbar[k];
                for (int k = 0; k < D; ++k) ta += Av[(size_t)tid * D + k] * qbar[k];
            }
        }
        rowp[tid] = tc; colp[tid] = ta;
    }
    __syncthreads();
    // <C^T R C> -> S1;  <C^T R C> + <A^T Q A> -> Pm        (node.py:213-227)
    // The means are staged, zero padded, in the LDS work matrix first: both operands of a product are then LDS reads without
    // bounds logic (fetching them from the arrays inside the product: 424k cycles per product, 3x this).
    double* S2 = S1 + BDP * BDP;
    STAGE_LOOP { const int k = idx >> 7, j = idx & 127; Pm[k * BLD + j] = C_at(k, j); }
    __syncthreads();
#ifdef PREP_STAMP
    ps_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (dense)      // <C>^T (E[R]<C>): the second operand straight from the array (the rare, untuned case)
        mm128(wave, lane, [&](int i, int k) { return Pm[k * BLD + i]; }, [&](int k, int j) { return RC_at(k, j); },
              [&](int i, int j, double v) { S1[(size_t)i * BDP + j] = (i < D && j < D) ? v + (i == j ? rowp[i] : 0.0) : 0.0; });
    else
    mm128(wave, lane, [&](int i, int k) { return Pm[k * BLD + i] * rbar[k]; }, [&](int k, int j) { return Pm[k * BLD + j]; },
          [&](int i, int j, double v) { S1[(size_t)i * BDP + j] = (i < D && j < D) ? v + (i == j ? rowp[i] : 0.0) : 0.0; }
#if defined(PREP_STAMP) && !defined(PREP_STAMP_NOMM)
          , ps_mm
#endif
          );
#ifdef PREP_STAMP
    ps_t = __builtin_amdgcn_s_memtime(); ps_acc[5] += ps_t - ps_t0;
#endif
    __syncthreads();
    STAGE_LOOP { const int k = idx >> 7, j = idx & 127; Pm[k * BLD + j] = A_at(k, j); }
    __syncthreads();
    if (dense)
        mm128(wave, lane, [&](int i, int k) { return Pm[k * BLD + i]; }, [&](int k, int j) { return QA_at(k, j); },
              [&](int i, int j, double v) { S2[(size_t)i * BDP + j] = (i < D && j < D) ? v + (i == j ? colp[i] : 0.0) : 0.0; });
    else
    mm128(wave, lane, [&](int i, int k) { return Pm[k * BLD + i] * qbar[k]; }, [&](int k, int j) { return Pm[k * BLD + j]; },
          [&](int i, int j, double v) { S2[(size_t)i * BDP + j] = (i < D && j < D) ? v + (i == j ? colp[i] : 0.0) : 0.0; });
    __syncthreads();
    // (the sum of the two moment matrices is formed in this pass: read inside the second product's stores, S1 cost every chain a trip to memory)
    STAGE_LOOP Pm[(idx >> 7) * BLD + (idx & 127)] = S1[idx] + S2[idx];
    __syncthreads();

    PSTAMP(0);
    // the three posterior precisions (gaussian.py:117), inverted one after the other (qcov, :118-119; q_ln_det, :120)
    const int ta = tid >> 4, tb = tid & 15;
    for (int cc = 0; cc < 3; ++cc) {
        const int c = cc == 0 ? 0 : (cc == 1 ? 2 : 1);             // the interior class last: its inverse stays in Pm
        double v[8][8];
#pragma unroll
        for (int ra = 0; ra < 8; ++ra)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int i = 8 * ta + ra, j = 8 * tb + cb;
                const bool in = i < D && j < D;
                const size_t at = (size_t)(i < D ? i : D - 1) * D + (j < D ? j : D - 1);     // (loads from clamped places, unconditional: no branch per element)
                double qd = 0.0;
                if (dense) { const double t = Qd[at]; qd = in ? t : 0.0; }
                else qd = (in && i == j) ? qbar[i] : 0.0;
                const double pad = (!in && i == j) ? 1.0 : 0.0;
                double x;
                if (c == 0) { const double t = a.x0_prec[at]; x = (in ? t : 0.0) + Pm[i * BLD + j]; }
                else if (c == 1) x = qd + Pm[i * BLD + j];
                else x = qd + S1[(size_t)i * BDP + j];
                v[ra][cb] = x + pad;
            }
        __syncthreads();
        PSTAMP(1);
        gj_wg128(v, D, tid, gjrc, pivs);
        PSTAMP(2);
        if (tid < 64) {
            double lp = 0.0;
            for (int k = tid; k < D; k += 64) {
                const double piv = pivs[k];
                if (!(piv > 0.0)) atomicOr(a.status, 1);
                lp += log(piv);
            }
            lp = wave_sum(lp);
            if (tid == 0) a.qld[(size_t)n * 3 + c] = 0.5 / (0.5 * lp);
        }
        // a thread's eight entries of a row are 64 contiguous bytes: with D a multiple of four they are stored as two 32-byte pieces
        // (element by element a wavefront's store touched 64 separate sectors and wrote a quarter of each)
        const bool vec = (D & 3) == 0;
        if (vec) {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                const int i = 8 * ta + ra;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int j0 = 8 * tb + 4 * hf;
                    const bool in = i < D && j0 < D;
                    const d4 val = in ? d4{v[ra][4 * hf], v[ra][4 * hf + 1], v[ra][4 * hf + 2], v[ra][4 * hf + 3]} : d4{0.0, 0.0, 0.0, 0.0};
                    if (in) *reinterpret_cast<d4*>(a.Sigma + ((size_t)n * 3 + c) * D * D + (size_t)i * D + j0) = val;
                    if (c == 0) *reinterpret_cast<d4*>(g + L.oS0 + (size_t)i * BDP + j0) = val;
                    if (c == 2) *reinterpret_cast<d4*>(g + L.oS2 + (size_t)i * BDP + j0) = val;
                    if (c == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Pm[i * BLD + j0 + e] = val[e];     // every thread has read its tile of Pm: barriers inside gj_wg128
                    }
                }
            }
        } else {
#pragma unroll
            for (int ra = 0; ra < 8; ++ra) {
                const int i = 8 * ta + ra;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const int j = 8 * tb + cb;
                    const bool in = i < D && j < D;
                    if (in) a.Sigma[((size_t)n * 3 + c) * D * D + (size_t)i * D + j] = v[ra][cb];
                    if (c == 0) g[L.oS0 + (size_t)i * BDP + j] = in ? v[ra][cb] : 0.0;
                    if (c == 2) g[L.oS2 + (size_t)i * BDP + j] = in ? v[ra][cb] : 0.0;
                    if (c == 1) Pm[i * BLD + j] = in ? v[ra][cb] : 0.0;          // every thread has read its tile of Pm: barriers inside gj_wg128
                }
            }
        }
        __syncthreads();
    }
    PSTAMP(1);
    if (tid < BDP) {
        g[L.oqr + tid] = qbar[tid]; g[L.oqr + BDP + tid] = rbar[tid];
        double s = 0.0;        // L0 m0: the Constant mean parent of X_0 through its Constant precision
        if (tid < D) for (int j = 0; j < D; ++j) s += a.x0_prec[(size_t)tid * D + j] * a.x0_mean[j];
        g[L.ow0 + tid] = s;
    }
    // gains of the interior class: F = Sigma <Q><A>, B = Sigma <A>^T<Q>, G = Sigma <C>^T<R>, each formed TRANSPOSED: the operand a
    // chain re-reads eight times over (B) is then Sigma, which sits in LDS, and the other factor is the A operand, read once per row
    // tile straight from the parameter arrays -- no staging pass, no operand fetched from global memory inside the chains (round 3
    // staged the right-hand factors in the global scratch and read them there: 190 000 cycles a product against 60 000 for a product
    // with both operands in LDS).  Element (i, j) of a transposed product is element (j, i) of the gain; same products, same sums.
    mm128(wave, lane, [&](int i, int k) { return dense ? QA_at(k, i) : qbar[k] * A_at(k, i); }, [&](int k, int j) { return Pm[j * BLD + k]; },
          [&](int i, int j, double v) { const bool in = i < D && j < D; g[L.oFn + pos_nat(j, i, BDS)] = in ? v : 0.0; S1[(size_t)j * BDP + i] = in ? v : 0.0; });
    mm128(wave, lane, [&](int i, int k) { return dense ? QA_at(i, k) : A_at(i, k) * qbar[i]; }, [&](int k, int j) { return Pm[j * BLD + k]; },
          [&](int i, int j, double v) { g[L.oBn + pos_nat(j, i, BDS)] = (i < D && j < D) ? v : 0.0; });
    mm128(wave, lane, [&](int l, int k) { return dense ? RC_at(l, k) : C_at(l, k) * rbar[l]; }, [&](int k, int j) { return Pm[j * BLD + k]; },
          [&](int l, int j, double v) { g[L.oGp + pos_perm(j, l, BDS)] = (j < D && l < K) ? v : 0.0; });
    __syncthreads();
    PSTAMP(3);
    // warm-up lengths: powers of F, and of B^T (inf-norm of powers of B^T = 1-norm of powers of B)
    STAGE_LOOP Pm[(idx >> 7) * BLD + (idx & 127)] = S1[idx];
    __syncthreads();
    int Jw = warmup128(Pm, tid, rowp);
    if (tid == 0) a.warm[n * 2 + 0] = Jw;
    __syncthreads();
    STAGE_LOOP {
        const int i = idx >> 7, j = idx & 127;
        Pm[i * BLD + j] = g[L.oBn + pos_nat(j, i, BDS)];       // B^T (zero padded by the store above)
    }
    __syncthreads();
    Jw = warmup128(Pm, tid, rowp);
    if (tid == 0) a.warm[n * 2 + 1] = Jw;
#ifdef PREP_STAMP
    PSTAMP(4);
    if (blockIdx.x == 100 && tid == 0)
        printf("first product: from the staged operands to its last store %llu (A operands %llu | B operands, waits %llu | 16 chains of 32 MFMAs %llu | stores %llu)\n", ps_acc[5], ps_mm[0], ps_mm[1], ps_mm[2], ps_mm[3]);
    if (blockIdx.x == 100 && tid == 0)
        printf("k_prep_big: moments (2 products) %llu | tiles in and out of the three inversions %llu | the three inversions %llu | gains (3 products) %llu | warm-up bounds (10 squarings) %llu\n",
               ps_acc[0], ps_acc[1], ps_acc[2], ps_acc[3], ps_acc[4]);
#endif
}

int launch_prep_big(pyvb_lds* h) {
    BigPrepArgs a;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var;
    a.Q_a = h->Q_a; a.Q_b = h->Q_b; a.R_a = h->R_a; a.R_b = h->R_b;
    a.x0_mean = h->pri.x0_mean; a.x0_prec = h->pri.x0_prec;
    a.Sigma = h->Sigma_new; a.qld = h->qld_x_new; a.gains = h->gains; a.scratch = h->scratch;
    a.warm = h->warm; a.status = h->status;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.noise = h->noise; a.L = h->L;
    a.dense = h->dense ? 1 : 0; a.Qbar = h->Qbar; a.QA = h->QA; a.RC = h->RC; a.trA = h->trA; a.trC = h->trC;
    const size_t lds = ((size_t)BDP * BLD + 4 * BDP + 2 * GJB_BUF + BDP) * sizeof(double);
    if (!h->big_attr_prep) {           // per handle: the attribute belongs to the device the handle lives on
        HIPCHK(hipFuncSetAttribute((const void*)k_prep_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->big_attr_prep = true;
    }
    {
        TimedLaunch tl(h, PYVB_K_PREP);
        hipLaunchKernelGGL(k_prep_big, dim3(h->N), dim3(256), lds, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

// ======================================================================================================================
// columns of A and C, residuals, noise update
// ======================================================================================================================
// k_cols.hip for up to 128 rows, columns in order (Gauss-Seidel); fuse bit 0: residuals of the noise node, bit 1: and its update.
// The rows of the matrix decouple (diagonal noise, diagonal column priors): a row belongs to TWO neighbouring lanes, each with one
// half of it (64 columns) in registers; a column's update is a dot product of the row with a row of G -- the two halves meet in
// one cross-lane add -- and a write of one register, which for a run-time column index is a chain of 64 selects on the lane that
// owns that half.  G sits in LDS (128 KB, read-only after the start), so the column loop has no barrier at all; what a column needs
// from global memory (its prior, the linear term, a known entry) is fetched two columns ahead; the block sums of a column (the log
// determinant of its precision, its number of known entries) are wavefront sums left in LDS and added up after the loop.
// (The first version kept the matrix as [col][row] in LDS, one thread per row: two wavefronts per CU, five barriers and four
// dependent global loads per column -- 7.4 ms at N = 1024, D = K = 128.)
#define CB_H 64         // columns per lane
__global__ void __launch_bounds__(256) k_cols_big_rows(ParamArgs a) {
    extern __shared__ double lds[];
    double* Gl = lds;                       // [128][128] zero padded
    double* plp = Gl + BDP * BDP;           // [4 wavefronts][128 columns] sums of log precision
    double* pkn = plp + 4 * BDP;            // [4][128] numbers of known entries
    double* gd = pkn + 4 * BDP;             // [128] the diagonal of G
    double* red = gd + BDP;                 // [4]
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int rows = WHICH == 0 ? D : K;
    const int row = tid >> 1, half = tid & 1, lane = tid & 63, wave = tid >> 6;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs; // [row][col], NaN = not known
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    const bool live = row < rows;
    const int lr = live ? row : 0;
    for (int idx = tid; idx < BDP * BDP; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        Gl[idx] = (i < D && j < D && i != j) ? G[(size_t)i * D + j] : 0.0;            // without the diagonal: a column's own entry does
    }                                                                                   // not enter its update; gd holds it
    if (tid < BDP) gd[tid] = tid < D ? G[(size_t)tid * D + tid] : 0.0;
    double Mr[CB_H];                        // columns 64 half .. 64 half + 63 of this row
#pragma unroll
    for (int j = 0; j < CB_H; ++j) { const int col = CB_H * half + j; Mr[j] = (live && col < D) ? M[(size_t)lr * D + col] : 0.0; }
    const double lam = live ? (WHICH == 0 ? a.Q_a[(size_t)n * D + lr] / a.Q_b[(size_t)n * D + lr]
                                          : a.R_a[(size_t)n * K + lr] / a.R_b[(size_t)n * K + lr]) : 0.0;
    __syncthreads();
    // the dot product of this row with row i of G without its diagonal entry: this lane's half, then the neighbour's
    auto rowdot = [&](int i) {
        const double* gr = Gl + i * BDP + CB_H * half;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int j = 0; j < CB_H; j += 2) {
            const d2 g = *reinterpret_cast<const d2*>(gr + j);
            acc0 = __builtin_fma(Mr[j], g[0], acc0);
            acc1 = __builtin_fma(Mr[j + 1], g[1], acc1);
        }
        const double s = acc0 + acc1;
        return s + __shfl_xor(s, 1, 64);
    };
    struct ColIn { double p0, pmv, hv, ob; };
    auto fetch = [&](int i) {
        ColIn c;
        const int ic = i < D ? i : D - 1;
        c.p0 = pp[(size_t)ic * rows + lr]; c.pmv = pm[(size_t)lr * D + ic]; c.hv = H[(size_t)lr * D + ic]; c.ob = obs[(size_t)lr * D + ic];
        return c;
    };
    ColIn c0 = fetch(a.c0), c1 = fetch(a.c0 + 1);
    for (int i = a.c0; i < a.c1; ++i) {
        const ColIn cur = c0;
        c0 = c1; c1 = fetch(i + 2);
        const double dot = rowdot(i);
        const double gii = gd[i];
        const double prec = cur.p0 + lam * gii;                                         // qprec  gaussian.py:117
        double var = 1.0 / prec;                                                        // qcov   gaussian.py:118-119
        double val = (cur.p0 * cur.pmv + lam * (cur.hv - dot)) * var;                   // qmu :122-123
        // known entries (Gaussian.observe on a column, LDS_knowns_in_A.py:73-74): conditioning a diagonal Gaussian on them
        // (gaussian.py:125-134) pins those entries and leaves the others alone; a column whose entries are all known is
        // never changed (gaussian.py:109-110)
        const bool known = live && (cur.ob == cur.ob);
        if (known) { val = cur.ob; var = 0.0; }
        const bool mine = live && half == 0;
        const double lp = wave_sum(mine ? log(prec) : 0.0);
        const double nk = wave_sum((mine && known) ? 1.0 : 0.0);
        if (lane == 0) { plp[wave * BDP + i] = lp; pkn[wave * BDP + i] = nk; }
        if (mine) V[(size_t)i * rows + row] = var;
        if (live && half == (i >> 6)) {
            const int jj = i & (CB_H - 1);
#pragma unroll
            for (int j = 0; j < CB_H; ++j) Mr[j] = (j == jj) ? val : Mr[j];
        }
    }
    __syncthreads();
    if (tid >= a.c0 && tid < a.c1 && tid < BDP) {
        const double lp = ((plp[tid] + plp[BDP + tid]) + plp[2 * BDP + tid]) + plp[3 * BDP + tid];
        const double nk = ((pkn[tid] + pkn[BDP + tid]) + pkn[2 * BDP + tid]) + pkn[3 * BDP + tid];
        if ((int)nk < rows) qld[tid] = 0.5 / (0.5 * lp);                                // quirk Q1, gaussian.py:120: of the whole precision
    }
    if (a.c0 < a.c1 && live) {
#pragma unroll
        for (int j = 0; j < CB_H; ++j) { const int col = CB_H * half + j; if (col >= a.c0 && col < a.c1 && col < D) M[(size_t)row * D + col] = Mr[j]; }
    }
    if (a.fuse & 1) {
        // res[k] = 1/2 own[k] + 1/2 (sum_ij M[k,i] G[i,j] M[k,j] + sum_i var_i[k] G[i,i]) - sum_i H[k,i] M[k,i]   (node.py:260-271)
        double e = 0.0, hm = 0.0;
        auto fetch2 = [&](int i, double& vv, double& hh) { const int ic = i < D ? i : D - 1; vv = V[(size_t)ic * rows + lr]; hh = H[(size_t)lr * D + ic]; };
        __syncthreads();                    // this workgroup's variances of all columns are in memory (written by the lanes with half == 0)
        double v0, h0, v1, h1;
        fetch2(0, v0, h0); fetch2(1, v1, h1);
        for (int i = 0; i < D; ++i) {
            const double vi = v0, hi = h0;
            v0 = v1; h0 = h1; fetch2(i + 2, v1, h1);
            const int jj = i & (CB_H - 1);
            double mi = 0.0;
#pragma unroll
            for (int j = 0; j < CB_H; ++j) mi = (j == jj) ? Mr[j] : mi;
            mi = __shfl(mi, (lane & ~1) | (i >> 6), 64);                                // from the lane that owns column i of this row
            const double s = rowdot(i) + mi * gd[i];
            e += mi * s + vi * gd[i];
            hm += hi * mi;
        }
        const double own = WHICH == 0 ? mo[MOM_DP(D, K) + lr] : a.Syy[(size_t)n * K + lr];
        double r = 0.5 * own + 0.5 * e - hm;
        const bool mine = live && half == 0;
        if (mine) (WHICH == 0 ? a.resQ : a.resR)[(size_t)n * rows + row] = r;
        if (a.fuse & 2) {
            const double* b0 = WHICH == 0 ? a.pri.Q_b0 : a.pri.R_b0;
            double* qb = (WHICH == 0 ? a.Q_b : a.R_b) + (size_t)n * rows;
            if (a.noise == PYVB_NOISE_GAMMA) {
                double t = wave_sum(mine ? r : 0.0);
                __syncthreads();
                if (lane == 0) red[wave] = t;
                __syncthreads();
                t = ((red[0] + red[1]) + red[2]) + red[3];
                if (mine) qb[row] = b0[0] + t;
            } else if (mine) {
                qb[row] = b0[row] + r;
            }
        }
    }
}

// ---- Round 4: the same pass on the matrix cores, blocked.
// What paced k_cols_big_rows (3.9 ms at N = 1024, D = K = 128, no MFMA at all) is that every column costs every wavefront a pass over
// a row of G in LDS (the broadcast reads of four wavefronts: ~1000 LDS cycles per column) and a 64-deep select chain.  Here a wavefront
// owns 32 ROWS of the matrix outright (row tiles 2w, 2w+1) and keeps, in MFMA accumulator layout (lane (q, c), register e of tile
// (m, nn) = element (16 m + 4 e + q, 16 nn + c)),
//     Mx  the matrix itself, and
//     S   its product with G-without-diagonal:  S[r][i] = sum_{j != i} M[r][j] G[j][i]  -- the "dot" of column i, for all i at once.
// S starts as M_old G_off (a 128^3 product, 512 MFMAs per wavefront) and is kept current as columns change: a block of 16 columns
// is swept column by column inside the 16 lanes that hold one row's 16 entries (the new value of column j is formed in lane c = j,
// its change delta_j goes to the row's other lanes by one cross-lane read, S[r][16 I + c] += delta_j G[j][c] -- 16 doubles of G per
// lane), and when the block is done its 32 x 16 panel of changes updates S for the other column tiles by MFMA (panel through 4 KB
// of the wavefront's own LDS into A-operand order).  No workgroup barrier in the pass; precisions, variances, logarithms are
// elementwise per block.  The residual of the noise node, sum_i M (S + M g_ii) + var g_ii, is elementwise on S = M_new G_off, formed afresh.
// Same formulas, other summation order than k_cols_big_rows (kept: PYVB_COLS_BIG=rows).
#define CBP 17      // row stride of a wavefront's panel
__global__ void __launch_bounds__(256) k_cols_big(ParamArgs a) {
    extern __shared__ double lds[];
#ifdef COLS_STAMP       // (profiles/build_variant.sh k_big cstamp "-DCOLS_STAMP": where a workgroup's time goes, shader-clock ticks)
    unsigned long long cs_t = __builtin_amdgcn_s_memtime(), cs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cs_acc[i] += t_ - cs_t; cs_t = t_; } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
    double* Gl = lds;                       // [128][128] zero padded, zero diagonal
    double* plp = Gl + BDP * BDP;           // [4 wavefronts][128 columns] sums of log precision
    double* pkn = plp + 4 * BDP;            // [4][128] numbers of known entries
    double* gd = pkn + 4 * BDP;             // [128] the diagonal of G
    double* red = gd + BDP;                 // [8]
    const int WHICH = a.which0 + blockIdx.y, n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int rows = WHICH == 0 ? D : K;
    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    double* panel = red + 8 + wave * (32 * CBP);
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs; // [row][col], NaN = not known
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
#pragma unroll 8
    for (int u = 0; u < BDP * BDP / 256; ++u) {         // (unconditional loads, eight and more in flight)
        const int idx = tid + 256 * u, i = idx >> 7, j = idx & 127;
        const double v = G[(size_t)(i < D ? i : D - 1) * D + (j < D ? j : D - 1)];
        Gl[(idx & ~127) | (j ^ ((i & 1) << 4))] = (i < D && j < D && i != j) ? v : 0.0;     // odd rows: neighbouring 16-column tiles swapped (see mma16)
    }
    if (tid < BDP) gd[tid] = tid < D ? G[(size_t)tid * D + tid] : 0.0;
    for (int idx = tid; idx < 8 * BDP; idx += 256) plp[idx] = 0.0;          // plp and pkn: a wavefront without rows leaves zeros
    __syncthreads();
    CSTAMP(0);
    const bool work = 32 * wave < rows;     // this wavefront has rows at all (K may be small)
    const int nbl = (D + 15) >> 4;          // column blocks in use
    // this lane's eight rows: (mm, e) -> 32 wave + 16 mm + 4 e + q
    bool live[2][4]; int rowc[2][4]; double lam[2][4];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 32 * wave + 16 * mm + 4 * e + q;
            live[mm][e] = row < rows; rowc[mm][e] = row < rows ? row : 0;
            lam[mm][e] = live[mm][e] ? (WHICH == 0 ? a.Q_a[(size_t)n * D + rowc[mm][e]] / a.Q_b[(size_t)n * D + rowc[mm][e]]
                                                   : a.R_a[(size_t)n * K + rowc[mm][e]] / a.R_b[(size_t)n * K + rowc[mm][e]]) : 0.0;
        }
    d4 S[2][BDT];
    // this lane's eight entries of column block J (accumulator layout) from the matrix in memory; zero beyond the matrix
    auto load_m = [&](d4 (&p)[2], int J) {
        const int col = 16 * J + c, colc = col < D ? col : D - 1;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {           // (unconditional loads from clamped places: a branch per entry otherwise)
                const double v = M[(size_t)rowc[mm][e] * D + colc];
                p[mm][e] = (live[mm][e] && col < D) ? v : 0.0;
            }
    };
    // S[.][nn] += P G_off[16 J .., 16 nn ..] for a 32 x 16 panel P given as A operands (av[mm][s4] = P[16 mm + c][4 s4 + q]); nn == skip
    // is left out, and so is nn < lo.  The B operands of a column tile are requested while the tile before it is multiplied.
    auto mma16 = [&](const double (&av)[2][4], int J, int skip, int lo) {
        // (G in LDS has no room for a padded row stride, and rows 1 KB apart share their banks: the four rows a B operand read
        // touches would meet in them.  Odd rows keep each pair of 16-column tiles swapped, so that the rows q = 0, 2 and q = 1, 3
        // of a read lie 32 banks apart.)
        const double* gj = Gl + (size_t)(16 * J + q) * BDP + c;
        const int sw = (q & 1) << 4;
        double b[2][4];
        auto fetch_b = [&](double (&bv)[4], int nn) {
            const int nc = nn < BDT ? nn : BDT - 1;
            const double* gp = gj + 16 * nc + ((nc & 1) ? -sw : sw);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) bv[s4] = gp[(size_t)(4 * s4) * BDP];
        };
        fetch_b(b[0], 0);
#pragma unroll
        for (int nn = 0; nn < BDT; ++nn) {
            fetch_b(b[(nn + 1) & 1], nn + 1);
            if (nn == skip || nn < lo || nn >= nbl) continue;       // wavefront-uniform
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)          // one chain after the other (accumulators taken in turn run at half the rate: mm128)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) S[mm][nn] = MFMA(av[mm][s4], b[nn & 1][s4], S[mm][nn]);
        }
    };
    // the same for a panel given in accumulator layout (p[mm][e]): through this wavefront's 4 KB of LDS into A-operand order
    auto rank16 = [&](const d4 (&p)[2], int J, int skip, int lo) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int e = 0; e < 4; ++e) panel[(16 * mm + 4 * e + q) * CBP + c] = p[mm][e];
        asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
        double av[2][4];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) av[mm][s4] = panel[(16 * mm + c) * CBP + 4 * s4 + q];
        mma16(av, J, skip, lo);
        asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    };
    const bool fuse1 = (a.fuse & 1) != 0;
    // S = M G_off for the matrix as it stands in memory, as mm128 forms a product: a row tile's 32 A operands straight from memory
    // (lane (q, c): row 16 mm + c of this wavefront's 32, columns 4 k + q), a column tile is ONE chain of 32 dependent MFMAs whose
    // B operands were requested while the chain before it ran
    auto full_product = [&]() {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int row = 32 * wave + 16 * mm + c;
            const double* mp = M + (size_t)(row < rows ? row : 0) * D;
            double av[BDS], bA[BDS], bB[BDS];
#pragma unroll
            for (int k = 0; k < BDS; ++k) {
                const int col = 4 * k + q;
                const double v = mp[col < D ? col : D - 1];
                av[k] = (row < rows && col < D) ? v : 0.0;
            }
            const double* g0 = Gl + (size_t)q * BDP + c;
            const int sw = (q & 1) << 4;
            auto fetch = [&](double (&bv)[BDS], int nn) {
                const int nc = nn < BDT ? nn : BDT - 1;
                const double* gp = g0 + 16 * nc + ((nc & 1) ? -sw : sw);
#pragma unroll
                for (int k = 0; k < BDS; ++k) bv[k] = gp[(size_t)(4 * k) * BDP];
            };
            auto chain = [&](const double (&bv)[BDS], d4& acc) {
                acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < BDS; ++k) acc = MFMA(av[k], bv[k], acc);
            };
            fetch(bA, 0);
#pragma unroll
            for (int nn = 0; nn < BDT; nn += 2) {
                fetch(bB, nn + 1);
                __builtin_amdgcn_sched_barrier(0);
                chain(bA, S[mm][nn]);
                __builtin_amdgcn_sched_barrier(0);
                fetch(bA, nn + 2);
                __builtin_amdgcn_sched_barrier(0);
                chain(bB, S[mm][nn + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (work) {
        full_product();
        CSTAMP(1);
        // ---- the pass, block by block
        const int I0 = a.c0 >> 4, I1 = (a.c1 + 15) >> 4;
        // what a block needs from memory (its entries of the matrix, of H, of the column priors, the known values) is requested
        // before the panel update of the block before it, which covers the trip
        double r_p0[2][4], r_pm[2][4], r_ob[2][4], r_h[2][4]; d4 r_m[2];
        auto load_raw = [&](int J) {
            const int cj = 16 * J + c, cjc = cj < D ? cj : D - 1;
            load_m(r_m, J);
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rowc[mm][e];
                    r_h[mm][e] = H[(size_t)r * D + cjc]; r_p0[mm][e] = pp[(size_t)cjc * rows + r];
                    r_pm[mm][e] = pm[(size_t)r * D + cjc]; r_ob[mm][e] = obs[(size_t)r * D + cjc];
                }
        };
        load_raw(I0 < nbl ? I0 : 0);
#pragma unroll 1
        for (int I = I0; I < I1 && I < nbl; ++I) {
            d4 Sc[2], Mc[2], Mo[2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) Mo[mm] = r_m[mm];
#pragma unroll
            for (int nn = 0; nn < BDT; ++nn)
                if (nn == I) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) Sc[mm] = S[mm][nn];
                }
            const int col = 16 * I + c, colc = col < D ? col : D - 1;
            const bool incol = col < D && col >= a.c0 && col < a.c1;
            const double gii = gd[col];
            // per entry: val = av + lv (hv - S)  with av = p0 pm / prec, lv = lam / prec  (qmu, gaussian.py:122-123: (p0 pm + lam (H - dot)) qcov);
            // a known entry (Gaussian.observe on a column, LDS_knowns_in_A.py:73-74; gaussian.py:125-134, :109-110) has av = its value,
            // lv = 0, a row beyond the matrix av = lv = 0
            double hv[2][4], av[2][4], lv[2][4];
            double nks = 0.0, mprod = 1.0;
            int esum = 0;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rowc[mm][e];
                    const double p0 = r_p0[mm][e], pmv = r_pm[mm][e], ob = r_ob[mm][e];
                    hv[mm][e] = r_h[mm][e];
                    const double prec = p0 + lam[mm][e] * gii;                          // qprec  gaussian.py:117
                    const double var = 1.0 / prec;                                      // qcov   gaussian.py:118-119
                    const bool known = ob == ob;
                    av[mm][e] = live[mm][e] ? (known ? ob : p0 * pmv * var) : 0.0;
                    lv[mm][e] = (live[mm][e] && !known) ? lam[mm][e] * var : 0.0;
                    const bool mine = live[mm][e] && incol;
                    // sum of log prec = log of the product of the mantissas + ln 2 x the sum of the exponents: one logarithm per lane and block
                    mprod *= mine ? __builtin_amdgcn_frexp_mant(prec) : 1.0;
                    esum += mine ? __builtin_amdgcn_frexp_exp(prec) : 0;
                    nks += (mine && known) ? 1.0 : 0.0;
                    if (mine) V[(size_t)colc * rows + r] = known ? 0.0 : var;
                    Mc[mm][e] = Mo[mm][e];
                }
            double lps = log(mprod) + 0.6931471805599453 * (double)esum;       // (a precision that is not positive: NaN, as the sum of logarithms)
            lps += __shfl_xor(lps, 16, 64); lps += __shfl_xor(lps, 32, 64);
            nks += __shfl_xor(nks, 16, 64); nks += __shfl_xor(nks, 32, 64);
            if (q == 0 && incol) { plp[wave * BDP + col] = lps; pkn[wave * BDP + col] = nks; }
            CSTAMP(2);
            const double* gcol = Gl + (size_t)(16 * I) * BDP;       // row 16 I + j: column (col) in even rows, (col ^ 16) in odd ones
            const int j0 = (a.c0 > 16 * I) ? a.c0 - 16 * I : 0;
            int j1 = a.c1 - 16 * I; if (j1 > 16) j1 = 16; if (j1 > D - 16 * I) j1 = D - 16 * I;
#pragma unroll 1
            for (int j = j0; j < j1; ++j) {
                const bool is = c == j;
                const double gjc = gcol[(size_t)j * BDP + (col ^ ((j & 1) << 4))];      // G[16 I + j][16 I + c], zero on the diagonal
                const int src = ((lane & 48) | j) << 2;
                // (in three rounds -- all changes, all cross-lane reads, all updates: entry by entry, each read waited for its own
                // trip through the LDS crossbar, eight trips a column)
                double dl[2][4], db[2][4];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double val = __builtin_fma(lv[mm][e], hv[mm][e] - Sc[mm][e], av[mm][e]);
                        dl[mm][e] = val - Mc[mm][e];                // (only lane c = j's is read below)
                        Mc[mm][e] = is ? val : Mc[mm][e];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        db[mm][e] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(dl[mm][e])), __builtin_amdgcn_ds_bpermute(src, __double2loint(dl[mm][e])));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Sc[mm][e] = __builtin_fma(db[mm][e], gjc, Sc[mm][e]);
            }
            CSTAMP(3);
            d4 dp[2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) dp[mm] = Mc[mm] - Mo[mm];
#pragma unroll
            for (int nn = 0; nn < BDT; ++nn)
                if (nn == I) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) S[mm][nn] = Sc[mm];
                }
            // the block's new values
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (live[mm][e] && incol) M[(size_t)rowc[mm][e] * D + col] = Mc[mm][e];
            __builtin_amdgcn_sched_barrier(0);
            load_raw(I + 1 < nbl ? I + 1 : I);
            __builtin_amdgcn_sched_barrier(0);
            rank16(dp, I, I, I + 1);        // the blocks still to come
            CSTAMP(4);
        }
    }
    __syncthreads();
    if (tid >= a.c0 && tid < a.c1 && tid < BDP) {
        const double lp = ((plp[tid] + plp[BDP + tid]) + plp[2 * BDP + tid]) + plp[3 * BDP + tid];
        const double nk = ((pkn[tid] + pkn[BDP + tid]) + pkn[2 * BDP + tid]) + pkn[3 * BDP + tid];
        if ((int)nk < rows) qld[tid] = 0.5 / (0.5 * lp);                                // quirk Q1, gaussian.py:120: of the whole precision
    }
    if (fuse1) {
        // res[k] = 1/2 own[k] + 1/2 (sum_ij M[k,i] G[i,j] M[k,j] + sum_i var_i[k] G[i,i]) - sum_i H[k,i] M[k,i]   (node.py:260-271)
        // S is formed afresh from the new matrix (behind the barrier above: this wavefront's own stores): the same sums whether the
        // columns were updated in this launch or in one before it
        CSTAMP(5);
        if (work && a.c0 < a.c1) full_product();
        CSTAMP(6);
        double rr[2][4];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double es = 0.0, hm = 0.0;
                if (work) {
#pragma unroll
                    for (int nn = 0; nn < BDT; ++nn) {
                        const int col = 16 * nn + c, colc = col < D ? col : D - 1;
                        const double g = gd[col];                               // zero beyond the matrix
                        const double mv = M[(size_t)rowc[mm][e] * D + colc];    // (this lane stored it, if it changed)
                        const double m = col < D ? mv : 0.0;
                        const double vi = V[(size_t)colc * rows + rowc[mm][e]], hi = H[(size_t)rowc[mm][e] * D + colc];
                        es += m * (S[mm][nn][e] + m * g) + vi * g;
                        hm += hi * m;
                    }
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { es += __shfl_xor(es, o, 64); hm += __shfl_xor(hm, o, 64); }
                const int r = rowc[mm][e];
                const double own = WHICH == 0 ? mo[MOM_DP(D, K) + r] : a.Syy[(size_t)n * K + r];
                rr[mm][e] = 0.5 * own + 0.5 * es - hm;
                if (work && live[mm][e] && c == 0) (WHICH == 0 ? a.resQ : a.resR)[(size_t)n * rows + r] = rr[mm][e];
            }
        if (a.fuse & 2) {
            const double* b0 = WHICH == 0 ? a.pri.Q_b0 : a.pri.R_b0;
            double* qb = (WHICH == 0 ? a.Q_b : a.R_b) + (size_t)n * rows;
            if (a.noise == PYVB_NOISE_GAMMA) {
                double t = 0.0;
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t += (work && live[mm][e] && c == 0) ? rr[mm][e] : 0.0;
                t = wave_sum(t);
                __syncthreads();
                if (lane == 0) red[wave] = t;
                __syncthreads();
                t = ((red[0] + red[1]) + red[2]) + red[3];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (work && live[mm][e] && c == 0) qb[rowc[mm][e]] = b0[0] + t;
            } else {
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (work && live[mm][e] && c == 0) qb[rowc[mm][e]] = b0[rowc[mm][e]] + rr[mm][e];
            }
        }
    }
#ifdef COLS_STAMP
    CSTAMP(7);
    if (blockIdx.x == 100 && tid == 0)
        printf("k_cols_big which %d: G to LDS %llu | M G_off %llu | block setup (loads, 1/x, log) %llu | 16 columns %llu | panel update %llu | to the barrier %llu | M G_off again %llu | residual %llu\n",
               WHICH, cs_acc[0], cs_acc[1], cs_acc[2], cs_acc[3], cs_acc[4], cs_acc[5], cs_acc[6], cs_acc[7]);
#endif
}

int launch_cols_big(pyvb_lds* h, int which, int c0, int c1, int fuse) {
    ParamArgs a = make_args(h);
    a.c0 = c0; a.c1 = c1; a.which0 = which == 1 ? 1 : 0; a.fuse = fuse;
    static const bool by_rows = [] { const char* e = getenv("PYVB_COLS_BIG"); return e && e[0] == 'r'; }();     // the kernel of round 3, for comparison
    const size_t lds_rows = ((size_t)BDP * BDP + 9 * BDP + 8) * sizeof(double);
    const size_t lds = ((size_t)BDP * BDP + 9 * BDP + 8 + 4 * 32 * CBP) * sizeof(double);
    if (!h->big_attr_cols) {
        HIPCHK(hipFuncSetAttribute((const void*)k_cols_big_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
        HIPCHK(hipFuncSetAttribute((const void*)k_cols_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->big_attr_cols = true;
    }
    TimedLaunch tl(h, PYVB_K_PARAMS);
    if (by_rows) hipLaunchKernelGGL(k_cols_big_rows, dim3(h->N, which == 2 ? 2 : 1), dim3(256), lds_rows, h->stream, a);
    else hipLaunchKernelGGL(k_cols_big, dim3(h->N, which == 2 ? 2 : 1), dim3(256), lds, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
