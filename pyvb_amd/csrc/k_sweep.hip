// k_sweep: one Gauss-Seidel sweep over all X_t of every replicate
//   [x.update() for x in Xs]  forward or reversed  (examples/Linear_Dynamic_System.py:70-73),
// i.e. Gaussian.update (gaussian.py:102-123) for 2T nodes with the messages of
// Multiplication.pass_up_m1_m2 / pass_down_Ex (node.py:182-242) folded into the gains of k_prep:
//   interior:  mu_t <- R mu_{t-dir} (just updated) + I mu_{t+dir} (previous sweep) + G y_t
//   forward:   R = F = Sigma <Q><A>,  I = B = Sigma <A>^T<Q>;   backward: R = B, I = F.
//
// Mapping to gfx950: one wavefront owns one replicate.  The interior time range is cut into 16
// segments which become the 16 columns of the B operand of v_mfma_f64_16x16x4_f64, so each step
// of the recurrence is a [DP x DP] x [DP x 16] product on the matrix cores.  The accumulator
// tile layout of that instruction (row = 4*reg + lane/16, col = lane%16) is exactly its B-operand
// layout for k-step = reg, so the new state feeds the next step straight from registers: no LDS,
// no cross-lane traffic.  R and I stay in VGPR/AGPRs as A operands for the whole sweep, G is read
// from LDS.  Segments other than the first do not know their starting state; because the
// recurrence is a contraction (spectral radius <= 1/2) they warm up from zero J steps early, with
// J from k_prep such that ||R^J|| <= 1e-18, or from t = first if that is nearer -- in which case
// the segment reproduces the sequential chain exactly.
#include "common.h"

struct SweepArgs {
    const double* Xold; double* Xnew; const double* Y; const double* gains; const int* warm;
    const double *A_mean, *C_mean;      // the boundary nodes read <A>, <C> themselves
    const double *QA, *RC;              // Wishart noise: <Q><A> [D][D] and <R><C> [K][D] per replicate (k_wishart.hip), else null
    double* trash;      // [N][256]: where lanes of inactive columns aim their (unconditional) stores
    double* U;          // [N][T][DP]: c_t = F mu_{t-1} + G y_t of the interior nodes, in accumulator order (see MODE)
    double* Sxx;        // [N][DP][DP]: sum over the interior nodes of mu_t mu_t^T, written by the MODE 2 sweep
    int N, T, D, K, dir;
    int W;              // wavefronts per replicate: each takes a contiguous part of the interior time range (grid.y)
    int keep_x;         // 0: the sweep that follows reads only c_t and the rows next to the far boundary, so the interior rows of Xnew are not written
    Layout L;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Xs[0].update() / Xs[T-1].update() (Gaussian.update gaussian.py:102-123 with the messages of
// Multiplication.pass_up_m1_m2 / pass_down_Ex, node.py:182-242), lane = row, one wavefront:
//   t = 0   : v = L0 m0 + <A>^T<Q> mu_1 + <C>^T<R> y_0         mu_0     = Sigma_0 v
//   t = T-1 : v = <Q><A> mu_{T-2}       + <C>^T<R> y_{T-1}     mu_{T-1} = Sigma_2 v
// i.e. qmu = qcov (sum of the m2 messages), as the reference has it.  nb(j) is entry j of the one
// neighbour's mean, vs 64 doubles of LDS.
// With Wishart noise the expected precisions are dense: QA = <Q><A>, RC = <R><C> take the place of the scaled rows
// (<A>^T<Q> = QA^T, <C>^T<R> = RC^T; the expectations are symmetric).
template <class NB>
__device__ __forceinline__ double boundary_update(bool first, const double* g, const Layout& L, const double* Am, const double* Cm,
                                                  int D, int K, int lane, NB nb, const double* y, double* vs,
                                                  const double* QA = nullptr, const double* RC = nullptr) {
    const double* qb = g + L.oqr;
    const double* rb = qb + 64;
    double v = 0.0;
    if (QA) {
        if (lane < D) {
            if (first) {
                v = g[L.ow0 + lane];
                for (int i = 0; i < D; ++i) v += QA[(size_t)i * D + lane] * nb(i);
            } else {
                for (int j = 0; j < D; ++j) v += QA[(size_t)lane * D + j] * nb(j);
            }
            for (int k = 0; k < K; ++k) v += RC[(size_t)k * D + lane] * y[k];
        }
    } else if (lane < D) {
        if (first) {
            v = g[L.ow0 + lane];
            for (int i = 0; i < D; ++i) v += Am[(size_t)i * D + lane] * (qb[i] * nb(i));
        } else {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += Am[(size_t)lane * D + j] * nb(j);
            v = qb[lane] * s;
        }
        for (int k = 0; k < K; ++k) v += Cm[(size_t)k * D + lane] * (rb[k] * y[k]);
    }
    vs[lane] = v;
    __syncthreads();
    const double* S = g + (first ? L.oS0 : L.oS2);
    double s = 0.0;
    if (lane < D)
        for (int j = 0; j < D; ++j) s += S[(size_t)j * L.DP + lane] * vs[j];      // Sigma is symmetric
    __syncthreads();
    return s;
}

// FULL: D == 16*DT and K == 16*KT (no padded rows/columns, 16-byte aligned rows)
// MODE: the parameters are frozen between the two sweeps of an iteration, and the backward update
//   mu_t <- B mu_{t+1}(new) + F mu_{t-1}(forward result) + G y_t
// contains c_t = F mu_{t-1} + G y_t, which the forward sweep has just formed on its way to mu_t
// (its recurrent term plus its observation term).  MODE 1 (forward) stores c_t, MODE 2 (the
// backward sweep that follows directly) reads it back instead of y_t and the forward state and
// runs ONE product per step instead of three; MODE 0 computes everything and keeps nothing.
// c rows are stored in accumulator order, [tile m][lane group q][register r], so a lane moves 32
// contiguous bytes per tile.
// SPLIT: several wavefronts per replicate (a.W > 1, few replicates); without it the part offsets fold to constants.
template <int DT, int KT, bool FULL, int MODE, bool SPLIT>
__global__ void __launch_bounds__(64) k_sweep(SweepArgs a) {
    constexpr int DS = 4 * DT, KS = 4 * KT, DP = 16 * DT;
    __shared__ double gl[MODE == 2 ? DP * 17 : DT * KS * 64];     // G as MFMA A operands; MODE 2: the state tile, transposed (below)
    __shared__ double xs[64];               // boundary state exchange
    __shared__ double vs[64];               // boundary_update scratch
    const int n = blockIdx.x, w = SPLIT ? blockIdx.y : 0, lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int T = a.T, D = a.D, K = a.K;
    const bool fwd = (a.dir == 0);
    const int sgn = fwd ? 1 : -1;
    const Layout& L = a.L;
    const double* g = a.gains + (size_t)n * L.gains_total;
    const double* Xo = a.Xold + (size_t)n * T * DP;      // state rows: stride DP, accumulator order
    double* Xn = a.Xnew + (size_t)n * T * DP;
    const double* Yn = a.Y + (size_t)n * T * K;

    // ---- operands that live for the whole sweep
    double rn[DT][DS], ip[DT][DS];
    {
        const double* Rn = g + (fwd ? L.oFn : L.oBn);
        const double* Ip = g + (fwd ? L.oBn : L.oFn);
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                rn[m][s] = Rn[(m * DS + s) * 64 + lane];
                ip[m][s] = (MODE == 2) ? 0.0 : Ip[(m * DS + s) * 64 + lane];
            }
        if constexpr (MODE != 2) {
            const double* Gp = g + L.oGp;
            for (int i = 0; i < DT * KS; ++i) gl[i * 64 + lane] = Gp[i * 64 + lane];
        }
    }

    // ---- the part of the interior this wavefront owns.  With few replicates the interior time range
    // (nodes 1 .. T-2, counted from the side the sweep starts at) is dealt out to a.W wavefronts per
    // replicate in contiguous parts of Lw nodes; every part is again cut into 16 segments.  All
    // segments but those that reach the chain's first node within J steps warm up from zero.
    const int Tint = T - 2;
    const int Lw = SPLIT ? ((((Tint + a.W - 1) / a.W) + 15) & ~15) : Tint;
    const int ow = SPLIT ? w * Lw : 0;                       // interior nodes before this part
    const int Tw = (Tint - ow < Lw) ? Tint - ow : Lw;        // interior nodes of this part (<= 0: none)
    const int J = a.warm[n * 2 + a.dir];

    // ---- first boundary node (t = 0 forward, T-1 backward): only the old neighbour.  Every wavefront
    // whose warm-up can reach it computes it; the first one stores it.
    const int t_first = fwd ? 0 : T - 1, t_last = fwd ? T - 1 : 0;
    const double* Am = a.A_mean + (size_t)n * D * D;
    const double* Cm = a.C_mean + (size_t)n * K * D;
    const double* QAm = a.QA ? a.QA + (size_t)n * D * D : nullptr;
    const double* RCm = a.RC ? a.RC + (size_t)n * K * D : nullptr;
    if (!SPLIT || ow <= J) {
        const double* xo = Xo + (size_t)(t_first + sgn) * DP;
        const double s = boundary_update(fwd, g, L, Am, Cm, D, K, lane, [&](int j) { return xo[xpos(j)]; },
                                         Yn + (size_t)t_first * K, vs, QAm, RCm);
        if (w == 0 && lane < DP) Xn[(size_t)t_first * DP + xpos(lane)] = (lane < D) ? s : 0.0;
        xs[lane] = (lane < D) ? s : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- interior: this part's nodes in 16 segments
    if (Tw > 0) {
        const int Lseg = (Tw + 15) >> 4;
        const int cL = c * Lseg;
        const int before = ow + cL;                              // interior nodes between the chain's start and this column
        const int jc = -(J < before ? J : before);               // first loop index of this column
        const int jstart = -((J < ow + 15 * Lseg) ? J : ow + 15 * Lseg);   // of the wave (its last column starts earliest)
        // starting state: the true boundary value when the warm-up reaches it, else zero
        d4 x[DT];
#pragma unroll
        for (int m = 0; m < DT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[m][r] = (jc == -before) ? xs[16 * m + 4 * r + q] : 0.0;

        // time index of this column at loop index j:  t = tbase + sgn * j
        const int tbase = fwd ? (1 + before) : (T - 2 - before);
        const int tsafe = fwd ? 1 : T - 2;      // an interior row that always exists: what inactive columns read
        auto active = [&](int j) { int tt = cL + j; return j >= jc && j < Lseg && tt < Tw; };
        // input registers: y_t and the old neighbour mean, as B operands in permuted k order.
        // Loads are unconditional and unmasked: an inactive column reads a valid row and computes
        // a value that the select after the step discards (MFMA columns do not mix), and padded
        // k positions meet zero matrix entries.  So a step's loads issue back to back with no
        // branch and no use until the next step.
        d2 yv[KS / 2];
        d4 mo[DT];
        auto load_y = [&](int j, d2* dst) {
            const double* p = Yn + (size_t)(active(j) ? tbase + sgn * j : tsafe) * K;
#pragma unroll
            for (int i = 0; i < KS / 2; ++i) {
                const int d0 = 8 * i + 2 * q;
                if constexpr (FULL) {
                    dst[i] = *reinterpret_cast<const d2*>(p + d0);
                } else {
                    dst[i][0] = p[d0 < K ? d0 : K - 1];
                    dst[i][1] = p[d0 + 1 < K ? d0 + 1 : K - 1];
                }
            }
        };
        auto load_o = [&](int j, d4* dst) {
            const double* p = Xo + (size_t)((active(j) ? tbase + sgn * j : tsafe) + sgn) * DP;
#pragma unroll
            for (int m = 0; m < DT; ++m) dst[m] = *reinterpret_cast<const d4*>(p + (m * 4 + q) * 4);
        };
        auto store_x = [&](double* out) {
#pragma unroll
            for (int m = 0; m < DT; ++m) *reinterpret_cast<d4*>(out + (m * 4 + q) * 4) = x[m];
        };
        // Stores are unconditional too (a branch around them would make the compiler's in-order
        // vmcnt bookkeeping conservative and stall on loads just issued): lanes with nothing to
        // store write to a per-replicate trash row.
        double* const trash = a.trash + (size_t)n * 256;
        double* out_pending = trash;
        // u_t rows (MODE 1 writes, MODE 2 reads), in accumulator order
        double* const Un = a.U + (size_t)n * T * DP;
        auto u_row = [&](int j) { return (active(j) && j >= 0) ? Un + (size_t)(tbase + sgn * j) * DP : trash; };
        auto load_u = [&](int j, d4* dst) {
            const double* p = Un + (size_t)(active(j) ? tbase + sgn * j : tsafe) * DP;
#pragma unroll
            for (int m = 0; m < DT; ++m) dst[m] = *reinterpret_cast<const d4*>(p + (m * 4 + q) * 4);
        };
        auto select_tile = [&](int m, bool act, const d4* acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[m][r] = act ? acc[m][r] : x[m][r];
        };
        auto pending_row = [&](int j, bool act) {
            return (act && j >= 0 && (a.keep_x || before + j == Tint - 1)) ? Xn + (size_t)(tbase + sgn * j) * DP : trash;
        };
        auto finish_step = [&](int j, const d4* acc) {
            const bool act = active(j);
#pragma unroll
            for (int m = 0; m < DT; ++m) select_tile(m, act, acc);
            out_pending = pending_row(j, act);
        };
        // acc += M v for a matrix resident in registers as A operands (R or I) and a state tile set v.  The MFMAs go
        // round robin over the row tiles' accumulators; dependent chains per accumulator, the next step's G y in the
        // shadow of the select and accumulators pinned in AGPRs were built and measured in round 2 -- same time, the
        // chip is at its power cap (profiles/r02/limits.txt; the variants: profiles/experiments/sweep_mfma_order.patch).
        auto mul_resident = [&](d4* acc, auto& M, const d4* v) {
#pragma unroll
            for (int s = 0; s < DS; ++s)
#pragma unroll
                for (int m = 0; m < DT; ++m) acc[m] = MFMA(M[m][s], v[s >> 2][s & 3], acc[m]);
        };
        if constexpr (MODE == 2) {
            // One product per step: the sweep is bound by the c_t stream, so c rows are fetched PF
            // steps ahead into a ring of registers (the loop is unrolled PF times so that every
            // load has a fixed destination).  The prologue issues the same load/store sequence as
            // PF loop steps (see below).
            constexpr int PF = 3;
            d4 ub[PF][DT];
            // The matrix pipe is idle about half of the time here (the sweep waits for HBM), so the sweep
            // also forms Sxx = sum_t mu_t mu_t^T over the interior nodes, which k_stats then does not have
            // to.  The 16 columns of a step are 16 time points: the sum over them is a product with the
            // time index as the MFMA k dimension, X X^T with X = [DP x 16].  Its operands want lane = row,
            // the state has lane = column: the tile goes through LDS (tl[row][17]: written in accumulator
            // order, read back in operand order one step later, while the step's own MFMAs run).
            double* const tl = gl;
            for (int i = lane; i < DP * 17; i += 64) tl[i] = 0.0;
            d4 sxx[DT][DT];
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int k = 0; k < DT; ++k) sxx[m][k] = d4{0.0, 0.0, 0.0, 0.0};
            auto tile_read = [&](double (*xt)[4]) {
#pragma unroll
                for (int m = 0; m < DT; ++m)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) xt[m][s4] = tl[(16 * m + c) * 17 + 4 * s4 + q];       // row 16m + lane%16, column 4 s + lane/16
            };
            auto tile_accumulate = [&](double (*xt)[4]) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int m = 0; m < DT; ++m)
#pragma unroll
                        for (int k = m; k < DT; ++k) sxx[m][k] = MFMA(xt[m][s4], xt[k][s4], sxx[m][k]);
            };
            auto tile_stage = [&](int m, bool keep) {       // tile m of the new states into LDS for the next step's tile_read
#pragma unroll
                for (int r = 0; r < 4; ++r) tl[(16 * m + 4 * r + q) * 17 + c] = keep ? x[m][r] : 0.0;
            };
#pragma unroll
            for (int p = 0; p < PF; ++p) { load_u(jstart + p, ub[p]); store_x(trash + 64 * p); }
            for (int j = jstart; j < Lseg; j += PF) {
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    d4 acc[DT];
                    double xt[DT][4];
                    tile_read(xt);                      // the previous step's states
#pragma unroll
                    for (int m = 0; m < DT; ++m) acc[m] = ub[p][m];
                    __builtin_amdgcn_sched_barrier(0);
                    load_u(j + p + PF, ub[p]);
                    store_x(out_pending);
                    __builtin_amdgcn_sched_barrier(0);
                    mul_resident(acc, rn, x);
                    const bool act = active(j + p);
                    const bool keep = act && j + p >= 0;      // warm-up steps and idle columns do not count
                    tile_accumulate(xt);
                    finish_step(j + p, acc);
#pragma unroll
                    for (int m = 0; m < DT; ++m) tile_stage(m, keep);
                }
            }
            {
                double xt[DT][4];
                tile_read(xt);
                tile_accumulate(xt);
            }
            store_x(out_pending);
            double* Sn = a.Sxx + ((size_t)n * (SPLIT ? a.W : 1) + w) * DP * DP;       // this wavefront's part of the sum (k_moments adds the parts)
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int k = m; k < DT; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 16 * m + 4 * e + q, col = 16 * k + c;     // accumulator element (row, col)
                        Sn[(size_t)row * DP + col] = sxx[m][k][e];
                        if (k > m) Sn[(size_t)col * DP + row] = sxx[m][k][e];
                    }
        } else {
        // Prologue: the same sequence of vector-memory operations as one loop iteration (loads, stores
        // -- into the trash row -- , loads), so that the compiler's in-order vmcnt bookkeeping at the
        // loop head sees the same number of younger operations on both incoming edges and the wait
        // for the first loads does not also cover the stores behind them.
        load_y(jstart, yv);
        store_x(trash);
        if constexpr (MODE == 1) store_x(trash + 64);  // stands for the c_t store
        load_o(jstart, mo);
        for (int j = jstart; j < Lseg; ++j) {
            d4 acc[DT];
            {
#pragma unroll
                for (int m = 0; m < DT; ++m) acc[m] = d4{0.0, 0.0, 0.0, 0.0};
                // G y_t.  The LDS offset is made opaque per iteration: G is loop invariant and the
                // compiler would otherwise hoist all of it into registers that R and I already fill.
                int goff = lane;
                asm volatile("" : "+v"(goff));
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int m = 0; m < DT; ++m) acc[m] = MFMA(gl[(m * KS + s) * 64 + goff], yv[s >> 1][s & 1], acc[m]);
                // Next step's y goes into the registers just consumed; it has the R and I blocks
                // (2/3 of a step) to arrive.  The store of the PREVIOUS step's state is issued right
                // behind those loads: vector-memory operations retire in order, so a wait for the
                // loads never includes the stores, and the stores get a whole step to drain before
                // the next loads queue up behind them.
                __builtin_amdgcn_sched_barrier(0);
                load_y(j + 1, yv);
                store_x(out_pending);
                __builtin_amdgcn_sched_barrier(0);
            }
            // R mu_{t-dir} (new): the previous accumulators are the B operands
            mul_resident(acc, rn, x);
            {
                if constexpr (MODE == 1) {      // c_t = G y_t + R mu_{t-1} for the backward sweep
                    __builtin_amdgcn_sched_barrier(0);
                    double* ur = u_row(j);
#pragma unroll
                    for (int m = 0; m < DT; ++m) *reinterpret_cast<d4*>(ur + (m * 4 + q) * 4) = acc[m];
                    __builtin_amdgcn_sched_barrier(0);
                }
                // I mu_{t+dir} (old)
                mul_resident(acc, ip, mo);
                __builtin_amdgcn_sched_barrier(0);
                load_o(j + 1, mo);
                __builtin_amdgcn_sched_barrier(0);
            }
            finish_step(j, acc);
        }
        store_x(out_pending);
        }
        // the column that holds the last interior node hands its state to the closing boundary step
        const int clast = (Tw - 1) / Lseg;
        __builtin_amdgcn_s_barrier();
        if (c == clast) {
#pragma unroll
            for (int m = 0; m < DT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[16 * m + 4 * r + q] = x[m][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    // ---- closing boundary node (t = T-1 forward, 0 backward): only the new neighbour; done by the
    // wavefront that owns the last interior node
    if (!SPLIT || w == ((Tint > 0) ? (Tint - 1) / Lw : 0)) {
        const double s = boundary_update(!fwd, g, L, Am, Cm, D, K, lane, [&](int j) { return xs[j]; },
                                         Yn + (size_t)t_last * K, vs, QAm, RCm);
        if (lane < DP) Xn[(size_t)t_last * DP + xpos(lane)] = (lane < D) ? s : 0.0;
    }
}

// Xs[t].update() alone, in place in the current buffer (neighbours as they are now).
struct StepArgs {
    double* X; const double* Y; const double* gains; const double *A_mean, *C_mean, *QA, *RC;
    int N, T, D, K, t;
    Layout L;
};

__global__ void __launch_bounds__(64) k_step(StepArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int T = a.T, D = a.D, K = a.K, t = a.t, DP = a.L.DP;
    const Layout& L = a.L;
    const double* g = a.gains + (size_t)n * L.gains_total;
    double* X = a.X + (size_t)n * T * DP;
    const double* y = a.Y + ((size_t)n * T + t) * K;
    const int cls = (t == 0) ? 0 : (t == T - 1 ? 2 : 1);
    const int row = lane % DP;
    if (cls != 1) {
        __shared__ double vs[64];
        const double* nbr = X + (size_t)(cls == 0 ? 1 : T - 2) * DP;
        const double s = boundary_update(cls == 0, g, L, a.A_mean + (size_t)n * D * D, a.C_mean + (size_t)n * K * D, D, K, lane,
                                         [&](int j) { return nbr[xpos(j)]; }, y, vs,
                                         a.QA ? a.QA + (size_t)n * D * D : nullptr, a.RC ? a.RC + (size_t)n * K * D : nullptr);
        if (lane < DP) X[(size_t)t * DP + xpos(lane)] = (lane < D) ? s : 0.0;
        return;
    }
    // interior node: rows of F, B, G picked out of the MFMA operand blocks (this is the rare path)
    const int DS = L.DS, KS = L.KS;
    double s = 0.0;
    for (int j = 0; j < D; ++j) s += g[L.oFn + pos_nat(row, j, DS)] * X[(size_t)(t - 1) * DP + xpos(j)];
    for (int j = 0; j < D; ++j) s += g[L.oBn + pos_nat(row, j, DS)] * X[(size_t)(t + 1) * DP + xpos(j)];
    for (int k = 0; k < K; ++k) s += g[L.oGp + pos_perm(row, k, KS)] * y[k];
    if (lane < DP) X[(size_t)t * DP + xpos(lane)] = (lane < D) ? s : 0.0;
}

template <int DT, int KT>
static int launch_sweep_t(pyvb_lds* h, const SweepArgs& a) {
    const bool full = h->D == 16 * DT && h->K == 16 * KT;
    const int mode = a.dir == PYVB_FORWARD ? 1 : (h->u_valid ? 2 : 0);
#define PYVB_SWEEP_CASE(F, M) do { if (h->W > 1) hipLaunchKernelGGL((k_sweep<DT, KT, F, M, true>), dim3(h->N, h->W), dim3(64), 0, h->stream, a); \
                                   else hipLaunchKernelGGL((k_sweep<DT, KT, F, M, false>), dim3(h->N), dim3(64), 0, h->stream, a); } while (0)
    if (full) { if (mode == 1) PYVB_SWEEP_CASE(true, 1); else if (mode == 2) PYVB_SWEEP_CASE(true, 2); else PYVB_SWEEP_CASE(true, 0); }
    else { if (mode == 1) PYVB_SWEEP_CASE(false, 1); else if (mode == 2) PYVB_SWEEP_CASE(false, 2); else PYVB_SWEEP_CASE(false, 0); }
#undef PYVB_SWEEP_CASE
    return PYVB_OK;
}

int launch_sweep(pyvb_lds* h, int direction, bool keep_x) {
    if (h->big) {       // k_big.hip: every state is written (keep_x ignored), no fused Sxx; the c_t cache as below
        int rc = launch_sweep_big(h, direction);
        if (rc) return rc;
        h->cur = 1 - h->cur;
        h->sxx_valid = false;
        h->u_valid = (direction == PYVB_FORWARD);
        return PYVB_OK;
    }
    SweepArgs a;
    a.keep_x = (keep_x || direction != PYVB_FORWARD) ? 1 : 0;
    a.W = h->W;
    a.Xold = h->X[h->cur]; a.Xnew = h->X[1 - h->cur]; a.Y = h->Y; a.gains = h->gains; a.warm = h->warm;
    a.trash = h->trash; a.U = h->U; a.Sxx = h->sxx; a.A_mean = h->A_mean; a.C_mean = h->C_mean;
    a.QA = h->dense ? h->QA : nullptr; a.RC = h->dense ? h->RC : nullptr;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.dir = direction; a.L = h->L;
    {
        TimedLaunch tl(h, direction == PYVB_FORWARD ? PYVB_K_SWEEP_FWD : PYVB_K_SWEEP_BWD);
        switch (h->L.DT * 10 + h->L.KT) {
            case 11: launch_sweep_t<1, 1>(h, a); break;
            case 12: launch_sweep_t<1, 2>(h, a); break;
            case 14: launch_sweep_t<1, 4>(h, a); break;
            case 21: launch_sweep_t<2, 1>(h, a); break;
            case 22: launch_sweep_t<2, 2>(h, a); break;
            case 24: launch_sweep_t<2, 4>(h, a); break;
            case 41: launch_sweep_t<4, 1>(h, a); break;
            case 42: launch_sweep_t<4, 2>(h, a); break;
            case 44: launch_sweep_t<4, 4>(h, a); break;
            default: pyvb_set_error("unsupported tile shape"); return PYVB_E_ARG;
        }
    }
    HIPCHK(hipGetLastError());
    h->cur = 1 - h->cur;
    // U holds c_t for the current gains and for THIS forward result; any other change of X drops it
    h->sxx_valid = direction == PYVB_BACKWARD && h->u_valid;     // that launch was the MODE 2 kernel: h->sxx is Sxx of the new states
    h->u_valid = (direction == PYVB_FORWARD);
    return PYVB_OK;
}

// API layout [N][T][D] <-> internal layout [N][T][DP] (accumulator order, zero padded)
struct PermArgs { const double* src; double* dst; size_t rows; int D, DP, to_internal; };
__global__ void __launch_bounds__(256) k_permute(PermArgs a) {
    // 64 threads per row, or 128 in the second shape class (DP = 128)
    const int per = a.DP > 64 ? 128 : 64;
    const size_t row = (size_t)blockIdx.x * (256 / per) + (threadIdx.x / per);
    const int d = threadIdx.x % per;
    if (row >= a.rows || d >= a.DP) return;
    if (a.to_internal) a.dst[row * a.DP + xpos(d)] = (d < a.D) ? a.src[row * a.D + d] : 0.0;
    else if (d < a.D) a.dst[row * a.D + d] = a.src[row * a.DP + xpos(d)];
}

int launch_permute(pyvb_lds* h, const double* src, double* dst, int to_internal) {
    PermArgs a; a.src = src; a.dst = dst; a.rows = (size_t)h->N * h->T; a.D = h->D; a.DP = h->L.DP; a.to_internal = to_internal;
    const size_t per_block = a.DP > 64 ? 2 : 4;
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((a.rows + per_block - 1) / per_block)), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_step(pyvb_lds* h, int t) {
    if (h->big) return launch_step_big(h, t);
    StepArgs a;
    a.X = h->X[h->cur]; a.Y = h->Y; a.gains = h->gains; a.A_mean = h->A_mean; a.C_mean = h->C_mean;
    a.QA = h->dense ? h->QA : nullptr; a.RC = h->dense ? h->RC : nullptr;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.t = t; a.L = h->L;
    TimedLaunch tl(h, PYVB_K_STEP);
    hipLaunchKernelGGL(k_step, dim3(h->N), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
