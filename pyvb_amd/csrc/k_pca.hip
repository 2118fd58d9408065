// VB-PCA with missing data (examples/PCA_missing_data.py:31-45), N data rows on one GPU.
//
// The reference updates 2N+q+2 node objects one by one.  Here the N rows are processed in 16-row
// tiles on the matrix cores (v_mfma_f64_16x16x4_f64), in two passes per iteration:
//   pass 1  [z.update() for z in Zs]     Z = X Gz^T - g0,  Gz = beta Sigma_z <W>^T     (gaussian.py:102-123,
//           node.py:203-227 for the message of Mult(W, z_n); all Z_n share Sigma_z)
//   pass 2  [x.update() for x in Xs]     missing entries <- <W><z_n> + <Mu>, variance 1/beta, known entries
//           pinned (gaussian.py:125-134 on a diagonal covariance), and, on the same tiles, the sums over n that
//           the W, Mu and Beta updates and the lower bound read: sum x z^T, sum z z^T, sum x, sum z, sum |x|^2.
// A wavefront of pass 2 owns two 16-column tiles of X: an imputed tile comes out of the MFMA in accumulator
// layout (row = 4*reg + lane/16, col = lane%16), which is exactly the A-operand layout of X^T for the product
// X^T Z, so the statistics take it straight from registers.
// Small single-workgroup kernels do the q x q / d-vector work (W columns, Sigma_z, Mu, Beta, lower bound).
#include "pca.h"
#include <cstdlib>
#include <type_traits>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define LN2PI 1.8378770664093453

struct PcaArgs {
    double* X; const unsigned char* M; double* xvar; const int* nmiss; double* Z;
    const double* Xdata; unsigned char* pinned;     // null unless some rows have not been conditioned on their observations yet
    double *W_mean, *W_var, *Mu_mean, *Mu_var, *Z_cov, *qld_W;
    const double *W_pm, *W_pp, *Mu_pm, *Mu_pp;
    double* scal; double* Gz; double* g0; double* sx_local;
    double* part; double* stats; double* aux; double* aux_tail; double* elbo; int* status;     // aux_tail: the [sum z | delta of sum x] vector
    long N, N_total, chunk_rows, lo_upd, hi_upd, n_part_missing, n_none_rows, row_offset;
    int d, q, DP, QP, DT, QT, nchunk, mode;
    int res_cached;     // PCA_ELBO: scal[PS_RES] holds the residual already
    int keep_z0;        // Z of global row 0 is final already (k_pca_pass12, k_pca_pass1)
    int z_deferred;     // PCA_PREPZ: also form sum z analytically;  PCA_X0: Z of row 0 is formed here
    int x0_prep;        // PCA_X0: first set the sum-z half of aux_tail: 1 from the statistics, 2 zero (a rank that does not own row 0); 0: the host has
    // lazy imputation (k_pca_pass12<.., LAZY>): the missing entries of rows [vin_lo, vin_hi) are not in X but stand for
    // <W>_x z_n + <Mu>_x with the Z in memory and the parameters as of the sweep that imputed them (W_x, Mu_x)
    double *W_x, *Mu_x; long vin_lo, vin_hi;
    int save_wx;        // k_pca_rowvar: copy W_mean, Mu_mean to W_x, Mu_x (the sweep before it imputed with them and stored nothing)
    PcaStatsLayout SL;
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// position of Gz[i][k] in the B-operand block of pass 1: k-step s = 4*(k/16) + k%4, lane group (k%16)/4,
// column tile t = i/16 -- a lane then reads 4 consecutive doubles of a row of X for 4 consecutive k-steps
__device__ __forceinline__ size_t gz_pos(int i, int k, int DS) {
    const int s = 4 * (k >> 4) + (k & 3), qk = (k & 15) >> 2;
    return ((size_t)((i >> 4) * DS + s) * 64) + qk * 16 + (i & 15);
}

// ---------------------------------------------------------------------------------------------------
// pass 1: Z <- X Gz^T - g0 for a chunk of rows; partial column sums of the new Z
// ---------------------------------------------------------------------------------------------------
__device__ static double bsum(double v, double* red);

template <int QT>
__global__ void __launch_bounds__(256) k_pca_pass1(PcaArgs a) {
    extern __shared__ double gl[];                       // Gz^T as B operands [QT][DS][64], shared by the 4 wavefronts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, qk = lane >> 4;
    const int DP = a.DP, QP = a.QP, DS = DP / 4;
    for (int i = threadIdx.x; i < QT * DS * 64; i += 256) gl[i] = a.Gz[i];
    __syncthreads();
    // the chunk's rows are split over the 4 wavefronts in multiples of 16
    const long c0 = (long)blockIdx.x * a.chunk_rows;
    const long c1 = (c0 + a.chunk_rows < a.N) ? c0 + a.chunk_rows : a.N;
    const long rpw = (((a.chunk_rows + 3) / 4) + 15) & ~15L;
    const long r0 = c0 + wave * rpw;
    const long r1 = (r0 + rpw < c1) ? r0 + rpw : c1;
    double g0c[QT], szc[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { g0c[t] = a.g0[16 * t + c]; szc[t] = 0.0; }
    for (long n0 = r0; n0 < r1; n0 += 16) {
        const long rowA = (n0 + c < a.N) ? n0 + c : a.N - 1;              // A operand: row = lane % 16
        const double* xr = a.X + rowA * DP + 4 * qk;
        d4 acc[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) acc[t] = d4{-g0c[t], -g0c[t], -g0c[t], -g0c[t]};
        for (int j = 0; j < DP / 16; ++j) {
            const d4 x4 = *reinterpret_cast<const d4*>(xr + 16 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t] = MFMA(x4[e], gl[(size_t)(t * DS + 4 * j + e) * 64 + lane], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = n0 + 4 * r + qk;                         // accumulator: row = 4*reg + lane/16
                if (row < r1 && !(a.keep_z0 && row == 0 && a.row_offset == 0)) { a.Z[row * QP + 16 * t + c] = acc[t][r]; szc[t] += acc[t][r]; }
            }
    }
    // column sums over the 4 lane groups -> part1[chunk][QP]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        double s = szc[t];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (qk == 0) a.aux[((size_t)blockIdx.x * 4 + wave) * QP + 16 * t + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// pass 2: impute + statistics.  One workgroup per row chunk; wavefront w owns columns [32w, 32w+32) of X as two
// INTERLEAVED 16-column tiles -- tile p holds columns 32w + 2c + p, c = lane % 16 -- so a lane's elements of the two
// accumulator tiles are two consecutive doubles of a row: X is read and written 16 B per lane, 256 contiguous bytes per
// row and wavefront, a whole 2 KB row per workgroup at a time (the column tiles of a row used to be separate
// wavefronts launched ~1000 chunks apart, 128 B per row each).  About 100 registers: four wavefronts per SIMD.
// ---------------------------------------------------------------------------------------------------
#define P2T 2           // tiles per wavefront
#ifndef P2_OCC
#define P2_OCC (QT == 1 ? (PIN ? 3 : 4) : 2)     // workgroups per CU the register budget is set for
#endif
// PIN: some rows still carry their initial mean at all entries and take their observations at their first update
// (pyvb_pca_set_unpinned_rows); its own instantiation, so that the usual one keeps its register budget
template <int QT, bool PIN>
__global__ void __launch_bounds__(256, P2_OCC) k_pca_pass2(PcaArgs a) {
    const int lane = threadIdx.x & 63, wave = 4 * blockIdx.y + (threadIdx.x >> 6), c = lane & 15, qk = lane >> 4;
    const int DP = a.DP, QP = a.QP, d = a.d, q = a.q;
    constexpr int QS = 4 * QT;
    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    const int col0 = 16 * P2T * wave + P2T * c;         // this lane's columns
    const bool colok = col0 < DP;                       // DP is a multiple of 16: both or none
    // B operands of the prediction  pred[n][dim] = sum_i Z[n][i] W[dim][i]:  B[k = i][col = c <-> dim = col0 + p]
    double wb[P2T][QS], mu[P2T];
#pragma unroll
    for (int p = 0; p < P2T; ++p) {
        const int dim = col0 + p;
#pragma unroll
        for (int s = 0; s < QS; ++s) { const int i = 4 * s + qk; wb[p][s] = (dim < d && i < q) ? a.W_mean[(size_t)dim * q + i] : 0.0; }
        mu[p] = dim < d ? a.Mu_mean[dim] : 0.0;
    }
    d4 sxz[P2T][QT], szz[QT][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int p = 0; p < P2T; ++p) sxz[p][t] = d4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < QT; ++u) szz[t][u] = d4{0, 0, 0, 0};
    }
    double sx[P2T], sxx = 0.0, sz[QT];
#pragma unroll
    for (int p = 0; p < P2T; ++p) sx[p] = 0.0;
#pragma unroll
    for (int t = 0; t < QT; ++t) sz[t] = 0.0;
    // chunk-relative 32-bit offsets (a chunk of X is a few MB); rows past the chunk read its last row and count for nothing;
    // a lane whose columns lie past DP (DP = 16 mod 32) reads column 0 and counts for nothing
    double* const Xc = a.X + (size_t)r0 * DP;
    const unsigned char* const Mc = a.M + (size_t)r0 * DP;
    const double* const Zc = a.Z + (size_t)r0 * QP;
    const unsigned nrows = (unsigned)(r1 - r0), colL = colok ? col0 : 0;
    const unsigned lo = a.lo_upd > r0 ? (unsigned)((a.lo_upd < r1 ? a.lo_upd : r1) - r0) : 0u;     // rows [lo, hi) of the chunk are updated
    const unsigned hi = a.hi_upd > r0 ? (unsigned)((a.hi_upd < r1 ? a.hi_upd : r1) - r0) : 0u;
    for (unsigned n0 = 0; n0 < nrows; n0 += 16) {
        // the rows of this lane's accumulator elements: n0 + 4r + qk
        d2 xo[4]; unsigned mk[4], xoff[4], rowc[4]; bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = n0 + 4 * r + qk;
            ok[r] = row < nrows;
            rowc[r] = ok[r] ? row : nrows - 1;
            xoff[r] = rowc[r] * DP + colL;
            xo[r] = *reinterpret_cast<const d2*>(Xc + xoff[r]);
            mk[r] = *reinterpret_cast<const unsigned short*>(Mc + xoff[r]);
        }
        // Z tile, A layout (row = lane%16, k = i) for the prediction ...
        const unsigned rowA = (n0 + c < nrows) ? n0 + c : nrows - 1;
        double za[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s) za[s] = Zc[rowA * QP + 4 * s + qk];
        // ... and B layout (k = row, col = i) for the statistics
        double zb[4][QT];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < QT; ++t) { const double v = Zc[rowc[s] * QP + 16 * t + c]; zb[s][t] = ok[s] ? v : 0.0; }
        d4 xn[P2T];                     // [p]: the tile of columns col0 + p; element r: row n0 + 4r + qk
#pragma unroll
        for (int p = 0; p < P2T; ++p) {
            d4 pred = d4{mu[p], mu[p], mu[p], mu[p]};
#pragma unroll
            for (int s = 0; s < QS; ++s) pred = MFMA(za[s], wb[p][s], pred);
            xn[p] = pred;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = n0 + 4 * r + qk;
            const bool live = ok[r] && colok;
            const bool rowupd = live && row >= lo && row < hi;
            d2 v = xo[r];
            bool any = false;
#pragma unroll
            for (int p = 0; p < P2T; ++p)
                if (rowupd && ((mk[r] >> (8 * p)) & 0xffu) == 0) { v[p] = xn[p][r]; any = true; }
            if (PIN && rowupd && !a.pinned[r0 + row]) {                     // first update of a row that still carries its initial
                const d2 dat = *reinterpret_cast<const d2*>(a.Xdata + (size_t)r0 * DP + xoff[r]);   // mean everywhere: the
#pragma unroll
                for (int p = 0; p < P2T; ++p)                                // observed entries take their data (gaussian.py:125-134)
                    if (((mk[r] >> (8 * p)) & 0xffu) != 0) v[p] = dat[p];
                any = true;
            }
            // whole rows go back, known entries with their own bits: full-line writes (a masked 8-byte store is a
            // read-modify-write at the memory side, measured 0.3 ms slower per pass)
#ifndef P2_GROUP
#define P2_GROUP 2
#endif
            bool st = rowupd;
            if (P2_GROUP > 0) {
                const unsigned long long b = __ballot(any);
                st = ((b >> (lane & ~(P2_GROUP - 1))) & ((1ull << P2_GROUP) - 1)) != 0;
            }
            if (st) *reinterpret_cast<d2*>(Xc + xoff[r]) = v;
#pragma unroll
            for (int p = 0; p < P2T; ++p) {
                const double x = live ? v[p] : 0.0;
                xn[p][r] = x; sx[p] += x; sxx += x * x;
            }
        }
        // statistics: an imputed tile is X^T's A operand (row = its column = lane%16, k = row index = lane/16)
#pragma unroll
        for (int p = 0; p < P2T; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < QT; ++t) sxz[p][t] = MFMA(xn[p][s], zb[s][t], sxz[p][t]);
        if (wave == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    sz[t] += zb[s][t];
#pragma unroll
                    for (int u = 0; u < QT; ++u) szz[t][u] = MFMA(zb[s][t], zb[s][u], szz[t][u]);
                }
        }
    }
    // partial sums of this chunk; accumulator row 4r + qk of tile p is column 32w + 2(4r + qk) + p
    double* P = a.part + (size_t)blockIdx.x * (a.SL.total + a.DT);
#pragma unroll
    for (int p = 0; p < P2T; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = 16 * P2T * wave + P2T * (4 * r + qk) + p;
            if (dim < DP) {
#pragma unroll
                for (int t = 0; t < QT; ++t) P[a.SL.oSxz + (size_t)dim * QP + 16 * t + c] = sxz[p][t][r];
            }
        }
#pragma unroll
    for (int p = 0; p < P2T; ++p) {
        double s = sx[p];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (qk == 0 && colok) P[a.SL.osx + col0 + p] = s;
    }
    sxx = wsum(sxx);
    if (lane == 0) P[a.SL.total + wave] = sxx;            // DT slots, one per wavefront here, the rest zero
    if (wave == 0 && lane >= (a.DT + P2T - 1) / P2T && lane < a.DT) P[a.SL.total + lane] = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            double s = sz[t];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (qk == 0) P[a.SL.osz + 16 * t + c] = s;
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[a.SL.oSzz + (size_t)(16 * t + 4 * r + qk) * QP + 16 * u + c] = szz[t][u][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// passes 1 and 2 in ONE sweep over X (the iteration's order W, Z, X_0, Mu, X_1.. allows it: the only thing between the Z and
// the X updates that needs a sum over all rows is Mu, and the sum it needs, sum_n <z_n>, is linear in sum_n <x_n>:
// Gz sum x - N g0, which k_pca_small(PCA_PREPZ) forms before any row is touched).  X is then read once per iteration.
//
// One workgroup per row chunk, wavefront w owns columns [32w, 32w + 32) of the chunk's rows, 16 rows (one MFMA tile) at a time:
//   1. its part of Z = X Gz^T over its 32 columns (X in A layout: lane = row, 4 consecutive columns per load) -> LDS
//   2. the parts are summed (thread = element), g0 subtracted, Z stored; both operand layouts of Z are read back from LDS
//   3. the prediction of its columns, TRANSPOSED product <W> Z^T with the rows of the <W> tile permuted (row m <-> dimension
//      4 (m % 4) + m / 4), so that the accumulator holds lane = row, register r = column 4 (lane / 16) + r: the layout X was
//      loaded in.  Missing entries take it; a lane stores its 32 bytes if one of them changed (whole sectors).
//   4. the tile goes through a per-wavefront LDS buffer into accumulator layout (row = 4 reg + lane / 16: the A operand of
//      X^T) for sum x z^T, sum x, sum |x|^2, exactly as k_pca_pass2.
// keep_z0: global row 0 has had its own update since Z was defined (Xs[0].update() comes before Mu in the crawl order): its z
// was stored by that step and is taken from Z instead of being recomputed from the changed row.
// ---------------------------------------------------------------------------------------------------
// A workgroup barrier that orders LDS traffic only: the wavefronts of k_pca_pass12 exchange data through LDS alone, and nothing here
// should make a wavefront wait for the tiles it has fetched ahead or for its write-back (a full __syncthreads() is a fence over
// global memory too; on gfx950 the compiler did not turn that into a wait for vmcnt in the builds inspected, and the measured time is
// the same -- the explicit form states the intent).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define P12_XS 36       // row stride (doubles) of the transposition buffer: 32 columns + pad, 32-byte aligned rows
// -DP12_STAMP (profiles/pca_stamps.py builds it as a variant library): wavefronts 0 and 4 of every workgroup add up, stage by
// stage, the time they spend (s_memtime ticks: core clock on this part) -- [s0, s1, wait at barrier A, s2, stage 3, stage 4 + fetch, wait at
// barrier B, whole kernel, steps]; pyvb_pca_debug_stamps copies the table out.  Not in the shipped library.
#ifdef P12_STAMP
__device__ unsigned long long g_p12_stamp[4096 * 2 * 12];
#define STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); st_acc[i] += _t - st_last; st_last = _t; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// LAZY (round 4): the imputed entries are NOT written back.  They are a function of what is stored anyway -- x_nk = (<W> z_n + <Mu>)_k
// with the z_n this sweep stores and the parameters it runs with -- so the next sweep recomputes them where it reads the row
// (stage 0 below: the row's previous z from Z, the previous parameters from W_x / Mu_x, the same transposed product as stage 3,
// the missing entries of the fetched tile take it) instead of this one writing 34 % of the 32-byte sectors of X (0.7 GB of the
// 0.87 GB the sweep wrote at N = 10^6 x 256, 10 % missing; storing the 8-byte entries alone is a read-modify-write at the memory
// side and slower still).  The host keeps track of which rows are in that state (pyvb_pca: xlazy, vlo, vhi) and k_pca_materialize
// -- the same product, bit for bit -- puts the entries into X when something other than the next sweep wants them.
template <int QT, bool PIN, bool LAZY = false>     // PIN as in k_pca_pass2
__global__ void __launch_bounds__(512) k_pca_pass12(PcaArgs a) {
    static_assert(!(LAZY && PIN), "rows that wait for their first update keep the write-back");
    extern __shared__ double lds12[];
    constexpr int RT = QT == 1 ? 2 : 1;         // tiles per step (register budget: 256 at two wavefronts per SIMD)
    constexpr int PF = 2 * RT;                  // register sets of X tiles: the tiles of this step and of the next
    const int nw = blockDim.x >> 6;
    double* zp = lds12;                                 // [RT][nw][QT][4][64] accumulator dumps of the partial products
    constexpr int ZBUF = RT * QT * 256, ZTBUF = RT * QT * 16 * 17;
    double* zf = zp + (size_t)RT * nw * QT * 256;       // 2 x [RT][QT][4][64] Z of a step's rows, accumulator layout
    double* zT = zf + 2 * ZBUF;                         // 2 x [RT][QP][17] the same, latent index major (the operand of the prediction)
    double* xt = zT + 2 * ZTBUF + (size_t)(threadIdx.x >> 6) * RT * 16 * P12_XS;      // [RT][16][P12_XS] per wavefront
    double* const mux = zT + 2 * ZTBUF + (size_t)nw * RT * 16 * P12_XS;               // LAZY: <Mu>_x [DP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, qk = lane >> 4;
    const int DP = a.DP, QP = a.QP, d = a.d, q = a.q, DS = DP / 4;
    constexpr int QS = 4 * QT;
    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    // ---- constant operands of this wavefront's two column tiles j = 2 wave + jj
    bool tok[2];
    double gz[2][4][QT], wa[2][QS];
    d4 mu4[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * wave + jj;
        tok[jj] = 16 * j < DP;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < QT; ++t) gz[jj][e][t] = tok[jj] ? a.Gz[(size_t)(t * DS + 4 * j + e) * 64 + lane] : 0.0;
        const int dimA = 16 * j + 4 * (c & 3) + (c >> 2);        // permuted row of the <W> tile (see 3. above)
#pragma unroll
        for (int s = 0; s < QS; ++s) { const int i = 4 * s + qk; wa[jj][s] = (dimA < d && i < q) ? a.W_mean[(size_t)dimA * q + i] : 0.0; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int dim = 16 * j + 4 * qk + e; mu4[jj][e] = dim < d ? a.Mu_mean[dim] : 0.0; }
    }
    // LAZY: the <W> tiles of the sweep that imputed the rows, same permuted layout as wa; <Mu>_x sits in LDS
    double wx[LAZY ? 2 : 1][LAZY ? QS : 1];
    if (LAZY) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int dimA = 16 * (2 * wave + jj) + 4 * (c & 3) + (c >> 2);
#pragma unroll
            for (int s = 0; s < QS; ++s) { const int i = 4 * s + qk; wx[jj][s] = (dimA < d && i < q) ? a.W_x[(size_t)dimA * q + i] : 0.0; }
        }
        for (int k = tid; k < DP; k += blockDim.x) mux[k] = k < d ? a.Mu_x[k] : 0.0;
    }
    const int col0 = 32 * wave + 2 * c;                 // this lane's two columns in the statistics (k_pca_pass2's interleaved tiles)
    const bool colok = col0 < DP;
    d4 sxz[P2T][QT], szz[QT][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int p = 0; p < P2T; ++p) sxz[p][t] = d4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < QT; ++u) szz[t][u] = d4{0, 0, 0, 0};
    }
    double sx[P2T] = {0.0, 0.0}, sxx = 0.0, sz[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) sz[t] = 0.0;
    double* const Xc = a.X + (size_t)r0 * DP;
    const unsigned char* const Mc = a.M + (size_t)r0 * DP;
    double* const Zc = a.Z + (size_t)r0 * QP;
    const unsigned nrows = (unsigned)(r1 - r0);
    const unsigned lo = a.lo_upd > r0 ? (unsigned)((a.lo_upd < r1 ? a.lo_upd : r1) - r0) : 0u;     // rows [lo, hi) of the chunk are updated
    const unsigned hi = a.hi_upd > r0 ? (unsigned)((a.hi_upd < r1 ? a.hi_upd : r1) - r0) : 0u;
    const bool z0_here = a.keep_z0 && a.row_offset == 0 && r0 == 0;
    const unsigned vlo = a.vin_lo > r0 ? (unsigned)((a.vin_lo < r1 ? a.vin_lo : r1) - r0) : 0u;    // LAZY: rows [vlo, vhi) of the chunk are to be recomputed
    const unsigned vhi = a.vin_hi > r0 ? (unsigned)((a.vin_hi < r1 ? a.vin_hi : r1) - r0) : 0u;
    double g0r[QT];         // step 2: a thread's elements all have latent index 16 t + tid % 16 (the workgroup is a multiple of 64 wide)
#pragma unroll
    for (int t = 0; t < QT; ++t) g0r[t] = a.g0[16 * t + (tid & 15)];
    // X in A layout for this wavefront: lane = row n0 + c, columns 32 wave + 16 jj + 4 qk .. + 3.  A tile of a workgroup is 36 KB
    // and a CU holds one workgroup: PF register sets per lane keep three tiles in flight behind the one being worked on
    // (with one, the sweep ran at the latency of a load per tile: 1.23 ms, 2.5 TB/s).
    d4 xq[PF][2]; unsigned mq[PF][2];
    auto fetch = [&](unsigned n0, d4 (&x)[2], unsigned (&m)[2]) {
        const unsigned row = (n0 + c < nrows) ? n0 + c : nrows - 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const unsigned off = row * DP + (tok[jj] ? 32 * wave + 16 * jj + 4 * qk : 0);
            x[jj] = *reinterpret_cast<const d4*>(Xc + off);
            m[jj] = *reinterpret_cast<const unsigned*>(Mc + off);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(16u * u < nrows ? 16u * u : 0u, xq[u], mq[u]);
    // LAZY, stage 0: the previous z of a step's rows as the operand of the transposed product (lane (qk, c): latent index 4 s + qk of
    // row c; every wavefront of the workgroup fetches the same 2 KB per tile, through L1), requested a step ahead -- right after
    // the step before has used the registers, two steps before stage 2 overwrites those rows of Z in place
    double zq[LAZY ? RT : 1][LAZY ? QS : 1];
    auto zfetch = [&](unsigned n0, double (&z)[LAZY ? QS : 1]) {
        const unsigned row = (n0 + c < nrows) ? n0 + c : nrows - 1;
#pragma unroll
        for (int s = 0; s < (LAZY ? QS : 1); ++s) z[s] = Zc[(size_t)row * QP + 4 * s + qk];
    };
    auto s0 = [&](unsigned nbase, auto U0) {
        constexpr int u0 = decltype(U0)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned n0 = nbase + 16 * rt, rowl = n0 + c;
            const bool rowv = rowl >= vlo && rowl < vhi;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                d4 pred = *reinterpret_cast<const d4*>(mux + (tok[jj] ? 32 * wave + 16 * jj + 4 * qk : 0));
#pragma unroll
                for (int s = 0; s < QS; ++s) pred = MFMA(wx[jj][s], zq[rt][s], pred);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rowv && ((mq[u0 + rt][jj] >> (8 * e)) & 0xffu) == 0) xq[u0 + rt][jj][e] = pred[e];
            }
            const unsigned nn = n0 + 16 * RT;
            zfetch(nn < nrows ? nn : n0, zq[rt]);
        }
    };
    if (LAZY) {
        __syncthreads();                                // mux
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) zfetch(16u * rt < nrows ? 16u * rt : 0u, zq[rt]);
    }
#ifdef P12_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    // A step takes RT tiles through the four stages between two barriers.  Measured with cycle stamps per tile (one tile per step, d = 256,
    // q = 16): stage 1 560, stage 2 790 (four of the eight wavefronts), stage 3 1900-2300, stage 4 1100-1500 cycles, against 3 x 1024
    // for the 48 MFMAs the two wavefronts of a SIMD issue: the sweep runs at the pace of its dependent chains (MFMA -> select -> LDS
    // -> MFMA), not of HBM; the write-back of the imputed entries (34 % of the 32-byte sectors) costs 0.22 of the 0.94 ms.
    // Software pipeline over the steps: stage 1 of step n + 1, a barrier, then stage 2 of step n + 1 (wavefronts 0..3) next to stages 3
    // and 4 of step n (all wavefronts), a barrier.  Z of a step lives in one of two LDS buffers (zsel).  (Stage 2 in a barrier interval
    // of its own, with half of the wavefronts idle, was a sixth of a step.)
    auto s1 = [&](unsigned nbase, auto U0) {
        constexpr int u0 = decltype(U0)::value;
        // ---- 1. this wavefront's part of Z
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            d4 zacc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                zacc[t] = d4{0, 0, 0, 0};
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) zacc[t] = MFMA(xq[u0 + rt][jj][e], gz[jj][e][t], zacc[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) zp[((size_t)((rt * nw + wave) * QT + t) * 4 + r) * 64 + lane] = zacc[t][r];
            }
        }
    };
    auto s2 = [&](unsigned nbase, int zsel) {
        double* const zfw = zf + zsel * ZBUF;
        double* const zTw = zT + zsel * ZTBUF;
        // ---- 2. sum of the parts: thread = (rt, t, r, lane) element of the accumulator layout
        for (int el = tid; el < RT * QT * 256; el += blockDim.x) {
            const int rt = el / (QT * 256), e2 = el % (QT * 256);
            const int t = e2 >> 8, r = (e2 >> 6) & 3, ln = e2 & 63, cc = ln & 15, qq = ln >> 4;
            double s = -(QT == 1 || t == 0 ? g0r[0] : g0r[QT - 1]);
            double pz[8];                               // all parts requested at once (a dependent loop costs eight LDS round trips)
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) pz[w2] = zp[((size_t)((rt * nw + (w2 < nw ? w2 : 0)) * QT + t) * 4 + r) * 64 + ln];
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) s += w2 < nw ? pz[w2] : 0.0;
            const unsigned row = nbase + 16 * rt + 4 * r + qq;
            if (row < nrows) {
                if (z0_here && row == 0) s = Zc[16 * t + cc];
                else Zc[(size_t)row * QP + 16 * t + cc] = s;
            }
            zfw[el] = row < nrows ? s : 0.0;
            zTw[(size_t)rt * QT * 16 * 17 + (16 * t + cc) * 17 + 4 * r + qq] = s;
        }
    };
    auto back = [&](unsigned nbase, auto U0, int zsel) {
        constexpr int u0 = decltype(U0)::value;
        const double* const zfr = zf + zsel * ZBUF;
        const double* const zTr = zT + zsel * ZTBUF;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned n0 = nbase + 16 * rt;
            d4 (&xa)[2] = xq[u0 + rt];
            unsigned (&ma)[2] = mq[u0 + rt];
            double za[QS];
#pragma unroll
            for (int s = 0; s < QS; ++s) za[s] = zTr[(size_t)rt * QT * 16 * 17 + (4 * s + qk) * 17 + c];
            // ---- 3. prediction, imputation, write-back (all tiles of the step, then stage 4 for all of them: the chains of
            // different tiles are independent and fill each other's waits)
            const unsigned rowl = n0 + c;
            const bool rowupd = rowl < nrows && rowl >= lo && rowl < hi;
            double* const xtr = xt + rt * 16 * P12_XS;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                d4 pred = mu4[jj];
#pragma unroll
                for (int s = 0; s < QS; ++s) pred = MFMA(wa[jj][s], za[s], pred);
                d4 v = xa[jj];
                bool any = false;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rowupd && ((ma[jj] >> (8 * e)) & 0xffu) == 0) { v[e] = pred[e]; any = true; }
                if (PIN && rowupd && tok[jj] && !a.pinned[r0 + rowl]) {          // first update of a row that still carries its initial mean
                    const d4 dat = *reinterpret_cast<const d4*>(a.Xdata + (size_t)(r0 + rowl) * DP + 32 * wave + 16 * jj + 4 * qk);
#pragma unroll
                    for (int e = 0; e < 4; ++e)                                  // everywhere: the observed entries take their data
                        if (((ma[jj] >> (8 * e)) & 0xffu) != 0) v[e] = dat[e];
                    any = true;
                }
                if (!LAZY && any && tok[jj]) *reinterpret_cast<d4*>(Xc + (size_t)rowl * DP + 32 * wave + 16 * jj + 4 * qk) = v;
                *reinterpret_cast<d4*>(xtr + c * P12_XS + 16 * jj + 4 * qk) = v;
            }
            // This register set is free from here on (stage 4 works on the LDS copy): the tile PF further on is requested NOW, in
            // front of the fences below, which no global load may be moved across -- at the end of stage 4, where this call used
            // to stand, the rows had a barrier's time to arrive before stage 1 of the next step asked for them (cycle stamps,
            // profiles/r04/pca_stamps_*.txt: 3000 of a step's 16000 cycles were that wait)
            const unsigned nn = n0 + 16 * PF;
            fetch(nn < nrows ? nn : n0, xa, ma);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned n0 = nbase + 16 * rt;
            double* const xtr = xt + rt * 16 * P12_XS;
            double zb[4][QT];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < QT; ++t) zb[s][t] = zfr[(size_t)(rt * QT + t) * 256 + s * 64 + lane];
            // ---- 4. statistics on the transposed tile: element r of tile p = row n0 + 4 r + qk, column col0 + p
            d4 xn[P2T];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const d2 v = *reinterpret_cast<const d2*>(xtr + (4 * r + qk) * P12_XS + 2 * c);
                const bool live = (n0 + 4 * r + qk < nrows) && colok;
#pragma unroll
                for (int p = 0; p < P2T; ++p) {
                    const double x = live ? v[p] : 0.0;
                    xn[p][r] = x; sx[p] += x; sxx += x * x;
                }
            }
#pragma unroll
            for (int p = 0; p < P2T; ++p)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < QT; ++t) sxz[p][t] = MFMA(xn[p][s], zb[s][t], sxz[p][t]);
            if (wave == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sz[t] += zb[s][t];
#pragma unroll
                        for (int u = 0; u < QT; ++u) szz[t][u] = MFMA(zb[s][t], zb[s][u], szz[t][u]);
                    }
            }
        }
        STAMP(5);
    };
    // prologue: Z of the first step
    if (LAZY) s0(0u, std::integral_constant<int, 0>{});
    s1(0u, std::integral_constant<int, 0>{});
    lds_barrier();
    s2(0u, 0);
    lds_barrier();
    STAMP(9);
    int zsel = 0;
    for (unsigned base = 0; base < nrows; base += 16 * PF) {
        {   // step at base (register sets 0..RT-1); the next one, if any, at base + 16 RT (sets RT..)
            const bool more = base + 16 * RT < nrows;
            if (LAZY && more) s0(base + 16 * RT, std::integral_constant<int, RT>{});
            STAMP(0);
            if (more) s1(base + 16 * RT, std::integral_constant<int, RT>{});
            STAMP(1);
            lds_barrier();
            STAMP(2);
            if (more) s2(base + 16 * RT, zsel ^ 1);
            STAMP(3);
            back(base, std::integral_constant<int, 0>{}, zsel);
            lds_barrier();
            STAMP(6);
            zsel ^= 1;
            if (!more) break;
        }
        {
            const unsigned b2 = base + 16 * RT;
            const bool more = b2 + 16 * RT < nrows;
            if (LAZY && more) s0(b2 + 16 * RT, std::integral_constant<int, 0>{});
            STAMP(0);
            if (more) s1(b2 + 16 * RT, std::integral_constant<int, 0>{});
            STAMP(1);
            lds_barrier();
            STAMP(2);
            if (more) s2(b2 + 16 * RT, zsel ^ 1);
            STAMP(3);
            back(b2, std::integral_constant<int, RT>{}, zsel);
            lds_barrier();
            STAMP(6);
            zsel ^= 1;
            if (!more) break;
        }
    }
#ifdef P12_STAMP
    if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_p12_stamp + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 12;
        for (int i = 0; i < 10; ++i) o[i] = st_acc[i];
        o[10] = __builtin_amdgcn_s_memtime() - st_begin;
        o[11] = (nrows + 16 * RT - 1) / (16 * RT);
    }
#endif
    // ---- partial sums of this chunk, laid out as k_pca_pass2's
    double* P = a.part + (size_t)blockIdx.x * (a.SL.total + a.DT);
#pragma unroll
    for (int p = 0; p < P2T; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = 16 * P2T * wave + P2T * (4 * r + qk) + p;
            if (dim < DP) {
#pragma unroll
                for (int t = 0; t < QT; ++t) P[a.SL.oSxz + (size_t)dim * QP + 16 * t + c] = sxz[p][t][r];
            }
        }
#pragma unroll
    for (int p = 0; p < P2T; ++p) {
        double s = sx[p];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (qk == 0 && colok) P[a.SL.osx + col0 + p] = s;
    }
    sxx = wsum(sxx);
    if (lane == 0) P[a.SL.total + wave] = sxx;            // DT slots, one per wavefront here, the rest zero
    if (wave == 0 && lane >= nw && lane < a.DT) P[a.SL.total + lane] = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            double s = sz[t];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (qk == 0) P[a.SL.osz + 16 * t + c] = s;
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[a.SL.oSzz + (size_t)(16 * t + 4 * r + qk) * QP + 16 * u + c] = szz[t][u][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The sweep of an iteration, second form (round 4): a wavefront owns WHOLE ROWS.
//
// k_pca_pass12 gives every wavefront of a workgroup 32 columns of the same 16 rows, so the eight partial products of a tile's Z
// meet in LDS and the workgroup walks in lock step, two barriers a step: its cycle stamps (profiles/r04/pca_stamps_*.txt) show a
// step of 32 rows taking 16 000 cycles of which the matrix pipe is busy 6 000 -- the rest is barrier skew, LDS round trips
// and the instruction streams of eight wavefronts that all do the same thing at the same time.  Here a wavefront takes a
// 16-row tile through all of it alone:
//   A. over the row's eight 32-column blocks: [the missing entries recomputed from the row's previous z, as LAZY above]
//      and Z += X Gz^T, one accumulator for the whole row -- no partial sums, no exchange;
//   B. z = Z - g0 stored; both operand forms of it: the accumulator IS the B operand of sum x z^T (k = row), the transposed one
//      (k = latent index, for the prediction) goes through 2 KB of this wavefront's own LDS;
//   C. over the blocks again (X a second time, from L2): prediction, imputation (nothing stored: LAZY), the 32-column piece
//      through this wavefront's transposition buffer into accumulator layout, sum x z^T / sum x / sum |x|^2.
// No workgroup barrier between the prologue and the final reduction: the four wavefronts of a workgroup -- one per SIMD, so that
// the 128 accumulator registers of the whole 256 x 16 sum x z^T fit beside the working set -- drift apart.  The operand tables
// (Gz^T, <W>, <W>_x as MFMA operands for all 16 column tiles: 96 KB) are in LDS, shared.  X is fetched through a ring of four
// register sets, three 32-column blocks ahead, across phases and tiles.
// Same operands, same products, same order of the sums over a tile's rows as k_pca_pass12; what differs is the order in
// which the tiles of a chunk are added up (per wavefront, then the wavefronts in order), i.e. rounding.
// MEASURED (N = 10^6 x 256, q = 16, one MI355X, profiles/r04/pca_rows_vs_columns.txt): correct at once (all PCA tests, 1.4e-13
// against the oracle) and SLOWER -- 1.11-1.19 ms per iteration against 1.03-1.04 for k_pca_pass12<.., LAZY>.  The matrix pipe needs
// 256 MFMAs x 64 cycles = 16 400 cycles per tile and SIMD (0.43 ms for the sweep); the tile takes 36 000.  With one wavefront per SIMD
// (256 + 172 registers: two do not fit) nobody fills the waits of the ~2 500 other instructions of a tile -- LDS operand reads, the
// byte-mask selects, 670 moves between the accumulator and the vector registers -- and the compiler does not interleave them with
// the MFMAs over blocks of this size (operand reads batched per block: 1.19 -> 1.11; bounds tests removed so that a phase is one
// basic block: 1.14).  Kept as PYVB_PCA_SWEEP=rows, with the fixtures run through it, as the record of the attempt; the sweep in
// use is k_pca_pass12<.., LAZY>.
#define PR_NW 4
__device__ __forceinline__ void wave_lds_sync() {
    // the LDS operations of one wavefront execute in order: what has to be kept in order is the compiler (no wait, in particular
    // none for the rows requested ahead)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// FULL: all 16 column tiles exist (d > 240): no bounds tests on them, so that a tile's phase is one basic block for the scheduler
template <bool FULL>
__global__ void __launch_bounds__(64 * PR_NW) k_pca_rows(PcaArgs a) {
    extern __shared__ double ldsr[];
    constexpr int QS = 4, NB = 8, RING = 4, AHEAD = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, qk = lane >> 4;
    const int DP = a.DP, d = a.d, q = a.q, DT = a.DT;
    constexpr int QP = 16;
    double* const gzL = ldsr;                               // [DT][4][64]   B operands of Z = X Gz^T (a.Gz as it is)
    double* const waL = gzL + (size_t)DT * 256;             // [DT][4][64]   <W> tiles, rows permuted (k_pca_pass12, 3.)
    double* const wxL = waL + (size_t)DT * 256;             // the same of <W>_x
    double* const muL = wxL + (size_t)DT * 256;             // [DP]
    double* const mxL = muL + DP;                           // [DP]
    double* const g0L = mxL + DP;                           // [16]
    double* const zTw = g0L + 16 + (size_t)wave * (16 * 17 + 16 * P12_XS);     // this wavefront's: z transposed [16][17]
    double* const xtw = zTw + 16 * 17;                                           // and the 32-column piece [16][P12_XS]
    for (int i = tid; i < DT * 256; i += 64 * PR_NW) {
        gzL[i] = a.Gz[i];
        const int j = i >> 8, s4 = (i >> 6) & 3, ln = i & 63, cc = ln & 15, qq = ln >> 4;
        const int dimA = 16 * j + 4 * (cc & 3) + (cc >> 2), li = 4 * s4 + qq;
        const bool in = dimA < d && li < q;
        waL[i] = in ? a.W_mean[(size_t)dimA * q + li] : 0.0;
        wxL[i] = in ? a.W_x[(size_t)dimA * q + li] : 0.0;
    }
    for (int k = tid; k < DP; k += 64 * PR_NW) { muL[k] = k < d ? a.Mu_mean[k] : 0.0; mxL[k] = k < d ? a.Mu_x[k] : 0.0; }
    if (tid < 16) g0L[tid] = a.g0[tid];
    __syncthreads();

    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    const unsigned nrows = (unsigned)(r1 - r0);
    const unsigned ntiles = (nrows + 15) >> 4;
    double* const Xc = a.X + (size_t)r0 * DP;
    const unsigned char* const Mc = a.M + (size_t)r0 * DP;
    double* const Zc = a.Z + (size_t)r0 * QP;
    const unsigned lo = a.lo_upd > r0 ? (unsigned)((a.lo_upd < r1 ? a.lo_upd : r1) - r0) : 0u;
    const unsigned hi = a.hi_upd > r0 ? (unsigned)((a.hi_upd < r1 ? a.hi_upd : r1) - r0) : 0u;
    const unsigned vlo = a.vin_lo > r0 ? (unsigned)((a.vin_lo < r1 ? a.vin_lo : r1) - r0) : 0u;
    const unsigned vhi = a.vin_hi > r0 ? (unsigned)((a.vin_hi < r1 ? a.vin_hi : r1) - r0) : 0u;
    const bool z0_here = a.keep_z0 && a.row_offset == 0 && r0 == 0;
    const double g0c = g0L[c];
    const double z0keep = z0_here ? Zc[c] : 0.0;        // read once, here: a load inside the loop would make every tile wait for all rows in flight

    d4 sxz[NB][2], szz = d4{0, 0, 0, 0};
    double sx[NB][2], sxx = 0.0, sz = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p) { sxz[b][p] = d4{0, 0, 0, 0}; sx[b][p] = 0.0; }

    // ---- the stream of X: position pos of a tile = block pos % 8 (0..7 phase A, 8..15 phase C); the set of position pos is pos % 4
    d4 rx[RING][2]; unsigned rm[RING][2];
    auto fetch = [&](d4 (&x)[2], unsigned (&m)[2], unsigned tile, int b) {
        const unsigned rw = 16u * tile + c;
        const unsigned row = rw < nrows ? rw : nrows - 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * b + jj;
            const unsigned off = row * DP + ((FULL || j < DT) ? 16 * j + 4 * qk : 0);
            x[jj] = *reinterpret_cast<const d4*>(Xc + off);
            m[jj] = *reinterpret_cast<const unsigned*>(Mc + off);
        }
    };
    auto zfetch = [&](double (&z)[QS], unsigned tile) {
        const unsigned rw = 16u * tile + c;
        const unsigned row = rw < nrows ? rw : nrows - 1;
#pragma unroll
        for (int s4 = 0; s4 < QS; ++s4) z[s4] = Zc[(size_t)row * QP + 4 * s4 + qk];
    };
    double zq[QS];                      // previous z of the tile's rows, operand of the recomputation
    unsigned t = wave;
    if (t < ntiles) {
        // in the order of the loop (z first, the blocks behind it): the counter of outstanding loads the compiler waits on at the
        // loop head is the minimum over the way in and the way round
        zfetch(zq, t);
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) fetch(rx[u], rm[u], t, u);
    }
    for (; t < ntiles; t += PR_NW) {
        const unsigned n0 = 16u * t, rowl = n0 + c;
        const unsigned tnext = (t + PR_NW < ntiles) ? t + PR_NW : t;
        const bool rowv = rowl >= vlo && rowl < vhi;
        const bool rowupd = rowl < nrows && rowl >= lo && rowl < hi;
        d4 zacc = d4{0, 0, 0, 0};
        d4 z;
        double za[QS];
#pragma unroll
        for (int pos = 0; pos < 2 * NB; ++pos) {
            const int b = pos & (NB - 1);
            d4 (&xa)[2] = rx[pos & (RING - 1)];
            unsigned (&ma)[2] = rm[pos & (RING - 1)];
            if (pos < NB) {
                // ---- A: recompute what the sweep before left unstored, then this block's part of Z.  All operands of the block
                // are requested from LDS first (one wait per block, not one per product: a single wavefront per SIMD has nobody
                // to hide an LDS round trip behind), the ones of the recomputation ahead of the ones of the Z product.
                if (FULL || 2 * b < DT) {
                    double wxo[2][QS], gzo[2][4]; d4 mxo[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = (FULL || 2 * b + jj < DT) ? 2 * b + jj : 2 * b;
                        mxo[jj] = *reinterpret_cast<const d4*>(mxL + 16 * j + 4 * qk);
#pragma unroll
                        for (int s4 = 0; s4 < QS; ++s4) wxo[jj][s4] = wxL[(j * 4 + s4) * 64 + lane];
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = (FULL || 2 * b + jj < DT) ? 2 * b + jj : 2 * b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) gzo[jj][e] = gzL[(j * 4 + e) * 64 + lane];
                    }
                    d4 pred[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {            // the two recomputations side by side (independent chains)
                        pred[jj] = mxo[jj];
                    }
#pragma unroll
                    for (int s4 = 0; s4 < QS; ++s4)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) pred[jj] = MFMA(wxo[jj][s4], zq[s4], pred[jj]);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (FULL || 2 * b + jj < DT) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (rowv && ((ma[jj] >> (8 * e)) & 0xffu) == 0) xa[jj][e] = pred[jj][e];
#pragma unroll
                            for (int e = 0; e < 4; ++e) zacc = MFMA(xa[jj][e], gzo[jj][e], zacc);
                        }
                    }
                }
            } else {
                // ---- C: prediction, imputation, statistics of this block
                if (FULL || 2 * b < DT) {
                    double wao[2][QS]; d4 muo[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = (FULL || 2 * b + jj < DT) ? 2 * b + jj : 2 * b;
                        muo[jj] = *reinterpret_cast<const d4*>(muL + 16 * j + 4 * qk);
#pragma unroll
                        for (int s4 = 0; s4 < QS; ++s4) wao[jj][s4] = waL[(j * 4 + s4) * 64 + lane];
                    }
                    d4 pred[2] = {muo[0], muo[1]};
#pragma unroll
                    for (int s4 = 0; s4 < QS; ++s4)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) pred[jj] = MFMA(wao[jj][s4], za[s4], pred[jj]);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        d4 v = xa[jj];
                        if (FULL || 2 * b + jj < DT) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (rowupd && ((ma[jj] >> (8 * e)) & 0xffu) == 0) v[e] = pred[jj][e];
                        } else v = d4{0, 0, 0, 0};
                        *reinterpret_cast<d4*>(xtw + c * P12_XS + 16 * jj + 4 * qk) = v;
                    }
                    wave_lds_sync();
                    d4 xn[2];
                    d2 v2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v2[r] = *reinterpret_cast<const d2*>(xtw + (4 * r + qk) * P12_XS + 2 * c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool live = (n0 + 4 * r + qk < nrows) && (FULL || 32 * b + 2 * c < DP);
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const double x = live ? v2[r][p] : 0.0;
                            xn[p][r] = x; sx[b][p] += x; sxx += x * x;
                        }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int p = 0; p < 2; ++p) sxz[b][p] = MFMA(xn[p][s4], z[s4], sxz[b][p]);
                    wave_lds_sync();                    // the piece has been read: the next block may overwrite it
                }
            }
            // the set of three positions on: same tile while it lasts, then the wavefront's next tile
            {
                const int np = pos + AHEAD;
                if (np < 2 * NB) fetch(rx[np & (RING - 1)], rm[np & (RING - 1)], t, np & (NB - 1));
                else fetch(rx[np & (RING - 1)], rm[np & (RING - 1)], tnext, np - 2 * NB);
            }
            if (pos == NB - 1) {
                // ---- B: the tile's z (accumulator layout: lane (qk, c), register r = row 4 r + qk, latent index c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned row = n0 + 4 * r + qk;
                    double v = zacc[r] - g0c;
                    const bool keep = z0_here && row == 0;      // z_0 was stored by Xs[0].update() itself (keep_z0)
                    v = keep ? z0keep : v;
                    if (row < nrows) { if (!keep) Zc[(size_t)row * QP + c] = v; }
                    else v = 0.0;
                    z[r] = v; sz += v;
                    zTw[c * 17 + 4 * r + qk] = v;
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) szz = MFMA(z[s4], z[s4], szz);
                zfetch(zq, tnext);                      // rows of this wavefront's next tile: nobody writes them before it does
                wave_lds_sync();
#pragma unroll
                for (int s4 = 0; s4 < QS; ++s4) za[s4] = zTw[(4 * s4 + qk) * 17 + c];
            }
        }
    }
    // ---- the wavefronts' sums into one per workgroup, in wavefront order (LDS: the tables are done with)
    __syncthreads();
    double* const red = ldsr;           // [Sxz 256 x 16 | sx 256 | Szz 16 x 16 | sz 16 | sxx]
    for (int w = 0; w < PR_NW; ++w) {
        if (wave == w) {
            const bool first = w == 0;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dim = 32 * b + 2 * (4 * r + qk) + p;
                        double* o = red + (size_t)dim * QP + c;
                        *o = first ? sxz[b][p][r] : *o + sxz[b][p][r];
                    }
                    double s = sx[b][p];
                    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                    if (qk == 0) { double* o = red + 4096 + 32 * b + 2 * c + p; *o = first ? s : *o + s; }
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) { double* o = red + 4352 + (4 * r + qk) * QP + c; *o = first ? szz[r] : *o + szz[r]; }
            double s = sz;
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (qk == 0) { double* o = red + 4608 + c; *o = first ? s : *o + s; }
            const double sq = wsum(sxx);
            if (lane == 0) { double* o = red + 4624; *o = first ? sq : *o + sq; }
        }
        __syncthreads();
    }
    double* P = a.part + (size_t)blockIdx.x * (a.SL.total + a.DT);
    for (int i = tid; i < DP * QP; i += 64 * PR_NW) P[a.SL.oSxz + i] = red[i];
    for (int i = tid; i < DP; i += 64 * PR_NW) P[a.SL.osx + i] = red[4096 + i];
    for (int i = tid; i < QP * QP; i += 64 * PR_NW) P[a.SL.oSzz + i] = red[4352 + i];
    if (tid < QP) P[a.SL.osz + tid] = red[4608 + tid];
    if (tid < a.DT) P[a.SL.total + tid] = tid == 0 ? red[4624] : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// Third form (round 4): a PAIR of wavefronts owns a 16-row tile, each wavefront one half of the columns.
//
// k_pca_rows showed that the barrier-free ownership computes the right thing but cannot keep the matrix pipe fed with one wavefront
// per SIMD -- and one is all that fits while a wavefront carries the whole 256 x 16 sum x z^T (128 accumulator registers).  Split
// over two wavefronts (columns [128 h, 128 h + 128), h = 0, 1) everything per wavefront halves -- accumulators, operands, the ring
// of X -- and eight wavefronts fit a CU, two to a SIMD (252 registers), four pairs walking through their tiles independently.
// What the pair has to exchange per tile is ONE 16 x 16 partial of Z: each wavefront leaves its partial in LDS, raises a flag, waits
// for the partner's (a spin on an LDS word, both are resident: no deadlock), and both form z = part_0 + part_1 - g0 in that order.
// No workgroup barrier between the prologue and the final reduction.  Everything else as k_pca_rows (phases A, B, C; LAZY).
#define PP_NW 8
#ifndef PP_ROT
#define PP_ROT 0
#endif
template <bool FULL>
__global__ void __launch_bounds__(64 * PP_NW) k_pca_pairs(PcaArgs a) {
    extern __shared__ double ldsr[];
    constexpr int QS = 4, NB = 4, RING = 4, AHEAD = 3;      // NB: the 32-column blocks of a wavefront (half a row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, qk = lane >> 4;
    const int pair = wave >> 1, half = wave & 1;
    const int DP = a.DP, d = a.d, q = a.q, DT = a.DT;
    constexpr int QP = 16;
    double* const gzL = ldsr;                               // [DT][4][64]   B operands of Z = X Gz^T (a.Gz as it is)
    double* const waL = gzL + (size_t)DT * 256;             // [DT][4][64]   <W> tiles, rows permuted (k_pca_pass12, 3.)
    double* const wxL = waL + (size_t)DT * 256;             // the same of <W>_x
    double* const muL = wxL + (size_t)DT * 256;             // [DP]
    double* const mxL = muL + DP;                           // [DP]
    double* const g0L = mxL + DP;                           // [16]
    double* const xtw = g0L + 32 + (size_t)wave * (16 * P12_XS);                // this wavefront's: the 32-column piece [16][P12_XS] ...
    double* const zTw = xtw;                                                     // ... and, before phase C needs it, z transposed [16][17]
    double* const xch = g0L + 32 + (size_t)PP_NW * (16 * P12_XS);               // [wave][4][64] the partial of Z a wavefront hands to its partner
    int* const flags = reinterpret_cast<int*>(xch + (size_t)PP_NW * 256);        // [wave][2]: tiles whose partial is ready / whose partner partial has been read
    if (tid < 2 * PP_NW) flags[tid] = 0;
#ifdef P12_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = st_t0;
#endif
    for (int i = tid; i < DT * 256; i += 64 * PP_NW) {
        gzL[i] = a.Gz[i];
        const int j = i >> 8, s4 = (i >> 6) & 3, ln = i & 63, cc = ln & 15, qq = ln >> 4;
        const int dimA = 16 * j + 4 * (cc & 3) + (cc >> 2), li = 4 * s4 + qq;
        const bool in = dimA < d && li < q;
        waL[i] = in ? a.W_mean[(size_t)dimA * q + li] : 0.0;
        wxL[i] = in ? a.W_x[(size_t)dimA * q + li] : 0.0;
    }
    for (int k = tid; k < DP; k += 64 * PP_NW) { muL[k] = k < d ? a.Mu_mean[k] : 0.0; mxL[k] = k < d ? a.Mu_x[k] : 0.0; }
    if (tid < 16) g0L[tid] = a.g0[tid];
    __syncthreads();

    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    const unsigned nrows = (unsigned)(r1 - r0);
    const unsigned ntiles = (nrows + 15) >> 4;
    double* const Xc = a.X + (size_t)r0 * DP;
    const unsigned char* const Mc = a.M + (size_t)r0 * DP;
    double* const Zc = a.Z + (size_t)r0 * QP;
    const unsigned lo = a.lo_upd > r0 ? (unsigned)((a.lo_upd < r1 ? a.lo_upd : r1) - r0) : 0u;
    const unsigned hi = a.hi_upd > r0 ? (unsigned)((a.hi_upd < r1 ? a.hi_upd : r1) - r0) : 0u;
    const unsigned vlo = a.vin_lo > r0 ? (unsigned)((a.vin_lo < r1 ? a.vin_lo : r1) - r0) : 0u;
    const unsigned vhi = a.vin_hi > r0 ? (unsigned)((a.vin_hi < r1 ? a.vin_hi : r1) - r0) : 0u;
    const bool z0_here = a.keep_z0 && a.row_offset == 0 && r0 == 0;
    // g0 and the kept z_0 stay in LDS (g0L[16..32)): as registers they were spilled, and a reload from scratch inside the loop is
    // a memory load that every row in flight has to land in front of
    if (wave == 0 && lane < 16) g0L[16 + lane] = z0_here ? Zc[lane] : 0.0;
    __syncthreads();

    d4 sxz[NB][2], szz = d4{0, 0, 0, 0};
    double sx[NB][2], sxx = 0.0, sz = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p) { sxz[b][p] = d4{0, 0, 0, 0}; sx[b][p] = 0.0; }

    // ---- the stream of X: position pos of a tile = block pos % 8 (0..7 phase A, 8..15 phase C); the set of position pos is pos % 4
    d4 rx[RING][2]; unsigned rm[RING][2];
    // workgroup b walks its chunk from tile (rot) on, round: the chunks lie a fixed 8 MB apart, and 256 workgroups in step at that
    // stride would keep asking the same few HBM channels
    const unsigned rot = ntiles ? (unsigned)((blockIdx.x * PP_ROT) % ntiles) : 0u;
    auto TI = [&](unsigned tl) { const unsigned u = tl + rot; return u >= ntiles ? u - ntiles : u; };
    auto fetch = [&](d4 (&x)[2], unsigned (&m)[2], unsigned tile, int b) {
        const unsigned rw = 16u * TI(tile) + c;
        const unsigned row = rw < nrows ? rw : nrows - 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * (NB * half + b) + jj;
            const unsigned off = row * DP + ((FULL || j < DT) ? 16 * j + 4 * qk : 0);
            x[jj] = *reinterpret_cast<const d4*>(Xc + off);
            m[jj] = *reinterpret_cast<const unsigned*>(Mc + off);
        }
    };
    auto zfetch = [&](double (&z)[QS], unsigned tile) {
        const unsigned rw = 16u * TI(tile) + c;
        const unsigned row = rw < nrows ? rw : nrows - 1;
#pragma unroll
        for (int s4 = 0; s4 < QS; ++s4) z[s4] = Zc[(size_t)row * QP + 4 * s4 + qk];
    };
    double zq[QS];                      // previous z of the tile's rows, operand of the recomputation
    unsigned t = pair;
    int seq = 0;                        // tiles of this pair so far
    if (t < ntiles) {
        // in the order of the loop (z first, the blocks behind it): the counter of outstanding loads the compiler waits on at the
        // loop head is the minimum over the way in and the way round
        zfetch(zq, t);
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) fetch(rx[u], rm[u], t, u);
    }
    for (; t < ntiles; t += PP_NW / 2) {
        const unsigned n0 = 16u * TI(t), rowl = n0 + c;
        const unsigned tnext = (t + PP_NW / 2 < ntiles) ? t + PP_NW / 2 : t;
        const bool rowv = rowl >= vlo && rowl < vhi;
        const bool rowupd = rowl < nrows && rowl >= lo && rowl < hi;
        d4 zacc = d4{0, 0, 0, 0};
        d4 z;
        double za[QS];
#pragma unroll
        for (int pos = 0; pos < 2 * NB; ++pos) {
            const int b = pos & (NB - 1);
            const int gb = NB * half + b;           // the block's place in the row
            d4 (&xa)[2] = rx[pos & (RING - 1)];
            unsigned (&ma)[2] = rm[pos & (RING - 1)];
            if (pos < NB) {
                // ---- A: recompute what the sweep before left unstored, then this block's part of Z.  All operands of the block
                // are requested from LDS first (one wait per block, not one per product: a single wavefront per SIMD has nobody
                // to hide an LDS round trip behind), the ones of the recomputation ahead of the ones of the Z product.
                // (two wavefronts share the SIMD here: the operands of one 16-column tile at a time, the other wavefront covers the
                // LDS round trip -- and the registers of a whole block's operands are not there to be had)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (FULL || 2 * gb + jj < DT) {
                        const int j = 2 * gb + jj;
                        double wxo[QS], gzo[4];
                        d4 pred = *reinterpret_cast<const d4*>(mxL + 16 * j + 4 * qk);
#pragma unroll
                        for (int s4 = 0; s4 < QS; ++s4) wxo[s4] = wxL[(j * 4 + s4) * 64 + lane];
#pragma unroll
                        for (int e = 0; e < 4; ++e) gzo[e] = gzL[(j * 4 + e) * 64 + lane];
#pragma unroll
                        for (int s4 = 0; s4 < QS; ++s4) pred = MFMA(wxo[s4], zq[s4], pred);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (rowv && ((ma[jj] >> (8 * e)) & 0xffu) == 0) xa[jj][e] = pred[e];
#pragma unroll
                        for (int e = 0; e < 4; ++e) zacc = MFMA(xa[jj][e], gzo[e], zacc);
                    }
                }
            } else {
                // ---- C: prediction, imputation, statistics of this block
                if (FULL || 2 * gb < DT) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        d4 v = xa[jj];
                        if (FULL || 2 * gb + jj < DT) {
                            const int j = 2 * gb + jj;
                            double wao[QS];
                            d4 pred = *reinterpret_cast<const d4*>(muL + 16 * j + 4 * qk);
#pragma unroll
                            for (int s4 = 0; s4 < QS; ++s4) wao[s4] = waL[(j * 4 + s4) * 64 + lane];
#pragma unroll
                            for (int s4 = 0; s4 < QS; ++s4) pred = MFMA(wao[s4], za[s4], pred);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (rowupd && ((ma[jj] >> (8 * e)) & 0xffu) == 0) v[e] = pred[e];
                        } else v = d4{0, 0, 0, 0};
                        *reinterpret_cast<d4*>(xtw + c * P12_XS + 16 * jj + 4 * qk) = v;
                    }
                    wave_lds_sync();
                    d4 xn[2];
                    d2 v2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v2[r] = *reinterpret_cast<const d2*>(xtw + (4 * r + qk) * P12_XS + 2 * c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool live = (n0 + 4 * r + qk < nrows) && (FULL || 32 * gb + 2 * c < DP);
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const double x = live ? v2[r][p] : 0.0;
                            xn[p][r] = x; sx[b][p] += x; sxx += x * x;
                        }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int p = 0; p < 2; ++p) sxz[b][p] = MFMA(xn[p][s4], z[s4], sxz[b][p]);
                    wave_lds_sync();                    // the piece has been read: the next block may overwrite it
                }
            }
            // the set of three positions on: same tile while it lasts, then the wavefront's next tile
            // (pinned between two scheduling barriers: at the register limit the compiler otherwise sinks these loads to just before
            // their use three positions later -- the listing showed waits for all but two or three loads in flight -- and the ring
            // hides nothing)
            {
                const int np = pos + AHEAD;
                __builtin_amdgcn_sched_barrier(0);
                if (np < 2 * NB) fetch(rx[np & (RING - 1)], rm[np & (RING - 1)], t, np & (NB - 1));
                else fetch(rx[np & (RING - 1)], rm[np & (RING - 1)], tnext, np - 2 * NB);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pos == NB - 1) {
                // ---- the pair's hand-over: my partial into my slot (once the partner has read the one before), flag up; the partner's
                ++seq;
                STAMP(0);
                while (__hip_atomic_load(&flags[2 * (wave ^ 1) + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq - 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[(size_t)wave * 256 + r * 64 + lane] = zacc[r];
                __hip_atomic_store(&flags[2 * wave], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(&flags[2 * (wave ^ 1)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq) __builtin_amdgcn_s_sleep(1);
                STAMP(1);
                d4 other;
#pragma unroll
                for (int r = 0; r < 4; ++r) other[r] = xch[(size_t)(wave ^ 1) * 256 + r * 64 + lane];
                __hip_atomic_store(&flags[2 * wave + 1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                // ---- B: the tile's z (accumulator layout: lane (qk, c), register r = row 4 r + qk, latent index c), the same bits in
                // both wavefronts: part_0 + part_1 - g0
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned row = n0 + 4 * r + qk;
                    double v = ((half == 0 ? zacc[r] : other[r]) + (half == 0 ? other[r] : zacc[r])) - g0L[c];
                    const bool keep = z0_here && row == 0;      // z_0 was stored by Xs[0].update() itself (keep_z0)
                    if (keep) v = g0L[16 + c];
                    if (row < nrows) { if (!keep && half == 0) Zc[(size_t)row * QP + c] = v; }
                    else v = 0.0;
                    z[r] = v;
                    if (half == 0) sz += v;
                    zTw[c * 17 + 4 * r + qk] = v;
                }
                if (half == 0) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) szz = MFMA(z[s4], z[s4], szz);
                }
                zfetch(zq, tnext);                      // rows of this pair's next tile: nobody writes them before wavefront 0 of the pair does
                wave_lds_sync();
#pragma unroll
                for (int s4 = 0; s4 < QS; ++s4) za[s4] = zTw[(4 * s4 + qk) * 17 + c];
                wave_lds_sync();                        // (z transposed shares its place with the 32-column piece of phase C)
                STAMP(2);
            }
        }
        STAMP(3);
    }
#ifdef P12_STAMP
    if (lane == 0 && wave < 2) {        // phase A | the wait for the partner | phase B | phase C | - | whole kernel in shader ticks, in 100 MHz ticks | tiles
        unsigned long long* o = g_p12_stamp + ((size_t)blockIdx.x * 2 + wave) * 12;
        for (int i = 0; i < 4; ++i) o[i] = st_acc[i];
        o[10] = __builtin_amdgcn_s_memtime() - st_t0; o[9] = __builtin_amdgcn_s_memrealtime() - st_r0; o[11] = (unsigned long long)seq;
    }
#endif
    // ---- the wavefronts' sums into one per workgroup, in wavefront order (LDS: the tables are done with)
    __syncthreads();
    double* const red = ldsr;           // [Sxz 256 x 16 | sx 256 | Szz 16 x 16 | sz 16 | sxx]
    for (int w = 0; w < PP_NW; ++w) {
        if (wave == w) {
            const bool first = w < 2;               // the first wavefront on each half of the columns
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dim = 32 * (NB * half + b) + 2 * (4 * r + qk) + p;
                        double* o = red + (size_t)dim * QP + c;
                        *o = first ? sxz[b][p][r] : *o + sxz[b][p][r];
                    }
                    double s = sx[b][p];
                    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                    if (qk == 0) { double* o = red + 4096 + 32 * (NB * half + b) + 2 * c + p; *o = first ? s : *o + s; }
                }
            if (half == 0) {                        // z belongs to the pair: its first wavefront has summed it
                const bool f0 = w == 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { double* o = red + 4352 + (4 * r + qk) * QP + c; *o = f0 ? szz[r] : *o + szz[r]; }
                double s = sz;
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                if (qk == 0) { double* o = red + 4608 + c; *o = f0 ? s : *o + s; }
            }
            const double sq = wsum(sxx);
            if (lane == 0) { double* o = red + 4624; *o = (w == 0) ? sq : *o + sq; }
        }
        __syncthreads();
    }
    double* P = a.part + (size_t)blockIdx.x * (a.SL.total + a.DT);
    for (int i = tid; i < DP * QP; i += 64 * PP_NW) P[a.SL.oSxz + i] = red[i];
    for (int i = tid; i < DP; i += 64 * PP_NW) P[a.SL.osx + i] = red[4096 + i];
    for (int i = tid; i < QP * QP; i += 64 * PP_NW) P[a.SL.oSzz + i] = red[4352 + i];
    if (tid < QP) P[a.SL.osz + tid] = red[4608 + tid];
    if (tid < a.DT) P[a.SL.total + tid] = tid == 0 ? red[4624] : 0.0;
}

// The missing entries of rows [vin_lo, vin_hi) into X: <W>_x z_n + <Mu>_x, formed exactly as stage 0 of k_pca_pass12<.., LAZY> forms
// them (same operands, same chain of MFMAs: bit for bit what the next sweep would have used).  Same mapping: a workgroup per row
// chunk, wavefront w on columns [32 w, 32 w + 32).
template <int QT>
__global__ void __launch_bounds__(512) k_pca_materialize(PcaArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, qk = lane >> 4;
    const int DP = a.DP, QP = a.QP, d = a.d, q = a.q;
    constexpr int QS = 4 * QT;
    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    const long lo = a.vin_lo > r0 ? a.vin_lo : r0, hi = a.vin_hi < r1 ? a.vin_hi : r1;
    if (lo >= hi) return;
    double wx[2][QS]; d4 mu4[2]; bool tok[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * wave + jj;
        tok[jj] = 16 * j < DP;
        const int dimA = 16 * j + 4 * (c & 3) + (c >> 2);
#pragma unroll
        for (int s = 0; s < QS; ++s) { const int i = 4 * s + qk; wx[jj][s] = (dimA < d && i < q) ? a.W_x[(size_t)dimA * q + i] : 0.0; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int dim = 16 * j + 4 * qk + e; mu4[jj][e] = dim < d ? a.Mu_x[dim] : 0.0; }
    }
    for (long n0 = r0 + ((lo - r0) & ~15L); n0 < hi; n0 += 16) {
        const long row = n0 + c;
        const long rowc = row < r1 ? row : r1 - 1;
        double z[QS];
#pragma unroll
        for (int s = 0; s < QS; ++s) z[s] = a.Z[(size_t)rowc * QP + 4 * s + qk];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (!tok[jj]) continue;
            const size_t off = (size_t)rowc * DP + 32 * wave + 16 * jj + 4 * qk;
            const unsigned m = *reinterpret_cast<const unsigned*>(a.M + off);
            d4 pred = mu4[jj];
#pragma unroll
            for (int s = 0; s < QS; ++s) pred = MFMA(wx[jj][s], z[s], pred);
            if (row >= lo && row < hi && m != 0x01010101u) {
                d4 v = *reinterpret_cast<const d4*>(a.X + off);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((m >> (8 * e)) & 0xffu) == 0) v[e] = pred[e];
                *reinterpret_cast<d4*>(a.X + off) = v;
            }
        }
    }
}

// the variances of the missing entries of the chunk's rows (1 / <beta> for the rows being updated) and their sums:
// sum_n #missing_n var_n, and sum over partially observed rows of #missing_n log var_n.  One workgroup per chunk.
// (256 or 1024 threads: with a chunk per CU a thread of 256 walked 15 rows one dependent trip to memory after the other, 24 us)
__global__ void __launch_bounds__(1024) k_pca_rowvar(PcaArgs a) {
    __shared__ double red[3][16];
    const int nt = blockDim.x;
    if (a.save_wx && blockIdx.x == 0) {         // the parameters the sweep before this launch imputed with (k_pca_pass12<.., LAZY>)
        for (int i = threadIdx.x; i < a.d * a.q; i += nt) a.W_x[i] = a.W_mean[i];
        for (int i = threadIdx.x; i < a.d; i += nt) a.Mu_x[i] = a.Mu_mean[i];
    }
    const long r0 = (long)blockIdx.x * a.chunk_rows;
    const long r1 = (r0 + a.chunk_rows < a.N) ? r0 + a.chunk_rows : a.N;
    const double var_new = a.scal[PS_BETA_B] / a.scal[PS_BETA_A];
    const double log_new = log(var_new), qld_new = 0.5 / (0.5 * a.d * log(1.0 / var_new));
    double sxv = 0.0, slv = 0.0, sql = 0.0;
    for (long row = r0 + threadIdx.x; row < r1; row += nt) {
        const int nm = a.nmiss[row];
        int cnt = nm;                       // entries of the row that carry this variance
        const bool upd = nm > 0 && row >= a.lo_upd && row < a.hi_upd;
        double v = var_new;
        if (upd) {
            a.xvar[row] = v;
            if (a.pinned) a.pinned[row] = 1;    // pass 2 (same stream, before this kernel) has conditioned the row just now
        } else {
            v = a.xvar[row];
            if (a.pinned && !a.pinned[row]) cnt = a.d;          // not updated yet: the initial covariance v I on all entries
        }
        sxv += cnt * v;
        if (nm > 0 && nm < a.d) slv += nm * (upd ? log_new : log(v));
        if (nm == a.d) sql += upd ? qld_new : 0.5 / (0.5 * a.d * log(1.0 / v));     // a latent row: qprec = I / v (gaussian.py:120, quirk Q1)
    }
    double* P = a.part + (size_t)blockIdx.x * (a.SL.total + a.DT);
    sxv = wsum(sxv); slv = wsum(slv); sql = wsum(sql);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sxv; red[1][threadIdx.x >> 6] = slv; red[2][threadIdx.x >> 6] = sql; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w = 0; w < nt / 64; ++w) t += red[threadIdx.x][w];
        P[threadIdx.x == 0 ? a.SL.osxv : (threadIdx.x == 1 ? a.SL.oslv : a.SL.osql)] = t;
    }
}

// sum the per-chunk partials in two deterministic stages (no atomics: results must not depend on timing)
//   what = 0: full statistics of pass 2 -> stats;  what = 1: [sum z (QP)] of pass 1 -> tail of aux
//   stage 0: slice y of the chunks -> red2[y][idx];  stage 1: the PCA_RED slices -> destination
__global__ void __launch_bounds__(256) k_pca_reduce(PcaArgs a, int what, int stage, double* red2) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = what == 1 ? (size_t)a.QP : a.SL.total;
    if (idx >= n) return;
    if (stage == 1) {
        double s = 0.0;
        for (int y = 0; y < PCA_RED; ++y) s += red2[(size_t)y * n + idx];
        if (what == 1) a.aux[(size_t)4 * a.nchunk * a.QP + idx] = s; else a.stats[idx] = s;
        if (what == 0 && idx >= a.SL.osx && idx < a.SL.osx + (size_t)a.DP) a.sx_local[idx - a.SL.osx] = s;      // this rank's own sum of x (before the all-reduce)
        return;
    }
    const int y = blockIdx.y;
    const int nch = what == 1 ? 4 * a.nchunk : a.nchunk;
    const int per = (nch + PCA_RED - 1) / PCA_RED;
    const int c0 = y * per, c1 = (c0 + per < nch) ? c0 + per : nch;
    double s = 0.0;
    if (what == 1) {
        for (int ch = c0; ch < c1; ++ch) s += a.aux[(size_t)ch * a.QP + idx];
    } else {
        const size_t stride = a.SL.total + a.DT;
        if (idx == a.SL.osxx) {
            for (int ch = c0; ch < c1; ++ch)
                for (int m = 0; m < a.DT; ++m) s += a.part[(size_t)ch * stride + a.SL.total + m];
        } else {
            for (int ch = c0; ch < c1; ++ch) s += a.part[(size_t)ch * stride + idx];
        }
    }
    red2[(size_t)y * n + idx] = s;
}

// ---------------------------------------------------------------------------------------------------
// small single-workgroup kernels (256 threads)
// ---------------------------------------------------------------------------------------------------
__device__ static double bsum(double v, double* red) {      // block-wide sum, 256 threads
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

// <W^T W>[i][j] for independent Gaussian columns (node.py:213-227 with an isotropic child precision) -> wtw [q][q] in LDS.
// <W> is staged into wst [d][q] first (one coalesced sweep instead of d dependent trips to L2 per thread) and stays there.
__device__ static void wtw_lds(const PcaArgs& a, double* wtw, double* wst) {
    const int d = a.d, q = a.q, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < d * q; idx += 256) wst[idx] = a.W_mean[idx];
    __syncthreads();
    for (int idx = threadIdx.x; idx < q * q; idx += 256) {
        const int i = idx / q, j = idx % q;
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < d; ++k) s += wst[k * q + i] * wst[k * q + j];
        wtw[idx] = s;
    }
    __syncthreads();
    // + sum_k var(W[k][i]) on the diagonal, a wavefront per column (i = wave, wave + 4, ..: up to eight columns, d <= 256: four
    // entries per lane); the loads of all of them are issued before the first sum (one trip to L2 instead of one per column)
    double wv[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = wave + 4 * u, k = lane + 64 * v;
            wv[u][v] = (i < q && k < d) ? a.W_var[(size_t)i * d + k] : 0.0;
        }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = wave + 4 * u;
        const double sv = wsum((wv[u][0] + wv[u][1]) + (wv[u][2] + wv[u][3]));
        if (lane == 0 && i < q) wtw[i * q + i] += sv;
    }
    __syncthreads();
}

// residual  sum_n tr[<x x^T> + <m m^T> - 2 <x><m>^T],  m = W z_n + Mu   (node.py:121-129, :260-271); all threads get it
__device__ static double residual(const PcaArgs& a, const double* wtw, const double* wst, double* red) {
    const int d = a.d, q = a.q, QP = a.QP;
    const double* S = a.stats;
    const double N = (double)a.N_total;
    double wz = 0.0;         // sum_ij <w_i^T w_j> Szz_ij, Szz = sum z z^T + N Sigma_z
    for (int idx = threadIdx.x; idx < q * q; idx += 256) {
        const int i = idx / q, j = idx % q;
        wz += wtw[idx] * (S[a.SL.oSzz + (size_t)i * QP + j] + N * a.Z_cov[idx]);
    }
    double mm = 0.0, cross = 0.0;
    for (int k = threadIdx.x; k < d; k += 256) {
        const double mu = a.Mu_mean[k];
        double wsz = 0.0, xzw = 0.0;
        for (int i = 0; i < q; ++i) { const double w = wst[k * q + i]; wsz += w * S[a.SL.osz + i]; xzw += w * S[a.SL.oSxz + (size_t)k * QP + i]; }
        mm += N * (mu * mu + a.Mu_var[k]) + 2.0 * wsz * mu;
        cross += xzw + S[a.SL.osx + k] * mu;
    }
    const double tot = bsum(wz + mm - 2.0 * cross, red);
    return S[a.SL.osxx] + S[a.SL.osxv] + tot;
}

// One instantiation per mode: these kernels run once per iteration on one CU, straight through, so their cost is the number of
// instruction-cache lines they touch and the round trips to L2 they wait for (measured: the W update took 43 us as a fully
// unrolled 32 x 32 register kernel, most of it instruction fetch); loops stay rolled, per-thread vectors live in LDS.
template <int MODE>
__device__ __forceinline__ void pca_small_body(const PcaArgs& a, double* sm, double* red, double* wst) {
    const int tid = threadIdx.x, d = a.d, q = a.q, QP = a.QP, DP = a.DP;
    const double* S = a.stats;
    const double beta = a.scal[PS_BETA_A] / a.scal[PS_BETA_B];
    const double N = (double)a.N_total;
    if constexpr (MODE == PCA_W) {
        // [w.update() for w in Ws]: rows of W decouple (isotropic beta, diagonal priors); thread = row k
        // message chain hstack -> Mult(W, z_n) -> Addition(., Mu) -> X_n: (beta I, beta (x_n - <Mu>))
        double* szz = sm;                                   // Szz = sum z z^T + N Sigma_z  [q][q]
        for (int idx = tid; idx < q * q; idx += 256) szz[idx] = S[a.SL.oSzz + (size_t)(idx / q) * QP + idx % q] + N * a.Z_cov[idx];
        __syncthreads();
        const int k = tid;
        // thread = row k; its row of <W> (wl), and per block of 16 columns the prior precisions -> new variances (pv) and the linear
        // terms -> log precisions (hl), in LDS columns of its own: the chain over the columns is a rolled loop without a global access
        double* wl = wst;                                   // [32][256]
        double* pv = sm + 32 * 32;                          // [16][256]
        double* hl = pv + 16 * 256;                         // [16][256]
        double* lred = hl + 16 * 256;                       // [4][32] wavefront sums of the log precisions
        const double muk = k < d ? a.Mu_mean[k] : 0.0;
#pragma unroll 8
        for (int i = 0; i < q; ++i) wl[i * 256 + tid] = k < d ? a.W_mean[(size_t)k * q + i] : 0.0;
        for (int i0 = 0; i0 < q; i0 += 16) {
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) {
                const int i = i0 + ii;
                if (i < q && k < d) {
                    const double pp = a.W_pp[(size_t)i * d + k];
                    pv[ii * 256 + tid] = pp;
                    hl[ii * 256 + tid] = pp * a.W_pm[(size_t)k * q + i] + beta * (S[a.SL.oSxz + (size_t)k * QP + i] - muk * S[a.SL.osz + i]);
                }
            }
            if (k < d) {
                for (int ii = 0; ii < 16 && i0 + ii < q; ++ii) {
                    const int i = i0 + ii;
                    const double* srow = szz + i * q;
                    double acc = 0.0;
#pragma unroll 4
                    for (int j = 0; j < q; ++j) acc += (j != i ? srow[j] : 0.0) * wl[j * 256 + tid];
                    const double prec = pv[ii * 256 + tid] + beta * srow[i];
                    wl[i * 256 + tid] = (hl[ii * 256 + tid] - beta * acc) / prec;
                    pv[ii * 256 + tid] = 1.0 / prec;
                    hl[ii * 256 + tid] = 0.5 * log(prec);
                }
            }
            for (int ii = 0; ii < 16 && i0 + ii < q; ++ii) {
                const int i = i0 + ii;
                if (k < d) { a.W_mean[(size_t)k * q + i] = wl[i * 256 + tid]; a.W_var[(size_t)i * d + k] = pv[ii * 256 + tid]; }
                const double v = wsum(k < d ? hl[ii * 256 + tid] : 0.0);
                if ((tid & 63) == 0) lred[(tid >> 6) * 32 + i] = v;
            }
        }
        __syncthreads();
        if (tid < q) a.qld_W[tid] = 0.5 / (((lred[tid] + lred[32 + tid]) + lred[64 + tid]) + lred[96 + tid]);      // gaussian.py:120 (quirk Q1)
    } else if constexpr (MODE == PCA_PREPZ) {
        // posterior of the Z_n: precision I + beta <W^T W>, shared by all n; Gz = beta Sigma_z <W>^T
        double* P = sm; double* Sg = sm + 64 * 64;
        wtw_lds(a, Sg, wst);
        double* Pn = sm + 32 * 32;                          // the second of the two buffers the elimination alternates between
        for (int idx = tid; idx < q * q; idx += 256) P[idx] = ((idx / q == idx % q) ? 1.0 : 0.0) + beta * Sg[idx];
        __syncthreads();
        // Gauss-Jordan inverse (SPD, no pivoting), out of one buffer into the other: one barrier per pivot; log det from the pivots
        double logdet = 0.0;
        for (int p = 0; p < q; ++p) {
            const double piv = P[p * q + p];
            if (tid == 0 && !(piv > 0.0)) atomicOr(a.status, 1);
            logdet += log(piv);
            const double dinv = 1.0 / piv;
            for (int idx = tid; idx < q * q; idx += 256) {
                const int i = idx / q, j = idx % q;
                const double cij = P[i * q + p], rpj = P[p * q + j];
                Pn[idx] = (i == p) ? ((j == p) ? dinv : rpj * dinv) : ((j == p) ? -cij * dinv : P[idx] - cij * rpj * dinv);
            }
            __syncthreads();
            double* tsw = P; P = Pn; Pn = tsw;
        }
        for (int idx = tid; idx < q * q; idx += 256) a.Z_cov[idx] = P[idx];
        if (tid == 0) a.scal[PS_QLD_Z] = 0.5 / (0.5 * logdet);
        // Gz[i][k] = beta sum_j Sigma_z[i][j] W[k][j], stored as pass 1's B operands (thread = position in the block: padded
        // positions get their zero in the same pass; Sigma_z is read through its transpose -- it is symmetric -- so that the lanes of a
        // wavefront, which differ in i, read consecutive words, and share the row of <W>);  g0 = Gz <Mu>
        const int DS = DP / 4;
        for (int pos = tid; pos < a.QT * DS * 64; pos += 256) {
            const int t = pos / (DS * 64), sk = (pos >> 6) % DS, ln = pos & 63;
            const int i = 16 * t + (ln & 15), k = 16 * (sk >> 2) + 4 * (ln >> 4) + (sk & 3);
            double sgz = 0.0;
            if (i < q && k < d)
                for (int j = 0; j < q; ++j) sgz += P[j * q + i] * wst[k * q + j];
            a.Gz[pos] = beta * sgz;
        }
        // t_j = sum_k W[k][j] <Mu>_k (first wavefront) and, for the deferred update, u_j = sum_k W[k][j] (sum x)_k (second wavefront):
        // what Mu.update() needs of the new Z is sum_n z_n = Gz sum_n x_n - N g0 = beta Sigma_z (u - N t), linear in the sum of x kept
        // from the last sweep (this rank's rows: sx_local and N are local, the result is all-reduced like the sum pass 1 delivered)
        double* mus = Sg + 64;                              // <Mu> staged
        double* sxl = mus + 256;                            // sum x staged
        double* tu = sxl + 256;                             // t [32], u [32]
        if (tid < d) { mus[tid] = a.Mu_mean[tid]; sxl[tid] = a.z_deferred ? a.sx_local[tid] : 0.0; }
        __syncthreads();
        if ((tid & 63) < q && tid < 128) {
            const double* vec = tid < 64 ? mus : sxl;
            const int j = tid & 63;
            double sv = 0.0;
            for (int k = 0; k < d; ++k) sv += wst[k * q + j] * vec[k];
            tu[(tid >> 6) * 32 + j] = sv;
        }
        __syncthreads();
        if (tid < QP) {
            double sg = 0.0, su = 0.0;
            if (tid < q)
                for (int j = 0; j < q; ++j) { sg += P[tid * q + j] * tu[j]; su += P[tid * q + j] * (tu[32 + j] - (double)a.N * tu[j]); }
            a.g0[tid] = beta * sg;
            if (a.z_deferred) a.aux_tail[tid] = beta * su;
        }
        if (a.z_deferred && tid < DP) a.aux_tail[QP + tid] = 0.0;     // no change of sum x
    } else if constexpr (MODE == PCA_X0) {
        // Xs[0].update() alone (the crawl order puts it before Mu): aux = [sz (QP) | delta of sum x (DP)]
        double* dsx = a.aux_tail + QP;
        if (a.x0_prep && tid < QP) a.aux_tail[tid] = a.x0_prep == 1 ? S[a.SL.osz + tid] : 0.0;     // the sum of z travels once, from the owner of row 0
        if (tid < DP) dsx[tid] = 0.0;
        __syncthreads();
        if (a.row_offset == 0 && a.z_deferred) {
            // Zs[0].update() has not been materialised (k_pca_pass12 does it for the other rows): z_0 = Gz x_0 - g0 from the row as
            // it stands, before it changes below; the pass keeps this value for row 0 (keep_z0)
            const double xk = tid < d ? a.X[tid] : 0.0;             // thread = column k; the sums over k through bsum, one latent index at a time
            for (int i = 0; i < QP; ++i) {
                const double s = bsum((i < q && tid < d) ? a.Gz[gz_pos(i, tid, DP / 4)] * xk : 0.0, red);
                if (tid == 0) a.Z[i] = i < q ? s - a.g0[i] : 0.0;
            }
            __threadfence_block();
            __syncthreads();
        }
        if (a.row_offset == 0 && a.nmiss[0] > 0) {
            const bool pin = a.pinned && !a.pinned[0];           // first update of a row still carrying its initial mean everywhere
            __syncthreads();
            if (tid < d && a.M[tid] == 0) {
                double pred = a.Mu_mean[tid];
                for (int i = 0; i < q; ++i) pred += a.W_mean[(size_t)tid * q + i] * a.Z[i];
                dsx[tid] = pred - a.X[tid];
                a.X[tid] = pred;
            } else if (tid < d && pin) {
                dsx[tid] = a.Xdata[tid] - a.X[tid];
                a.X[tid] = a.Xdata[tid];
            }
            if (tid == 0) { a.xvar[0] = 1.0 / beta; if (pin) a.pinned[0] = 1; }
        }
        __syncthreads();
        if (tid < DP) a.sx_local[tid] += dsx[tid];          // this rank's own sum of x follows its row
    } else if constexpr (MODE == PCA_APPLY) {
        // after the all-reduce of aux: the new sum of z replaces the old one, the sum of x moves by the delta
        if (tid < QP) a.stats[a.SL.osz + tid] = a.aux_tail[tid];
        if (tid < DP) a.stats[a.SL.osx + tid] += a.aux_tail[QP + tid];
    } else if constexpr (MODE == PCA_MU) {
        // Mu.update(): N Addition children, each sends (beta I, beta (x_n - <W><z_n>))
        double lp = 0.0;
        if (tid < d) {
            const int k = tid;
            const double pp = a.Mu_pp[k], prec = pp + N * beta;
            double wsz = 0.0;
            for (int i = 0; i < q; ++i) wsz += a.W_mean[(size_t)k * q + i] * S[a.SL.osz + i];
            a.Mu_mean[k] = (pp * a.Mu_pm[k] + beta * (S[a.SL.osx + k] - wsz)) / prec;
            a.Mu_var[k] = 1.0 / prec;
            lp = 0.5 * log(prec);
        }
        lp = bsum(lp, red);
        if (tid == 0) a.scal[PS_QLD_MU] = 0.5 / lp;
    } else if constexpr (MODE == PCA_BETA) {
        // Beta.update(): Gamma, traces (nodes_todo.py:130-138)
        wtw_lds(a, sm, wst);
        const double res = residual(a, sm, wst, red);
        if (tid == 0) { a.scal[PS_BETA_B] = a.scal[PS_BETA_B0] + 0.5 * res; a.scal[PS_RES] = res; }
    } else if constexpr (MODE == PCA_ELBO) {
        double res;
        if (a.res_cached) {
            res = a.scal[PS_RES];
        } else {
            wtw_lds(a, sm, wst);
            res = residual(a, sm, wst, red);
        }
        const double qa = a.scal[PS_BETA_A], qb = a.scal[PS_BETA_B];
        const double lnd_beta = d * (log(qa) - log(qb));                  // Gamma.pass_down_lndet (quirk Q2)
        // X_n (gaussian.py:136-151)
        double LX = N * (-0.5 * d * LN2PI + 0.5 * lnd_beta) - 0.5 * beta * res;
        LX -= 0.5 * (double)a.n_part_missing * LN2PI - 0.5 * S[a.SL.oslv] - 0.5 * (double)a.n_part_missing;
        if (a.n_none_rows > 0) LX += (double)a.n_none_rows * (0.5 * d * LN2PI + 0.5 * d) + 0.5 * S[a.SL.osql];
        // Z_n against Constant(0), Constant(I)
        double tr = 0.0;
        if (tid < q) tr = S[a.SL.oSzz + (size_t)tid * QP + tid] + N * a.Z_cov[tid * q + tid];
        tr = bsum(tr, red);
        const double LZ = N * (-0.5 * q * LN2PI) - 0.5 * tr + N * (0.5 * q * LN2PI + 0.5 * a.scal[PS_QLD_Z] + 0.5 * q);
        // W columns and Mu against their Constant parents: thread = row (d <= 256) adds its terms of all columns, one block sum
        double lw = 0.0;
        if (tid < d) {
            const int k = tid;
            for (int i = 0; i < q; ++i) {
                const double pp = a.W_pp[(size_t)i * d + k], w = a.W_mean[(size_t)k * q + i], pm = a.W_pm[(size_t)k * q + i];
                lw += 0.5 * log(pp) - 0.5 * pp * (w * w + a.W_var[(size_t)i * d + k] + pm * pm - 2.0 * w * pm);
            }
        }
        if (tid < q) lw += 0.5 * a.qld_W[tid] + 0.5 * d;        // (- d/2 ln 2 pi of the prior term + d/2 ln 2 pi of the entropy cancel)
        const double LW = bsum(lw, red);
        double lm = 0.0;
        if (tid < d) {
            const double pp = a.Mu_pp[tid], mu = a.Mu_mean[tid], pm = a.Mu_pm[tid];
            lm = 0.5 * log(pp) - 0.5 * pp * (mu * mu + a.Mu_var[tid] + pm * pm - 2.0 * mu * pm);
        }
        const double LM = bsum(lm, red) - 0.5 * d * LN2PI + 0.5 * d * LN2PI + 0.5 * a.scal[PS_QLD_MU] + 0.5 * d;
        if (tid == 0) {
            const double a0 = a.scal[PS_BETA_A0], b0 = a.scal[PS_BETA_B0];
            const double Elnx = a.scal[PS_DIGAMMA_A] - log(qb);
            double LB = (a0 - 1.0) * Elnx - a.scal[PS_LGAMMA_A0] + a0 * log(b0) - b0 * beta;
            LB -= (qa - 1.0) * Elnx - a.scal[PS_LGAMMA_A] + qa * log(qb) - qb * beta;
            a.elbo[0] = LW; a.elbo[1] = LZ; a.elbo[2] = LX; a.elbo[3] = LM; a.elbo[4] = LB;
        }
    }
}

// The small steps of an iteration come in runs (W, Z-prepare | X_0, apply, Mu | Beta, bound): a run is one launch, its steps
// separated by a barrier and a fence (the steps talk through global memory; they are the same code that runs alone).
template <int... MODES>
__global__ void __launch_bounds__(256) k_pca_small(PcaArgs a) {
    __shared__ double sm[64 * 64 + 64 * 64 + 64 + 1088], red[4], wst[256 * 32];     // wst: <W> [d][q] staged by wtw_lds
    int first = 1;
#ifdef SMALL_STAMP      // (bash profiles/build_pca_variant.sh sstamp "-DSMALL_STAMP": where the single-workgroup steps spend their microseconds)
    unsigned long long st = __builtin_amdgcn_s_memrealtime();
    ((first ? (void)(first = 0) : (__threadfence(), __syncthreads()), pca_small_body<MODES>(a, sm, red, wst), __syncthreads(),
      (threadIdx.x == 0 ? (void)printf("small mode %d: %.2f us\n", MODES, (__builtin_amdgcn_s_memrealtime() - st) / 100.0) : (void)0),
      st = __builtin_amdgcn_s_memrealtime()), ...);
#else
    ((first ? (void)(first = 0) : (__threadfence(), __syncthreads()), pca_small_body<MODES>(a, sm, red, wst)), ...);
#endif
}

// q_ln_det (gaussian.py:120, quirk Q1) of every X_n that has no observed entry -- a latent node with qprec = I / var_n --
// and NaN for the rows that have one (observed and partially observed nodes never set it)
__global__ void __launch_bounds__(256) k_pca_rowqld(PcaArgs a, double* out) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.N) return;
    out[row] = (a.nmiss[row] == a.d) ? 0.5 / (0.5 * a.d * log(1.0 / a.xvar[row])) : __builtin_nan("");
}

// ---------------------------------------------------------------------------------------------------
static PcaArgs pca_args(pyvb_pca* h) {
    PcaArgs a;
    a.X = h->X; a.M = h->M; a.xvar = h->xvar; a.nmiss = h->nmiss; a.Z = h->Z; a.Xdata = h->Xdata; a.pinned = h->pinned;
    a.W_mean = h->W_mean; a.W_var = h->W_var; a.Mu_mean = h->Mu_mean; a.Mu_var = h->Mu_var; a.Z_cov = h->Z_cov; a.qld_W = h->qld_W;
    a.W_pm = h->W_pm; a.W_pp = h->W_pp; a.Mu_pm = h->Mu_pm; a.Mu_pp = h->Mu_pp;
    a.scal = h->scal; a.Gz = h->Gz; a.g0 = h->g0; a.sx_local = h->sx_local; a.part = h->part; a.stats = h->stats; a.aux = h->aux; a.aux_tail = h->aux + (size_t)4 * h->nchunk * h->QP; a.elbo = h->elbo; a.status = h->status;
    a.N = h->N; a.N_total = h->N_total; a.chunk_rows = h->chunk_rows; a.lo_upd = 0; a.hi_upd = 0;
    a.n_part_missing = h->n_part_missing; a.n_none_rows = h->n_none_rows; a.row_offset = h->row_offset;
    a.d = h->d; a.q = h->q; a.DP = h->DP; a.QP = h->QP; a.DT = h->DT; a.QT = h->QT; a.nchunk = h->nchunk; a.mode = 0; a.SL = h->SL;
    a.res_cached = h->res_valid ? 1 : 0;
    a.keep_z0 = 0; a.z_deferred = 0; a.x0_prep = 0;
    a.W_x = h->W_x; a.Mu_x = h->Mu_x; a.vin_lo = a.vin_hi = 0; a.save_wx = 0;
    return a;
}

int pca_launch_small(pyvb_pca* h, int mode) {
    PcaArgs a = pca_args(h); a.mode = mode;
    a.z_deferred = h->z_pending ? 1 : 0;
    if (mode == PCA_RUN_MID) a.x0_prep = 1;             // no communicator: this rank owns row 0
    if (mode == PCA_RUN_TAIL) a.res_cached = 1;         // the bound follows the Beta update of the same launch
    switch (mode) {
#define PCA_SMALL(M, ...) case M: hipLaunchKernelGGL((k_pca_small<__VA_ARGS__>), dim3(1), dim3(256), 0, h->stream, a); break
        PCA_SMALL(PCA_W, PCA_W); PCA_SMALL(PCA_PREPZ, PCA_PREPZ); PCA_SMALL(PCA_MU, PCA_MU); PCA_SMALL(PCA_BETA, PCA_BETA);
        PCA_SMALL(PCA_ELBO, PCA_ELBO); PCA_SMALL(PCA_X0, PCA_X0); PCA_SMALL(PCA_APPLY, PCA_APPLY);
        // runs of pyvb_pca_iterate without a communicator (with one, an all-reduce sits between the steps)
        PCA_SMALL(PCA_RUN_HEAD, PCA_W, PCA_PREPZ, PCA_APPLY);
        PCA_SMALL(PCA_RUN_MID, PCA_X0, PCA_APPLY, PCA_MU);
        PCA_SMALL(PCA_RUN_TAIL, PCA_BETA, PCA_ELBO);
#undef PCA_SMALL
        default: pyvb_set_error("no such small kernel"); return PYVB_E_ARG;
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int pca_launch_pass1(pyvb_pca* h) {
    { int rc = pca_materialize_x(h); if (rc) return rc; }
    PcaArgs a = pca_args(h);
    a.keep_z0 = h->z0_done ? 1 : 0;
    const size_t lds = (size_t)h->QT * (h->DP / 4) * 64 * sizeof(double);
    if (h->QT == 1) hipLaunchKernelGGL(k_pca_pass1<1>, dim3(h->nchunk), dim3(256), lds, h->stream, a);
    else hipLaunchKernelGGL(k_pca_pass1<2>, dim3(h->nchunk), dim3(256), lds, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int pca_launch_pass2(pyvb_pca* h, long lo_upd, long hi_upd) {
    { int rc = pca_materialize_x(h); if (rc) return rc; }
    PcaArgs a = pca_args(h); a.lo_upd = lo_upd; a.hi_upd = hi_upd;
    const unsigned nw = (h->DT + P2T - 1) / P2T;          // wavefronts per row chunk: 32 columns each, four to a workgroup
    const dim3 grid(h->nchunk, (nw + 3) / 4), block(64 * (nw < 4 ? nw : 4));
    const bool pin = h->Xdata != nullptr;
    if (h->QT == 1) { if (pin) hipLaunchKernelGGL((k_pca_pass2<1, true>), grid, block, 0, h->stream, a); else hipLaunchKernelGGL((k_pca_pass2<1, false>), grid, block, 0, h->stream, a); }
    else { if (pin) hipLaunchKernelGGL((k_pca_pass2<2, true>), grid, block, 0, h->stream, a); else hipLaunchKernelGGL((k_pca_pass2<2, false>), grid, block, 0, h->stream, a); }
    hipLaunchKernelGGL(k_pca_rowvar, dim3(h->nchunk), dim3(a.chunk_rows >= 2048 ? 1024 : 256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    h->part_chunks = h->nchunk;
    return PYVB_OK;
}

// The missing entries of the rows a lazy sweep imputed (pyvb_pca: xlazy) into X, for everything that reads X other than the next sweep
int pca_materialize_x(pyvb_pca* h) {
    if (!h->xlazy) return PYVB_OK;
    PcaArgs a = pca_args(h); a.vin_lo = h->vlo; a.vin_hi = h->vhi;
    const unsigned nw = (h->DT + P2T - 1) / P2T;
    if (h->QT == 1) hipLaunchKernelGGL(k_pca_materialize<1>, dim3(h->nchunk), dim3(64 * nw), 0, h->stream, a);
    else hipLaunchKernelGGL(k_pca_materialize<2>, dim3(h->nchunk), dim3(64 * nw), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    h->xlazy = false;
    return PYVB_OK;
}

// Z and X updates and the statistics in one sweep (k_pca_pass12); rows [lo_upd, hi_upd) of X are updated.  The standard sweep
// of an iteration -- all rows, or all but row 0, which the crawl order updates on its own -- leaves the imputed entries to be
// recomputed by the next one (LAZY); any other range, latent dimensions beyond 16 (the register budget) and rows that still
// wait for their first update take the write-back.
int pca_launch_pass12(pyvb_pca* h, long lo_upd, long hi_upd) {
    const bool pin = h->Xdata != nullptr;
    const bool lazy = h->lazy_ok && h->QT == 1 && !pin && lo_upd <= 1 && hi_upd == h->N && hi_upd > lo_upd
                      && (!h->xlazy || (lo_upd <= h->vlo && h->vhi <= hi_upd));
    int rc;
    if (!lazy && (rc = pca_materialize_x(h))) return rc;
    PcaArgs a = pca_args(h); a.lo_upd = lo_upd; a.hi_upd = hi_upd;
    a.keep_z0 = h->z0_done ? 1 : 0;
    if (lazy && h->xlazy) { a.vin_lo = h->vlo; a.vin_hi = h->vhi; }
    a.save_wx = lazy ? 1 : 0;
    if (lazy && h->rows_ok) {
        // the row-owning sweeps (k_pca_rows, k_pca_pairs) on their own partition of the rows: one workgroup per CU, equal shares of 16-row tiles
        a.nchunk = h->nchunkB; a.chunk_rows = h->chunk_rowsB;
        const bool pairs = h->rows_ok == 2;
        size_t need = 3 * (size_t)h->DT * 256 + 2 * (size_t)h->DP + 16 +
                      (pairs ? 16 + PP_NW * (16 * P12_XS) + PP_NW * 256 + 8 : PR_NW * (16 * 17 + 16 * P12_XS));
        if (need < 4640) need = 4640;                   // the final reduction's buffer
        const size_t lds = need * sizeof(double);
        if (!h->rows_attr_set) {
            HIPCHK(hipFuncSetAttribute((const void*)k_pca_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void*)k_pca_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void*)k_pca_pairs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPCHK(hipFuncSetAttribute((const void*)k_pca_pairs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            h->rows_attr_set = true;
        }
        if (pairs) {
            if (h->DT == 16) hipLaunchKernelGGL(k_pca_pairs<true>, dim3(a.nchunk), dim3(64 * PP_NW), lds, h->stream, a);
            else hipLaunchKernelGGL(k_pca_pairs<false>, dim3(a.nchunk), dim3(64 * PP_NW), lds, h->stream, a);
        }
        else if (h->DT == 16) hipLaunchKernelGGL(k_pca_rows<true>, dim3(a.nchunk), dim3(64 * PR_NW), lds, h->stream, a);
        else hipLaunchKernelGGL(k_pca_rows<false>, dim3(a.nchunk), dim3(64 * PR_NW), lds, h->stream, a);
        hipLaunchKernelGGL(k_pca_rowvar, dim3(a.nchunk), dim3(a.chunk_rows >= 2048 ? 1024 : 256), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        h->part_chunks = a.nchunk;
        h->xlazy = true; h->vlo = lo_upd; h->vhi = hi_upd;
        return PYVB_OK;
    }
    h->part_chunks = h->nchunk;
    const unsigned nw = (h->DT + P2T - 1) / P2T;          // wavefronts per workgroup: 32 columns each
    const size_t rt = h->QT == 1 ? 2 : 1;                 // k_pca_pass12: RT
    const size_t lds = (rt * ((size_t)nw * h->QT * 256 + 2 * (size_t)h->QT * 256 + 2 * (size_t)h->QT * 16 * 17 + (size_t)nw * 16 * P12_XS) + (lazy ? h->DP : 0)) * sizeof(double);
    const dim3 grid(h->nchunk), block(64 * nw);
    if (lazy) hipLaunchKernelGGL((k_pca_pass12<1, false, true>), grid, block, lds, h->stream, a);
    else if (h->QT == 1) { if (pin) hipLaunchKernelGGL((k_pca_pass12<1, true>), grid, block, lds, h->stream, a); else hipLaunchKernelGGL((k_pca_pass12<1, false>), grid, block, lds, h->stream, a); }
    else { if (pin) hipLaunchKernelGGL((k_pca_pass12<2, true>), grid, block, lds, h->stream, a); else hipLaunchKernelGGL((k_pca_pass12<2, false>), grid, block, lds, h->stream, a); }
    hipLaunchKernelGGL(k_pca_rowvar, dim3(h->nchunk), dim3(a.chunk_rows >= 2048 ? 1024 : 256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    if (lazy) { h->xlazy = true; h->vlo = lo_upd; h->vhi = hi_upd; }
    return PYVB_OK;
}

#ifdef P12_STAMP
extern "C" int pyvb_pca_debug_stamps(unsigned long long* out, int nblocks) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p12_stamp), (size_t)nblocks * 2 * 12 * sizeof(unsigned long long)));
    return PYVB_OK;
}
#endif

int pca_launch_rowqld(pyvb_pca* h, double* out) {
    PcaArgs a = pca_args(h);
    hipLaunchKernelGGL(k_pca_rowqld, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, a, out);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int pca_launch_reduce(pyvb_pca* h, int what) {
    PcaArgs a = pca_args(h);
    if (what == 0 && h->part_chunks > 0) a.nchunk = h->part_chunks;     // the partition of the sweep that wrote the partials
    const size_t n = what == 1 ? (size_t)h->QP : h->SL.total;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_pca_reduce, dim3(nb, PCA_RED), dim3(256), 0, h->stream, a, what, 0, h->red2);
    hipLaunchKernelGGL(k_pca_reduce, dim3(nb), dim3(256), 0, h->stream, a, what, 1, h->red2);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
