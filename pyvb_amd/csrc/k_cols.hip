// k_cols: the column Gaussians of A and C, the residuals their noise nodes need, and the noise update.
//   [a.update() for a in As] / [c.update() for c in Cs]:
//       Gaussian.update gaussian.py:102-123 fed by hstack.pass_up_m1_m2 nodes_todo.py:43-62
//   residuals  sum over children of  1/2 diag<x x^T> + 1/2 diag<mu mu^T> - diag(<x><mu>^T)
//       (nodes_todo.py:138, :190 with Multiplication.pass_down_ExxT node.py:260-271)
//   Gamma.update / DiagonalGamma.update  nodes_todo.py:130-138, :187-190
//
// With diagonal noise precisions and diagonal column priors every row of A (of C) only interacts
// with itself: lane k of the wavefront owns row k.  The pass over the columns is a Gauss-Seidel
// sweep (column i sees the new columns 0..i-1 and the old columns i+1..): sequential in i.  It is
// blocked by 16 columns.  For a block the products with everything as it stood before the block,
//       P[k][i] = sum_j M[k][j] G[i][j],
// are one [64 x 64] x [64 x 16] product on v_mfma_f64_16x16x4_f64; inside the block only the
// corrections by the columns already renewed in it remain,
//       sum_{j != i} M[k][j] G[i][j] = P[k][i] - Mold[k][i] G[i][i] + sum_{j<i, j in block} (Mnew - Mold)[k][j] G[i][j],
// at most 15 multiply-adds per column and lane.  The residuals need diag(M G M^T): the same
// product once more with the renewed M, then a row-wise dot product.
#include "params.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MS 72       // row stride of Mb: A-operand reads and lane-per-row reads are conflict free
#define GS 68       // row stride of Gb and Sb

// 16 consecutive entries of this lane's row (row-contiguous per lane); columns >= D read as `fill`
__device__ __forceinline__ void load_row16(const double* row, int col0, int D, bool vec, double fill, double* out) {
    if (vec) {      // D % 4 == 0: groups of four are either wholly inside or wholly outside
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (col0 + 4 * g < D) {
                const d4 v = *reinterpret_cast<const d4*>(row + col0 + 4 * g);
                out[4 * g] = v[0]; out[4 * g + 1] = v[1]; out[4 * g + 2] = v[2]; out[4 * g + 3] = v[3];
            } else {
                out[4 * g] = out[4 * g + 1] = out[4 * g + 2] = out[4 * g + 3] = fill;
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = col0 + u;
            const double v = row[c < D ? c : D - 1];
            out[u] = c < D ? v : fill;
        }
    }
}

__global__ void __launch_bounds__(64) k_cols(ParamArgs a) {
    const int WHICH = a.which0 + blockIdx.y;
    __shared__ double Mb[64 * MS];      // Mb[col * MS + row], zero padded to 64 x 64
    __shared__ double Gb[16 * GS];      // rows 16b .. 16b+15 of G, zero padded: Gb[ii * GS + j] = G[16b+ii][j]
    __shared__ double Sb[16 * GS];      // Sb[ii * GS + row]: P of the block, then the precisions of column ii
    const int n = blockIdx.x, lane = threadIdx.x, D = a.D, K = a.K, c = lane & 15, q = lane >> 4;
    const int rows = WHICH == 0 ? D : K;
    double* M = (WHICH == 0 ? a.A_mean : a.C_mean) + (size_t)n * rows * D;
    double* V = (WHICH == 0 ? a.A_var : a.C_var) + (size_t)n * D * rows;
    double* qld = (WHICH == 0 ? a.qld_A : a.qld_C) + (size_t)n * D;
    const double* pm = WHICH == 0 ? a.pri.A_pm : a.pri.C_pm;    // [row][col]
    const double* pp = WHICH == 0 ? a.pri.A_pp : a.pri.C_pp;    // [col][row]
    const double* obs = WHICH == 0 ? a.pri.A_obs : a.pri.C_obs; // [row][col], NaN = not observed
    const double* mo = a.mom + (size_t)n * mom_total(D, K);
    const double* G = mo + (WHICH == 0 ? MOM_GA(D, K) : MOM_GC(D, K));
    const double* H = mo + (WHICH == 0 ? MOM_HA(D, K) : MOM_HC(D, K));
    const bool live = lane < rows;
    const int lr = live ? lane : 0;                 // clamped row for the per-lane loads
    const int lc = lane < D ? lane : D - 1;         // clamped column for the coalesced row loads
    const bool vec = (D & 3) == 0;
    const int nb = (D + 15) >> 4;
    const double* Mrow = M + (size_t)lr * D;
    const double* Hrow = H + (size_t)lr * D;

    // ---- this lane's row of M into LDS
    for (int b = 0; b < 4; ++b) {
        double m[16] = {};
        if (b < nb) load_row16(Mrow, 16 * b, D, vec, 0.0, m);
#pragma unroll
        for (int u = 0; u < 16; ++u) Mb[(16 * b + u) * MS + lane] = (b < nb && live) ? m[u] : 0.0;
    }
    const double lam = live ? (WHICH == 0 ? a.Q_a[(size_t)n * D + lane] / a.Q_b[(size_t)n * D + lane]
                                          : a.R_a[(size_t)n * K + lane] / a.R_b[(size_t)n * K + lane]) : 0.0;

    auto stage_G = [&](int b) {         // coalesced: lane = element j of the row
        double g[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = 16 * b + u;
            g[u] = G[(size_t)(i < D ? i : D - 1) * D + lc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) Gb[u * GS + lane] = (16 * b + u < D && lane < D) ? g[u] : 0.0;
    };
    auto product = [&]() {              // Sb[ii][k] = sum_j Mb[j][k] Gb[ii][j]
        d4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = d4{0.0, 0.0, 0.0, 0.0};
        // all 16 k-steps whatever D is: Mb and Gb are zero beyond it, and a fixed trip count unrolls
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const double bop = Gb[c * GS + 4 * s + q];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA(Mb[(4 * s + q) * MS + 16 * rt + c], bop, acc[rt]);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Sb[c * GS + 16 * rt + 4 * r + q] = acc[rt][r];
    };

    // ---- the columns
    if (a.c0 < a.c1) {
        for (int b = a.c0 >> 4; b <= (a.c1 - 1) >> 4; ++b) {
            __syncthreads();
            stage_G(b);
            double p0[16], m0[16], hk[16], ob[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = 16 * b + u;
                p0[u] = pp[(size_t)(i < D ? i : D - 1) * rows + lr];
            }
            load_row16(pm + (size_t)lr * D, 16 * b, D, vec, 0.0, m0);
            load_row16(Hrow, 16 * b, D, vec, 0.0, hk);
            load_row16(obs + (size_t)lr * D, 16 * b, D, vec, 0.0, ob);
            __syncthreads();
            product();
            __syncthreads();
            // everything that does not depend on the columns renewed in this block, so that the chain from
            // one column to the next is a handful of multiply-adds
            double gd[16], var[16], base[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                gd[u] = Gb[u * GS + 16 * b + u];
                const double prec = p0[u] + lam * gd[u];                                // qprec  gaussian.py:117
                var[u] = 1.0 / prec;                                                    // qcov   gaussian.py:118-119
                base[u] = p0[u] * m0[u] + lam * hk[u];
                p0[u] = prec;
            }
            double dl[16];
            int mynk = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = 16 * b + u;
                dl[u] = 0.0;
                if (i >= a.c0 && i < a.c1) {            // wave-uniform
                    const double xo = Mb[i * MS + lane];
                    double acc = Sb[u * GS + lane] - xo * gd[u];
#pragma unroll
                    for (int v = 0; v < u; ++v) acc += dl[v] * Gb[u * GS + 16 * b + v];
                    double val = (base[u] - lam * acc) * var[u], vr = var[u];           // qmu  gaussian.py:119-123
                    // known entries (Gaussian.observe on a column, LDS_knowns_in_A.py:73-74): conditioning a
                    // diagonal Gaussian on them (gaussian.py:125-134) pins those entries and leaves the others
                    // alone; a column whose entries are all known is thereby never changed (gaussian.py:109-110)
                    const bool known = live && (ob[u] == ob[u]);
                    if (known) { val = ob[u]; vr = 0.0; }
                    const int nknown = __popcll(__ballot(known));
                    mynk = (lane == u) ? nknown : mynk;
                    if (live) {
                        dl[u] = val - xo;
                        Mb[i * MS + lane] = val;
                        V[(size_t)i * rows + lane] = vr;
                    }
                    Sb[u * GS + lane] = live ? p0[u] : 1.0;
                } else {
                    Sb[u * GS + lane] = 1.0;
                    mynk = (lane == u) ? rows : mynk;   // no q_ln_det for a column that was not updated
                }
            }
            __syncthreads();
            // q_ln_det of the block's columns (gaussian.py:120, quirk Q1): 0.5 / ln prod_k sqrt(prec_k).
            // Lane (ii, quarter) multiplies 16 mantissas and adds 16 exponents; one log per column.
            {
                double mant = 1.0; int ex = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int e;
                    mant *= frexp(Sb[c * GS + 16 * q + r], &e);
                    ex += e;
                }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    mant *= __shfl_xor(mant, o, 64);
                    ex += __shfl_xor(ex, o, 64);
                }
                const int i = 16 * b + c;
                if (q == 0 && i >= a.c0 && i < a.c1 && mynk < rows)
                    qld[i] = 0.5 / (0.5 * (log(mant) + (double)ex * 0.6931471805599453));
            }
        }
        __syncthreads();
        // this lane's row back to memory
        if (live) {
            for (int b = a.c0 >> 4; b <= (a.c1 - 1) >> 4; ++b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cc = 16 * b + 4 * g;
                    if (vec) {
                        if (cc < D) {
                            d4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = Mb[(cc + r) * MS + lane];
                            *reinterpret_cast<d4*>(M + (size_t)lane * D + cc) = v;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (cc + r < D) M[(size_t)lane * D + cc + r] = Mb[(cc + r) * MS + lane];
                    }
                }
            }
        }
    }

    // ---- residuals of the noise node: res[k] = 1/2 own[k] + 1/2 <mu mu^T>[k,k] - (H M^T)[k,k]
    if (a.fuse & 1) {
        // <mu mu^T>[k,k] = sum_ij M[k,i] G[i,j] M[k,j] + sum_i var_i[k] G[i,i]      node.py:260-271
        double e0 = 0.0, e1 = 0.0, hm0 = 0.0, hm1 = 0.0;
        for (int b = 0; b < nb; ++b) {
            __syncthreads();
            stage_G(b);
            double hk[16], vv[16];
            load_row16(Hrow, 16 * b, D, vec, 0.0, hk);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = 16 * b + u;
                vv[u] = V[(size_t)(i < D ? i : D - 1) * rows + lr];
            }
            __syncthreads();
            product();
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                const int i = 16 * b + u;
                const double x0 = Mb[i * MS + lane], x1 = Mb[(i + 1) * MS + lane];
                e0 += x0 * Sb[u * GS + lane] + (i < D ? vv[u] * Gb[u * GS + i] : 0.0);
                e1 += x1 * Sb[(u + 1) * GS + lane] + (i + 1 < D ? vv[u + 1] * Gb[(u + 1) * GS + i + 1] : 0.0);
                hm0 += x0 * hk[u];
                hm1 += x1 * hk[u + 1];
            }
        }
        const double own = WHICH == 0 ? mo[MOM_DP(D, K) + lr] : a.Syy[(size_t)n * K + lr];
        double r = 0.5 * own + 0.5 * (e0 + e1) - (hm0 + hm1);
        if (live) (WHICH == 0 ? a.resQ : a.resR)[(size_t)n * rows + lane] = r;
        if (a.fuse & 2) {       // Gamma.update / DiagonalGamma.update: same arithmetic as k_noise
            const double* b0 = WHICH == 0 ? a.pri.Q_b0 : a.pri.R_b0;
            double* qb = (WHICH == 0 ? a.Q_b : a.R_b) + (size_t)n * rows;
            r = live ? r : 0.0;
            if (a.noise == PYVB_NOISE_GAMMA) {
                r = wave_sum(r);
                if (live) qb[lane] = b0[0] + r;
            } else if (live) {
                qb[lane] = b0[lane] + r;
            }
        }
    }
}

// which: 0 = A / Q, 1 = C / R, 2 = both in one launch (they are independent given the statistics)
// fuse: bit 0 = also the residuals of the noise node, bit 1 = and its update
int launch_cols(pyvb_lds* h, int which, int c0, int c1, int fuse) {
    if (h->big) return launch_cols_big(h, which, c0, c1, fuse);
    ParamArgs a = make_args(h);
    a.c0 = c0; a.c1 = c1; a.which0 = which == 1 ? 1 : 0; a.fuse = fuse;
    TimedLaunch tl(h, PYVB_K_PARAMS);
    hipLaunchKernelGGL(k_cols, dim3(h->N, which == 2 ? 2 : 1), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}

int launch_resid(pyvb_lds* h, int which) { return launch_cols(h, which, 0, 0, 1); }
