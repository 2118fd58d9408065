// k_prep: everything that depends only on the parameter posteriors and is shared by all
// X_t of one replicate: <A^T Q A>, <C^T R C>, the three posterior precisions of the states,
// their inverses / q_ln_det, the gain matrices, and the warm-up length of the segmented sweeps.
//
// Reference work replaced (per X_t.update(), 2T times per iteration in the reference):
//   Multiplication.pass_up_m1_m2, hstack branch    node.py:213-227   (<A^T Q A>, the D^4 tensor)
//   qprec = pprec + sum m1 ; cho_factor ; cho_solve gaussian.py:117-119
//   q_ln_det (quirk Q1)                             gaussian.py:120
//
// One 256-thread workgroup per replicate, the work matrices in LDS (row stride DP+2 doubles, which
// makes the MFMA A-operand reads bank-conflict free), <A> and <C> read from global memory.  All D x D products run on
// v_mfma_f64_16x16x4_f64, one row tile per wavefront.  The posterior precisions are inverted in
// place by Gauss-Jordan elimination without pivoting (they are symmetric positive definite); its
// pivots are the squares of the Cholesky diagonal, which gives the reference's q_ln_det.
#include "common.h"
#include "gj.h"

struct PrepArgs {
    const double *A_mean, *A_var, *C_mean, *C_var, *Q_a, *Q_b, *R_a, *R_b, *x0_mean, *x0_prec;
    double *Sigma, *qld, *gains, *scratch;
    // Wishart noise (DENSE): E[Q] [D][D], E[Q] <A> [D][D], E[R] <C> [K][D], tr(S_i E[Q]) [D], tr(S'_i E[R]) [D] per replicate
    const double *Qbar, *QA, *RC, *trA, *trC;
    int *warm, *status;
    int N, T, D, K;
    Layout L;
};

#define PREP_THREADS 256
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// C = A * B on the matrix cores: C is [16*MT x 16*NT], the contraction runs over 4*KS.
// a_at(i, k) / b_at(k, j) fetch operand elements (from LDS), store(i, j, v) consumes results.
// Wavefront w owns row tiles w, w+4, ...
template <int MT, int NT, int KS, class FA, class FB, class FS>
__device__ __forceinline__ void mm(int wave, int lane, FA a_at, FB b_at, FS store) {
    const int r = lane & 15, q = lane >> 4;
    for (int m = wave; m < MT; m += 4) {
        d4 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = d4{0, 0, 0, 0};
#pragma unroll 4
        for (int s = 0; s < KS; ++s) {
            const double av = a_at(16 * m + r, 4 * s + q);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = MFMA(av, b_at(4 * s + q, 16 * n + r), acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) store(16 * m + 4 * e + q, 16 * n + r, acc[n][e]);
    }
}

// The same product in two phases, for operands that live in the LDS matrix the result goes to (or that is about to be
// reused): row tile m of the product into registers, and -- after whatever barrier the caller needs -- out of them.
template <int NT, int KS, class FA, class FB>
__device__ __forceinline__ void mm_acc(int m, int lane, FA a_at, FB b_at, d4 (&acc)[NT]) {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = d4{0, 0, 0, 0};
#pragma unroll 4
    for (int s = 0; s < KS; ++s) {
        const double av = a_at(16 * m + r, 4 * s + q);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = MFMA(av, b_at(4 * s + q, 16 * n + r), acc[n]);
    }
}
template <int NT, class FS>
__device__ __forceinline__ void mm_out(int m, int lane, const d4 (&acc)[NT], FS store) {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) store(16 * m + 4 * e + q, 16 * n + r, acc[n][e]);
}

// max_i sum_j |A_ij|: four threads per row (16 columns each), rows and then wavefronts combined by
// shuffles; two barriers.  red: 4 doubles of LDS.
__device__ __forceinline__ double inf_norm(const double* A, int LD, int D, int tid, double* red) {
    const int row = tid >> 2, part = tid & 3;
    double s = 0.0;
    if (row < D) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = 16 * part + u;
            if (j < D) s += fabs(A[row * LD + j]);
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) s = fmax(s, __shfl_xor(s, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    return m;
}

// Number of recurrence steps after which the influence of the starting state of
// x_t = M x_{t-1} + c_t is below 1e-18 relative.  Any induced norm is submultiplicative, so with
// n_k = ||M^(2^k)|| (k = 2..5, from repeated squaring) every J = 4 a + 8 b + 16 c + 32 d has
// ||M^J|| <= n_2^a n_3^b n_4^c n_5^d; the smallest such J under the tolerance is taken.
// M is in W1 (padded with zeros) on entry; W1/W2 are clobbered.
// (Forced inline, as everything on the hot path of this kernel: as a called function it is subject to the calling convention --
// saved registers, operands through memory; k_big.hip's warmup128 ran with ~1000 scratch accesses per squaring that way.)
template <int DT>
__device__ __forceinline__ int warmup_length(double* W1, double* W2, int LD, int D, int tid, double* red) {
    constexpr int DS = 4 * DT;
    const double lntol = -41.4465316738928;   // ln(1e-18)
    const int wave = tid >> 6, lane = tid & 63;
    double l2 = 1.0, l3 = 1.0, l4 = 1.0, l5 = 1.0;   // ln n_k; +1 marks "not contracting at this power"
    double* src = W1; double* dst = W2;
#pragma unroll
    for (int k = 1; k <= 5; ++k) {
        mm<DT, DT, DS>(wave, lane,
                       [&](int i, int kk) { return src[i * LD + kk]; },
                       [&](int kk, int j) { return src[kk * LD + j]; },
                       [&](int i, int j, double v) { dst[i * LD + j] = v; });      // M^(2^k)
        __syncthreads();
        if (k >= 2) {
            const double nrm = inf_norm(dst, LD, D, tid, red);
            const double l = nrm < 1.0 ? ((nrm > 0.0) ? log(nrm) : -1e300) : 1.0;
            if (k == 2) l2 = l; else if (k == 3) l3 = l; else if (k == 4) l4 = l; else l5 = l;
        }
        double* t = src; src = dst; dst = t;
    }
    int best = 1 << 30;
    if (tid == 0) {         // only thread 0 stores the result
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

// DENSE: the noise precisions are Wishart nodes (nodes_todo.py:205-234), their expectations dense matrices; the
// products with them (k_wishart.hip: k_dense_pre) replace the row scalings of the diagonal case.
template <int DT, int KT, bool DENSE>
__global__ void __launch_bounds__(PREP_THREADS) k_prep(PrepArgs a) {
    constexpr int DP = 16 * DT, DS = 4 * DT, KS = 4 * KT, LD = DP + 2;
#ifdef PREP64_STAMP     // (profiles/build_variant.sh k_prep p64 "-DPREP64_STAMP": where a workgroup's time goes, shader-clock ticks)
    unsigned long long qs_t = __builtin_amdgcn_s_memtime(), qs_acc[6] = {0, 0, 0, 0, 0, 0};
#define QSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); qs_acc[i] += t_ - qs_t; qs_t = t_; } while (0)
#else
#define QSTAMP(i) do { } while (0)
#endif
    // <A> and <C> are MFMA operands straight from global memory (L2-resident, 32 KB each): keeping them in
    // LDS too would put the workgroup over half of the CU's 160 KB and halve the occupancy of a kernel
    // that is all latency.  Zero padded by the accessors.
    __shared__ double P[DP * LD];        // posterior precision -> covariance
    __shared__ double W[DP * LD];        // work
    __shared__ double qbar[64], rbar[64], rowp[64], colp[64], gjbuf[3 * 2 * GJ_BUF + 192];
    const int n = blockIdx.x, tid = threadIdx.x, D = a.D, K = a.K;
    const int wave = tid >> 6, lane = tid & 63;
    const Layout& L = a.L;
    const double* Am = a.A_mean + (size_t)n * D * D;
    const double* Av = a.A_var + (size_t)n * D * D;
    const double* Cm = a.C_mean + (size_t)n * K * D;
    const double* Cv = a.C_var + (size_t)n * D * K;
    double* g = a.gains + (size_t)n * L.gains_total;

    if (tid < 64) {
        qbar[tid] = (!DENSE && tid < D) ? a.Q_a[(size_t)n * D + tid] / a.Q_b[(size_t)n * D + tid] : 0.0;
        rbar[tid] = (!DENSE && tid < K) ? a.R_a[(size_t)n * K + tid] / a.R_b[(size_t)n * K + tid] : 0.0;
    }
    const double* Qd = DENSE ? a.Qbar + (size_t)n * D * D : nullptr;
    const double* QAd = DENSE ? a.QA + (size_t)n * D * D : nullptr;
    const double* RCd = DENSE ? a.RC + (size_t)n * K * D : nullptr;
    auto QA_at = [&](int k, int j) { const double v = QAd[(k < D ? k : D - 1) * D + (j < D ? j : D - 1)]; return (k < D && j < D) ? v : 0.0; };
    auto RC_at = [&](int k, int j) { const double v = RCd[(k < K ? k : K - 1) * D + (j < D ? j : D - 1)]; return (k < K && j < D) ? v : 0.0; };
    for (int idx = tid; idx < DP * LD; idx += PREP_THREADS) {
        W[idx] = 0.0;
        P[idx] = 0.0;
    }
    auto A_at = [&](int i, int j) { const double v = Am[(i < D ? i : D - 1) * D + (j < D ? j : D - 1)]; return (i < D && j < D) ? v : 0.0; };
    auto C_at = [&](int k, int j) { const double v = Cm[(k < K ? k : K - 1) * D + (j < D ? j : D - 1)]; return (k < K && j < D) ? v : 0.0; };
    __syncthreads();
    // traces of the column covariances against the noise expectations (diagonal of node.py:223-227)
    if (tid < D) {
        double tc = 0.0, ta = 0.0;
        if constexpr (DENSE) {
            tc = a.trC[(size_t)n * D + tid]; ta = a.trA[(size_t)n * D + tid];
        } else {
            for (int k = 0; k < K; ++k) tc += Cv[tid * K + k] * rbar[k];
            for (int k = 0; k < D; ++k) ta += Av[tid * D + k] * qbar[k];
        }
        rowp[tid] = tc; colp[tid] = ta;
    }
    __syncthreads();
    // <C^T R C> -> W;  <C^T R C> + <A^T Q A> -> P     (node.py:213-227)
    // STAGE: <C>, then <A>, are copied into the matrix that is free at that moment and the products run out of LDS (an
    // operand fetched from L2 inside the k loop makes a 64^3 product three times as long: profiles/r02/limits.txt section 8).
    constexpr bool STAGE = !DENSE && KT <= DT;
    if constexpr (STAGE) {
        constexpr int KP = 16 * KT;
        for (int idx = tid; idx < KP * DP; idx += PREP_THREADS) P[(idx / DP) * LD + idx % DP] = C_at(idx / DP, idx % DP);
        __syncthreads();
        d4 acc[DT];
        if (wave < DT) {
            mm_acc<DT, KS>(wave, lane, [&](int i, int k) { return P[k * LD + i] * rbar[k]; }, [&](int k, int j) { return P[k * LD + j]; }, acc);
            mm_out<DT>(wave, lane, acc, [&](int i, int j, double v) { if (i < D && j < D) { if (i == j) v += rowp[i]; W[i * LD + j] = v; } });
        }
        __syncthreads();
        for (int idx = tid; idx < DP * DP; idx += PREP_THREADS) P[(idx / DP) * LD + idx % DP] = A_at(idx / DP, idx % DP);
        __syncthreads();
        if (wave < DT)
            mm_acc<DT, DS>(wave, lane, [&](int i, int k) { return P[k * LD + i] * qbar[k]; }, [&](int k, int j) { return P[k * LD + j]; }, acc);
        __syncthreads();            // every read of the staged <A> is done: P takes the result
        if (wave < DT)
            mm_out<DT>(wave, lane, acc, [&](int i, int j, double v) {
                const bool in = i < D && j < D;
                if (in && i == j) v += colp[i];
                P[i * LD + j] = in ? W[i * LD + j] + v : 0.0;
            });
        __syncthreads();
    } else {
    mm<DT, DT, KS>(wave, lane,
                   [&](int i, int k) { return DENSE ? C_at(k, i) : C_at(k, i) * rbar[k]; },
                   [&](int k, int j) { if constexpr (DENSE) return RC_at(k, j); else return C_at(k, j); },
                   [&](int i, int j, double v) { if (i < D && j < D) { if (i == j) v += rowp[i]; W[i * LD + j] = v; } });
    __syncthreads();
    mm<DT, DT, DS>(wave, lane,
                   [&](int i, int k) { return DENSE ? A_at(k, i) : A_at(k, i) * qbar[k]; },
                   [&](int k, int j) { if constexpr (DENSE) return QA_at(k, j); else return A_at(k, j); },
                   [&](int i, int j, double v) { if (i < D && j < D) { if (i == j) v += colp[i]; P[i * LD + j] = W[i * LD + j] + v; } });
    __syncthreads();
    }

    QSTAMP(0);
    // the three posterior precisions, qprec = pprec + (m1 from Mult(C,.) + m1 from Mult(A,.))  gaussian.py:117,
    // inverted together in registers (qcov, gaussian.py:118-119)
    double sig[3][16];
    const int ta = tid >> 4, tb = tid & 15;                         // owner of the 4 x 4 tile (4 ta + ra, 4 tb + cb)
#pragma unroll
    for (int ra = 0; ra < 4; ++ra)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int i = 4 * ta + ra, j = 4 * tb + cb, u = 4 * ra + cb;
            const bool in = i < D && j < D;
            const double mc = in ? W[i * LD + j] : 0.0, mac = in ? P[i * LD + j] : 0.0;
            double qd;
            if constexpr (DENSE) qd = in ? Qd[i * D + j] : 0.0; else qd = (in && i == j) ? qbar[i] : 0.0;
            const double pad = (!in && i == j) ? 1.0 : 0.0;         // identity in the padding keeps pivots finite
            const double x0p = a.x0_prec[(i < D ? i : D - 1) * D + (j < D ? j : D - 1)];       // (unconditional, from a clamped place)
            sig[0][u] = (in ? x0p : 0.0) + mac + pad;
            sig[1][u] = qd + mac + pad;
            sig[2][u] = qd + mc + pad;
        }
    __syncthreads();
    QSTAMP(1);
    gj_inverse<3>(sig, D, tid, gjbuf, gjbuf + 3 * 2 * GJ_BUF);
    QSTAMP(2);
    if (tid < 64) {                                                 // q_ln_det, gaussian.py:120 (quirk Q1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double lp = 0.0;
            if (tid < D) {
                const double piv = gjbuf[3 * 2 * GJ_BUF + c * 64 + tid];
                if (!(piv > 0.0)) atomicOr(a.status, 1);
                lp = log(piv);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 64);
            if (tid == 0) a.qld[(size_t)n * 3 + c] = 0.5 / (0.5 * lp);
        }
    }
    if ((D & 3) == 0) {         // a thread's four entries of a row are 32 contiguous bytes: one store (element by element they were 48 scattered ones)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) {
                const int i = 4 * ta + ra, j0 = 4 * tb;
                if (i < D && j0 < D)
                    *reinterpret_cast<d4*>(a.Sigma + ((size_t)n * 3 + c) * D * D + (size_t)i * D + j0) = d4{sig[c][4 * ra], sig[c][4 * ra + 1], sig[c][4 * ra + 2], sig[c][4 * ra + 3]};
            }
    } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ra = 0; ra < 4; ++ra)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int i = 4 * ta + ra, j = 4 * tb + cb;
                if (i < D && j < D) a.Sigma[((size_t)n * 3 + c) * D * D + i * D + j] = sig[c][4 * ra + cb];
            }
    }

    // the boundary classes are used as they are (one matrix-vector chain per sweep, k_sweep.hip): Sigma_0,
    // Sigma_2, the noise expectations and L0 m0 go into the block
#pragma unroll
    for (int ra = 0; ra < 4; ++ra) {
        const int i = 4 * ta + ra, j0 = 4 * tb;
        if (i < DP && j0 < DP) {
            d4 s0, s2;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const bool in = i < D && j0 + cb < D;
                s0[cb] = in ? sig[0][4 * ra + cb] : 0.0;
                s2[cb] = in ? sig[2][4 * ra + cb] : 0.0;
            }
            *reinterpret_cast<d4*>(g + L.oS0 + (size_t)i * DP + j0) = s0;
            *reinterpret_cast<d4*>(g + L.oS2 + (size_t)i * DP + j0) = s2;
        }
    }
    if (tid < 64) { g[L.oqr + tid] = qbar[tid]; g[L.oqr + 64 + tid] = rbar[tid]; }
    if (tid < DP) {      // L0 m0: the Constant mean parent of X_0 through its Constant precision
        double s = 0.0;
        if (tid < D) for (int j = 0; j < D; ++j) s += a.x0_prec[tid * D + j] * a.x0_mean[j];
        g[L.ow0 + tid] = s;
    }
    QSTAMP(1);
    // gains of the interior class
#pragma unroll
    for (int ra = 0; ra < 4; ++ra)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int i = 4 * ta + ra, j = 4 * tb + cb;
            if (i < D && j < D) P[i * LD + j] = sig[1][4 * ra + cb];
        }
    __syncthreads();
    if constexpr (STAGE) {
        constexpr int KP = 16 * KT;
        // <Q><A> staged in W: F = Sigma (<Q><A>), B = Sigma (<Q><A>)^T (the expectations are symmetric)
        for (int idx = tid; idx < DP * DP; idx += PREP_THREADS) W[(idx / DP) * LD + idx % DP] = qbar[idx / DP] * A_at(idx / DP, idx % DP);
        __syncthreads();
        d4 facc[DT], acc[DT], gacc[KT];
        if (wave < DT) {
            mm_acc<DT, DS>(wave, lane, [&](int i, int k) { return P[i * LD + k]; }, [&](int k, int j) { return W[k * LD + j]; }, facc);
            mm_out<DT>(wave, lane, facc, [&](int i, int j, double v) { if (i < D && j < D) g[L.oFn + pos_nat(i, j, DS)] = v; });
            mm_acc<DT, DS>(wave, lane, [&](int i, int k) { return P[i * LD + k]; }, [&](int k, int j) { return W[j * LD + k]; }, acc);
            mm_out<DT>(wave, lane, acc, [&](int i, int j, double v) { if (i < D && j < D) g[L.oBn + pos_nat(i, j, DS)] = v; });
        }
        __syncthreads();
        // <R><C> staged in W: G = Sigma (<R><C>)^T
        for (int idx = tid; idx < KP * DP; idx += PREP_THREADS) W[(idx / DP) * LD + idx % DP] = rbar[idx / DP] * C_at(idx / DP, idx % DP);
        __syncthreads();
        if (wave < DT) {
            mm_acc<KT, DS>(wave, lane, [&](int i, int k) { return P[i * LD + k]; }, [&](int k, int l) { return W[l * LD + k]; }, gacc);
            mm_out<KT>(wave, lane, gacc, [&](int i, int l, double v) { if (i < D && l < K) g[L.oGp + pos_perm(i, l, KS)] = v; });
        }
        __syncthreads();
        // F, zero padded, into W: the matrix whose powers give the forward warm-up length
        if (wave < DT) mm_out<DT>(wave, lane, facc, [&](int i, int j, double v) { W[i * LD + j] = (i < D && j < D) ? v : 0.0; });
    } else
    {
        // Sigma <Q><A>: multiplies the mean of X_{t-1}
        mm<DT, DT, DS>(wave, lane,
                       [&](int i, int k) { return P[i * LD + k]; },
                       [&](int k, int j) { if constexpr (DENSE) return QA_at(k, j); else return qbar[k] * A_at(k, j); },
                       [&](int i, int j, double v) {
                           if (i < D && j < D) { g[L.oFn + pos_nat(i, j, DS)] = v; W[i * LD + j] = v; }
                       });
        // Sigma <A>^T<Q>: multiplies the mean of X_{t+1}
        mm<DT, DT, DS>(wave, lane,
                       [&](int i, int k) { return P[i * LD + k]; },
                       [&](int k, int j) { if constexpr (DENSE) return QA_at(j, k); else return A_at(j, k) * qbar[j]; },
                       [&](int i, int j, double v) {
                           if (i < D && j < D) g[L.oBn + pos_nat(i, j, DS)] = v;
                       });
        // Sigma <C>^T<R>: multiplies y_t
        mm<DT, KT, DS>(wave, lane,
                       [&](int i, int k) { return P[i * LD + k]; },
                       [&](int k, int l) { if constexpr (DENSE) return RC_at(l, k); else return C_at(l, k) * rbar[l]; },
                       [&](int i, int l, double v) {
                           if (i < D && l < K) g[L.oGp + pos_perm(i, l, KS)] = v;
                       });
    }
    __syncthreads();

    QSTAMP(3);
    // warm-up lengths of the segmented sweeps.  W holds F (forward recurrence matrix); the backward
    // one, B, is read back transposed (powers of B^T in the inf-norm = powers of B in the 1-norm).
    int J = warmup_length<DT>(W, P, LD, D, tid, rowp);
    if (tid == 0) a.warm[n * 2 + 0] = J;
    __syncthreads();
    for (int idx = tid; idx < DP * LD; idx += PREP_THREADS) {
        const int i = idx / LD, j = idx % LD;
        W[idx] = (i < D && j < D) ? g[L.oBn + pos_nat(j, i, DS)] : 0.0;       // B^T
    }
    __syncthreads();
    J = warmup_length<DT>(W, P, LD, D, tid, rowp);
    if (tid == 0) a.warm[n * 2 + 1] = J;
#ifdef PREP64_STAMP
    QSTAMP(4);
    if (blockIdx.x == 100 && tid == 0)
        printf("k_prep: moments %llu | tiles in and out %llu | three inversions %llu | gains %llu | warm-up bounds %llu\n", qs_acc[0], qs_acc[1], qs_acc[2], qs_acc[3], qs_acc[4]);
#endif
}

template <int DT, int KT>
static void launch_prep_t(pyvb_lds* h, const PrepArgs& a) {
    if (h->dense) hipLaunchKernelGGL((k_prep<DT, KT, true>), dim3(h->N), dim3(PREP_THREADS), 0, h->stream, a);
    else hipLaunchKernelGGL((k_prep<DT, KT, false>), dim3(h->N), dim3(PREP_THREADS), 0, h->stream, a);
}

int launch_prep(pyvb_lds* h) {
    if (h->big) return launch_prep_big(h);
    PrepArgs a;
    a.A_mean = h->A_mean; a.A_var = h->A_var; a.C_mean = h->C_mean; a.C_var = h->C_var;
    a.Q_a = h->Q_a; a.Q_b = h->Q_b; a.R_a = h->R_a; a.R_b = h->R_b;
    a.x0_mean = h->pri.x0_mean; a.x0_prec = h->pri.x0_prec;
    a.Sigma = h->Sigma_new; a.qld = h->qld_x_new; a.gains = h->gains; a.scratch = h->scratch;
    a.warm = h->warm; a.status = h->status;
    a.Qbar = h->Qbar; a.QA = h->QA; a.RC = h->RC; a.trA = h->trA; a.trC = h->trC;
    a.N = h->N; a.T = h->T; a.D = h->D; a.K = h->K; a.L = h->L;
    {
        TimedLaunch tl(h, PYVB_K_PREP);
        switch (h->L.DT * 10 + h->L.KT) {
            case 11: launch_prep_t<1, 1>(h, a); break;
            case 12: launch_prep_t<1, 2>(h, a); break;
            case 14: launch_prep_t<1, 4>(h, a); break;
            case 21: launch_prep_t<2, 1>(h, a); break;
            case 22: launch_prep_t<2, 2>(h, a); break;
            case 24: launch_prep_t<2, 4>(h, a); break;
            case 41: launch_prep_t<4, 1>(h, a); break;
            case 42: launch_prep_t<4, 2>(h, a); break;
            case 44: launch_prep_t<4, 4>(h, a); break;
            default: pyvb_set_error("unsupported tile shape"); return PYVB_E_ARG;
        }
    }
    HIPCHK(hipGetLastError());
    return PYVB_OK;
}
