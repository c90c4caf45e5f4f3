// fit_kernels.hip -- GP fit on the GPU, all float64.
//
//   kernel matrix   sklearn/gaussian_process/kernels.py RBF :1553-1560, Matern :1708-1738,
//                   Product :966, Sum :866, WhiteKernel :1402; _gpr.py:346-347 (jitter)
//   Cholesky        _gpr.py:349 (scipy.linalg.cholesky -> LAPACK dpotrf): right-looking blocked
//                   factorisation, NB = 64: LDS diagonal-block potf2 (+ its inverse), MFMA panel
//                   solve, MFMA trailing update
//   inverse factor  Linv = L^-1 by recursive doubling on MFMA (feeds the candidate sweep, which
//                   replaces solve_triangular at _gpr.py:454 by a triangular contraction)
//   alpha, LML      _gpr.py:360-364 and :584-613
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

#include "chol64.hpp"
#include "gemm64_glds.hpp"
#include "gemm_nt_glds.hpp"
#include "mfma_gemm.hpp"
#include "pairwise.hpp"
#include <string.h>

#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// ------------------------------------------------------------------------------------------
// Kernel matrix: the 64x64 tiles on and below the diagonal (nothing reads the ones above it).
// Padding rows/cols (>= N) form an identity block so the padded factor is [[L,0],[0,I]].
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void kernel_matrix_kernel(
    const double *__restrict__ Xs, double *__restrict__ K, int N, int Np, int Dp,
    double constant, double noise, double jitter) {
    __shared__ double Ct[PwCfg<double>::DC][PwCfg<double>::LD];
    __shared__ double Xt[PwCfg<double>::DC][PwCfg<double>::LD];
    int bx = blockIdx.x;
    int tm = (int)((sqrtf(8.0f * (float)bx + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= bx) ++tm;
    while (tm * (tm + 1) / 2 > bx) --tm;
    const int tn = bx - tm * (tm + 1) / 2;
    const int i0 = tm * PW_T, j0 = tn * PW_T;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

    double d2[4][4];
    pairwise_sqdist<double>(Xs, i0, Np, Xs, j0, Np, Dp, Ct, Xt, d2);

#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + 4 * ty + a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + 4 * tx + b;
            double v;
            if (i == j) {
                // np.fill_diagonal(K, 1); constant * K; + noise; + jitter
                v = (i < N) ? ((constant * 1.0 + noise) + jitter) : 1.0;
            } else if (i < N && j < N) {
                v = kernel_value<double, KIND>(d2[a][b], constant);
            } else {
                v = 0.0;
            }
            K[(long)i * Np + j] = v;   // tiles above the diagonal are never read: left unwritten
        }
    }
}

// Diagonal block (factor + inverse) of panel o and, in the same launch, the panel solve of every
// row block below it.  VAR selects the factorisation: 4 / 8 = variant B with that many columns
// per barrier (4 is the default), 0 = variant A (the round-1 form, kept for A/B runs).
template <int VAR>
__global__ __launch_bounds__(256) void panel_kernel(double *__restrict__ K, int Np, int o,
                                                        double *__restrict__ Dinv,
                                                        double *__restrict__ Linv,
                                                        double *__restrict__ Lstage,
                                                        double *__restrict__ scal,
                                                        int *__restrict__ flag, double tiny) {
    // one LDS arena: the factorisation's column / row buffers, later the panel solve's two
    // 64 x 66 operand tiles
    __shared__ __attribute__((aligned(16))) double panel_lds[2 * NB * (NB + 2)];
    double *logs = panel_lds + 4 * 8 * NB;
    const int tid = threadIdx.x;
    const int tc = tid >> 4, tr = tid & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // holds tc = 4*wave .. 4*wave+3
    double a[4][4], x[4][4];
    const double *src = K + (long)(o + 4 * tr) * Np + o + 4 * tc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const d2_t v0 = *reinterpret_cast<const d2_t *>(src + (long)i * Np);
        const d2_t v1 = *reinterpret_cast<const d2_t *>(src + (long)i * Np + 2);
        a[i][0] = v0[0]; a[i][1] = v0[1]; a[i][2] = v1[0]; a[i][3] = v1[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = (tr == tc && i == j) ? 1.0 : 0.0;
    }
    // Workgroup 0 publishes L_kk.  The other workgroups of this launch READ A_kk from K(o, o)
    // whenever they happen to be dispatched, so L_kk must not land there before the launch is
    // over: it is parked in Lstage[k] and moved into K by workgroup 0 of the NEXT panel's launch
    // (stream order = all readers done).  The last panel runs alone (gridDim.x == 1) and writes
    // in place.
    const bool alone = gridDim.x == 1;
    // row blocks below the diagonal: fetch the A_ik tile now, so its latency hides behind the
    // factorisation (it is only needed by the panel solve at the end)
    double *Ablk = K + (long)(o + NB * blockIdx.x) * Np + o;
    d2_t apre[8];
    if (blockIdx.x > 0) {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8;
            apre[p8] = *reinterpret_cast<const d2_t *>(Ablk + (long)(idx >> 5) * Np + (idx & 31) * 2);
        }
    }
    if (VAR == 3 || VAR == 38) {
        double *Ldst = nullptr;
        if (blockIdx.x == 0) Ldst = alone ? K + (long)o * Np + o : Lstage + (long)(o / NB) * NB * NB;
        if (VAR == 38) factor64_v3<8>(a, panel_lds, o, Ldst, alone ? (long)Np : (long)NB, flag, tiny);
        else factor64_v3<4>(a, panel_lds, o, Ldst, alone ? (long)Np : (long)NB, flag, tiny);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = a[i][j];          // the live tile ended as the X tile
    } else if (VAR == 8) {
        factor64_steps<8>(a, x, panel_lds, o, flag, tiny);
    } else if (VAR == 4) {
        factor64_steps<4>(a, x, panel_lds, o, flag, tiny);
    } else {
        factor64_steps4(a, x, panel_lds, o, flag, tiny);
    }
    if (blockIdx.x > 0) {
        // ---- panel solve for row block blockIdx.x - 1 below the diagonal: L_ik = A_ik * X^T ----
        // (every workgroup of the launch factored the diagonal block redundantly above, so X is
        // already in this workgroup's registers; no second launch, no round trip through HBM)
        constexpr int LDP = NB + 2;
        double (*As)[LDP] = reinterpret_cast<double (*)[LDP]>(panel_lds);
        double (*Xs)[LDP] = As + NB;
        __syncthreads();   // colbuf/xbuf (aliased by panel_lds) are no longer read
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8;
            *reinterpret_cast<d2_t *>(&As[idx >> 5][(idx & 31) * 2]) = apre[p8];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                Xs[4 * tr + i][4 * tc + j] = ((4 * tc + j) <= (4 * tr + i)) ? x[i][j] : 0.0;
        __syncthreads();
        // 4 waves, each a 32x32 quadrant of the 64x64 block: 2x2 fragments of v_mfma_f64_16x16x4
        using MF = Mfma<double>;
        const int lane = tid & 63;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
        d4_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
#pragma unroll
        for (int ks = 0; ks < NB; ks += 8) {
            d2_t av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&As[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Xs[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(av[i][e], bv[j][e], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ablk[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)] = acc[i][j][r];
        return;
    }
    // Workgroup 0 publishes the result: X to Dinv[k] and to the diagonal of Linv, L_kk (zeros above
    // the diagonal) to K or Lstage (see the top of the kernel).
    double *dstK = alone ? K + (long)(o + 4 * tr) * Np + o + 4 * tc
                         : Lstage + (long)(o / NB) * NB * NB + (4 * tr) * NB + 4 * tc;
    const long ldK = alone ? (long)Np : (long)NB;
    double *dstL = Linv + (long)(o + 4 * tr) * Np + o + 4 * tc;
    double *dstD = Dinv + (long)(o / NB) * NB * NB + (4 * tr) * NB + 4 * tc;
    if (o > 0) {   // the previous panel's parked L_kk
        const double *prev = Lstage + (long)(o / NB - 1) * NB * NB + (4 * tr) * NB + 4 * tc;
        double *pk = K + (long)(o - NB + 4 * tr) * Np + (o - NB) + 4 * tc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[(long)i * Np + j] = prev[i * NB + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool low = (4 * tc + j) <= (4 * tr + i);
            if (VAR != 3 && VAR != 38) dstK[(long)i * ldK + j] = low ? a[i][j] : 0.0;   // variant C stored L during the loop
            const double xv = low ? x[i][j] : 0.0;
            dstL[(long)i * Np + j] = xv;
            dstD[i * NB + j] = xv;
        }
    // sum(log(diag L)) of this block, fixed-order tree in one wave
    if (VAR == 3 || VAR == 38) {
        // variant C left 1 / L[j][j] in LDS: log L[j][j] = -log(rs[j])
        __syncthreads();
        if (tid < 64) {
            double s = -log(panel_lds[CHOL64_RS_OFF + tid]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (tid == 0) scal[0] += s;
        }
        return;
    }
    if (tr == tc) logs[tr] = (log(a[0][0]) + log(a[1][1])) + (log(a[2][2]) + log(a[3][3]));
    __syncthreads();
    if (tid < 64) {
        double s = tid < 16 ? logs[tid] : 0.0;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (tid == 0) scal[0] += s;
    }
}

// Panel kernel on variant D of the diagonal-block factorisation (chol64.hpp: block in LDS, one wave
// factors + inverts 16 x 16 pivot blocks, MFMA for the rest).  Same contract as panel_kernel.
__global__ __launch_bounds__(256) void panel_d_kernel(double *__restrict__ K, int Np, int o,
                                                      double *__restrict__ Dinv,
                                                      double *__restrict__ Linv,
                                                      double *__restrict__ Lstage,
                                                      double *__restrict__ scal,
                                                      int *__restrict__ flag, double tiny) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + NB + 32];
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double (*Tb)[CH_LD] = Xt + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool alone = gridDim.x == 1;          // see panel_kernel: where L_kk may be written
    const d2_t z2 = {0.0, 0.0};
    const double *Akk = K + (long)o * Np + o;
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        *reinterpret_cast<d2_t *>(&At[r][c2]) = *reinterpret_cast<const d2_t *>(Akk + (long)r * Np + c2);
        *reinterpret_cast<d2_t *>(&Xt[r][c2]) = z2;
    }
    double *Ablk = K + (long)(o + NB * blockIdx.x) * Np + o;
    d2_t apre[8];
    if (blockIdx.x > 0) {   // the A_ik tile of this workgroup's row block, fetched behind the factorisation
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8;
            apre[p8] = *reinterpret_cast<const d2_t *>(Ablk + (long)(idx >> 5) * Np + (idx & 31) * 2);
        }
    }
    __syncthreads();
    factor64_v4(At, Xt, Tb, rsbuf, o, flag, tiny);
    if (blockIdx.x > 0) {
        // ---- panel solve: L_ik = A_ik X^T, X already sits in LDS ----
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8;
            *reinterpret_cast<d2_t *>(&Tb[idx >> 5][(idx & 31) * 2]) = apre[p8];
        }
        __syncthreads();
        using MF = Mfma<double>;
        const int lane = tid & 63;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
        d4_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
#pragma unroll
        for (int ks = 0; ks < NB; ks += 8) {
            d2_t av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&Tb[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Xt[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(av[i][e], bv[j][e], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ablk[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)] = acc[i][j][r];
        return;
    }
    // ---- workgroup 0 publishes: L_kk (zeros above the diagonal) to K or Lstage, X to Dinv and Linv ----
    double *dstK = alone ? K + (long)o * Np + o : Lstage + (long)(o / NB) * NB * NB;
    const long ldK = alone ? (long)Np : (long)NB;
    double *dstL = Linv + (long)o * Np + o;
    double *dstD = Dinv + (long)(o / NB) * NB * NB;
    const double *prev = Lstage + (long)(o / NB - 1) * NB * NB;
    double *pk = K + (long)(o - NB) * Np + (o - NB);
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        if (o > 0)   // the previous panel's parked L_kk
            *reinterpret_cast<d2_t *>(pk + (long)r * Np + c2) = *reinterpret_cast<const d2_t *>(prev + r * NB + c2);
        d2_t lv = *reinterpret_cast<const d2_t *>(&At[r][c2]);
        lv[0] = (c2 <= r) ? lv[0] : 0.0;
        lv[1] = (c2 + 1 <= r) ? lv[1] : 0.0;
        *reinterpret_cast<d2_t *>(dstK + (long)r * ldK + c2) = lv;
        const d2_t xv = *reinterpret_cast<const d2_t *>(&Xt[r][c2]);
        *reinterpret_cast<d2_t *>(dstL + (long)r * Np + c2) = xv;
        *reinterpret_cast<d2_t *>(dstD + r * NB + c2) = xv;
    }
    if (tid < 64) {   // sum(log(diag L)), fixed-order tree
        double s = log(At[tid][tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (tid == 0) scal[0] += s;
    }
}

// z[i] = sum_{j<=i} Linv[i][j] * v[j]   (one wave per row)
__global__ __launch_bounds__(256) void gemv_lower_rows_kernel(const double *__restrict__ Linv,
                                                              const double *__restrict__ v,
                                                              double *__restrict__ z, int Np) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Np) return;
    double s = 0.0;
    const double *row = Linv + (long)i * Np;
    for (int jj = lane; jj <= i; jj += 64) s = fma(row[jj], v[jj], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) z[i] = s;
}

// partial[y][j] = sum over the y-th row slice of Linv[i][j] * z[i], i >= j
// (64 columns x GEMV_RS row slices per grid; 4 row lanes per column inside a block)
constexpr int GEMV_RS = 16;
__global__ __launch_bounds__(256) void gemv_lower_cols_kernel(const double *__restrict__ Linv,
                                                              const double *__restrict__ z,
                                                              double *__restrict__ partial, int Np) {
    __shared__ double red[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, j = j0 + c;
    const int rows = Np - j0;                       // rows j0 .. Np-1 can be non-zero
    const int per = (rows + GEMV_RS - 1) / GEMV_RS;
    const int i0 = j0 + blockIdx.y * per;
    const int i1 = (i0 + per < Np) ? i0 + per : Np;
    double s = 0.0;
    for (int i = i0 + rg; i < i1; i += 4)
        if (i >= j) s = fma(Linv[(long)i * Np + j], z[i], s);
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0) partial[(long)blockIdx.y * Np + j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// alpha[j] = sum_y partial[y][j] (fixed order);  scal[1] = yn . alpha   (single block)
__global__ __launch_bounds__(256) void alpha_finish_kernel(const double *__restrict__ partial,
                                                           const double *__restrict__ yn,
                                                           double *__restrict__ alpha,
                                                           double *__restrict__ scal, int Np,
                                                           const int *__restrict__ flag = nullptr,
                                                           double *__restrict__ res_host = nullptr) {
    __shared__ double red[256];
    double d = 0.0;
    for (int j = threadIdx.x; j < Np; j += 256) {
        double a = 0.0;
#pragma unroll
        for (int y = 0; y < GEMV_RS; ++y) a += partial[(long)y * Np + j];
        alpha[j] = a;
        d = fma(yn[j], a, d);
    }
    red[threadIdx.x] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scal[1] = red[0];
        if (res_host) {   // device-mapped host memory: the fit's scalars land there without a D2H copy
            res_host[0] = scal[0];
            res_host[1] = red[0];
            res_host[2] = (double)*flag;
        }
    }
}

// ---- the same two products taken row block by row block (the inverse behind the panel chain) ----
// z[i] = sum_{j<=i} Linv[i][j] * v[j] for the rows from row0 on (one workgroup per row, four loads in
// flight per lane) and, for an f32 sweep, the f32 copy of that row (all Np columns: the zeros right of
// the diagonal included)
__global__ __launch_bounds__(256) void rowblock_finish_kernel(const double *__restrict__ Linv,
                                                              const double *__restrict__ v,
                                                              double *__restrict__ z,
                                                              float *__restrict__ Linv32, int Np, int row0) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = row0 + blockIdx.x;
    const double *row = Linv + (long)i * Np;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int jj = tid;
    for (; jj + 768 <= i; jj += 1024) {
        const double a0 = row[jj], a1 = row[jj + 256], a2 = row[jj + 512], a3 = row[jj + 768];
        s0 = fma(a0, v[jj], s0);
        s1 = fma(a1, v[jj + 256], s1);
        s2 = fma(a2, v[jj + 512], s2);
        s3 = fma(a3, v[jj + 768], s3);
    }
    for (; jj <= i; jj += 256) s0 = fma(row[jj], v[jj], s0);
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[tid >> 6] = s;
    if (Linv32) {
        float *o = Linv32 + (long)i * Np;
        for (int j2 = tid * 2; j2 < Np; j2 += 512) {
            const d2_t x = *reinterpret_cast<const d2_t *>(row + j2);
            float2 y;
            y.x = (float)x[0];
            y.y = (float)x[1];
            *reinterpret_cast<float2 *>(o + j2) = y;
        }
    }
    __syncthreads();
    if (tid == 0) z[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[s][j] = sum over the 128-row slice s = i / 128 of Linv[i][j] * z[i], i >= j  (64 columns per
// workgroup, blockIdx.y = slice inside the row block that starts at row0); zsq[s] = sum of z[i]^2
// over the slice (yn . alpha = z . z)
constexpr int GEMV_SLICE = 128;
__global__ __launch_bounds__(256) void rowblock_cols_kernel(const double *__restrict__ Linv,
                                                            const double *__restrict__ z,
                                                            double *__restrict__ partial,
                                                            double *__restrict__ zsq, int Np, int row0) {
    __shared__ double red[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const int i0 = row0 + blockIdx.y * GEMV_SLICE;
    const double *col = Linv + j;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < GEMV_SLICE / 4; q += 8) {   // rows i0 + rg + 4 (q + u): eight loads in flight
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + rg + 4 * (q + u);
            a[u] = (i >= j) ? col[(long)i * Np] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            s0 = fma(a[u], z[i0 + rg + 4 * (q + u)], s0);
            s1 = fma(a[u + 1], z[i0 + rg + 4 * (q + u + 1)], s1);
        }
    }
    red[rg][c] = s0 + s1;
    __syncthreads();
    if (rg == 0) partial[(long)(i0 / GEMV_SLICE) * Np + j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (blockIdx.x == 0 && threadIdx.x < 64) {   // fixed-order tree over the slice's 128 rows
        const double za = z[i0 + threadIdx.x], zb = z[i0 + 64 + threadIdx.x];
        double q = fma(za, za, zb * zb);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        if (threadIdx.x == 0) zsq[i0 / GEMV_SLICE] = q;
    }
}

// alpha[j] = sum of partial[s][j] over the slices that reach column j, s = j / 128 ... (fixed order:
// four interleaved groups of slices, then the groups); workgroup 0 also leaves scal[1] = yn . alpha
// = z . z = sum of zsq (fixed order) and the fit's scalars
__global__ __launch_bounds__(256) void alpha_finish_sliced_kernel(const double *__restrict__ partial,
                                                                  const double *__restrict__ zsq,
                                                                  double *__restrict__ alpha,
                                                                  double *__restrict__ scal, int Np,
                                                                  const int *__restrict__ flag,
                                                                  double *__restrict__ res_host) {
    __shared__ double red[4][64];
    const int ns = Np / GEMV_SLICE;
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    double a = 0.0;
    for (int y = j / GEMV_SLICE + rg; y < ns; y += 4) a += partial[(long)y * Np + j];
    red[rg][c] = a;
    __syncthreads();
    if (rg == 0) alpha[j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double d = 0.0;
        for (int y = threadIdx.x; y < ns; y += 64) d += zsq[y];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        if (threadIdx.x == 0) {
            scal[1] = d;
            if (res_host) {
                res_host[0] = scal[0];
                res_host[1] = d;
                res_host[2] = (double)*flag;
            }
        }
    }
}

// U[blk] = Linv[blk]^T for every 128x128 diagonal block (32x32 LDS tiles)
__global__ __launch_bounds__(256) void transpose_diag128_kernel(const double *__restrict__ Linv,
                                                                double *__restrict__ U, int Np) {
    __shared__ double t[32][33];
    const int blk = blockIdx.x, tile = blockIdx.y;          // 16 tiles per block
    const int tr = tile >> 2, tc = tile & 3;
    const long base = (long)blk * 128 * ((long)Np + 1);
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = y + 8 * k;
        t[r][x] = Linv[base + (long)(32 * tr + r) * Np + 32 * tc + x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = y + 8 * k;
        U[base + (long)(32 * tc + r) * Np + 32 * tr + x] = t[x][r];
    }
}

// The 64 -> 128 level of the inverse in ONE launch (round 5; it was three: two 64 x 64 products on the generic template
// and the transpose): workgroup b takes the 128-block at row o + 128 b,
//     T = L21 X11,   Linv21 = -X22 T,   U[block] = Linv[block]^T
// with the tiles in LDS and the products on tile_mma64 (the template's k order: the same bits).  Saves two launches
// (~12 us) per fit where the inverse is not hidden behind the panel chain (Np <= 512: every evaluation of a mid-size
// hyper-parameter fit, C1).
__global__ __launch_bounds__(256) void level64_fused_kernel(const double *__restrict__ K, double *__restrict__ Linv,
                                                            double *__restrict__ U, int Np, long o) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD];
    typedef double (*tile_t)[CH_LD];
    tile_t A = reinterpret_cast<tile_t>(lds), B = A + NB, Tt = B + NB;
    const int tid = threadIdx.x;
    const long d = (o + (long)blockIdx.x * 2 * NB) * ((long)Np + 1);
    const double *L21 = K + d + (long)NB * Np;
    const double *X11 = Linv + d;
    const double *X22 = Linv + d + (long)NB * Np + NB;
    double *Ub = U + d;
    // A = L21 [i][k];  B[j][k] = X11[k][j] (= U11[j][k]: stored as the block's upper-left tile of U on the way)
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        *reinterpret_cast<d2_t *>(&A[r][c2]) = *reinterpret_cast<const d2_t *>(L21 + (long)r * Np + c2);
        const d2_t v = *reinterpret_cast<const d2_t *>(X11 + (long)r * Np + c2);
        B[c2][r] = v[0];
        B[c2 + 1][r] = v[1];
    }
    __syncthreads();
    d4_t acc[2][2];
    acc_zero(acc);
    tile_mma64(A, B, acc);                                   // T[i][j] = sum_k L21[i][k] X11[k][j]
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {                         // U11 = X11^T, and zeros below it (the transpose of Linv's zero upper-right tile)
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        *reinterpret_cast<d2_t *>(Ub + (long)r * Np + c2) = *reinterpret_cast<const d2_t *>(&B[r][c2]);
        const d2_t z = {0.0, 0.0};
        *reinterpret_cast<d2_t *>(Ub + (long)(NB + r) * Np + c2) = z;
    }
    acc_foreach(acc, [&](int r, int c, double v) { Tt[c][r] = v; });   // T^T: the second product's B operand, [j][k] = T[k][j]
    __syncthreads();                                         // A's readers are done, Tt is complete
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        *reinterpret_cast<d2_t *>(&A[r][c2]) = *reinterpret_cast<const d2_t *>(X22 + (long)r * Np + c2);
    }
    __syncthreads();
    acc_zero(acc);
    tile_mma64(A, Tt, acc);                                  // (X22 T)[i][j]
    double *L21inv = Linv + d + (long)NB * Np;
    acc_foreach(acc, [&](int r, int c, double v) {
        L21inv[(long)r * Np + c] = -v;                       // Linv21
        Ub[(long)c * Np + NB + r] = -v;                      // ... and its transpose, the block's upper-right tile of U
    });
    for (int idx = tid; idx < NB * NB; idx += 256) {         // U22 = X22^T
        const int r = idx >> 6, c = idx & 63;
        Ub[(long)(NB + r) * Np + NB + c] = A[c][r];
    }
}

__global__ void f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i];
}

hipError_t launch_f64_to_f32(Context &c, const double *in, float *out, long n) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(blocks), dim3(256), 0, c.stream, in, out, n);
    return hipGetLastError();
}

// ---- one-row append: K' = [[K, k], [k^T, kappa]] ------------------------------------
// k[j] = c * k0(x_new, x_j) for j < n_old, 0 beyond
template <int KIND>
__global__ __launch_bounds__(256) void kvec_kernel(const double *__restrict__ Xs, double *__restrict__ k,
                                                   int n_old, int Np, int Dp, double constant) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    double v = 0.0;
    if (j < n_old) {
        const double *xn = Xs + (long)n_old * Dp;
        const double *xj = Xs + (long)j * Dp;
        double d2 = 0.0;
        for (int d = 0; d < Dp; ++d) {
            const double df = xn[d] - xj[d];
            d2 = fma(df, df, d2);
        }
        v = kernel_value<double, KIND>(d2, constant);
    }
    k[j] = v;
}

// pivot of the new row: lambda^2 = kappa - l.l ; writes L row, scal[2] = 1/lambda, scal[3] = log(lambda)
__global__ __launch_bounds__(256) void append_pivot_kernel(const double *__restrict__ l,
                                                           double *__restrict__ K, double *__restrict__ scal,
                                                           int *__restrict__ flag, int n_old, int Np,
                                                           double kappa, double tiny) {
    __shared__ double red[256];
    double s = 0.0;
    for (int j = threadIdx.x; j < n_old; j += 256) s = fma(l[j], l[j], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    double piv = kappa - red[0];
    if (!(piv > tiny) || !isfinite(piv)) {
        if (threadIdx.x == 0) *flag = n_old + 1;
        piv = 1.0;
    }
    const double lam = sqrt(piv);
    double *row = K + (long)n_old * Np;
    for (int j = threadIdx.x; j < Np; j += 256) row[j] = (j < n_old) ? l[j] : (j == n_old ? lam : 0.0);
    if (threadIdx.x == 0) { scal[2] = 1.0 / lam; scal[3] = log(lam); }
}

// Linv row n_old = [-(Linv^T l) / lambda, 1 / lambda, 0...] from the column partials
__global__ __launch_bounds__(256) void append_inv_row_kernel(const double *__restrict__ partial,
                                                             const double *__restrict__ scal,
                                                             double *__restrict__ Linv,
                                                             float *__restrict__ Linv32,
                                                             const double *__restrict__ Xs,
                                                             float *__restrict__ Xs32, int n_old, int Np,
                                                             int Dp) {
    const double rl = scal[2];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < Np; j += gridDim.x * 256) {
        double v = 0.0;
        if (j < n_old) {
            double a = 0.0;
#pragma unroll
            for (int y = 0; y < GEMV_RS; ++y) a += partial[(long)y * Np + j];
            v = -a * rl;
        } else if (j == n_old) {
            v = rl;
        }
        Linv[(long)n_old * Np + j] = v;
        if (Linv32) Linv32[(long)n_old * Np + j] = (float)v;
        if (Xs32 && j < Dp) Xs32[(long)n_old * Dp + j] = (float)Xs[(long)n_old * Dp + j];
    }
}

hipError_t launch_fit_append(Context &c, int n_old) {
    hipStream_t s = c.stream;
    const int Np = (int)c.Np, Dp = (int)c.Dp;
    TGP_TRY(hipMemsetAsync(c.d_flag, 0, sizeof(int), s));
    const dim3 kg((Np + 255) / 256);
    switch (c.kernel) {
        case TGP_RBF: hipLaunchKernelGGL(kvec_kernel<TGP_RBF>, kg, dim3(256), 0, s, c.d_Xs, c.d_t1, n_old, Np, Dp, c.constant); break;
        case TGP_MATERN12: hipLaunchKernelGGL(kvec_kernel<TGP_MATERN12>, kg, dim3(256), 0, s, c.d_Xs, c.d_t1, n_old, Np, Dp, c.constant); break;
        case TGP_MATERN32: hipLaunchKernelGGL(kvec_kernel<TGP_MATERN32>, kg, dim3(256), 0, s, c.d_Xs, c.d_t1, n_old, Np, Dp, c.constant); break;
        default: hipLaunchKernelGGL(kvec_kernel<TGP_MATERN52>, kg, dim3(256), 0, s, c.d_Xs, c.d_t1, n_old, Np, Dp, c.constant); break;
    }
    TGP_TRY(hipGetLastError());
    // l = Linv k  (the new row of L), then its pivot
    hipLaunchKernelGGL(gemv_lower_rows_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, c.d_Linv, c.d_t1, c.d_t2, Np);
    TGP_TRY(hipGetLastError());
    const double kappa = (c.constant * 1.0 + c.noise) + c.jitter;
    const double tiny = 8.0 * 2.220446049250313e-16 * kappa;
    hipLaunchKernelGGL(append_pivot_kernel, dim3(1), dim3(256), 0, s, c.d_t2, c.d_K, c.d_scal, c.d_flag, n_old, Np, kappa, tiny);
    TGP_TRY(hipGetLastError());
    // new row of Linv = [-(Linv^T l) / lambda, 1 / lambda]
    hipLaunchKernelGGL(gemv_lower_cols_kernel, dim3(Np / 64, GEMV_RS), dim3(256), 0, s, c.d_Linv, c.d_t2, c.d_W, Np);
    TGP_TRY(hipGetLastError());
    hipLaunchKernelGGL(append_inv_row_kernel, dim3(16), dim3(256), 0, s, c.d_W, c.d_scal, c.d_Linv,
                       c.dtype != TGP_F64 ? c.d_Linv32 : nullptr, c.d_Xs, c.dtype != TGP_F64 ? c.d_Xs32 : nullptr,
                       n_old, Np, Dp);
    TGP_TRY(hipGetLastError());
    // alpha = Linv^T (Linv yn) with the re-normalised targets, yn . alpha
    hipLaunchKernelGGL(gemv_lower_rows_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, c.d_Linv, c.d_yn, c.d_z, Np);
    TGP_TRY(hipGetLastError());
    hipLaunchKernelGGL(gemv_lower_cols_kernel, dim3(Np / 64, GEMV_RS), dim3(256), 0, s, c.d_Linv, c.d_z, c.d_W, Np);
    TGP_TRY(hipGetLastError());
    hipLaunchKernelGGL(alpha_finish_kernel, dim3(1), dim3(256), 0, s, c.d_W, c.d_yn, c.d_alpha, c.d_scal, Np, nullptr, nullptr);
    TGP_TRY(hipGetLastError());
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------
template <int BM, int BN, bool BK_MAJOR, int KR, int TMAP>
static hipError_t launch_gemm64(hipStream_t s, int device, const GemmArgs &g, int nblocks, int batch) {
    constexpr int BK = 16;
    // every NT product (both operands K-contiguous) goes to the direct-to-LDS kernel of gemm64_glds.hpp (round 4);
    // TGP_GEMM64=reg keeps the register-staged template (A/B; same k order, bit-identical results).  The two NN
    // products of the inverse's 64 -> 128 level (K = 64) stay on the template.
    // A debug build (make debug: -DTGP_DEBUG_KERNELS, libturbogp_dbg.so -- never the shipped library) adds the
    // routes tools/repeat_fit.py bisected the k-loop race with: reg-trail / glds-trail = only the trailing update on
    // the template / on the direct-to-LDS kernel; wait0 = its barriers drain every DMA; round4-war = its k-loop as it
    // was before the LDS reads were awaited in front of the barrier (one fit in ten wrong at N = 5000).
    if constexpr (BK_MAJOR && BM == 64 && BN == 64 && (KR == KR_FULL || KR == KR_LOWER_A || KR == KR_UPPER_A)) {
        bool reg = tuning().gemm64_reg;
#ifdef TGP_DEBUG_KERNELS
        static const char *sel = getenv("TGP_GEMM64") ? getenv("TGP_GEMM64") : "";
        constexpr bool trail = KR == KR_FULL && TMAP == TM_LOWER;
        reg = reg || (!strcmp(sel, "reg-trail") && trail) || (!strcmp(sel, "glds-trail") && !trail);
        if (!reg && !strcmp(sel, "wait0")) return launch_gemm64_glds<KR, TMAP, 3, 4>(s, device, g, nblocks, batch);
        if (!reg && !strcmp(sel, "round4-war")) return launch_gemm64_glds<KR, TMAP, 3, 5>(s, device, g, nblocks, batch);
#endif
        if (!reg) return launch_gemm64_glds<KR, TMAP>(s, device, g, nblocks, batch);
    }
    auto kern = mfma_gemm_kernel<double, BM, BN, BK, BK_MAJOR, KR, TMAP, EP_STORE>;
    constexpr size_t lds = gemm_lds_bytes<double, BM, BN, BK>();
    static LdsOptIn opt_in;
    TGP_TRY(opt_in.ensure(reinterpret_cast<const void *>(kern), device, lds));
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, batch), dim3(256), lds, s, g);
    return hipGetLastError();
}

// Rank-64 update inside an outer block: A[i][j] -= L_ik * L_jk^T for the row blocks i below panel k
// and the ncol column blocks j right of it (j <= i; tiles above the diagonal are skipped).  One
// 64 x 64 tile per workgroup, both 64 x 64 operands fetched in one round trip, 16 MFMA k-steps.
// The generic template spends most of its ~12 us on pipeline prologue for so short a k-range.
constexpr int R64_LDP = NB + 2;
__device__ __forceinline__ void rank64_tile(double *__restrict__ K, int Np, int o, int bi, int bj,
                                            double (*As)[R64_LDP], double (*Bs)[R64_LDP]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *A = K + (long)(o + NB * (1 + bi)) * Np + o;
    const double *B = K + (long)(o + NB * (1 + bj)) * Np + o;
    double *C = K + (long)(o + NB * (1 + bi)) * Np + o + NB * (1 + bj);
    d2_t ra[8], rb[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int idx = tid + 256 * p;
        ra[p] = *reinterpret_cast<const d2_t *>(A + (long)(idx >> 5) * Np + (idx & 31) * 2);
        rb[p] = *reinterpret_cast<const d2_t *>(B + (long)(idx >> 5) * Np + (idx & 31) * 2);
    }
    using MF = Mfma<double>;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    d4_t acc[2][2];   // starts as the old C tile: the MFMAs then subtract in place
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[i][j][r] = C[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int idx = tid + 256 * p;
        *reinterpret_cast<d2_t *>(&As[idx >> 5][(idx & 31) * 2]) = ra[p];
        *reinterpret_cast<d2_t *>(&Bs[idx >> 5][(idx & 31) * 2]) = rb[p];
    }
    __syncthreads();
    const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
#pragma unroll
    for (int ks = 0; ks < NB; ks += 8) {
        d2_t av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&As[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Bs[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(-av[i][e], bv[j][e], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)] = acc[i][j][r];
}

__global__ __launch_bounds__(256, 2) void rank64_update_kernel(double *__restrict__ K, int Np, int o, int ncol) {
    __shared__ __attribute__((aligned(16))) double As[NB][R64_LDP];
    __shared__ __attribute__((aligned(16))) double Bs[NB][R64_LDP];
    const int bi = blockIdx.x / ncol, bj = blockIdx.x - bi * ncol;
    if (bj > bi) return;
    rank64_tile(K, Np, o, bi, bj, As, Bs);
}

// ------------------------------------------------------------------------------------------
// The panel chain with the pivot taken off the update's back (TGP_PANEL_LA, default on):
//   pivot_update_kernel(o, upd)   workgroup 0: the diagonal block at o -- with upd its pending
//                                 rank-64 update from the panel at o - NB applied in LDS first --
//                                 factored and inverted by one wave + MFMA (chol64.hpp), L_kk
//                                 written into K, X = L_kk^-1 into Dinv and Linv;
//                                 every other workgroup: one 64 x 64 tile of the rank-64 update
//                                 of the panel at o - NB (all tiles but that diagonal block)
//   panel_solve_kernel(o)         L_ik = A_ik X^T for the row blocks below, X read from Dinv
// Per panel the chain is solve (4 us) + pivot (14 us) with the update (7 us) hidden beside the
// pivot, instead of panel_d_kernel (16 us: every workgroup factoring the same block before its
// solve) followed by the update (7 us).  Same arithmetic in the same order: bit-identical results.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pivot_update_kernel(double *__restrict__ K, int Np, int o, int upd, int ncol,
                                                           double *__restrict__ Dinv, double *__restrict__ Linv,
                                                           double *__restrict__ scal, int *__restrict__ flag,
                                                           double tiny) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + NB + 32];
    static_assert(CH_LD == R64_LDP, "the update tiles reuse the pivot's LDS");
    if (blockIdx.x > 0) {
        // tile t of the update with the panel at o - NB, skipping the diagonal block (workgroup 0 has it)
        const int t = blockIdx.x - 1;
        const int bi = t / ncol, bj = t - bi * ncol;
        if (bj > bi || (bi == 0 && bj == 0)) return;
        rank64_tile(K, Np, o - NB, bi, bj, reinterpret_cast<double (*)[R64_LDP]>(lds),
                    reinterpret_cast<double (*)[R64_LDP]>(lds + NB * R64_LDP));
        return;
    }
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double (*Tb)[CH_LD] = Xt + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const d2_t z2 = {0.0, 0.0};
    double *Akk = K + (long)o * Np + o;
    if (upd) {
        // A_kk -= L_r L_r^T, L_r = rows o .. o+63 of the previous panel: the arithmetic of rank64_tile(0, 0)
        const double *Lr = K + (long)o * Np + (o - NB);
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&Tb[r][c2]) = *reinterpret_cast<const d2_t *>(Lr + (long)r * Np + c2);
            *reinterpret_cast<d2_t *>(&Xt[r][c2]) = z2;
        }
        using MF = Mfma<double>;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        d4_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = Akk[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)];
        __syncthreads();
        const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
#pragma unroll
        for (int ks = 0; ks < NB; ks += 8) {
            d2_t av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&Tb[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Tb[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(-av[i][e], bv[j][e], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    At[wm0 + 16 * i + MF::c_row(lane, r)][wn0 + 16 * j + MF::c_col(lane)] = acc[i][j][r];
    } else {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&At[r][c2]) = *reinterpret_cast<const d2_t *>(Akk + (long)r * Np + c2);
            *reinterpret_cast<d2_t *>(&Xt[r][c2]) = z2;
        }
    }
    __syncthreads();
    factor64_v4(At, Xt, Tb, rsbuf, o, flag, tiny);
    // publish: L_kk (zeros above the diagonal) into K, X to Dinv and Linv
    double *dstL = Linv + (long)o * Np + o;
    double *dstD = Dinv + (long)(o / NB) * NB * NB;
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        d2_t lv = *reinterpret_cast<const d2_t *>(&At[r][c2]);
        lv[0] = (c2 <= r) ? lv[0] : 0.0;
        lv[1] = (c2 + 1 <= r) ? lv[1] : 0.0;
        *reinterpret_cast<d2_t *>(Akk + (long)r * Np + c2) = lv;
        const d2_t xv = *reinterpret_cast<const d2_t *>(&Xt[r][c2]);
        *reinterpret_cast<d2_t *>(dstL + (long)r * Np + c2) = xv;
        *reinterpret_cast<d2_t *>(dstD + r * NB + c2) = xv;
    }
    if (tid < 64) {   // sum(log(diag L)), fixed-order tree
        double s = log(At[tid][tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (tid == 0) scal[0] += s;
    }
}

__global__ __launch_bounds__(256, 2) void panel_solve_kernel(double *__restrict__ K, int Np, int o,
                                                          const double *__restrict__ Dinv) {
    __shared__ __attribute__((aligned(16))) double Xt[NB][CH_LD];
    __shared__ __attribute__((aligned(16))) double Tb[NB][CH_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Ablk = K + (long)(o + NB * (1 + blockIdx.x)) * Np + o;
    const double *X = Dinv + (long)(o / NB) * NB * NB;
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        *reinterpret_cast<d2_t *>(&Tb[r][c2]) = *reinterpret_cast<const d2_t *>(Ablk + (long)r * Np + c2);
        *reinterpret_cast<d2_t *>(&Xt[r][c2]) = *reinterpret_cast<const d2_t *>(X + r * NB + c2);
    }
    __syncthreads();
    using MF = Mfma<double>;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
    d4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
#pragma unroll
    for (int ks = 0; ks < NB; ks += 8) {
        d2_t av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&Tb[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Xt[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(av[i][e], bv[j][e], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ablk[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)] = acc[i][j][r];
}

// ------------------------------------------------------------------------------------------
// ONE launch per panel (round 3, TGP_PANEL_FUSE, default on): the panel solve folded into the
// launch that carries the next pivot and the rank-64 update, so the chain per panel is one launch
// instead of two (panel_solve_kernel 6 us + pivot_update_kernel 18 us -> 19 us).
//   mode 0  (first panel of an outer block, pivot at o)   workgroup 0: the pivot alone;
//           workgroup 1 + i: copies the unsolved block A_i of the panel into Apan_out[i]
//   mode 1  (panel at o, its X = L_kk^-1 in Dinv)
//           workgroup 0: L_r = A_r X^T for the first row block below (written in place), the diagonal
//           block at o + NB minus L_r L_r^T in LDS, its factor and inverse, published;
//           workgroup 1 + t, tile (bi, bj) of the update: L_bi = A_bi X^T and L_bj = A_bj X^T solved
//           here (each tile solves its own two operands: no hand-over between workgroups), then
//           C_ij -= L_bi L_bj^T.  The tiles of column 0 (bj = 0) also write L_bi into K in place and
//           leave their updated tile -- the NEXT panel's unsolved block -- in Apan_out as well.
// Why the copy: the tiles read the unsolved A blocks of the panel while the column-0 tiles write
// the solved ones over them.  Everybody reads the panel from Apan_in (written by the previous
// launch, read-only here), so the in-place stores race with nothing.  Two slots used in turn.
// X is not staged in LDS: every wave reads the 32 rows of X its quadrant multiplies with straight
// into registers (B fragments of v_mfma_f64_16x16x4, 16 ds-free loads per lane from L2).
// Same MFMA sequences as panel_solve_kernel / rank64_tile / pivot_update_kernel: bit-identical L.
// ------------------------------------------------------------------------------------------
constexpr unsigned FUSED_STAMP_STRIDE = 2048;   // debug stamps: entries per launch
// where a wave runs: HW_ID (wave / SIMD / CU / SH / SE ids ...) in the low word, XCC_ID in the high one (debug stamps only)
__device__ __forceinline__ unsigned long long hw_where() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // hwreg(HW_REG_HW_ID, 0, 32)
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // hwreg(HW_REG_XCC_ID, 0, 32)
    return ((unsigned long long)xcc << 32) | hw;
}
// debug (TGP_STAMP_FILE): which CUs a stream reaches -- many short workgroups, each leaving where it ran
__global__ void hw_probe_kernel(unsigned long long *__restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = hw_where() | (1ull << 63);   // (bit 63: the slot was written)
}
__global__ __launch_bounds__(256) void fused_panel_kernel(double *__restrict__ K, int Np, int o, int mode, int ncol,
                                                          const double *__restrict__ Apan_in,
                                                          double *__restrict__ Apan_out,
                                                          double *__restrict__ Dinv, double *__restrict__ Linv,
                                                          double *__restrict__ scal, int *__restrict__ flag,
                                                          double tiny, unsigned long long *__restrict__ stamp) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + NB + 32];
    // debug stamps (TGP_STAMP_FILE; null otherwise): per workgroup [start, end] in 10 ns ticks at
    // stamp[8 + 2 b], workgroup 0's phases at stamp[0..4]
    // (a launch's slot is FUSED_STAMP_STRIDE entries: workgroups past its end do not stamp)
    if (blockIdx.x >= (FUSED_STAMP_STRIDE - 8) / 2) stamp = nullptr;
    struct Stamp {
        unsigned long long *p; int b;
        __device__ ~Stamp() { if (p && threadIdx.x == 0) p[8 + 2 * b + 1] = wall_clock64(); }
    } stamp_guard{stamp, (int)blockIdx.x};
    if (stamp && threadIdx.x == 0) stamp[8 + 2 * blockIdx.x] = wall_clock64();
    if (stamp && threadIdx.x == 0 && blockIdx.x == 0) stamp[3] = hw_where();   // the pivot workgroup's CU (round 6)
    using MF = Mfma<double>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
    const double *X = Dinv + (long)(o / NB) * NB * NB;
    if (blockIdx.x > 0) {
        const int t = blockIdx.x - 1;
        if (mode == 0) {   // copy the unsolved block t of the panel at o
            const double *src = K + (long)(o + NB * (1 + t)) * Np + o;
            double *dst = Apan_out + (long)t * NB * NB;
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) {
                const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
                *reinterpret_cast<d2_t *>(dst + r * NB + c2) = *reinterpret_cast<const d2_t *>(src + (long)r * Np + c2);
            }
            return;
        }
        const int bi = t / ncol, bj = t - bi * ncol;
        if (bj > bi || (bi == 0 && bj == 0)) return;
        double (*As)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
        double (*Bs)[CH_LD] = As + NB;
        const double *Ai = Apan_in + (long)bi * NB * NB;
        const double *Aj = Apan_in + (long)bj * NB * NB;
        double *C = K + (long)(o + NB * (1 + bi)) * Np + o + NB * (1 + bj);
        const bool diag = bi == bj;
        d2_t ra[8], rb[8], xb[2][8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = tid + 256 * p;
            ra[p] = *reinterpret_cast<const d2_t *>(Ai + (idx >> 5) * NB + (idx & 31) * 2);
            rb[p] = *reinterpret_cast<const d2_t *>(Aj + (idx >> 5) * NB + (idx & 31) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 8; ++s)
                xb[j][s] = *reinterpret_cast<const d2_t *>(X + (wn0 + 16 * j + fidx) * NB + 8 * s + fkg);
        d4_t acc[2][2];   // the old C tile: the update's MFMAs subtract in place
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = C[(long)(wm0 + 16 * i + MF::c_row(lane, r)) * Np + wn0 + 16 * j + MF::c_col(lane)];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = tid + 256 * p;
            *reinterpret_cast<d2_t *>(&As[idx >> 5][(idx & 31) * 2]) = ra[p];
            *reinterpret_cast<d2_t *>(&Bs[idx >> 5][(idx & 31) * 2]) = rb[p];
        }
        __syncthreads();
        d4_t li[2][2], lj[2][2];
        acc_zero(li);
        acc_zero(lj);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            d2_t av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = *reinterpret_cast<const d2_t *>(&As[wm0 + 16 * i + fidx][8 * s + fkg]);
                bv[i] = *reinterpret_cast<const d2_t *>(&Bs[wm0 + 16 * i + fidx][8 * s + fkg]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        li[i][j] = MF::mma(av[i][e], xb[j][s][e], li[i][j]);
                        if (!diag) lj[i][j] = MF::mma(bv[i][e], xb[j][s][e], lj[i][j]);
                    }
        }
        __syncthreads();   // every wave has read the unsolved tiles
        double *Lout = K + (long)(o + NB * (1 + bi)) * Np + o;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = wm0 + 16 * i + MF::c_row(lane, r), cc = wn0 + 16 * j + MF::c_col(lane);
                    As[rr][cc] = li[i][j][r];
                    if (!diag) Bs[rr][cc] = lj[i][j][r];
                    if (bj == 0) Lout[(long)rr * Np + cc] = li[i][j][r];   // L_bi in place (nobody reads K's panel in this launch)
                }
        __syncthreads();
        double (*Bu)[CH_LD] = diag ? As : Bs;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            d2_t av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&As[wm0 + 16 * i + fidx][8 * s + fkg]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Bu[wn0 + 16 * j + fidx][8 * s + fkg]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(-av[i][e], bv[j][e], acc[i][j]);
        }
        double *Nout = Apan_out + (long)(bi - 1) * NB * NB;   // (bj == 0 implies bi >= 1 here)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = wm0 + 16 * i + MF::c_row(lane, r), cc = wn0 + 16 * j + MF::c_col(lane);
                    C[(long)rr * Np + cc] = acc[i][j][r];
                    if (bj == 0) Nout[rr * NB + cc] = acc[i][j][r];        // the next panel's unsolved block bi - 1
                }
        return;
    }
    // ---- workgroup 0: the pivot ----
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double (*Tb)[CH_LD] = Xt + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const d2_t z2 = {0.0, 0.0};
    const int op = mode ? o + NB : o;          // the block factored here
    double *Akk = K + (long)op * Np + op;
    if (mode) {
        // Both products skip what is known to be zero -- exact zeros whose products only ever add +-0 to
        // a sum, so the results stay bit-identical to the full products:
        //   solve   L_r = A_r X^T, X lower triangular: column block jb of the result needs k < 16 (jb + 1).
        //           Wave w takes row fragment w and all four column fragments: 40 MFMAs instead of 64.
        //   update  A_kk -= L_r L_r^T: only the ten 16 x 16 fragments on and below the diagonal are ever
        //           read again (the factorisation reads the diagonal blocks and what lies below them).
        //           Waves 0, 1 take three fragments, waves 2, 3 two: 48 MFMAs instead of 64.
        d2_t ra[8];
        d2_t xb0[2], xb1[4], xb2[6], xb3[8];       // B fragments of X's row blocks 0 .. 3: k-steps s < 2 (jb + 1)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int idx = tid + 256 * p;
            ra[p] = *reinterpret_cast<const d2_t *>(Apan_in + (idx >> 5) * NB + (idx & 31) * 2);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) xb0[s] = *reinterpret_cast<const d2_t *>(X + (0 + fidx) * NB + 8 * s + fkg);
#pragma unroll
        for (int s = 0; s < 4; ++s) xb1[s] = *reinterpret_cast<const d2_t *>(X + (16 + fidx) * NB + 8 * s + fkg);
#pragma unroll
        for (int s = 0; s < 6; ++s) xb2[s] = *reinterpret_cast<const d2_t *>(X + (32 + fidx) * NB + 8 * s + fkg);
#pragma unroll
        for (int s = 0; s < 8; ++s) xb3[s] = *reinterpret_cast<const d2_t *>(X + (48 + fidx) * NB + 8 * s + fkg);
        // this wave's fragments (row block fi, column block fj) of the diagonal update
        const int nfr = wave < 2 ? 3 : 2;
        int fi[3], fj[3];
        fi[0] = wave == 0 ? 3 : (wave == 1 ? 3 : (wave == 2 ? 2 : 1));   fj[0] = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 2 : 1));
        fi[1] = wave == 0 ? 3 : (wave == 1 ? 2 : (wave == 2 ? 1 : 0));   fj[1] = wave == 0 ? 1 : 0;
        fi[2] = wave == 0 ? 3 : 2;                                        fj[2] = wave == 0 ? 2 : 1;
        d4_t acc[3];
#pragma unroll
        for (int f = 0; f < 3; ++f)
            if (f < nfr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[f][r] = Akk[(long)(16 * fi[f] + MF::c_row(lane, r)) * Np + 16 * fj[f] + MF::c_col(lane)];
            }
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&Tb[r][c2]) = ra[p8];
            *reinterpret_cast<d2_t *>(&Xt[r][c2]) = z2;
        }
        __syncthreads();
        if (stamp && tid == 0) stamp[0] = wall_clock64();   // loads landed
        d4_t li[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) li[jb][r] = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const d2_t av = *reinterpret_cast<const d2_t *>(&Tb[16 * wave + fidx][8 * s + fkg]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (s < 2) li[0] = MF::mma(av[e], xb0[s < 2 ? s : 0][e], li[0]);
                if (s < 4) li[1] = MF::mma(av[e], xb1[s < 4 ? s : 0][e], li[1]);
                if (s < 6) li[2] = MF::mma(av[e], xb2[s < 6 ? s : 0][e], li[2]);
                li[3] = MF::mma(av[e], xb3[s][e], li[3]);
            }
        }
        __syncthreads();
        double *Lout = K + (long)op * Np + o;   // row block 0 of the panel at o, in place
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * wave + MF::c_row(lane, r), cc = 16 * jb + MF::c_col(lane);
                Tb[rr][cc] = li[jb][r];
                Lout[(long)rr * Np + cc] = li[jb][r];
            }
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 3; ++f)
            if (f < nfr) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const d2_t av = *reinterpret_cast<const d2_t *>(&Tb[16 * fi[f] + fidx][8 * s + fkg]);
                    const d2_t bv = *reinterpret_cast<const d2_t *>(&Tb[16 * fj[f] + fidx][8 * s + fkg]);
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[f] = MF::mma(-av[e], bv[e], acc[f]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) At[16 * fi[f] + MF::c_row(lane, r)][16 * fj[f] + MF::c_col(lane)] = acc[f][r];
            }
    } else {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&At[r][c2]) = *reinterpret_cast<const d2_t *>(Akk + (long)r * Np + c2);
            *reinterpret_cast<d2_t *>(&Xt[r][c2]) = z2;
        }
    }
    __syncthreads();
    if (stamp && tid == 0) stamp[1] = wall_clock64();       // the block is in LDS, updated
    factor64_v4(At, Xt, Tb, rsbuf, op, flag, tiny);
    if (stamp && tid == 0) stamp[2] = wall_clock64();       // factored and inverted
    double *dstL = Linv + (long)op * Np + op;
    double *dstD = Dinv + (long)(op / NB) * NB * NB;
#pragma unroll
    for (int p8 = 0; p8 < 8; ++p8) {
        const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
        d2_t lv = *reinterpret_cast<const d2_t *>(&At[r][c2]);
        lv[0] = (c2 <= r) ? lv[0] : 0.0;
        lv[1] = (c2 + 1 <= r) ? lv[1] : 0.0;
        *reinterpret_cast<d2_t *>(Akk + (long)r * Np + c2) = lv;
        const d2_t xv = *reinterpret_cast<const d2_t *>(&Xt[r][c2]);
        *reinterpret_cast<d2_t *>(dstL + (long)r * Np + c2) = xv;
        *reinterpret_cast<d2_t *>(dstD + r * NB + c2) = xv;
    }
    if (tid < 64) {   // sum(log(diag L)), fixed-order tree
        double s = log(At[tid][tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (tid == 0) scal[0] += s;
    }
}

// 64-bit zero fill (the factor buffers exceed 4 GiB from N = 23170 on)
__global__ __launch_bounds__(256) void zero_fill_kernel(double2 *__restrict__ p, long n2) {
    const double2 z = {0.0, 0.0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) p[i] = z;
}

// ---- the two streams of a device ----------------------------------------------------------
// ONE pair per device for the whole process, shared by every handle on that device: the main
// stream (every call runs on it, in order) and the background stream of the fit (the inverse
// factor's GEMMs behind the panel chain), confined to a subset of the CUs (TGP_BG_CUS of the
// 256) so that the chain's small launches always find free CUs.
// Why shared: whether two hardware queues really run side by side depends on where the driver
// puts them.  With a stream pair per handle, every few handles one pair ended up time-sliced
// instead of concurrent and its fit took twice as long (N = 1000: 0.51 -> 1.1-1.3 ms in the
// ninth live handle; every CU-masked stream is a hardware queue of its own).  One pair, probed
// once: probe_wait spins (bounded, 300 us) on the main stream for a flag that probe_set raises
// from the background stream; if it times out the two are serialised and the background stream
// is created again (another queue), at most three times.
// The calls of this library are synchronous, so handles sharing a stream lose nothing but the
// overlap of two threads' calls on one device.
__global__ void probe_wait_kernel(int *flag, int *result, long long max_ticks) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    int seen = 0;
    while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) &&
           (long long)__builtin_readcyclecounter() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(32);
    *result = seen;
}
__global__ void probe_set_kernel(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static hipError_t create_bg_stream(int device, hipStream_t *out, int cus_env, int cu0 = 0) {
    // measured at N = 4096 on the 256-CU part: 2.50 ms with 192 CUs, 2.55 with 224, 2.61 with 128, 2.66 unmasked
    // -> three quarters of whatever this device (or partition: CPX / DPX modes expose fewer CUs per
    // device) reports, unless TGP_BG_CUS names a count; 0 or >= the device's count = unmasked
    const int bg_env = cus_env;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) {
        (void)hipGetLastError();
        ncu = 0;
    }
    const int bg_cus = bg_env >= 0 ? bg_env : (ncu * 3) / 4;
    hipStream_t st = nullptr;
    if (bg_cus > 0 && bg_cus < ncu) {
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int i = 0; i < bg_cus; ++i) {
            const int cu = (cu0 + i) % ncu;
            mask[(size_t)(cu >> 5)] |= 1u << (cu & 31);
        }
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            st = nullptr;
        }
    }
    if (!st) TGP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = st;
    return hipSuccess;
}

// Round 6, measured and NOT kept: the XCD of the pivot workgroup out of the background / third streams' masks.  Stamps of
// the panel chain with the pivot workgroup's HW_ID (profiles/r06_pivot_cu_stamps.txt): workgroup 0 of every launch on a
// stream lands on ONE XCD (the dispatcher deals a launch's workgroups round-robin over the XCDs from a start that is fixed
// per queue), on any of its 32 CUs; the mask of bits [0, 192) covers 24 CUs of every XCD (bit i = XCD i mod 8, CU i / 8 of
// it); and every panel of five fits whose factor phase took 19-29 us instead of 11 had its pivot workgroup on a CU the
// background stream's GEMMs run on -- none of those on the other CUs did.  The cause, then; but no mask cures it: an XCD
// with NO bit set comes back fully enabled, and with ONE CU left in it an eighth of every background launch's
// workgroups -- they are dealt over the XCDs whatever each has to offer -- queue on that CU (fit at N = 4096: 12.6 ms
// instead of 2.15; tools/microbench/cu_mask_probe.hip, profiles/r06_pivot_cu_stamps.txt).  Masks have to leave every XCD
// the same number of CUs; which of an XCD's CUs the pivot workgroup gets is the dispatcher's choice.

// 1 = the two streams overlap, 0 = serialised
static hipError_t streams_overlap(hipStream_t main, hipStream_t bg, int *overlap) {
    int *d = nullptr;
    TGP_TRY(hipMalloc((void **)&d, 2 * sizeof(int)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(int), main);
    if (e == hipSuccess) e = hipStreamSynchronize(main);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, main, d, d + 1, 700000LL);   // ~300 us of shader clocks
        hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, bg, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(main);
    if (e == hipSuccess) e = hipStreamSynchronize(bg);
    int h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    *overlap = h[1];
    return e;
}

// The pair lives while a handle on the device does (tgp_create takes a reference, tgp_destroy
// drops it) and whatever is left is destroyed by an atexit handler, i.e. before the HIP
// runtime's own tear-down: a CU-masked stream still alive in the static destructors crashed
// rocprofv3 runs at exit.
namespace {
struct StreamPair { hipStream_t main = nullptr, bg = nullptr, pre = nullptr; bool pre_failed = false; int refs = 0;
                    int bg_ok = -1, pre_ok = -1; };   // what the probes said: 1 runs beside the main stream, 0 serialised, -1 not probed
std::mutex g_pair_mu;
StreamPair g_pairs[64];
// the background stream on loan to the fit of a private-stream handle (below)
std::atomic<bool> g_bg_on_loan[64];
std::atomic<int> g_private_fits[64];
bool g_pair_atexit = false;

void destroy_pair(int dev) {   // g_pair_mu held
    StreamPair &p = g_pairs[dev];
    if (!p.main && !p.bg && !p.pre) return;
    if (hipSetDevice(dev) == hipSuccess) {
        if (p.pre) { (void)hipStreamSynchronize(p.pre); (void)hipStreamDestroy(p.pre); }
        if (p.bg) { (void)hipStreamSynchronize(p.bg); (void)hipStreamDestroy(p.bg); }
        if (p.main) { (void)hipStreamSynchronize(p.main); (void)hipStreamDestroy(p.main); }
    }
    p.main = p.bg = p.pre = nullptr;
    p.pre_failed = false;
}
void destroy_all_pairs() {
    std::lock_guard<std::mutex> lock(g_pair_mu);
    for (int d = 0; d < 64; ++d) destroy_pair(d);
}
}  // namespace

// The device's background stream on loan to the fit of a handle on a PRIVATE stream (the workers of a hyper-parameter
// fit).  Such a fit runs its share of the inverse in line, because the device has ONE background stream and inverses of
// several fits queued on it wait for each other -- but a fit that is the only private-stream fit in flight (the last
// start of a hyper-parameter fit still running, a lone worker) can have that stream to itself: it then issues the
// launches a handle on the shared stream issues (same bits), the inverse behind the panel chain (an evaluation at
// N = 1000: 0.72 -> 0.57 ms, what the caller's handle takes).  Only when alone: with one holder and one fit in line both
// lose (0.84 / 1.18 ms against 0.785 / 0.785 both in line); and more background streams, one per worker, were measured
// too (round 5: two workers 0.64 / 0.64, but with three two of them ran one behind the other, 1.31 / 0.65 / 1.31, whatever
// GPU_MAX_HW_QUEUES said) -- not kept.  private_fit_begin returns the stream on loan or null; never blocks.
hipStream_t private_fit_begin(int device, bool may_borrow) {
    if (device < 0 || device >= 64) return nullptr;
    const int in_flight = ++g_private_fits[device];
    if (!may_borrow || in_flight != 1) return nullptr;
    bool expected = false;
    if (!g_bg_on_loan[device].compare_exchange_strong(expected, true)) return nullptr;
    hipStream_t bg = nullptr;
    if (device_streams(device, nullptr, &bg) == hipSuccess && bg) return bg;
    (void)hipGetLastError();
    g_bg_on_loan[device].store(false);
    return nullptr;
}
void private_fit_end(int device, bool held) {
    if (device < 0 || device >= 64) return;
    if (held) g_bg_on_loan[device].store(false);
    --g_private_fits[device];
}

// main != null: take a reference (tgp_create); bg != null: the background stream.
// BOTH streams are created together, at the first call (round 4).  The runtime deals its streams round-robin
// onto a fixed number of hardware queues (4 per process unless GPU_MAX_HW_QUEUES says otherwise) and never
// re-deals: a background stream created later, as the 5th stream of the process behind three private streams
// of a threaded hyper-parameter fit (tgp_set_private_stream), landed on the MAIN stream's queue, the inverse
// ran after the panel chain instead of beside it, and every fit of the process took twice as long from then
// on (N = 2048: 1.00 -> 2.02 ms; tools/ab_private_streams.py reproduces it, and shows it gone with 8 queues).
// Created back to back the two always sit on neighbouring queues, whatever was created before them.
// Round 5: a THIRD stream of the set, created with the other two -- the front of the next sweep inside a fit
// (tgp_set_overlap): candidate scaling, the first cross-kernel, the contraction's early row tiles.  CU-masked like the
// background stream (TGP_PRE_CUS), so the panel chain's small launches keep a quarter of the device to themselves.
hipError_t device_streams(int device, hipStream_t *main, hipStream_t *bg, hipStream_t *pre) {
    std::lock_guard<std::mutex> lock(g_pair_mu);
    StreamPair &p = g_pairs[device & 63];
    if (!g_pair_atexit) { g_pair_atexit = true; atexit(destroy_all_pairs); }
    if (!p.main) TGP_TRY(hipStreamCreateWithFlags(&p.main, hipStreamNonBlocking));
    const bool probe = tuning().bg_probe != 0;

    if (!p.bg) {
        for (int attempt = 0; attempt < 3; ++attempt) {
            hipStream_t st = nullptr;
            TGP_TRY(create_bg_stream(device, &st, tuning().bg_cus));
            int ok = 1;
            if (probe) TGP_TRY(streams_overlap(p.main, st, &ok));
            if (ok || attempt == 2) { p.bg = st; p.bg_ok = probe ? ok : -1; break; }
            (void)hipStreamDestroy(st);
        }
    }
    if (!p.pre && !p.pre_failed) {
        // the third stream has to run beside BOTH others: on a process with too few hardware queues (GPU_MAX_HW_QUEUES,
        // 4 by default; turbo_amd/_lib.py asks for 8 before the runtime initialises) it can land on the main stream's
        // queue, and a sweep front "inside" the fit would then simply lengthen it.  Probed like the background stream;
        // if no attempt overlaps, the device has no third stream and tgp_set_overlap is a no-op on it.
        for (int attempt = 0; attempt < 3 && !p.pre; ++attempt) {
            hipStream_t st = nullptr;
            TGP_TRY(create_bg_stream(device, &st, tuning().pre_cus, tuning().pre_cu0));
            int ok1 = 1, ok2 = 1;
            if (probe) {
                TGP_TRY(streams_overlap(p.main, st, &ok1));
                TGP_TRY(streams_overlap(p.bg, st, &ok2));
            }
            if (ok1 && ok2) { p.pre = st; p.pre_ok = probe ? 1 : -1; break; }
            (void)hipStreamDestroy(st);
        }
        if (!p.pre) { p.pre_failed = true; p.pre_ok = 0; }
    }
    if (main) { *main = p.main; ++p.refs; }
    if (bg) *bg = p.bg;
    if (pre) *pre = p.pre;
    return hipSuccess;
}

// what the probes of the device's shared streams said when they were created (tgp_stream_status)
void device_stream_status(int device, int *bg_ok, int *pre_ok) {
    std::lock_guard<std::mutex> lock(g_pair_mu);
    const StreamPair &p = g_pairs[device & 63];
    if (bg_ok) *bg_ok = p.bg_ok;
    if (pre_ok) *pre_ok = p.pre_ok;
}

// tgp_destroy: the last handle on a device takes the pair with it
void device_streams_release(int device) {
    std::lock_guard<std::mutex> lock(g_pair_mu);
    StreamPair &p = g_pairs[device & 63];
    if (p.refs > 0 && --p.refs == 0) destroy_pair(device & 63);
}

static hipError_t ensure_lookahead(Context &c, size_t nev) {
    if (!c.stream_bg) TGP_TRY(device_streams(c.device, nullptr, &c.stream_bg, &c.stream_pre));
    while (c.ev_la.size() < nev) {
        hipEvent_t e;
        TGP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));   // (hipEventDisableSystemFence on top: 2.49 vs 2.51 ms, not worth the weaker visibility)
        c.ev_la.push_back(e);
    }
    return hipSuccess;
}

// First launch of a fit: Linv = 0, the pivot flag and the two scalars = 0, and -- when the inputs
// were staged in device-mapped host memory (src = [Xs | yn | ls]) -- their copy into HBM, so that a
// fit issues no memcpy and no memset of its own.
__global__ __launch_bounds__(256) void fit_prologue_kernel(double2 *__restrict__ linv, long n2,
                                                           const double *__restrict__ src,
                                                           double *__restrict__ Xs, long nxs,
                                                           double *__restrict__ yn, long nyn,
                                                           double *__restrict__ ls, long nls, int Dp,
                                                           int *__restrict__ flag, double *__restrict__ scal,
                                                           unsigned long long *__restrict__ stamp) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    if (stamp && t == 0) *stamp = wall_clock64();      // the start tick of a polled call (doorbell.hpp)
    if (src) {
        // src holds the RAW inputs (round 5): X / length_scale (sklearn kernels.py:1556, :1711) is taken here -- the same
        // correctly rounded IEEE division the host loop did, without 131 072 of them on the host's critical path at C3
        const double *lsrc = src + nxs + nyn;
        for (long i = t; i < nxs; i += stride) {
            const int d = (int)(i % Dp);
            Xs[i] = d < (int)nls ? src[i] / lsrc[d] : 0.0;
        }
        for (long i = t; i < nyn; i += stride) yn[i] = src[nxs + i];
        for (long i = t; i < nls; i += stride) ls[i] = src[nxs + nyn + i];
    }
    if (t == 0) { *flag = 0; scal[0] = 0.0; scal[1] = 0.0; }
    const double2 z = {0.0, 0.0};
    for (long i = t; i < n2; i += stride) linv[i] = z;
}

// The last launch of a polled call that is a CHAIN of kernels (the blocked fit, its LML gradient): a kernel of its own
// behind them on the call's stream -- everything they wrote, the scalars in mapped host memory included, is out -- that
// rings the doorbell.  One 2-us launch for two event records and a stream synchronisation.
__global__ void ring_kernel(Bell bell) { bell_ring(bell, 1); }
hipError_t launch_ring(Context &c, const Bell &bell) {
    hipLaunchKernelGGL(ring_kernel, dim3(1), dim3(64), 0, c.stream, bell);
    return hipGetLastError();
}

hipError_t launch_fit(Context &c, const double *staged_in, double *res_host, bool zero_linv, unsigned long long *start_stamp) {
    hipStream_t s = c.stream;
    const int N = (int)c.N, Np = (int)c.Np, Dp = (int)c.Dp;
    const long NN = (long)Np * Np;

    {
        long blocks = NN / 2 / 256;
        if (blocks > 4096) blocks = 4096;
        // Linv = 0: only when something other than zeros can sit above the diagonal or in the rows this fit
        // skips (a fresh allocation, another leading dimension, an older fit with more rows): everything on
        // and below the diagonal of the rows it does not skip is written by this fit anyway (27 us at Np = 4096,
        // 96 us at 8192 when it has to run)
        if (!zero_linv) blocks = std::min<long>(blocks, std::max<long>(1, ((long)Np * Dp + 255) / 256));
        hipLaunchKernelGGL(fit_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           reinterpret_cast<double2 *>(c.d_Linv), zero_linv ? NN / 2 : 0L, staged_in, c.d_Xs, (long)Np * Dp,
                           c.d_yn, (long)Np, c.d_ls, (long)c.D, Dp, c.d_flag, c.d_scal, start_stamp);
        TGP_TRY(hipGetLastError());
    }

    // the f32 copy of Xs right away (it used to be the fit's last launch): the cross-kernel of an f32 sweep that
    // starts inside this fit reads it
    if (c.dtype != TGP_F64) {
        hipLaunchKernelGGL(f64_to_f32_kernel, dim3(64), dim3(256), 0, s, c.d_Xs, c.d_Xs32,
                           (long)Np * Dp);
        TGP_TRY(hipGetLastError());
    }
    // ---- the front of the next sweep, beside everything that follows (tgp_set_overlap; sweep_kernels.hip) ----
    hipStream_t spre = nullptr;
    int pre_budget128 = 0;
    if (c.pre.issue > 0) {
        if (!c.stream_pre) TGP_TRY(device_streams(c.device, nullptr, &c.stream_bg, &c.stream_pre));
        spre = c.stream_pre;   // (null: this process has no hardware queue left for a third stream -- no front)
    }
    if (spre) {
        TGP_TRY(hipEventRecord(c.pre.ev_in, s));
        TGP_TRY(hipStreamWaitEvent(spre, c.pre.ev_in, 0));
        TGP_TRY(presweep_front(c, spre));
        if (c.pre.issue >= 2 && c.pre.front) {
            // how many 128-row tiles of launch pair 0 the fit takes: a quarter of the rows unless TGP_PRE_TILES says otherwise
            const int n128 = (N + 127) / 128;
            pre_budget128 = tuning().pre_tiles >= 0 ? tuning().pre_tiles : n128 / 4;
        }
    }

    // ---- K ----
    {
        const int nt = Np / PW_T;
        const dim3 grid(nt * (nt + 1) / 2);
        switch (c.kernel) {
            case TGP_RBF: hipLaunchKernelGGL(kernel_matrix_kernel<TGP_RBF>, grid, dim3(256), 0, s, c.d_Xs, c.d_K, N, Np, Dp, c.constant, c.noise, c.jitter); break;
            case TGP_MATERN12: hipLaunchKernelGGL(kernel_matrix_kernel<TGP_MATERN12>, grid, dim3(256), 0, s, c.d_Xs, c.d_K, N, Np, Dp, c.constant, c.noise, c.jitter); break;
            case TGP_MATERN32: hipLaunchKernelGGL(kernel_matrix_kernel<TGP_MATERN32>, grid, dim3(256), 0, s, c.d_Xs, c.d_K, N, Np, Dp, c.constant, c.noise, c.jitter); break;
            default: hipLaunchKernelGGL(kernel_matrix_kernel<TGP_MATERN52>, grid, dim3(256), 0, s, c.d_Xs, c.d_K, N, Np, Dp, c.constant, c.noise, c.jitter); break;
        }
        TGP_TRY(hipGetLastError());
    }
    // ---- blocked Cholesky, two levels ----
    // Panels of NB = 64 columns (LDS factorisation + MFMA panel solve) are grouped into outer
    // blocks of OB = 512: inside an outer block a panel only updates the remaining columns of
    // that block (narrow, K = 64); the trailing matrix gets ONE rank-OB update per outer block
    // on the direct-to-LDS NT kernel (OB/16 k-tiles per tile instead of OB/64 launches of 4).
    // outer block (multiple of 256).  With the inverse behind the chain: 256 wins from Np = 1024 to
    // 3072 (1.06 vs 1.10 ms at N = 2048), 512 at N = 4096 (2.53 vs 2.59) and beyond.
    const int OB_env = tuning().ob;
    // (round 3, with the fused panel launches: 512 is now equal or better from Np = 1536 on -- 2560: 1.293 vs 1.334 ms,
    // 3072: 1.617 vs 1.713 -- and 256 only wins where 512 leaves a half-empty last block: Np = 1280 0.618 vs 0.667)
    // (round 4, after the f64 kernels got faster: 1024 from Np = 6144 on -- 6144: 4.66 -> 4.48 ms, 8192: 8.87 -> 8.62 --
    // half as many trailing updates and event hops; still 512 at 4096: 2.20 vs 2.26)
    // (round 5, late: up to Np = 1024 ONE outer block -- no trailing update, the inverse level by level behind the last
    // panel.  With the one-launch panels the in-block updates cost no more than the rank-256 ones did, and the chain
    // loses its hand-overs to the background stream: fit 0.455 -> 0.417 / 0.452 -> 0.439 ms at N = 900 / 1000, an
    // evaluation of the hyper-parameter objective at N = 1000 0.562 -> 0.544 ms on the caller's handle and, on the
    // pooled workers whose inverse runs in line, 0.73 -> 0.55 alone, 0.88 -> 0.63 three side by side; Np = 768 equal)
    int OB = OB_env ? OB_env : (Np <= 1024 ? Np : (Np <= 1280 ? 256 : (Np >= 6144 ? 1024 : 512)));
    // The last, partial outer block builds its own inverse by merging halves (inverse_block below): its length has to
    // be a power of two.  Np is a multiple of 256, so 256 and 512 always leave 0 or 256; 1024 can leave 768 (Np = 6912,
    // 7936, 8960: `invalid configuration argument` from a merge of zero pairs before this check) -> 512 there.
    if (const int tail = Np % OB; (tail & (tail - 1)) != 0) OB = 512;
    const double tiny = 8.0 * 2.220446049250313e-16 * ((c.constant + c.noise) + c.jitter);
    // Rows >= N are padding: K is the identity there, so its factor is the identity too and the
    // panels, panel rows and trailing tiles that hold nothing but padding are skipped (their L
    // stays as kernel_matrix_kernel wrote it; Linv stays zero, which is all the sweep needs since
    // the cross-kernel slab is zero in those columns).
    const int Nr = ((N + NB - 1) / NB) * NB;   // real rows, rounded up to whole panels
    // ---- the inverse factor: [[A,0],[C,B]]^-1 = [[Ai,0],[-Bi*C*Ai,Bi]] ----
    // Level 64 -> 128 on the k-major GEMM template; from 128 up the transpose U = Linv^T is kept
    // alongside so both products of a merge are NT:
    //     T^T   = U11 * L21^T            (U11 upper triangular: k from the tile's own row on)  -> W
    //     Linv21 = -Linv22 * (T^T)^T     (Linv22 lower triangular), stored to Linv and, transposed, to U
    // level64(st, o, pairs): the `pairs` 128-blocks from row o on, one batched launch per product.
    const int panel_var = tuning().panel;   // A/B: 5 = variant D (block in LDS, MFMA), 3 / 38 = variant C with 4 / 8 columns per barrier, 4 / 8 = variant B, 0 = round-1 form
    auto level64 = [&](hipStream_t st, long o, int pairs) -> hipError_t {
        const long bs64 = (long)2 * NB * ((long)Np + 1);
        const long d = o * ((long)Np + 1);
        GemmArgs t{};   // T = L21 * L11inv -> W
        t.A = c.d_K + d + (long)NB * Np; t.lda = Np; t.strideA = bs64;
        t.B = c.d_Linv + d; t.ldb = Np; t.strideB = bs64;
        t.C = c.d_W + d + (long)NB * Np; t.ldc = Np; t.strideC = bs64;
        t.ntm = t.ntn = 1; t.K = NB; t.alpha = 1.0; t.beta = 0.0;
        GemmArgs u{};   // Linv21 = -L22inv * T
        u.A = c.d_Linv + d + (long)NB * Np + NB; u.lda = Np; u.strideA = bs64;
        u.B = c.d_W + d + (long)NB * Np; u.ldb = Np; u.strideB = bs64;
        u.C = c.d_Linv + d + (long)NB * Np; u.ldc = Np; u.strideC = bs64;
        u.ntm = u.ntn = 1; u.K = NB; u.alpha = -1.0; u.beta = 0.0;
        if (tuning().level64_fused) {
            hipLaunchKernelGGL(level64_fused_kernel, dim3(pairs), dim3(256), 0, st, c.d_K, c.d_Linv, c.d_U, Np, o);
            return hipGetLastError();
        }
        TGP_TRY((launch_gemm64<64, 64, false, KR_LOWER_B, TM_FULL>(st, c.device, t, 1, pairs)));
        TGP_TRY((launch_gemm64<64, 64, false, KR_LOWER_A, TM_FULL>(st, c.device, u, 1, pairs)));
        hipLaunchKernelGGL(transpose_diag128_kernel, dim3(pairs, 16), dim3(256), 0, st, c.d_Linv + d,
                           c.d_U + d, Np);
        return hipGetLastError();
    };
    // Merges with a leading block up to MERGE64 (few, very unequal 128-tiles) go to the 64 x 64
    // register-staged template like the small trailing updates.  Measured: fit 4.52 -> 4.13 ms
    // at N = 4096, 0.54 -> 0.46 ms at N = 512; the 4096-level of N = 8192 is faster on the
    // 128-tile direct-to-LDS kernel (13.39 vs 13.59 ms), hence 2048.
    const int MERGE64 = tuning().merge64;
    // leading block [o, o+a), trailing block [o+a, o+a+b); a, b multiples of 128 (64 with small = true)
    auto merge_t = [&](hipStream_t st, long o, int a, int b, int nprob, long bstride, bool small) -> hipError_t {
        if (small || a <= MERGE64) {
            GemmArgs tt{};   // T^T (a x b) -> W[o.., o+a..]
            tt.A = c.d_U + o * Np + o; tt.lda = Np; tt.strideA = bstride;
            tt.B = c.d_K + (o + a) * Np + o; tt.ldb = Np; tt.strideB = bstride;
            tt.C = c.d_W + o * Np + (o + a); tt.ldc = Np; tt.strideC = bstride;
            tt.ntm = a / NB; tt.ntn = b / NB; tt.K = a; tt.alpha = 1.0; tt.beta = 0.0;
            return launch_gemm64<64, 64, true, KR_UPPER_A, TM_FULL>(st, c.device, tt, tt.ntm * tt.ntn, nprob);
        }
        GemmNtArgs tt{};
        tt.A = c.d_U + o * Np + o; tt.lda = Np; tt.strideA = bstride;
        tt.B = c.d_K + (o + a) * Np + o; tt.ldb = Np; tt.strideB = bstride;
        tt.C = c.d_W + o * Np + (o + a); tt.ldc = Np; tt.strideC = bstride;
        tt.Ct = nullptr;
        tt.ntm = a / 128; tt.ntn = b / 128; tt.K = a; tt.alpha = 1.0; tt.beta = 0.0;
        return launch_gemm_nt_glds<double, KN_UPPER_A, TM_FULL>(st, c.device, tt, tt.ntm * tt.ntn, nprob);
    };
    auto merge_u = [&](hipStream_t st, long o, int a, int b, int nprob, long bstride, bool small) -> hipError_t {
        if (small || a <= MERGE64) {
            GemmArgs uu{};   // Linv21 (b x a) and its transpose into U
            uu.A = c.d_Linv + (o + a) * Np + (o + a); uu.lda = Np; uu.strideA = bstride;
            uu.B = c.d_W + o * Np + (o + a); uu.ldb = Np; uu.strideB = bstride;
            uu.C = c.d_Linv + (o + a) * Np + o; uu.ldc = Np; uu.strideC = bstride;
            uu.Ct = c.d_U + o * Np + (o + a); uu.ldct = Np; uu.strideCt = bstride;
            uu.ntm = b / NB; uu.ntn = a / NB; uu.K = b; uu.alpha = -1.0; uu.beta = 0.0;
            return launch_gemm64<64, 64, true, KR_LOWER_A, TM_FULL>(st, c.device, uu, uu.ntm * uu.ntn, nprob);
        }
        GemmNtArgs uu{};
        uu.A = c.d_Linv + (o + a) * Np + (o + a); uu.lda = Np; uu.strideA = bstride;
        uu.B = c.d_W + o * Np + (o + a); uu.ldb = Np; uu.strideB = bstride;
        uu.C = c.d_Linv + (o + a) * Np + o; uu.ldc = Np; uu.strideC = bstride;
        uu.Ct = c.d_U + o * Np + (o + a); uu.ldct = Np; uu.strideCt = bstride;
        uu.ntm = b / 128; uu.ntn = a / 128; uu.K = b; uu.alpha = -1.0; uu.beta = 0.0;
        return launch_gemm_nt_glds<double, KN_LOWER_A, TM_FULL>(st, c.device, uu, uu.ntm * uu.ntn, nprob);
    };
    auto merge = [&](hipStream_t st, long o, int a, int b, int nprob, long bstride) -> hipError_t {
        TGP_TRY(merge_t(st, o, a, b, nprob, bstride, false));
        return merge_u(st, o, a, b, nprob, bstride, false);
    };
    // (1) level by level over the whole matrix, after the factorisation: all complete pairs of a
    // level share one batched launch; an odd segment out (Np is a multiple of 256, not necessarily
    // a power of two) is merged separately when it finds a partner.
    auto inverse_levels = [&](hipStream_t st) -> hipError_t {
        TGP_TRY(level64(st, 0, Np / (2 * NB)));
        int nfull = Np / 128;  // complete segments of size sz
        int tail = 0;          // size of the trailing odd segment (0 = none)
        for (int sz = 128; nfull + (tail ? 1 : 0) > 1; sz *= 2) {
            const int pairs = nfull / 2;
            if (pairs > 0) TGP_TRY(merge(st, 0, sz, sz, pairs, (long)2 * sz * ((long)Np + 1)));
            if (nfull & 1) {
                const long o = (long)(nfull - 1) * sz;
                if (tail) {   // odd full segment + tail -> new tail
                    TGP_TRY(merge(st, o, sz, tail, 1, 0));
                    tail += sz;
                } else {
                    tail = sz;
                }
            }
            nfull = pairs;
        }
        return hipSuccess;
    };
    // (2) outer block by outer block, BEHIND the panel chain on the background stream: once the
    // panels of block [O, E) are final,
    //     X_bb              the block's own inverse (levels 64, 128, 256 inside the block)
    //     Linv[O:E, 0:O]  = -X_bb * T[O:E, 0:O]                       (T^T sits in W above the diagonal)
    //     T[E:, 0:E]     +=  L[E:, O:E] * Linv[O:E, 0:E]              (every later row block's share)
    // so that after the last panel only the last block's X_bb and row block are left to do.
    // `inner` = the block's own inverse X_bb (false: already there, see inverse_inner_all)
    auto inverse_block = [&](hipStream_t st, long O, long E, bool inner = true) -> hipError_t {
        const int len = (int)(E - O);
        if (inner) {
            TGP_TRY(level64(st, O, len / 128));
            for (int sz = 128; sz < len; sz *= 2)
                TGP_TRY(merge(st, O, sz, sz, len / (2 * sz), (long)2 * sz * ((long)Np + 1)));
        }
        if (O > 0) TGP_TRY(merge_u(st, 0, (int)O, len, 1, 0, true));
        if (E < Np) {
            if (O > 0) {   // rows 0:O of T^T: W[0:O, E:] += U[0:O, O:E] * L[E:, O:E]^T
                GemmArgs g{};
                g.A = c.d_U + O; g.lda = Np;
                g.B = c.d_K + E * Np + O; g.ldb = Np;
                g.C = c.d_W + E; g.ldc = Np;
                g.ntm = (int)(O / NB); g.ntn = (int)((Np - E) / NB); g.K = len; g.alpha = 1.0; g.beta = 1.0;
                TGP_TRY((launch_gemm64<64, 64, true, KR_FULL, TM_FULL>(st, c.device, g, g.ntm * g.ntn, 1)));
            }
            TGP_TRY(merge_t(st, O, len, (int)(Np - E), 1, 0, true));   // rows O:E, first contribution
        }
        return hipSuccess;
    };
    // The X_bb of EVERY outer block at once, after the factorisation: the launches of inverse_block's first part batched
    // over the blocks (a pair of 64-, 128-, ... row segments never straddles an outer block: OB and the tail are powers
    // of two times 128) -- the same tiles with the same operands in the same k order, i.e. the same bits, in 1 + 2 log2(OB /
    // 128) launches instead of that many per block.  For a fit that cannot put its inverse behind the chain (a handle on a
    // private stream): 40 -> 25 launches at Np = 2048, 80 -> 45 at 4096.
    auto inverse_inner_all = [&](hipStream_t st) -> hipError_t {
        TGP_TRY(level64(st, 0, Np / (2 * NB)));
        for (int sz = 128; sz < OB; sz *= 2) {
            const int pairs = Np / (2 * sz);
            if (pairs > 0) TGP_TRY(merge(st, 0, sz, sz, pairs, (long)2 * sz * ((long)Np + 1)));
        }
        return hipSuccess;
    };
    // ... and with a row block of Linv final: its rows of z = Linv yn, its share of alpha = Linv^T z,
    // its f32 copy for an f32 sweep
    auto finish_block = [&](hipStream_t st, long O, long E) -> hipError_t {
        hipLaunchKernelGGL(rowblock_finish_kernel, dim3((unsigned)(E - O)), dim3(256), 0, st, c.d_Linv, c.d_yn,
                           c.d_z, c.dtype != TGP_F64 ? c.d_Linv32 : nullptr, Np, (int)O);
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(rowblock_cols_kernel, dim3((unsigned)(E / 64), (unsigned)((E - O) / GEMV_SLICE)), dim3(256), 0, st,
                           c.d_Linv, c.d_z, c.d_apart, c.d_apart + (long)(Np / GEMV_SLICE) * Np, Np, (int)O);
        return hipGetLastError();
    };
    const int bginv_on = tuning().bginv;
    // up to Np = 9216: beyond, the inverse's products want the 128-tile direct-to-LDS kernel and the whole
    // chip (N = 10000: 17.7 vs 17.2 ms, 16384: 63.0 vs 57.5, 20000: 114 vs 101 for the level-by-level path)
    const int bginv_max = tuning().bginv_max;
    const bool bginv = bginv_on && Np > OB && Np <= bginv_max;
    const int nblk = (Np + OB - 1) / OB;
    // A handle on a PRIVATE stream (the workers of a threaded hyper-parameter fit) keeps its share of the inverse on that
    // stream: the device has ONE background stream, and the workers' inverses queued on it one behind the other while
    // every worker's chain waited for its own -- three starts side by side at N = 1000 took as long as one after the
    // other (28.5 vs 28.6 ms; 19.1 with the inverses in line, N = 2048 101 -> 83).  Same launches in the same order of
    // arithmetic, so the results are those of the shared-stream path bit for bit.
    const bool bg_shared = c.stream_own == nullptr || c.bg_lease != nullptr;   // (bg_lease: the background stream this fit holds on loan)
    if (bginv && bg_shared) TGP_TRY(ensure_lookahead(c, (size_t)2 * nblk));
    const hipStream_t sbg = c.bg_lease ? c.bg_lease : c.stream_bg;
    // debug: TGP_STAMP_FILE=path makes every fused panel launch leave in-kernel time stamps (10 ns
    // ticks) and the fit dump them there (tools/stamp_summary.py reads the file); Np <= 8192 only
    constexpr size_t STAMP_STRIDE = FUSED_STAMP_STRIDE;
    const char *stamp_path = tuning().stamp_file.empty() ? nullptr : tuning().stamp_file.c_str();
    // (debug only.  The buffer belongs to the handle -- free_fit releases it -- so fits on several handles, e.g.
    // the threaded hyper-parameter starts, stamp buffers of their own; the FILE is the last finisher's)
    unsigned long long *stamp_dev = nullptr;
    if (stamp_path && Np <= 8192) {
        if (!c.d_stamp) TGP_TRY(hipMalloc((void **)&c.d_stamp, 2 * 128 * STAMP_STRIDE * sizeof(unsigned long long)));
        TGP_TRY(hipMemsetAsync(c.d_stamp, 0, 2 * 128 * STAMP_STRIDE * sizeof(unsigned long long), s));
        stamp_dev = c.d_stamp;
    }
    for (int O = 0; O < Nr; O += OB) {
        for (int kk = 0; kk < OB / NB; ++kk) {
            const int o = O + kk * NB;
            if (o >= Nr) break;
            const int rem = (Nr - o - NB) / NB;   // real block rows below
            const int panel_la = tuning().panel_la;
            const int panel_fuse = tuning().panel_fuse;
            // Decided per OUTER BLOCK by the tiles of its first update.  Each fused tile is three f64 products on
            // a CU of its own (10 us against 3 for a plain update tile), so a block with hundreds of them holds
            // CUs the background stream's inverse wants: measured on one box, fit ms at N = 2048 / 4096 / 8192 with
            // the limit at 0 (never): 1.037 / 2.379 / 9.997, 128: 1.004 / 2.373 / 9.953, 256: 1.010 / 2.428 / 9.967,
            // no limit: 1.000 / 2.418 / 10.185 -- the fused launch is kept for blocks of up to 128 tiles
            // (everything up to N = 2048, the last two outer blocks beyond).  TGP_PANEL_FUSE_TILES overrides.
            // Round 4 (MFMAs in VGPR form: a fused tile is shorter): 384 -- N = 2048 0.949 -> 0.903 ms, 3072 1.506 -> 1.427,
            // 4096 / 6144 / 8192 unchanged; 512: 6144 +2 %, no limit: 8192 +3 %.
            const int panel_fuse_tiles = tuning().panel_fuse_tiles;
            const int rem0 = (Nr - O - NB) / NB;                                  // row blocks below the block's first panel
            const int tiles0 = rem0 * (OB / NB - 1 < rem0 ? OB / NB - 1 : rem0);  // tiles of its first update
            if (panel_fuse && tiles0 <= panel_fuse_tiles && panel_la && panel_var == 5) {
                // one launch per panel: see fused_panel_kernel.  Slot kk & 1 of Apan holds this panel's
                // unsolved blocks; the last panel of an outer block (no update follows) is solved in place.
                const long slot = (long)(Np / NB) * NB * NB;
                int ncol = OB / NB - 1 - kk;
                if (ncol > rem) ncol = rem;
                unsigned long long *st0 = stamp_dev ? stamp_dev + (size_t)(2 * (o / NB)) * STAMP_STRIDE : nullptr;
                if (kk == 0) {
                    hipLaunchKernelGGL(fused_panel_kernel, dim3(1 + (ncol > 0 ? rem : 0)), dim3(256), 0, s, c.d_K, Np, o, 0, 1,
                                       c.d_Apan, c.d_Apan, c.d_Dinv, c.d_Linv, c.d_scal, c.d_flag, tiny, st0);
                    TGP_TRY(hipGetLastError());
                }
                if (rem == 0) break;
                if (ncol > 0) {
                    hipLaunchKernelGGL(fused_panel_kernel, dim3(1 + rem * ncol), dim3(256), 0, s, c.d_K, Np, o, 1, ncol,
                                       c.d_Apan + (kk & 1) * slot, c.d_Apan + ((kk + 1) & 1) * slot, c.d_Dinv,
                                       c.d_Linv, c.d_scal, c.d_flag, tiny, st0 ? st0 + STAMP_STRIDE : nullptr);
                } else {
                    hipLaunchKernelGGL(panel_solve_kernel, dim3(rem), dim3(256), 0, s, c.d_K, Np, o, c.d_Dinv);
                }
                TGP_TRY(hipGetLastError());
                continue;
            }
            if (panel_la && panel_var == 5) {
                // the pivot off the update's back: see pivot_update_kernel.  The first panel of an outer block
                // factors its pivot alone (the trailing update before it has touched everything); every
                // later pivot was factored beside the previous panel's update.
                if (kk == 0) {
                    hipLaunchKernelGGL(pivot_update_kernel, dim3(1), dim3(256), 0, s, c.d_K, Np, o, 0, 1, c.d_Dinv, c.d_Linv,
                                       c.d_scal, c.d_flag, tiny);
                    TGP_TRY(hipGetLastError());
                }
                if (rem == 0) break;
                hipLaunchKernelGGL(panel_solve_kernel, dim3(rem), dim3(256), 0, s, c.d_K, Np, o, c.d_Dinv);
                TGP_TRY(hipGetLastError());
                int ncol = OB / NB - 1 - kk;
                if (ncol > rem) ncol = rem;
                if (ncol > 0) {
                    hipLaunchKernelGGL(pivot_update_kernel, dim3(1 + rem * ncol), dim3(256), 0, s, c.d_K, Np, o + NB, 1, ncol,
                                       c.d_Dinv, c.d_Linv, c.d_scal, c.d_flag, tiny);
                    TGP_TRY(hipGetLastError());
                }
                continue;
            }
            // diagonal block (factor + inverse) and, in the same launch, the panel solve of every
            // row block below it
            auto pk = panel_kernel<3>;
            if (panel_var == 38) pk = panel_kernel<38>;
            else if (panel_var == 8) pk = panel_kernel<8>;
            else if (panel_var == 4) pk = panel_kernel<4>;
            else if (panel_var == 0) pk = panel_kernel<0>;
            if (panel_var == 5) pk = panel_d_kernel;
            hipLaunchKernelGGL(pk,
                               dim3(rem + 1), dim3(256), 0, s, c.d_K, Np, o, c.d_Dinv,
                               c.d_Linv, c.d_Dinv + (long)(Np / NB) * NB * NB, c.d_scal, c.d_flag, tiny);   // second half of Dinv: where a diagonal block waits
            TGP_TRY(hipGetLastError());
            if (rem == 0) break;
            double *panel = c.d_K + (long)(o + NB) * Np + o;
            int ncol = OB / NB - 1 - kk;          // panels left inside this outer block
            if (ncol > rem) ncol = rem;           // ... that exist (last, partial outer block)
            const bool inner_generic = tuning().inner_generic;
            if (ncol > 0 && !inner_generic) {   // A[:, o+64 : O+OB] -= L_:k * L_jk^T
                hipLaunchKernelGGL(rank64_update_kernel, dim3(rem * ncol), dim3(256), 0, s, c.d_K, Np, o, ncol);
                TGP_TRY(hipGetLastError());
            } else if (ncol > 0) {
                GemmArgs g{};
                g.A = panel; g.lda = Np;
                g.B = panel; g.ldb = Np;
                g.C = c.d_K + (long)(o + NB) * Np + (o + NB); g.ldc = Np;
                g.ntm = rem; g.ntn = ncol; g.K = NB; g.alpha = -1.0; g.beta = 1.0;
                TGP_TRY((launch_gemm64<64, 64, true, KR_FULL, TM_FULL>(s, c.device, g, rem * ncol, 1)));
            }
        }
        if (bginv && O + OB < Np) {   // every block but the last: its share of the inverse goes behind the chain
            const int b = O / OB;
            if (bg_shared) {
                TGP_TRY(hipEventRecord(c.ev_la[2 * b], s));
                TGP_TRY(hipStreamWaitEvent(sbg, c.ev_la[2 * b], 0));
                TGP_TRY(inverse_block(sbg, O, O + OB));
                TGP_TRY(finish_block(sbg, O, O + OB));
                TGP_TRY(hipEventRecord(c.ev_la[2 * b + 1], sbg));
                if (spre && pre_budget128 > c.pre.rows128) {   // rows < O + OB of Linv (and its f32 copy) are final behind that event
                    TGP_TRY(hipStreamWaitEvent(spre, c.ev_la[2 * b + 1], 0));
                    TGP_TRY(presweep_rows(c, spre, O + OB, pre_budget128));
                }
            }   // (a handle on a private stream without the background stream on loan: after the last panel, below)
        }
        const int R = ((Nr - O - OB + 127) / 128) * 128;   // real trailing rows in whole 128-tiles (<= Np - O - OB)
        // Trailing matrices up to TRAIL64 rows have too few 128-tiles to fill the chip and each tile
        // then runs its whole k-range alone on a CU: use 64 x 64 tiles (4x the tiles, a quarter of
        // the critical path) on the register-staged template.  Measured: fit 4.73 -> 4.50 ms at
        // N = 4096, 14.35 -> 13.59 ms at N = 8192; from N = 20000 on the 128-tile direct-to-LDS
        // kernel wins again (448 vs 419 ms at N = 33000), hence the threshold.
        const int TRAIL64 = tuning().trail64;
        if (R > 0 && R <= TRAIL64) {
            GemmArgs g{};
            g.A = c.d_K + (long)(O + OB) * Np + O; g.lda = Np;
            g.B = g.A; g.ldb = Np;
            g.C = c.d_K + (long)(O + OB) * Np + (O + OB); g.ldc = Np;
            g.ntm = g.ntn = R / NB; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
            const int nt = R / NB;
            TGP_TRY((launch_gemm64<64, 64, true, KR_FULL, TM_LOWER>(s, c.device, g, nt * (nt + 1) / 2, 1)));
        } else if (R > 0) {   // A[i][j] -= L[i][O:O+OB] * L[j][O:O+OB]^T, i >= j >= O+OB
            GemmNtArgs g{};
            g.A = c.d_K + (long)(O + OB) * Np + O; g.lda = Np;
            g.B = g.A; g.ldb = Np;
            g.C = c.d_K + (long)(O + OB) * Np + (O + OB); g.ldc = Np;
            g.Ct = nullptr;
            g.ntm = g.ntn = R / 128; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
            const int nt = R / 128;
            TGP_TRY((launch_gemm_nt_glds<double, KN_FULL, TM_LOWER>(s, c.device, g, nt * (nt + 1) / 2, 1)));
        }
    }
    if (bginv) {
        const long O = (long)(nblk - 1) * OB;
        if (bg_shared) {
            TGP_TRY(hipStreamWaitEvent(s, c.ev_la[2 * (nblk - 2) + 1], 0));
            TGP_TRY(inverse_block(s, O, Np));
        } else {
            // in line, on the private stream: every block's own inverse in batched launches, then block by block what
            // depends on the blocks before it -- the arithmetic of the background schedule, operation for operation
            TGP_TRY(inverse_inner_all(s));
            for (long Ob = 0; Ob < O; Ob += OB) {
                TGP_TRY(inverse_block(s, Ob, Ob + OB, false));
                TGP_TRY(finish_block(s, Ob, Ob + OB));
            }
            TGP_TRY(inverse_block(s, O, Np, false));
        }
        TGP_TRY(finish_block(s, O, Np));
        hipLaunchKernelGGL(alpha_finish_sliced_kernel, dim3(Np / 64), dim3(256), 0, s, c.d_apart,
                           c.d_apart + (long)(Np / GEMV_SLICE) * Np, c.d_alpha, c.d_scal, Np, c.d_flag, res_host);
        TGP_TRY(hipGetLastError());
    } else {
        TGP_TRY(inverse_levels(s));
        // ---- alpha = Linv^T (Linv yn),  yn . alpha ----
        hipLaunchKernelGGL(gemv_lower_rows_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, c.d_Linv,
                           c.d_yn, c.d_z, Np);
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gemv_lower_cols_kernel, dim3(Np / 64, GEMV_RS), dim3(256), 0, s, c.d_Linv,
                           c.d_z, c.d_W, Np);   // W is free again after the inverse
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(alpha_finish_kernel, dim3(1), dim3(256), 0, s, c.d_W, c.d_yn, c.d_alpha,
                           c.d_scal, Np, c.d_flag, res_host);
        TGP_TRY(hipGetLastError());
        if (c.dtype != TGP_F64) {
            hipLaunchKernelGGL(f64_to_f32_kernel, dim3(2048), dim3(256), 0, s, c.d_Linv, c.d_Linv32, NN);
            TGP_TRY(hipGetLastError());
        }
    }
    if (spre) {
        TGP_TRY(hipEventRecord(c.pre.ev, spre));
        c.pre.pending = true;
    }
    if (stamp_dev) {
        std::vector<unsigned long long> hst(2 * 128 * STAMP_STRIDE);
        TGP_TRY(hipStreamSynchronize(s));
        TGP_TRY(hipMemcpy(hst.data(), stamp_dev, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *fh = fopen(stamp_path, "wb")) {
            fwrite(hst.data(), sizeof(unsigned long long), hst.size(), fh);
            fclose(fh);
        }
        // round 6: the CUs each of the fit's streams reaches (main | background | third), HW_PROBE_WGS workgroups each, in
        // <path>.cus -- so that tools/stamp_summary.py can say whether a panel's pivot workgroup sat on a CU the
        // background stream's GEMMs can run on
        constexpr int HW_PROBE_WGS = 16384;
        unsigned long long *pd = nullptr;
        TGP_TRY(hipMalloc((void **)&pd, 3 * HW_PROBE_WGS * sizeof(unsigned long long)));
        TGP_TRY(hipMemset(pd, 0, 3 * HW_PROBE_WGS * sizeof(unsigned long long)));
        const hipStream_t probe_on[3] = {s, sbg, c.stream_pre};
        for (int k = 0; k < 3; ++k) {
            if (!probe_on[k]) continue;
            hipLaunchKernelGGL(hw_probe_kernel, dim3(HW_PROBE_WGS), dim3(64), 0, probe_on[k], pd + (size_t)k * HW_PROBE_WGS);
            TGP_TRY(hipGetLastError());
            TGP_TRY(hipStreamSynchronize(probe_on[k]));
        }
        std::vector<unsigned long long> hp(3 * HW_PROBE_WGS);
        TGP_TRY(hipMemcpy(hp.data(), pd, hp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(pd);
        const std::string cus = std::string(stamp_path) + ".cus";
        if (FILE *fh = fopen(cus.c_str(), "wb")) {
            fwrite(hp.data(), sizeof(unsigned long long), hp.size(), fh);
            fclose(fh);
        }
    }
    return hipSuccess;
}

}  // namespace tgp
