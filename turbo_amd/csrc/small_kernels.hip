// small_kernels.hip -- the small-problem path: N <= 128 training points (the reference's own
// demo regime: demos/Branin-Hoo.ipynb cells 5-8 run 4..30 trials; turbo/plotting/trials.py:371,
// 448, 574-577 predict 200 .. 10^4 points per stored model).
//
//   small_fit_kernel    ONE workgroup does what launch_fit spreads over ~25 launches: kernel
//                       matrix (sklearn kernels.py:1553-1560 / 1708-1738, _gpr.py:346-347),
//                       Cholesky (_gpr.py:349) as one or two 64 x 64 blocks (chol64.hpp) with the
//                       off-diagonal block on MFMA, the inverse factor, alpha (_gpr.py:360-364)
//                       and the two scalars of the log marginal likelihood (_gpr.py:609-611).
//                       Inputs are read from, and the scalars written to, pinned host memory:
//                       no memcpy call, one launch, one stream synchronisation.
//   small_sweep_kernel  posterior mean / variance / acquisition (_gpr.py:443-494;
//                       turbo/modules/acquisition_functions.py:147-158, 225-247, 336-358) for 64
//                       candidates per workgroup in ONE launch: cross-kernel tile in LDS, the
//                       triangular contraction on MFMA against the (L2-resident) inverse factor,
//                       the same epilogue arithmetic as finalize_kernel.  Always f64.
#include <hip/hip_runtime.h>

#include <atomic>
#include <math.h>

#include <algorithm>

#include "chol64.hpp"
#include "lbfgs_wave.hpp"
#include "lds_opt_in.hpp"
#include "pairwise.hpp"
#include "small_grad.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

constexpr int T_LD = NB + 2;                       // padded row of a 64 x 64 LDS tile
constexpr int T_SZ = NB * T_LD;                    // doubles per tile
typedef double (*tile_t)[T_LD];

// thread tile (factor layout: rows 4tr.., cols 4tc..) -> LDS tile, optionally transposed
__device__ __forceinline__ void regs_to_tile(const double (&a)[4][4], tile_t T, bool transpose) {
    const int tid = threadIdx.x, tc = tid >> 4, tr = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (transpose) T[4 * tc + j][4 * tr + i] = a[i][j];
            else T[4 * tr + i][4 * tc + j] = a[i][j];
        }
}

// 64 x 64 tile of the kernel matrix in the factor layout: rows r0.., cols c0.. of K = c k(X, X) +
// (noise + jitter) I with the exact diagonal and identity padding beyond N (as kernel_matrix_kernel)
template <int KIND>
__device__ __forceinline__ void kmat_tile(const double *__restrict__ Xs, int r0, int c0, int N, int Np, int Dp,
                                          double constant, double noise, double jitter,
                                          double (*Ct)[PwCfg<double>::LD], double (*Xt)[PwCfg<double>::LD],
                                          double (&a)[4][4]) {
    const int tid = threadIdx.x, tc = tid >> 4, tr = tid & 15;
    // pairwise_sqdist gives thread (tx = tid & 15, ty = tid >> 4) the distances of P rows 4ty.. to Q
    // rows 4tx..; with P = the tile's COLUMN points and Q = its ROW points that is exactly the
    // transpose of this thread's (rows 4tr.., cols 4tc..) tile
    double d2[4][4];
    pairwise_sqdist<double>(Xs, c0, Np, Xs, r0, Np, Dp, Ct, Xt, d2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = r0 + 4 * tr + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = c0 + 4 * tc + j;
            double v;
            if (gi == gj) v = (gi < N) ? ((constant * 1.0 + noise) + jitter) : 1.0;
            else if (gi < N && gj < N) v = kernel_value<double, KIND>(d2[j][i], constant);
            else v = 0.0;
            a[i][j] = v;
        }
    }
}

// the same with the pairwise staging buffers placed in a given (free) LDS tile
template <int KIND>
__device__ __forceinline__ void kmat_tile_nolds(const double *__restrict__ Xs, int r0, int c0, int N, int Np, int Dp,
                                                double constant, double noise, double jitter, tile_t stage,
                                                double (&a)[4][4]) {
    double (*Ct)[PwCfg<double>::LD] = reinterpret_cast<double (*)[PwCfg<double>::LD]>(reinterpret_cast<double *>(stage));
    double (*Xt)[PwCfg<double>::LD] = Ct + PwCfg<double>::DC;
    kmat_tile<KIND>(Xs, r0, c0, N, Np, Dp, constant, noise, jitter, Ct, Xt, a);
}

struct SmallFitArgs {
    const double *in;        // pinned host (mapped): xs[Nin * Dp] | yn[Nin] | ls[D],  Nin = 64 or 128 rows
    double *Xs, *yn, *ls;    // device copies the later sweeps read (Xs: Np x Dp, yn: Np, ls: D)
    double *K, *Linv, *alpha;
    float *Xs32, *Linv32;    // f32 copies for f32 handles, or null
    double *res;             // pinned host (mapped): [sum log diag L, yn . alpha, first bad pivot + 1]
    int N, D, Dp, Np;
    int zero_to;             // rows / columns [Nin, zero_to) of Linv may hold an older factor: cleared here
    double constant, noise, jitter, tiny;
    Bell bell;               // polled completion (doorbell.hpp); word == null: the caller synchronises the stream
    int legacy;              // TGP_SMALL_LIVE=0 (A/B): round 5's body -- the pivot chain walks the padding, the targets are fetched in front of alpha
};

constexpr int SF_SCRATCH = 2304;                                // doubles: factorisation buffers + vectors
constexpr size_t SMALL_FIT_LDS = (size_t)(SF_SCRATCH + 4 * T_SZ) * sizeof(double);

// sm: SMALL_FIT_LDS bytes of LDS; sflag_p: one int of LDS (the first failing pivot + 1)
template <int KIND>
__device__ __forceinline__ void small_fit_body(const SmallFitArgs &p, double *sm, int *sflag_p) {
    int &sflag = *sflag_p;
    double *scratch = sm;                                       // chol64 buffers [0, 2112), then yn / z / alpha vectors
    tile_t T0 = reinterpret_cast<tile_t>(sm + SF_SCRATCH);
    tile_t T1 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + T_SZ);
    tile_t T2 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + 2 * T_SZ);
    tile_t T3 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + 3 * T_SZ);
    const int tid = threadIdx.x, tc = tid >> 4, tr = tid & 15;
    const int N = p.N, Np = p.Np, Dp = p.Dp;
    const int nblk = (N + NB - 1) / NB;                         // 1 or 2 diagonal blocks
    const int Nin = nblk * NB;

    // ---- inputs: pinned host -> device buffers (later sweeps / appends read them) ------------
    if (tid == 0) sflag = 0;
    // The inputs sit in device-mapped HOST memory: a load is a PCIe round trip (~1.7 us).  Round 5's three copy loops --
    // load, wait, store, each -- and the targets' second fetch made FOUR of them in a row, 7 us of a 20 us fit (phase
    // stamps, profiles/r06_short_calls.txt).  Now every load of the common sizes is issued before the first wait: the
    // first 1024 scaled coordinates (four per thread), the targets, the length scales; larger inputs follow in a loop.
    double yn_lo = 0.0, yn_hi = 0.0;
    if (!p.legacy) {
        const int nx = Nin * Dp;
        double x4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x4[k] = (tid + 256 * k < nx) ? p.in[tid + 256 * k] : 0.0;
        const double yv = (tid < Nin) ? p.in[nx + tid] : 0.0;
        const double lv = (tid < p.D) ? p.in[nx + Nin + tid] : 0.0;
        if (tid < 64) { yn_lo = p.in[nx + tid]; if (nblk == 2) yn_hi = p.in[nx + NB + tid]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            if (i < Np * Dp) {
                p.Xs[i] = x4[k];
                if (p.Xs32) p.Xs32[i] = (float)x4[k];
            }
        }
        for (int i = tid + 1024; i < Np * Dp; i += 256) {
            const double v = (i < nx) ? p.in[i] : 0.0;
            p.Xs[i] = v;
            if (p.Xs32) p.Xs32[i] = (float)v;
        }
        if (tid < Np) p.yn[tid] = yv;                          // (Np <= 256: one pass)
        for (int i = tid + 256; i < Np; i += 256) p.yn[i] = 0.0;
        if (tid < p.D) p.ls[tid] = lv;
        for (int i = tid + 256; i < p.D; i += 256) p.ls[i] = p.in[nx + Nin + i];
    } else {
        for (int i = tid; i < Np * Dp; i += 256) {
            const double v = (i < Nin * Dp) ? p.in[i] : 0.0;
            p.Xs[i] = v;
            if (p.Xs32) p.Xs32[i] = (float)v;
        }
        for (int i = tid; i < Np; i += 256) p.yn[i] = (i < Nin) ? p.in[Nin * Dp + i] : 0.0;
        for (int i = tid; i < p.D; i += 256) p.ls[i] = p.in[Nin * Dp + Nin + i];
    }
    // the inverse factor is zero outside the corner this kernel writes; only the band an older,
    // larger factor may have left behind needs clearing (the host tracks its extent)
    if (p.zero_to > Nin) {
        const int Z = p.zero_to;
        const d2_t z2 = {0.0, 0.0};
        for (int i = tid; i < Z * (Z / 2); i += 256) {
            const int r = i / (Z / 2), c = 2 * (i - r * (Z / 2));
            if (r >= Nin || c >= Nin) {
                *reinterpret_cast<d2_t *>(p.Linv + (long)r * Np + c) = z2;
                if (p.Linv32) { p.Linv32[(long)r * Np + c] = 0.f; p.Linv32[(long)r * Np + c + 1] = 0.f; }
            }
        }
    }
    __syncthreads();   // Xs is read back below by this same workgroup
    // (phase stamps of a polled call, thread 0: bell.word[3..6] = inputs staged | first kernel-matrix tile in LDS |
    // first block factored | fit done -- four posted 8-byte writes; tools/bench_short_calls.py --phases prints them)
    unsigned long long *const phase = (p.bell.word && blockIdx.x == 0 && tid == 0) ? p.bell.word : nullptr;
    if (phase) phase[3] = wall_clock64();

    double (*Ct)[PwCfg<double>::LD] = reinterpret_cast<double (*)[PwCfg<double>::LD]>(reinterpret_cast<double *>(T3));
    double (*Xt)[PwCfg<double>::LD] = Ct + PwCfg<double>::DC;   // the pairwise staging lives in T3 until T3 is needed
    double *vyn = scratch + 2112, *vz = vyn + 64;               // 2 x 64 doubles inside the scratch region

    // helpers on whole LDS tiles (256 threads, 16 elements each)
    auto tile_zero = [&](tile_t T) {
        for (int idx = tid; idx < NB * NB; idx += 256) T[idx >> 6][idx & 63] = 0.0;
    };
    auto store_L = [&](tile_t T, int r0, int c0) {              // lower triangle of a diagonal block of L
        if (!p.K) return;
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            p.K[(long)(r0 + r) * Np + c0 + c] = (c <= r) ? T[r][c] : 0.0;
        }
    };
    auto store_linv = [&](tile_t T, int r0, int c0, double sign) {
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const long off = (long)(r0 + r) * Np + c0 + c;
            const double v = sign * T[r][c];
            p.Linv[off] = v;
            if (p.Linv32) p.Linv32[off] = (float)v;
        }
    };

    double a[4][4], x11[4][4];
    // ---- block (0, 0): A11 -> T0, factor in LDS (chol64.hpp variant D): T0 = L11, T1 = X11 = L11^-1 ----
    kmat_tile<KIND>(p.Xs, 0, 0, N, Np, Dp, p.constant, p.noise, p.jitter, Ct, Xt, a);
    regs_to_tile(a, T0, false);
    tile_zero(T1);
    __syncthreads();
    if (phase) phase[4] = wall_clock64();
    factor64_v4(T0, T1, T2, scratch, 0, &sflag, p.tiny, (N < NB && !p.legacy) ? N : NB);   // (the pivot chain stops at the last live 16-column block)
    if (phase) phase[5] = wall_clock64();
    double sumlog = 0.0;
    if (tid < 64) sumlog = log(T0[tid][tid]);
    store_L(T0, 0, 0);
    store_linv(T1, 0, 0, 1.0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) x11[i][j] = T1[4 * tr + i][4 * tc + j];     // keep X11 in registers too

    if (nblk == 2) {
        // ---- block (1, 0): L21 = A21 X11^T ----
        __syncthreads();                                        // T0 / T1 readers above are done; T3 (pairwise staging) free
        kmat_tile<KIND>(p.Xs, NB, 0, N, Np, Dp, p.constant, p.noise, p.jitter, Ct, Xt, a);
        regs_to_tile(a, T2, false);                             // T2 = A21
        __syncthreads();
        d4_t acc[2][2];
        acc_zero(acc);
        tile_mma64(T2, T1, acc);                                // A21 * X11^T
        __syncthreads();                                        // pairwise staging in T3 is no longer read
        acc_foreach(acc, [&](int r, int c, double v) {
            T3[r][c] = v;                                       // T3 = L21
            if (p.K) p.K[(long)(NB + r) * Np + c] = v;
        });
        // ---- block (1, 1): A22 - L21 L21^T -> T2, factor: T2 = L22, T0 = X22 ----
        kmat_tile_nolds<KIND>(p.Xs, NB, NB, N, Np, Dp, p.constant, p.noise, p.jitter, T0, a);   // staging in T0 (L11 is stored)
        __syncthreads();
        regs_to_tile(a, T2, false);                             // T2 = A22 (A21 has been consumed)
        acc_zero(acc);
        tile_mma64(T3, T3, acc);
        __syncthreads();
        acc_foreach(acc, [&](int r, int c, double v) { T2[r][c] -= v; });
        tile_zero(T0);
        __syncthreads();
        factor64_v4(T2, T0, T1, scratch, NB, &sflag, p.tiny, p.legacy ? NB : N - NB);   // (T1 = scratch tile of the merges: X11 lives on in registers)
        if (tid < 64) sumlog += log(T2[tid][tid]);
        store_L(T2, NB, NB);
        store_linv(T0, NB, NB, 1.0);
        // the upper-right corner block of L and of Linv is zero
        for (int i = tid; i < NB * NB; i += 256) {
            const long off = (long)(i >> 6) * Np + NB + (i & 63);
            if (p.K) p.K[off] = 0.0;
            p.Linv[off] = 0.0;
            if (p.Linv32) p.Linv32[off] = 0.f;
        }
        // ---- X21 = -X22 (L21 X11) ----
        __syncthreads();
        regs_to_tile(x11, T1, true);                            // T1 = X11^T
        __syncthreads();
        acc_zero(acc);
        tile_mma64(T3, T1, acc);                                // P = L21 * X11
        __syncthreads();                                        // L22 (T2) is stored, T1 / T3 have been read
        acc_foreach(acc, [&](int r, int c, double v) { T2[c][r] = v; });   // T2 = P^T
        regs_to_tile(x11, T1, false);                           // T1 = X11 again, for alpha
        __syncthreads();
        acc_zero(acc);
        tile_mma64(T0, T2, acc);                                // X22 * P
        acc_foreach(acc, [&](int r, int c, double v) { T3[r][c] = -v; });  // T3 = X21 (L21 has been consumed)
        __syncthreads();
        store_linv(T3, NB, 0, 1.0);
    }
    __syncthreads();

    // ---- alpha = Linv^T (Linv yn), yn . alpha -- X11 in T1, X21 in T3, X22 in T0 ----
    double *vyn2 = scratch, *vz2 = scratch + 64, *valpha = scratch + 128;   // the factorisation buffers are free now
    if (tid < 64) {
        if (p.legacy) { yn_lo = p.in[Nin * Dp + tid]; if (nblk == 2) yn_hi = p.in[Nin * Dp + NB + tid]; }
        vyn[tid] = yn_lo;
        if (nblk == 2) vyn2[tid] = yn_hi;
    }
    __syncthreads();
    {
        // z: 4 lanes per row, 16 columns each, fixed-order reduce
        const int r = tid >> 2, q = tid & 3;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int c = 16 * q + cc;
            s1 = fma(T1[r][c], vyn[c], s1);
            if (nblk == 2) s2 = fma(T3[r][c], vyn[c], fma(T0[r][c], vyn2[c], s2));
        }
        s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
        if (q == 0) { vz[r] = s1; vz2[r] = s2; }
    }
    __syncthreads();
    {
        // alpha_c = sum_r Linv[r][c] z_r: 4 lanes per column, 16 rows each
        const int c = tid >> 2, q = tid & 3;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = 16 * q + rr;
            s1 = fma(T1[r][c], vz[r], s1);
            if (nblk == 2) { s1 = fma(T3[r][c], vz2[r], s1); s2 = fma(T0[r][c], vz2[r], s2); }
        }
        s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
        if (q == 0) { valpha[c] = s1; valpha[64 + c] = s2; }
    }
    __syncthreads();
    for (int i = tid; i < Np; i += 256) p.alpha[i] = (i < Nin) ? valpha[i] : 0.0;
    if (tid < 64) {
        double ya = vyn[tid] * valpha[tid] + (nblk == 2 ? vyn2[tid] * valpha[64 + tid] : 0.0);
        double sl = sumlog;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ya += __shfl_xor(ya, off, 64); sl += __shfl_xor(sl, off, 64); }
        if (tid == 0) { p.res[0] = sl; p.res[1] = ya; p.res[2] = (double)sflag; }
    }
    if (phase) phase[6] = wall_clock64();
}

template <int KIND>
__global__ __launch_bounds__(256) void small_fit_kernel(SmallFitArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int sflag;
    bell_start(p.bell);
    small_fit_body<KIND>(p, sm, &sflag);
    bell_ring(p.bell, 1);
}

// one workgroup per model (tgp_predict_batch): the argument records live in pinned host memory
template <int KIND>
__global__ __launch_bounds__(256) void small_fit_batch_kernel(const SmallFitArgs *__restrict__ args) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int sflag;
    const SmallFitArgs p = args[blockIdx.x];
    small_fit_body<KIND>(p, sm, &sflag);
}

static SmallFitArgs small_fit_args(Context &c, const Bell &bell) {
    SmallFitArgs a{};
    a.in = c.d_pin_in; a.Xs = c.d_Xs; a.yn = c.d_yn; a.ls = c.d_ls;
    a.K = c.d_K; a.Linv = c.d_Linv; a.alpha = c.d_alpha;
    a.Xs32 = c.dtype != TGP_F64 ? c.d_Xs32 : nullptr;
    a.Linv32 = c.dtype != TGP_F64 ? c.d_Linv32 : nullptr;
    a.res = c.d_pin_out;
    a.N = (int)c.N; a.D = (int)c.D; a.Dp = (int)c.Dp; a.Np = (int)c.Np;
    // Linv is known to be zero from row / column c.linv_extent on (for the leading dimension it
    // was written with); a change of the leading dimension moves everything: clear it all
    a.zero_to = (c.linv_ld == c.Np) ? (int)std::min<int64_t>(c.linv_extent, c.Np) : (int)c.Np;
    a.constant = c.constant; a.noise = c.noise; a.jitter = c.jitter;
    a.tiny = 8.0 * 2.220446049250313e-16 * ((c.constant + c.noise) + c.jitter);
    a.bell = bell;
    a.legacy = tuning().small_live == 0 ? 1 : 0;
    return a;
}

hipError_t launch_small_fit(Context &c, const Bell &bell) {
    const SmallFitArgs a = small_fit_args(c, bell);
    void (*k)(SmallFitArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_fit_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_fit_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_fit_kernel<TGP_MATERN32>; break;
        default: k = small_fit_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, SMALL_FIT_LDS));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), SMALL_FIT_LDS, c.stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// small_hyper_kernel: the hyper-parameter fit of a small problem (N <= 128) in ONE launch -- what
// GaussianProcessRegressor.fit does with optimizer='fmin_l_bfgs_b' (_gpr.py:296-337: L-BFGS-B on
// -LML over theta = log(constant, length scale(s), noise), the first start = the current theta,
// the others drawn by the caller).  One workgroup per start runs its whole optimisation: scaled
// inputs for the trial theta -> small_fit_body -> small_grad_body (each block pair in turn) ->
// one step of the projected L-BFGS of lbfgs_wave.hpp (wave 0, lane = hyper-parameter), until the
// start has converged.  The starts run side by side, and nothing goes back to the host in between.
// ------------------------------------------------------------------------------------------
struct SmallHyperArgs {
    const double *X, *yn;            // (N, D) inputs as given, (N) normalised targets
    const double *theta0;            // (S, P) starts, P = 2 + n_ls: log constant, log length scale(s), log noise
    const double *blo, *bhi;         // (P) bounds in log space
    double *ws;                      // S x ws_stride doubles of device memory
    double *theta_out, *f_out, *info;   // (S, P), (S) = -LML there, (3 S): status, accepted steps, evaluations
    long ws_stride;
    int N, D, Dp, n_ls, max_iter;
    double jitter, pgtol, ftol;
    // 64 < N <= 128: the three block pairs of the gradient on THREE workgroups per start (wgs = 3), each
    // running the fit redundantly and one pair, behind a barrier of their own (bar[start], counting up)
    int wgs;
    unsigned *bar;                   // S counters, zero at launch
    double *shares;                  // S x 2 x 3 x SMALL_GRAD_OUT_STRIDE: the pairs' sums, two slots used in turn
    int legacy;                      // TGP_SMALL_LIVE=0: SmallFitArgs::legacy of every evaluation
};

__host__ __device__ inline long hyper_even(long v) { return (v + 1) & ~1L; }
// per start: in | Xs | yn | ls | Linv | alpha | res (8) | gradient shares (3 x 72) | history (2 x RF_MEM x 64) | rho |
//            the start's own copy of X and the targets (the arguments may be device-mapped host memory)
// (per WORKGROUP; small_hyper_workspace_doubles() is what a start needs)
__host__ __device__ inline long small_hyper_ws_doubles(int N, int D, int Dp) {
    const long Nin = ((N + NB - 1) / NB) * NB;
    return hyper_even(Nin * Dp + Nin + D) + hyper_even(Nin * Dp) + Nin + hyper_even(D) + Nin * Nin + Nin + 8 +
           3 * SMALL_GRAD_OUT_STRIDE + 2 * RF_MEM * 64 + RF_MEM + hyper_even((long)N * D) + Nin;
}
// LDS: the larger of the two bodies' needs, then the optimiser's history (2 x RF_MEM x 64 + RF_MEM doubles:
// in global memory the two-loop recursion's dependent loads made the step 10 us of a 40 us evaluation)
constexpr size_t SMALL_HYPER_BODY_LDS = SMALL_FIT_LDS > SMALL_GRAD_LDS ? SMALL_FIT_LDS : SMALL_GRAD_LDS;
constexpr size_t SMALL_HYPER_LDS = SMALL_HYPER_BODY_LDS + (size_t)(2 * RF_MEM * 64 + RF_MEM) * sizeof(double);
static_assert(SMALL_HYPER_LDS + 1024 <= 160 * 1024, "small_hyper_kernel: LDS (dynamic + th / flags) exceeds 160 KB");

// The two bodies as REAL calls: inlined into the optimiser's loop they shared one register allocation
// with it and with each other, and the kernel spilled 150-167 VGPRs to scratch.  The LDS pointers
// travel as address-space-3 pointers, so the callees keep addressing LDS with ds_ instructions, and
// neither callee declares LDS of its own (what broke the first attempt at this: DESIGN.md section 8).
typedef __attribute__((address_space(3))) double *lds_dptr;
typedef __attribute__((address_space(3))) int *lds_iptr;
template <int KIND>
__device__ __noinline__ void small_fit_call(SmallFitArgs a, lds_dptr sm3, lds_iptr flag3) {
    small_fit_body<KIND>(a, (double *)sm3, (int *)flag3);
}
template <int KIND>
__device__ __noinline__ void small_grad_call(SmallGradArgs a, int pr, lds_dptr sm3) {
    small_grad_body<KIND>(a, pr, (double *)sm3);
}

// ------------------------------------------------------------------------------------------
// small_fit_grad_kernel (round 6): ONE evaluation of the hyper-parameter objective of a small problem --
// the fit and the LML gradient (tgp_fit_grad; turbo/modules/surrogates.py:313-318 -> _gpr.py:584-650) -- in
// ONE launch with a polled completion.  Round 5 took two launches (small_fit_kernel, then one workgroup per
// block pair of the gradient), two event records and a stream synchronisation: 34 us a call around ~30 us
// of kernels.  Here workgroup q owns block pair q of the gradient (one pair up to N = 64, three up to 128)
// and runs the fit itself first: workgroup 0 writes the handle's resident state, the others a private copy
// (same inputs, same code: the same bytes), so nobody waits for anybody.  Bodies, arithmetic and the order of
// the host's sums are small_fit_kernel's and small_grad_kernel's: the results are theirs bit for bit.
// ------------------------------------------------------------------------------------------
struct SmallFitGradArgs {
    SmallFitArgs fit;        // workgroup 0's view: the handle's buffers, res in device-mapped host memory
    double *gout;            // device-mapped host memory: workgroup q's sums at gout + q * SMALL_GRAD_OUT_STRIDE
    double *ws;              // device memory: the private fit state of workgroups 1 and 2
    long ws_stride;
    int ard;
};
__host__ __device__ inline long small_fit_grad_ws_doubles(int Nin, int D, int Dp) {
    return hyper_even((long)Nin * Dp) + Nin + hyper_even(D) + (long)Nin * Nin + Nin + 8;
}

template <int KIND>
__global__ __launch_bounds__(256) void small_fit_grad_kernel(SmallFitGradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int sflag;
    bell_start(p.fit.bell);
    const int q = blockIdx.x;
    SmallFitArgs fa = p.fit;
    const int Nin = ((fa.N + NB - 1) / NB) * NB;
    if (q > 0) {
        double *w = p.ws + (long)(q - 1) * p.ws_stride;
        fa.Xs = w; w += hyper_even((long)Nin * fa.Dp);
        fa.yn = w; w += Nin;
        fa.ls = w; w += hyper_even(fa.D);
        fa.Linv = w; w += (long)Nin * Nin;
        fa.alpha = w; w += Nin;
        fa.res = w;
        fa.K = nullptr; fa.Xs32 = nullptr; fa.Linv32 = nullptr;
        fa.Np = Nin; fa.zero_to = 0;
    }
    // (both bodies INLINED: as non-inlined callees -- what the optimiser loop of small_hyper_kernel needs -- they cost this
    // kernel a 796-byte stack per lane, i.e. a scratch allocation of ~300 MB on the queue the first time it ran)
    small_fit_body<KIND>(fa, sm, &sflag);
    SmallGradArgs ga{};
    ga.Xs = fa.Xs; ga.alpha = fa.alpha; ga.Linv = fa.Linv; ga.out = p.gout;
    ga.N = fa.N; ga.Np = fa.Np; ga.Dp = fa.Dp; ga.ard = p.ard;
    __syncthreads();
    small_grad_body<KIND>(ga, q, sm);
    bell_ring(p.fit.bell, gridDim.x);
}

hipError_t launch_small_fit_grad(Context &c, bool ard, double *gout_host, const Bell &bell) {
    SmallFitGradArgs a{};
    a.fit = small_fit_args(c, bell);
    a.gout = gout_host;
    a.ard = ard ? 1 : 0;
    const int Nin = (int)((c.N + NB - 1) / NB) * NB;
    const unsigned npair = c.N > NB ? 3 : 1;
    a.ws = c.d_sfg;
    a.ws_stride = small_fit_grad_ws_doubles(Nin, (int)c.D, (int)c.Dp);
    void (*k)(SmallFitGradArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_fit_grad_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_fit_grad_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_fit_grad_kernel<TGP_MATERN32>; break;
        default: k = small_fit_grad_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, SMALL_HYPER_BODY_LDS));
    hipLaunchKernelGGL(k, dim3(npair), dim3(256), SMALL_HYPER_BODY_LDS, c.stream, a);
    return hipGetLastError();
}
// bytes of c.d_sfg the launch above needs (two private fit states at the largest small problem: N = 128, Dp = 64)
size_t small_fit_grad_ws_bytes() { return 2 * (size_t)small_fit_grad_ws_doubles(2 * NB, 64, 64) * sizeof(double); }

constexpr unsigned long long HYPER_BAR_BUDGET_TICKS = 200000000ull;   // 2 s of wall_clock64() at its constant 100 MHz

template <int KIND>
__global__ __launch_bounds__(256) void small_hyper_kernel(SmallHyperArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double th[64];
    __shared__ int done;
    __shared__ int sflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N, D = p.D, Dp = p.Dp, P = 2 + p.n_ls;
    const int Nin = ((N + NB - 1) / NB) * NB;
    const int wgs = p.wgs;
    const int s = (int)blockIdx.x / wgs, q = (int)blockIdx.x - s * wgs;   // start, and this workgroup's block pair when wgs = 3
    double *in = p.ws + (long)blockIdx.x * p.ws_stride;                    // every workgroup has a workspace of its own
    __shared__ int bar_fail;
    double *Xs = in + hyper_even((long)Nin * Dp + Nin + D);
    double *ynb = Xs + hyper_even((long)Nin * Dp);
    double *lsb = ynb + Nin;
    double *Linv = lsb + hyper_even(D);
    double *alpha = Linv + (long)Nin * Nin;
    double *res = alpha + Nin;
    double *gout = res + 8;
    double (*Sv)[64] = reinterpret_cast<double (*)[64]>(sm + SMALL_HYPER_BODY_LDS / sizeof(double));
    double (*Yv)[64] = Sv + RF_MEM;
    double *rh = reinterpret_cast<double *>(Yv + RF_MEM);
    double *Xr = gout + 3 * SMALL_GRAD_OUT_STRIDE + 2 * RF_MEM * 64 + RF_MEM, *yr = Xr + hyper_even((long)N * D);
    for (int i = tid; i < N * D; i += 256) Xr[i] = p.X[i];
    for (int i = tid; i < N; i += 256) yr[i] = p.yn[i];
    const bool on = lane < P;
    const int li = on ? lane : 0;
    const double lo_i = p.blo[li], hi_i = p.bhi[li];
    double th_i = on ? rf_clip(p.theta0[(long)s * P + li], lo_i, hi_i) : 0.0;    // (wave 0's copy is the one that counts)
    if (wave == 0) {
        th[lane] = th_i;
        if (lane < RF_MEM) rh[lane] = 0.0;
        if (lane == 0) done = 0;
    }
    RfWave w{};
    int evals = 0;
    const bool ard = p.n_ls > 1;
    const int npair = Nin > NB ? 3 : 1;
    for (int it = 0; it <= p.max_iter; ++it) {
        __syncthreads();
        const double constant = exp(th[0]), noise = exp(th[P - 1]);
        // the fit's inputs for this theta: X / length scale (zero padded), targets, length scales
        for (int i = tid; i < Nin * Dp; i += 256) {
            const int r = i / Dp, d = i - r * Dp;
            in[i] = (r < N && d < D) ? Xr[(long)r * D + d] / exp(th[1 + (ard ? d : 0)]) : 0.0;
        }
        for (int i = tid; i < Nin; i += 256) in[(long)Nin * Dp + i] = (i < N) ? yr[i] : 0.0;
        for (int d = tid; d < D; d += 256) in[(long)Nin * Dp + Nin + d] = exp(th[1 + (ard ? d : 0)]);
        __syncthreads();
        SmallFitArgs fa{};
        fa.in = in; fa.Xs = Xs; fa.yn = ynb; fa.ls = lsb; fa.K = nullptr; fa.Linv = Linv; fa.alpha = alpha;
        fa.Xs32 = nullptr; fa.Linv32 = nullptr; fa.res = res;
        fa.N = N; fa.D = D; fa.Dp = Dp; fa.Np = Nin; fa.zero_to = 0;
        fa.constant = constant; fa.noise = noise; fa.jitter = p.jitter;
        fa.tiny = 8.0 * 2.220446049250313e-16 * ((constant + noise) + p.jitter);
        fa.legacy = p.legacy;
        small_fit_call<KIND>(fa, (lds_dptr)sm, (lds_iptr)&sflag);
        SmallGradArgs ga{};
        ga.Xs = Xs; ga.alpha = alpha; ga.Linv = Linv; ga.out = gout;
        ga.N = N; ga.Np = Nin; ga.Dp = Dp; ga.ard = ard ? 1 : 0;
        const double *gsrc = gout;
        if (wgs == 1) {
            for (int pr = 0; pr < npair; ++pr) {
                __syncthreads();
                small_grad_call<KIND>(ga, pr, (lds_dptr)sm);
            }
            __syncthreads();
        } else {
            // this workgroup's pair into the start's slot of this iteration, then the start's barrier:
            // a counter that only goes up (3 per iteration), a bounded spin (every wave leaves: on a
            // time-out the start ends with status 3), agent-scope fences either side so that the other
            // two workgroups' sums -- written on other CUs, possibly behind another XCD's L2 -- are seen.
            // PROGRESS: a start's three workgroups are consecutive block ids and the dispatcher hands
            // workgroups out in order, so whenever one of the three is resident the other two are either
            // resident or next in line for the CUs the finished starts free -- that, not the host's
            // 3 S <= CU-count test (which HSA_CU_MASK or another process on the card can falsify), is why
            // the spin ends.  Should it not (the budget below), the host relaunches with one workgroup per
            // start and hands nothing of the failed launch to the caller.
            double *slot = p.shares + ((long)s * 2 + (it & 1)) * 3 * SMALL_GRAD_OUT_STRIDE;
            ga.out = slot;
            gsrc = slot;
            __syncthreads();
            small_grad_call<KIND>(ga, q, (lds_dptr)sm);
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                const unsigned target = 3u * (unsigned)(it + 1);
                __hip_atomic_fetch_add(p.bar + s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                // bounded by TIME, not by an iteration count: wall_clock64() ticks at the device's constant
                // 100 MHz, HYPER_BAR_BUDGET_TICKS = 2 s -- an evaluation takes ~150 us, so this only ends a
                // launch whose partner workgroups never got a CU
                int ok = 0;
                const unsigned long long t0 = wall_clock64();
                do {
                    if (__hip_atomic_load(p.bar + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                } while (wall_clock64() - t0 < HYPER_BAR_BUDGET_TICKS);
                bar_fail = ok ? 0 : 1;
            }
            __syncthreads();
            __threadfence();
        }
        if (wave == 0) {
            // -LML and its gradient in log space (tgp_fit_grad's arithmetic: _gpr.py:609-611, :643-647)
            double phit, gt_i = 0.0;
            if (res[2] != 0.0) {
                phit = INFINITY;                         // not positive definite: -inf likelihood, zero gradient (_gpr.py:586-589)
            } else {
                phit = 0.5 * res[1] + res[0] + 0.5 * (double)N * 1.8378770664093453;   // log(2 pi)
                double g = 0.0;
                if (on) {
                    const int slot = lane == 0 ? 0 : (lane == P - 1 ? 2 : (ard ? 3 + (lane - 1) : 1));
                    double sh = __hip_atomic_load(gsrc + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int pr = 1; pr < npair; ++pr)
                        sh += __hip_atomic_load(gsrc + pr * SMALL_GRAD_OUT_STRIDE + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) g = 0.5 * constant * sh;
                    else if (lane == P - 1) g = 0.5 * noise * sh;
                    else g = ard ? constant * sh : 0.5 * constant * sh;
                }
                gt_i = -g;
            }
            (void)rf_wave_step(w, th_i, gt_i, phit, it == 0, on, lane, lo_i, hi_i, p.pgtol, p.ftol, Sv, Yv, rh);
            ++evals;
            th[lane] = th_i;
            if (wgs > 1 && bar_fail) w.status = 3;      // the start's barrier timed out: give up (reported as an error)
            if (lane == 0) done = (w.status != 0) ? 1 : 0;
        }
        __syncthreads();
        if (done) break;
    }
    if (wave == 0 && q == 0) {
        if (on) p.theta_out[(long)s * P + li] = w.x_i[0];
        if (lane == 0) {
            p.f_out[s] = w.phi;
            p.info[3 * s] = (double)w.status;
            p.info[3 * s + 1] = (double)w.iters;
            p.info[3 * s + 2] = (double)evals;
        }
    }
}

// per start: three workgroup workspaces + the two slots of gradient shares; S barrier counters in front of everything
constexpr long HYPER_BAR_DOUBLES = 64;
static long hyper_start_doubles(int N, int D, int Dp) { return 3 * small_hyper_ws_doubles(N, D, Dp) + 2 * 3 * SMALL_GRAD_OUT_STRIDE; }
long small_hyper_workspace_doubles(int N, int D, int Dp) { return hyper_start_doubles(N, D, Dp) + HYPER_BAR_DOUBLES; }

hipError_t launch_small_hyper(Context &c, int kernel, const double *d_X, const double *d_yn, const double *d_theta0,
                              const double *d_blo, const double *d_bhi, int S, int N, int D, int Dp, int n_ls,
                              int max_iter, double jitter, double *d_ws, double *d_theta, double *d_f, double *d_info,
                              bool one_wg_per_start) {
    SmallHyperArgs a{};
    a.X = d_X; a.yn = d_yn; a.theta0 = d_theta0; a.blo = d_blo; a.bhi = d_bhi;
    a.theta_out = d_theta; a.f_out = d_f; a.info = d_info;
    a.ws_stride = small_hyper_ws_doubles(N, D, Dp);
    a.N = N; a.D = D; a.Dp = Dp; a.n_ls = n_ls; a.max_iter = max_iter;
    a.jitter = jitter; a.pgtol = 1e-5; a.ftol = 2.220446049250313e-09;   // SciPy's L-BFGS-B defaults (factr 1e7)
    a.legacy = tuning().small_live == 0 ? 1 : 0;
    // Three workgroups per start when the gradient has three block pairs (64 < N <= 128) AND all 3 S
    // workgroups can be resident at once (one per CU at this LDS size): the barrier between a start's
    // three must never wait for a workgroup that has no CU.  TGP_HYPER_WGS=1 keeps one workgroup per start.
    const int wgs_env = tuning().hyper_wgs;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess) { (void)hipGetLastError(); ncu = 0; }
    a.wgs = (N > NB && 3 * S <= ncu && wgs_env != 1 && !one_wg_per_start) ? 3 : 1;
    // [S counters | S x (wgs workspaces) | S x 2 x 3 shares]  (S <= 64 counters fit the 64 doubles in front)
    a.bar = reinterpret_cast<unsigned *>(d_ws);
    a.ws = d_ws + HYPER_BAR_DOUBLES;
    a.shares = a.ws + (long)S * 3 * a.ws_stride;
    TGP_TRY(hipMemsetAsync(a.bar, 0, (size_t)HYPER_BAR_DOUBLES * sizeof(double), c.stream));
    void (*k)(SmallHyperArgs);
    switch (kernel) {
        case TGP_RBF: k = small_hyper_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_hyper_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_hyper_kernel<TGP_MATERN32>; break;
        default: k = small_hyper_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, SMALL_HYPER_LDS));
    hipLaunchKernelGGL(k, dim3((unsigned)(S * a.wgs)), dim3(256), SMALL_HYPER_LDS, c.stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Sweep for N <= 128: 64 candidates per workgroup, everything in one launch.
// ------------------------------------------------------------------------------------------
struct SmallSweepArgs {
    const double *cand;      // (M, D) raw candidates: device memory, or pinned host memory (mapped)
    const double *ls, *Xs, *Linv, *alpha;
    double *mu, *sigma, *acqv;        // nullable outputs (device, or pinned host memory)
    double *bval; long long *bidx; long long *counters;
    long M;
    int N, D, Dp, Np;
    double constant, kss, y_mean, y_std;
    int acq; double sf, incumbent, param;
};

__device__ __forceinline__ double ndtr_small(double a) {
    // scipy.special.ndtr (cephes ndtr.c), as finalize_kernel
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    const double y = 0.5 * erfc(z);
    return x > 0 ? 1.0 - y : y;
}

constexpr size_t SMALL_SWEEP_LDS = (size_t)(4 * T_SZ + 1024) * sizeof(double);

template <int KIND>
__device__ __forceinline__ void small_sweep_body(const SmallSweepArgs &p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    tile_t Ks0 = reinterpret_cast<tile_t>(sm);                  // cross-kernel, candidates x training points 0..63
    tile_t Ks1 = reinterpret_cast<tile_t>(sm + T_SZ);           //                                   ... 64..127
    tile_t Lt = reinterpret_cast<tile_t>(sm + 2 * T_SZ);        // one 64 x 64 block of Linv
    double *stage = sm + 3 * T_SZ;                              // scaled candidates (64 x Dp chunk) + pairwise staging
    double *red = sm + 4 * T_SZ;                                // [2][64] row-half partial sums, [64] means, arg-max scratch
    const int tid = threadIdx.x;
    const int N = p.N, Np = p.Np, D = p.D, Dp = p.Dp;
    const int nblk = (N + NB - 1) / NB;
    const long c0 = (long)blockIdx.x * NB;

    // the (up to three) 64 x 64 blocks of Linv, requested now and parked in registers: they arrive
    // behind the cross-kernel phase instead of three round trips in front of the MFMA phase
    d2_t lpre[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t > 0 && nblk < 2) break;
        const int rb = t > 0 ? 1 : 0, cb = t == 2 ? 1 : 0;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            lpre[t][p8] = *reinterpret_cast<const d2_t *>(p.Linv + (long)(rb * NB + r) * Np + cb * NB + c2);
        }
    }

    // ---- cross-kernel tile(s): Ks[c][j] = constant * k(cand_c / ls, Xs_j), 0 for j >= N ----
    // direct sum of squared differences, 16 dimensions per pass (pairwise.hpp says why)
    double (*Ct)[PwCfg<double>::LD] = reinterpret_cast<double (*)[PwCfg<double>::LD]>(stage);
    double (*Xt)[PwCfg<double>::LD] = Ct + PwCfg<double>::DC;
    const int tx = tid & 15, ty = tid >> 4;                     // thread: candidates 4ty.., points 4tx..
    double mupart[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < nblk; ++b) {
        double d2[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) d2[a][e] = 0.0;
        for (int d0 = 0; d0 < Dp; d0 += 16) {
            __syncthreads();
            // stage 64 candidates (scaled on the fly: X / length_scale) and 64 training points, transposed
            for (int idx = tid; idx < NB * 16; idx += 256) {
                const int r = idx >> 4, d = d0 + (idx & 15);
                const long gc = c0 + r;
                Ct[idx & 15][r] = (d < D && gc < p.M) ? p.cand[gc * D + d] / p.ls[d] : 0.0;
                Xt[idx & 15][r] = (d < Dp) ? p.Xs[(long)(b * NB + r) * Dp + d] : 0.0;
            }
            __syncthreads();
            pw_accumulate<double>(Ct, Xt, Dp - d0, d2);
        }
        tile_t Ks = b == 0 ? Ks0 : Ks1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = b * NB + 4 * tx + e;
                const double kv = (j < N) ? kernel_value<double, KIND>(d2[a][e], p.constant) : 0.0;
                Ks[4 * ty + a][4 * tx + e] = kv;
                mupart[a] = fma(kv, p.alpha[j], mupart[a]);
            }
    }
    // mean partials: the 16 lanes tx = 0..15 of a row group hold the same 4 candidates
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = mupart[a];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (tx == 0) red[128 + 4 * ty + a] = s;
    }

    // ---- q_c = || Linv Ks_c ||^2: V = Linv_blk * Ks^T on MFMA, squares summed per column ----
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double q[2] = {0.0, 0.0};                                   // this lane's two candidate columns (j = 0, 1)
    auto put_L = [&](tile_t dst, int t) {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&dst[r][c2]) = lpre[t][p8];
        }
    };
    auto add_squares = [&](const d4_t (&acc)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s = fma(acc[i][j][r], acc[i][j][r], s);
            q[j] += s;
        }
    };
    {
        tile_t Lu = reinterpret_cast<tile_t>(stage);            // the staging tile is free from here on
        d4_t acc[2][2];
        acc_zero(acc);
        __syncthreads();
        put_L(Lt, 0);
        if (nblk == 2) put_L(Lu, 1);
        __syncthreads();
        tile_mma64(Lt, Ks0, acc);                               // rows 0..63
        add_squares(acc);
        if (nblk == 2) {
            acc_zero(acc);
            tile_mma64(Lu, Ks0, acc);
            __syncthreads();
            put_L(Lt, 2);
            __syncthreads();
            tile_mma64(Lt, Ks1, acc);                           // rows 64..127
            add_squares(acc);
        }
    }
    // lanes l, l+16, l+32, l+48 share a column; then the two row-halves (wave >> 1) through LDS
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double s = q[j];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) red[(wave >> 1) * NB + (wave & 1) * 32 + 16 * j + lane] = s;
    }
    __syncthreads();

    // ---- epilogue: as finalize_kernel, one candidate per thread of the first wave ----
    double best = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    int clamped = 0;
    if (tid < NB) {
        const long gc = c0 + tid;
        if (gc < p.M) {
            const double qv = red[tid] + red[NB + tid];
            double var = p.kss - qv;
            if (var < 0.0) { var = 0.0; clamped = 1; }
            const double mu = p.y_std * red[128 + tid] + p.y_mean;
            const double sigma = sqrt(var * (p.y_std * p.y_std));
            double a = 0.0;
            if (p.acq == TGP_ACQ_UCB) {
                a = p.sf * mu + p.param * sigma;
            } else if (p.acq == TGP_ACQ_SIGMA) {
                a = sigma;
            } else if (p.acq == TGP_ACQ_PI || p.acq == TGP_ACQ_EI) {
                if (sigma != 0.0) {
                    const double diff = p.sf * (mu - p.incumbent) - p.param;
                    const double Z = diff / sigma;
                    if (p.acq == TGP_ACQ_PI) {
                        a = ndtr_small(Z);
                    } else {
                        const double pdf = exp(-(Z * Z) / 2.0) / 2.5066282746310002;
                        a = diff * ndtr_small(Z) + sigma * pdf;
                    }
                }
            }
            if (p.mu) p.mu[gc] = mu;
            if (p.sigma) p.sigma[gc] = sigma;
            if (p.acqv) p.acqv[gc] = a;
            if (p.acq != TGP_ACQ_NONE) { bi = gc; if (!isnan(a)) best = a; }
        }
        // arg-max of the 64 candidates (value, then lowest index) and the clamp count, in-wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(best, o, 64);
            const long long i2 = __shfl_xor(bi, o, 64);
            clamped += __shfl_xor(clamped, o, 64);
            if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if (tid == 0) {
            if (p.bval) { p.bval[blockIdx.x] = best; p.bidx[blockIdx.x] = bi; }
            if (clamped) atomicAdd((unsigned long long *)&p.counters[1], (unsigned long long)clamped);
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void small_sweep_kernel(SmallSweepArgs p) { small_sweep_body<KIND>(p); }

// blockIdx.y = model (tgp_predict_batch)
template <int KIND>
__global__ __launch_bounds__(256) void small_sweep_batch_kernel(const SmallSweepArgs *__restrict__ args) {
    const SmallSweepArgs p = args[blockIdx.y];
    small_sweep_body<KIND>(p);
}

// ------------------------------------------------------------------------------------------
// Fit of a model with N <= 256 in ONE workgroup ("mid" batch fit, round 4): tgp_predict_batch with stored models
// of 128 < N <= 256 -- the plot path walks the recorder's trials and predicts one grid with every trial's model
// (turbo/plotting/trials.py:371, 448, 574-577), which otherwise re-fits every stored model in turn (0.13 ms each
// through the blocked path, a dozen launches).  Here T models are T workgroups of one launch, each doing what
// small_fit_body does for two blocks, for up to four: kernel matrix tiles (sklearn kernels.py:1553-1560 /
// 1708-1738, _gpr.py:346-347) into the model's workspace, a right-looking Cholesky on 64 x 64 blocks (_gpr.py:349;
// diagonal blocks factored and inverted in LDS by chol64.hpp, panel and update products on MFMA from LDS tiles,
// the tiles themselves in the L2-resident workspace), the inverse factor block row by block row
// (Linv_ij = -X_i sum_k L_ik Linv_kj), alpha (_gpr.py:360-364) and the two LML scalars (_gpr.py:609-611).
// SmallFitArgs as for the small kernels, with Np = 256 and K = the model's (Np, Np) work matrix (mandatory).
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void mid_fit_batch_kernel(const SmallFitArgs *__restrict__ args) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int sflag;
    const SmallFitArgs p = args[blockIdx.x];
    double *scratch = sm;                                       // chol64 buffers [0, 2112), then vectors
    tile_t T0 = reinterpret_cast<tile_t>(sm + SF_SCRATCH);
    tile_t T1 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + T_SZ);
    tile_t T2 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + 2 * T_SZ);
    tile_t T3 = reinterpret_cast<tile_t>(sm + SF_SCRATCH + 3 * T_SZ);
    const int tid = threadIdx.x, tc = tid >> 4, tr = tid & 15;
    const int N = p.N, Np = p.Np, Dp = p.Dp;
    const int nblk = (N + NB - 1) / NB;                         // 1 .. 4 diagonal blocks
    const int Nin = nblk * NB;
    double *Kb = p.K;

    // ---- inputs: pinned host -> the model's workspace; Linv = 0 (the workspaces are recycled) ----
    if (tid == 0) sflag = 0;
    for (int i = tid; i < Np * Dp; i += 256) p.Xs[i] = (i < Nin * Dp) ? p.in[i] : 0.0;
    for (int i = tid; i < Np; i += 256) p.yn[i] = (i < Nin) ? p.in[Nin * Dp + i] : 0.0;
    for (int i = tid; i < p.D; i += 256) p.ls[i] = p.in[Nin * Dp + Nin + i];
    {
        const d2_t z2 = {0.0, 0.0};
        for (int i = tid; i < Np * (Np / 2); i += 256) *reinterpret_cast<d2_t *>(p.Linv + 2 * (long)i) = z2;
    }
    __syncthreads();   // Xs is read back below by this same workgroup

    // whole 64 x 64 tiles between the workspace (leading dimension Np) and LDS; 256 threads, 8 double2 each
    auto load_tile = [&](tile_t T, const double *src) {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            *reinterpret_cast<d2_t *>(&T[r][c2]) = *reinterpret_cast<const d2_t *>(src + (long)r * Np + c2);
        }
    };
    auto load_tile_t = [&](tile_t T, const double *src) {       // T[c][r] = src[r][c]
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            const d2_t v = *reinterpret_cast<const d2_t *>(src + (long)r * Np + c2);
            T[c2][r] = v[0];
            T[c2 + 1][r] = v[1];
        }
    };
    auto store_tile = [&](const tile_t T, double *dst, bool lower_only) {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            d2_t v = *reinterpret_cast<const d2_t *>(&T[r][c2]);
            if (lower_only) { v[0] = (c2 <= r) ? v[0] : 0.0; v[1] = (c2 + 1 <= r) ? v[1] : 0.0; }
            *reinterpret_cast<d2_t *>(dst + (long)r * Np + c2) = v;
        }
    };
    auto tile_zero = [&](tile_t T) {
        for (int idx = tid; idx < NB * NB; idx += 256) T[idx >> 6][idx & 63] = 0.0;
    };

    // ---- kernel matrix: the tiles on and below the diagonal into the work matrix ----
    for (int bi = 0; bi < nblk; ++bi)
        for (int bj = 0; bj <= bi; ++bj) {
            double a[4][4];
            kmat_tile_nolds<KIND>(p.Xs, bi * NB, bj * NB, N, Np, Dp, p.constant, p.noise, p.jitter, T3, a);
            double *dst = Kb + (long)(bi * NB) * Np + bj * NB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d2_t v0, v1;
                v0[0] = a[i][0]; v0[1] = a[i][1]; v1[0] = a[i][2]; v1[1] = a[i][3];
                *reinterpret_cast<d2_t *>(dst + (long)(4 * tr + i) * Np + 4 * tc) = v0;
                *reinterpret_cast<d2_t *>(dst + (long)(4 * tr + i) * Np + 4 * tc + 2) = v1;
            }
            __syncthreads();                                    // the staging tile is reused by the next tile
        }

    // ---- right-looking Cholesky on 64 x 64 blocks ----
    double sumlog = 0.0;
    for (int k = 0; k < nblk; ++k) {
        double *Kkk = Kb + (long)(k * NB) * Np + k * NB;
        __syncthreads();
        load_tile(T0, Kkk);
        tile_zero(T1);
        __syncthreads();
        factor64_v4(T0, T1, T2, scratch, k * NB, &sflag, p.tiny);      // T0 = L_kk, T1 = X_k = L_kk^-1
        if (tid < 64) sumlog += log(T0[tid][tid]);
        store_tile(T0, Kkk, true);
        store_tile(T1, p.Linv + (long)(k * NB) * Np + k * NB, false);
        // panel: L_ik = A_ik X_k^T
        for (int i = k + 1; i < nblk; ++i) {
            double *Kik = Kb + (long)(i * NB) * Np + k * NB;
            __syncthreads();                                    // T2's readers of the previous turn are done
            load_tile(T2, Kik);
            __syncthreads();
            d4_t acc[2][2];
            acc_zero(acc);
            tile_mma64(T2, T1, acc);
            acc_foreach(acc, [&](int r, int c, double v) { Kik[(long)r * Np + c] = v; });
        }
        // trailing update: A_ij -= L_ik L_jk^T, i >= j > k
        for (int i = k + 1; i < nblk; ++i)
            for (int j = k + 1; j <= i; ++j) {
                __syncthreads();                                // the panel's stores are visible; T2 / T3 free
                load_tile(T2, Kb + (long)(i * NB) * Np + k * NB);
                if (j != i) load_tile(T3, Kb + (long)(j * NB) * Np + k * NB);
                __syncthreads();
                d4_t acc[2][2];
                acc_zero(acc);
                tile_mma64(T2, j != i ? T3 : T2, acc);
                double *Kij = Kb + (long)(i * NB) * Np + j * NB;
                acc_foreach(acc, [&](int r, int c, double v) { Kij[(long)r * Np + c] -= v; });
            }
    }

    // ---- the inverse factor below the diagonal, block row by block row ----
    for (int i = 1; i < nblk; ++i) {
        __syncthreads();
        load_tile(T1, p.Linv + (long)(i * NB) * Np + i * NB);           // X_i
        for (int j = 0; j < i; ++j) {
            d4_t accs[2][2];
            acc_zero(accs);
            for (int k = j; k < i; ++k) {
                __syncthreads();
                load_tile(T2, Kb + (long)(i * NB) * Np + k * NB);         // L_ik [r][k]
                load_tile_t(T3, p.Linv + (long)(k * NB) * Np + j * NB);   // Linv_kj^T [c][k]
                __syncthreads();
                tile_mma64(T2, T3, accs);                                 // S += L_ik Linv_kj
            }
            __syncthreads();
            acc_foreach(accs, [&](int r, int c, double v) { T2[c][r] = v; });   // S^T
            __syncthreads();
            d4_t acc[2][2];
            acc_zero(acc);
            tile_mma64(T1, T2, acc);                                      // X_i S
            double *Lij = p.Linv + (long)(i * NB) * Np + j * NB;
            acc_foreach(acc, [&](int r, int c, double v) { Lij[(long)r * Np + c] = -v; });
        }
    }
    __syncthreads();

    // ---- alpha = Linv^T (Linv yn), yn . alpha; one row / column per thread, fixed order ----
    double *vz = scratch, *vyn = scratch + 256;
    vyn[tid] = (tid < Nin) ? p.in[Nin * Dp + tid] : 0.0;
    __syncthreads();
    {
        double s = 0.0;
        if (tid < Nin) {
            const double *row = p.Linv + (long)tid * Np;
            for (int c = 0; c <= tid; ++c) s = fma(row[c], vyn[c], s);
        }
        vz[tid] = s;
    }
    __syncthreads();
    double al = 0.0;
    if (tid < Nin)
        for (int r = tid; r < Nin; ++r) al = fma(p.Linv[(long)r * Np + tid], vz[r], al);
    p.alpha[tid] = al;                                                   // (Np = 256 = the workgroup)
    double *red = scratch + 512;
    red[tid] = vyn[tid] * al;
    if (tid < 64) red[256 + tid] = sumlog;
    __syncthreads();
    if (tid == 0) {
        double ya = 0.0, sl = 0.0;
        for (int i = 0; i < 256; ++i) ya += red[i];
        for (int i = 0; i < 64; ++i) sl += red[256 + i];
        p.res[0] = sl; p.res[1] = ya; p.res[2] = (double)sflag;
    }
}

// ------------------------------------------------------------------------------------------
// Sweep for 128 < N <= 512 ("mid", round 4): north_star's fused posterior kernel where the cross-kernel
// tile still fits the LDS -- the reference's larger everyday sizes and the whole plot path
// (turbo/modules/surrogates.py:332-338 -> sklearn _gpr.py:443-494; turbo/plotting/trials.py:574-577).
// One launch for any M: CPW candidates per workgroup of 8 waves (64 for N <= 256, 32 for N <= 512),
//   1. cross-kernel tile Ks[CPW candidates][N training points] in LDS (direct sum of squared differences,
//      two 64-point blocks at a time, one per half of the workgroup) with the mean K* alpha on the way;
//   2. V = Linv Ks^T on v_mfma_f64_16x16x4: wave w owns the 16-row strips w and 15 - w (and 16 + w, 31 - w
//      above N = 256) of the lower-triangular inverse factor -- k < 16 (s + 1) for strip s, so every wave has
//      the same 17 (+ 49) units of work and only the non-zero k-range is touched; A fragments straight from
//      the L2-resident Linv into registers, B fragments from the tile; V never exists, its squares are summed
//      per candidate in a fixed order;
//   3. variance, EI / PI / UCB as finalize_kernel, the workgroup's arg-max -- and the LAST workgroup to finish
//      (a ticket counter) reduces the per-workgroup winners and packs the winner record, so there is no second
//      launch.
// The blocked fit has left Linv with leading dimension Np (256 or 512) and zeros beyond N.  With 32 candidates
// per workgroup every workgroup streams the factor's triangle (1 MB at N = 512) from L2 once: right for the
// plot path's 10^2 .. 10^4 points, not for C1's 65 536 -- the general sweep keeps the large batches.
// ------------------------------------------------------------------------------------------
template <int CPW> struct MidCfg {
    static constexpr int MAXN = CPW == 64 ? 4 * NB : 8 * NB;
    static constexpr int LDK = MAXN + 2;                              // padded row of the cross-kernel tile (doubles)
    static constexpr int CT_LD = CPW + 2;                             // staged candidates, [dim][candidate]
    static constexpr int XT_LD = PwCfg<double>::LD;                   // staged training points, [dim][point]
    static constexpr int STAGE = 16 * CT_LD + 2 * 16 * XT_LD;         // Ct, Xt of the two halves
    static constexpr size_t LDS = (size_t)(CPW * LDK + STAGE) * sizeof(double);
    static_assert(STAGE >= 8 * CPW + 2 * CPW && STAGE >= 2 * 512, "the reduction scratch reuses the staging area");
};

struct MidFinal {            // what the last workgroup needs to finish the launch (argmax_final_kernel's arguments)
    double *best; double *winner; double *res_host; long long global_offset;
    Bell bell;               // a polled call (round 6): the last workgroup rings when the record is out
};

template <int KIND, int CPW>
__device__ __forceinline__ void mid_sweep_body(const SmallSweepArgs &p, const MidFinal &f) {
    using Cfg = MidCfg<CPW>;
    constexpr int LDK = Cfg::LDK, CT_LD = Cfg::CT_LD, XT_LD = Cfg::XT_LD;
    constexpr int CA = CPW / 16;                                      // candidates per thread of the cross-kernel phase
    constexpr int NCF = CPW / 16;                                     // 16-candidate column fragments of the contraction
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double (*Ks)[LDK] = reinterpret_cast<double (*)[LDK]>(sm);
    double *stage = sm + CPW * LDK;
    double (*Ct)[CT_LD] = reinterpret_cast<double (*)[CT_LD]>(stage);
    const int tid = threadIdx.x, half = tid >> 8, t256 = tid & 255;
    double (*Xt)[XT_LD] = reinterpret_cast<double (*)[XT_LD]>(stage + 16 * CT_LD) + half * 16;
    const int N = p.N, Np = p.Np, D = p.D, Dp = p.Dp;
    const int nblk = (N + NB - 1) / NB;
    const long c0 = (long)blockIdx.x * CPW;
    const int tx = t256 & 15, ty = t256 >> 4;                         // candidates CA ty.., points 4 tx.. of this half's block

    // ---- 1. cross-kernel tile and the mean ----
    double mupart[CA];
#pragma unroll
    for (int a = 0; a < CA; ++a) mupart[a] = 0.0;
    for (int b0 = 0; b0 < nblk; b0 += 2) {
        const int b = b0 + half;
        double d2[CA][4];
#pragma unroll
        for (int a = 0; a < CA; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) d2[a][e] = 0.0;
        for (int d0 = 0; d0 < Dp; d0 += 16) {
            __syncthreads();
            for (int idx = tid; idx < (CPW + 2 * NB) * 16; idx += 512) {
                const int dd = idx & 15, d = d0 + dd, r = idx >> 4;
                if (r < CPW) {
                    const long gc = c0 + r;
                    Ct[dd][r] = (d < D && gc < p.M) ? p.cand[gc * D + d] / p.ls[d] : 0.0;   // X / length_scale
                } else {
                    const int h = (r - CPW) >> 6, rr = (r - CPW) & 63, bb = b0 + h;
                    (reinterpret_cast<double (*)[XT_LD]>(stage + 16 * CT_LD) + h * 16)[dd][rr] =
                        (d < Dp && bb < nblk) ? p.Xs[(long)(bb * NB + rr) * Dp + d] : 0.0;
                }
            }
            __syncthreads();
            if (b < nblk) {
                const int dn = Dp - d0 < 16 ? Dp - d0 : 16;
                for (int d = 0; d < dn; ++d) {
                    double cv[CA], xv[4];
#pragma unroll
                    for (int a = 0; a < CA; ++a) cv[a] = Ct[d][CA * ty + a];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = Xt[d][4 * tx + e];
#pragma unroll
                    for (int a = 0; a < CA; ++a)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double df = cv[a] - xv[e];
                            d2[a][e] = fma(df, df, d2[a][e]);
                        }
                }
            }
        }
        if (b < nblk) {
#pragma unroll
            for (int a = 0; a < CA; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = b * NB + 4 * tx + e;
                    const double kv = (j < N) ? kernel_value<double, KIND>(d2[a][e], p.constant) : 0.0;
                    Ks[CA * ty + a][j] = kv;
                    mupart[a] = fma(kv, p.alpha[j], mupart[a]);
                }
        }
    }
    // mean: the 16 lanes tx of a row group hold the same candidates; then the two halves, in a fixed order
#pragma unroll
    for (int a = 0; a < CA; ++a) {
        double s = mupart[a];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mupart[a] = s;
    }
    __syncthreads();                                                  // the tile is complete, the staging area is free
    double (*qred)[CPW] = reinterpret_cast<double (*)[CPW]>(stage);   // [8 waves][CPW candidates]
    double *mred = stage + 8 * CPW;                                   // [2 halves][CPW]
    if (tx == 0) {
#pragma unroll
        for (int a = 0; a < CA; ++a) mred[half * CPW + CA * ty + a] = mupart[a];
    }

    // ---- 2. q_c = || Linv Ks_c ||^2 ----
    using MF = Mfma<double>;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
    double q[NCF];                                                    // this lane's candidate columns 16 j + (lane & 15)
#pragma unroll
    for (int j = 0; j < NCF; ++j) q[j] = 0.0;
    const int nstrips = nblk * 4;
#pragma unroll
    for (int which = 0; which < (CPW == 64 ? 2 : 4); ++which) {
        // strips w, 15 - w (and 16 + w, 31 - w): the same number of k-units for every wave
        const int s = which == 0 ? wave : (which == 1 ? 15 - wave : (which == 2 ? 16 + wave : 31 - wave));
        if (s >= nstrips) continue;                                   // rows beyond the model: Linv is zero there
        const int nsteps = 2 * (s + 1);                               // 8-wide k-steps: k < 16 (s + 1)
        const double *arow = p.Linv + (long)(16 * s + fidx) * Np + fkg;
        d4_t acc[NCF];
#pragma unroll
        for (int j = 0; j < NCF; ++j) acc[j] = (d4_t){0.0, 0.0, 0.0, 0.0};
        // A fragments of chunk c (64 k) in registers, chunks c + 1 and c + 2 in flight: the factor comes from L2 /
        // the Infinity Cache at 1-2 us per round trip, a chunk's MFMAs take ~1 us
        d2_t a_cur[8], a_nxt[8], a_nx2[8];
        auto fetch = [&](d2_t (&dst)[8], int c) {
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                dst[st] = (d2_t){0.0, 0.0};
                if (8 * c + st < nsteps) dst[st] = *reinterpret_cast<const d2_t *>(arow + 64 * c + 8 * st);
            }
        };
        fetch(a_cur, 0);
        fetch(a_nxt, 1);
        for (int c = 0; 8 * c < nsteps; ++c) {
            fetch(a_nx2, c + 2);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (8 * c + st < nsteps) {
#pragma unroll
                    for (int j = 0; j < NCF; ++j) {
                        const d2_t bv = *reinterpret_cast<const d2_t *>(&Ks[16 * j + fidx][64 * c + 8 * st + fkg]);
                        acc[j] = MF::mma(a_cur[st][0], bv[0], acc[j]);
                        acc[j] = MF::mma(a_cur[st][1], bv[1], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int st = 0; st < 8; ++st) { a_cur[st] = a_nxt[st]; a_nxt[st] = a_nx2[st]; }
        }
#pragma unroll
        for (int j = 0; j < NCF; ++j) {
            double ssq = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) ssq = fma(acc[j][r], acc[j][r], ssq);
            q[j] += ssq;
        }
    }
    // lanes l, l + 16, l + 32, l + 48 hold four rows each of the same column; then the eight waves through LDS
#pragma unroll
    for (int j = 0; j < NCF; ++j) {
        double ssq = q[j];
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) qred[wave][16 * j + lane] = ssq;
    }
    __syncthreads();

    // ---- 3. epilogue: as finalize_kernel / small_sweep_kernel, one candidate per thread of the first wave ----
    __shared__ int is_last;
    if (tid < 64) {
        double best = -INFINITY;
        long long bi = 0x7fffffffffffffffLL;
        int clamped = 0;
        const long gc = c0 + tid;
        if (tid < CPW && gc < p.M) {
            double qv = qred[0][tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) qv += qred[w][tid];
            double var = p.kss - qv;
            if (var < 0.0) { var = 0.0; clamped = 1; }
            const double mu = p.y_std * (mred[tid] + mred[CPW + tid]) + p.y_mean;
            const double sigma = sqrt(var * (p.y_std * p.y_std));
            double a = 0.0;
            if (p.acq == TGP_ACQ_UCB) {
                a = p.sf * mu + p.param * sigma;
            } else if (p.acq == TGP_ACQ_SIGMA) {
                a = sigma;
            } else if (p.acq == TGP_ACQ_PI || p.acq == TGP_ACQ_EI) {
                if (sigma != 0.0) {
                    const double diff = p.sf * (mu - p.incumbent) - p.param;
                    const double Z = diff / sigma;
                    if (p.acq == TGP_ACQ_PI) {
                        a = ndtr_small(Z);
                    } else {
                        const double pdf = exp(-(Z * Z) / 2.0) / 2.5066282746310002;
                        a = diff * ndtr_small(Z) + sigma * pdf;
                    }
                }
            }
            if (p.mu) p.mu[gc] = mu;
            if (p.sigma) p.sigma[gc] = sigma;
            if (p.acqv) p.acqv[gc] = a;
            if (p.acq != TGP_ACQ_NONE) { bi = gc; if (!isnan(a)) best = a; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(best, o, 64);
            const long long i2 = __shfl_xor(bi, o, 64);
            clamped += __shfl_xor(clamped, o, 64);
            if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if (tid == 0) {
            if (clamped) atomicAdd((unsigned long long *)&p.counters[1], (unsigned long long)clamped);
            is_last = 0;
            if (p.bval) {      // (a batched predict has no arg-max and nothing for a last workgroup to do)
                p.bval[blockIdx.x] = best;
                p.bidx[blockIdx.x] = bi;
                // the ticket: release my partials, and whoever draws the last one sees everybody's.  (Every store of
                // this epilogue -- means, deviations, acquisition values -- came from THIS wave, so the fence's wait
                // covers them; a polled call's outputs are mapped host memory: system scope then)
                if (f.bell.word) __threadfence_system(); else __threadfence();
                const unsigned long long ticket = atomicAdd((unsigned long long *)&p.counters[2], 1ull);
                is_last = ticket == (unsigned long long)gridDim.x - 1ull;
            }
        }
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // ---- the last workgroup: (value, lowest index) over all workgroups, exactly argmax_final_kernel ----
    double *sv = stage;                                               // [512]
    long long *si = reinterpret_cast<long long *>(stage + 512);       // [512]
    double v = -INFINITY;
    long long i = 0x7fffffffffffffffLL;
    if (p.acq != TGP_ACQ_NONE) {
        for (long b = tid; b < (long)gridDim.x; b += 512) {
            const double v2 = __hip_atomic_load(p.bval + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long i2 = __hip_atomic_load(p.bidx + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
        }
    }
    sv[tid] = v;
    si[tid] = i;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) {
            const double v2 = sv[tid + o];
            const long long i2 = si[tid + o];
            if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        f.best[0] = sv[0];
        p.counters[0] = si[0];
        p.counters[2] = 0;                                            // the ticket counter is handed back at zero
        if (f.res_host) {                                             // zero-copy result record: [best value, best index, clamp count]
            const long long nc = __hip_atomic_load(&p.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f.res_host[0] = sv[0];
            f.res_host[1] = (double)((si[0] >= p.M) ? 0 : si[0]);
            f.res_host[2] = (double)nc;
            p.counters[1] = 0;
        }
    }
    if (f.winner && p.acq != TGP_ACQ_NONE) {
        const long long wi = (si[0] >= p.M) ? 0 : si[0];
        if (tid == 0) { f.winner[0] = sv[0]; f.winner[1] = (double)(f.global_offset + wi); }
        for (int d = tid; d < D; d += 512) f.winner[2 + d] = p.cand[wi * D + d];
    }
    bell_ring(f.bell, 1);     // (the last workgroup only: everybody else has left)
}

template <int KIND, int CPW>
__global__ __launch_bounds__(512) void mid_sweep_kernel(SmallSweepArgs p, MidFinal f) { mid_sweep_body<KIND, CPW>(p, f); }

// blockIdx.y = model (tgp_predict_batch with stored models of 128 < N <= 256): predict only
template <int KIND>
__global__ __launch_bounds__(512) void mid_sweep_batch_kernel(const SmallSweepArgs *__restrict__ args) {
    const SmallSweepArgs p = args[blockIdx.y];
    mid_sweep_body<KIND, 64>(p, MidFinal{nullptr, nullptr, nullptr, 0, Bell{nullptr, 0, nullptr}});
}

// candidates per workgroup of the one-launch sweep that serves this model and batch, or 0 (the general sweep)
int mid_sweep_cpw(const Context &c, int64_t M) {
    const bool off = tuning().mid == 0;       // A/B: the general sweep instead
    // Above N = 256 (32 candidates per workgroup, every workgroup streaming the factor's triangle from L2, one
    // workgroup per CU) the kernel only pays for batches of up to ~16 k candidates.  Device ms of an EI sweep, this
    // kernel against the general sweep (tools/gpu/r4_mid_device_ms.sh):
    //   N = 300: M = 4096 0.047 / 0.108, 16 384 0.095 / 0.131, 32 768 0.180 / 0.173
    //   N = 500: M = 4096 0.082 / 0.129, 16 384 0.162 / 0.173, 32 768 0.310 / 0.251;  C1 (M = 65 536) 0.57 / 0.43
    // A choice that depends on M would make a candidate's value depend (in the last bits) on the size of the batch it
    // travels in -- and with it the winner of a sharded sweep on the number of GPUs.  So the default above N = 256 is
    // the general sweep for every M, and this kernel is OPT-IN for those sizes: TGP_MID_MAXM = the largest batch that
    // takes it (read at every call, so a plotting script can switch it on around its predict loop).
    const long maxm = tuning_mid_maxm_now();
    if (off || !c.fitted || c.small || c.N <= 2 * NB) return 0;
    // (the kernel is f64 whatever the handle's dtype -- like the N <= 128 kernels: FunctionInstance.sweep_dtype says so --
    // and wants ~155 KB of LDS per workgroup: a device that cannot grant that takes the general sweep)
    static std::atomic<int> lds_ok[64];                       // 0 unknown, 1 yes, -1 no
    int ok = lds_ok[c.device & 63].load(std::memory_order_relaxed);
    if (ok == 0) {
        int a = 0, b = 0;
        if (hipDeviceGetAttribute(&a, hipDeviceAttributeSharedMemPerBlockOptin, c.device) != hipSuccess) { (void)hipGetLastError(); a = 0; }
        if (hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerBlock, c.device) != hipSuccess) { (void)hipGetLastError(); b = 0; }
        const size_t cap = (size_t)(a > b ? a : b);           // (an attribute the runtime does not report leaves the path on)
        ok = (cap == 0 || (cap >= MidCfg<64>::LDS && cap >= MidCfg<32>::LDS)) ? 1 : -1;
        lds_ok[c.device & 63].store(ok, std::memory_order_relaxed);
    }
    if (ok < 0) return 0;
    if (c.N <= 4 * NB && c.Np == 4 * NB) return 64;
    if (c.N <= 8 * NB && c.Np == 8 * NB && M <= maxm) return 32;
    return 0;
}

template <int CPW>
static hipError_t launch_mid_sweep_as(Context &c, const SmallSweepArgs &a, const MidFinal &f) {
    void (*k)(SmallSweepArgs, MidFinal);
    switch (c.kernel) {
        case TGP_RBF: k = mid_sweep_kernel<TGP_RBF, CPW>; break;
        case TGP_MATERN12: k = mid_sweep_kernel<TGP_MATERN12, CPW>; break;
        case TGP_MATERN32: k = mid_sweep_kernel<TGP_MATERN32, CPW>; break;
        default: k = mid_sweep_kernel<TGP_MATERN52, CPW>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, MidCfg<CPW>::LDS));
    const unsigned nblk = (unsigned)((c.M + CPW - 1) / CPW);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(512), MidCfg<CPW>::LDS, c.stream, a, f);
    return hipGetLastError();
}

hipError_t launch_mid_sweep(Context &c, const double *cand, int acq, double sf, double incumbent,
                            double param, double *mu, double *sigma, double *acqv, double *res_host, const Bell &bell) {
    SmallSweepArgs a{};
    a.cand = cand; a.ls = c.d_ls; a.Xs = c.d_Xs; a.Linv = c.d_Linv; a.alpha = c.d_alpha;
    a.mu = mu; a.sigma = sigma; a.acqv = acqv;
    a.bval = c.d_bval; a.bidx = c.d_bidx; a.counters = c.d_besti;
    a.M = (long)c.M; a.N = (int)c.N; a.D = (int)c.D; a.Dp = (int)c.Dp; a.Np = (int)c.Np;
    a.constant = c.constant; a.kss = c.constant + c.noise; a.y_mean = c.y_mean; a.y_std = c.y_std;
    a.acq = acq; a.sf = sf; a.incumbent = incumbent; a.param = param;
    const MidFinal f{c.d_best, c.d_winner, res_host, (long long)c.winner_offset, bell};
    return mid_sweep_cpw(c, c.M) == 64 ? launch_mid_sweep_as<64>(c, a, f) : launch_mid_sweep_as<32>(c, a, f);
}

hipError_t launch_small_sweep(Context &c, const double *cand, int acq, double sf, double incumbent,
                              double param, double *mu, double *sigma, double *acqv) {
    SmallSweepArgs a{};
    a.cand = cand; a.ls = c.d_ls; a.Xs = c.d_Xs; a.Linv = c.d_Linv; a.alpha = c.d_alpha;
    a.mu = mu; a.sigma = sigma; a.acqv = acqv;
    a.bval = c.d_bval; a.bidx = c.d_bidx; a.counters = c.d_besti;
    a.M = (long)c.M; a.N = (int)c.N; a.D = (int)c.D; a.Dp = (int)c.Dp; a.Np = (int)c.Np;
    a.constant = c.constant; a.kss = c.constant + c.noise; a.y_mean = c.y_mean; a.y_std = c.y_std;
    a.acq = acq; a.sf = sf; a.incumbent = incumbent; a.param = param;
    void (*k)(SmallSweepArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_sweep_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_sweep_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_sweep_kernel<TGP_MATERN32>; break;
        default: k = small_sweep_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, SMALL_SWEEP_LDS));
    const unsigned nblk = (unsigned)((c.M + NB - 1) / NB);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), SMALL_SWEEP_LDS, c.stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Many stored models x one batch of points (the plot path: turbo/plotting/trials.py:574-577 walks
// the recorder's trials and predicts a grid with each trial's model).  T small fits as ONE launch
// of T workgroups, T sweeps as ONE launch of (M / 64, T) workgroups.
// ------------------------------------------------------------------------------------------
size_t small_fit_args_bytes() { return sizeof(SmallFitArgs); }
size_t small_sweep_args_bytes() { return sizeof(SmallSweepArgs); }

void fill_small_batch_args(void *fit_args, void *sweep_args, int64_t t, const double *in_dev, double *ws_dev,
                           double *res_dev, long long *counters_dev, const double *cand_dev, double *mu_dev,
                           double *sigma_dev, int64_t N, int64_t D, int64_t Dp, int64_t M, double constant,
                           double noise, double jitter, double y_mean, double y_std) {
    constexpr int NPB = 2 * NB;                         // leading dimension of a batch model
    SmallFitArgs &f = reinterpret_cast<SmallFitArgs *>(fit_args)[t];
    double *Xs = ws_dev, *yn = Xs + NPB * Dp, *ls = yn + NPB, *Linv = ls + Dp, *alpha = Linv + NPB * NPB;   // Dp keeps Linv 16-byte aligned
    f = SmallFitArgs{};
    f.legacy = tuning().small_live == 0 ? 1 : 0;
    f.in = in_dev; f.Xs = Xs; f.yn = yn; f.ls = ls; f.K = nullptr; f.Linv = Linv; f.alpha = alpha;
    f.Xs32 = nullptr; f.Linv32 = nullptr; f.res = res_dev;
    f.N = (int)N; f.D = (int)D; f.Dp = (int)Dp; f.Np = NPB;
    f.zero_to = NPB;                                    // batch workspaces are recycled: clear every time
    f.constant = constant; f.noise = noise; f.jitter = jitter;
    f.tiny = 8.0 * 2.220446049250313e-16 * ((constant + noise) + jitter);
    SmallSweepArgs &w = reinterpret_cast<SmallSweepArgs *>(sweep_args)[t];
    w = SmallSweepArgs{};
    w.cand = cand_dev; w.ls = ls; w.Xs = Xs; w.Linv = Linv; w.alpha = alpha;
    w.mu = mu_dev; w.sigma = sigma_dev; w.acqv = nullptr;
    w.bval = nullptr; w.bidx = nullptr; w.counters = counters_dev;
    w.M = (long)M; w.N = (int)N; w.D = (int)D; w.Dp = (int)Dp; w.Np = NPB;
    w.constant = constant; w.kss = constant + noise; w.y_mean = y_mean; w.y_std = y_std;
    w.acq = TGP_ACQ_NONE; w.sf = 1.0; w.incumbent = 0.0; w.param = 0.0;
}

int64_t small_batch_ws_doubles(int64_t D, int64_t Dp) { (void)D; return 2 * NB * Dp + 2 * NB + Dp + 4 * NB * NB + 2 * NB; }

// models of 128 < N <= 256: [Xs (256, Dp) | yn 256 | ls Dp | K (256, 256) | Linv (256, 256) | alpha 256]
int64_t mid_batch_ws_doubles(int64_t D, int64_t Dp) { (void)D; return 4 * NB * Dp + 4 * NB + Dp + 2 * 16 * NB * NB + 4 * NB; }

void fill_mid_batch_args(void *fit_args, void *sweep_args, int64_t t, const double *in_dev, double *ws_dev,
                         double *res_dev, long long *counters_dev, const double *cand_dev, double *mu_dev,
                         double *sigma_dev, int64_t N, int64_t D, int64_t Dp, int64_t M, double constant,
                         double noise, double jitter, double y_mean, double y_std) {
    constexpr int NPB = 4 * NB;
    SmallFitArgs &f = reinterpret_cast<SmallFitArgs *>(fit_args)[t];
    double *Xs = ws_dev, *yn = Xs + NPB * Dp, *ls = yn + NPB, *K = ls + Dp, *Linv = K + NPB * NPB, *alpha = Linv + NPB * NPB;
    f = SmallFitArgs{};
    f.legacy = tuning().small_live == 0 ? 1 : 0;
    f.in = in_dev; f.Xs = Xs; f.yn = yn; f.ls = ls; f.K = K; f.Linv = Linv; f.alpha = alpha;
    f.Xs32 = nullptr; f.Linv32 = nullptr; f.res = res_dev;
    f.N = (int)N; f.D = (int)D; f.Dp = (int)Dp; f.Np = NPB;
    f.zero_to = NPB;
    f.constant = constant; f.noise = noise; f.jitter = jitter;
    f.tiny = 8.0 * 2.220446049250313e-16 * ((constant + noise) + jitter);
    SmallSweepArgs &w = reinterpret_cast<SmallSweepArgs *>(sweep_args)[t];
    w = SmallSweepArgs{};
    w.cand = cand_dev; w.ls = ls; w.Xs = Xs; w.Linv = Linv; w.alpha = alpha;
    w.mu = mu_dev; w.sigma = sigma_dev; w.acqv = nullptr;
    w.bval = nullptr; w.bidx = nullptr; w.counters = counters_dev;
    w.M = (long)M; w.N = (int)N; w.D = (int)D; w.Dp = (int)Dp; w.Np = NPB;
    w.constant = constant; w.kss = constant + noise; w.y_mean = y_mean; w.y_std = y_std;
    w.acq = TGP_ACQ_NONE; w.sf = 1.0; w.incumbent = 0.0; w.param = 0.0;
}

hipError_t launch_mid_batch(Context &c, int kernel, const void *fit_args_dev, const void *sweep_args_dev, int64_t T, int64_t M) {
    void (*kf)(const SmallFitArgs *);
    void (*ks)(const SmallSweepArgs *);
    switch (kernel) {
        case TGP_RBF: kf = mid_fit_batch_kernel<TGP_RBF>; ks = mid_sweep_batch_kernel<TGP_RBF>; break;
        case TGP_MATERN12: kf = mid_fit_batch_kernel<TGP_MATERN12>; ks = mid_sweep_batch_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: kf = mid_fit_batch_kernel<TGP_MATERN32>; ks = mid_sweep_batch_kernel<TGP_MATERN32>; break;
        default: kf = mid_fit_batch_kernel<TGP_MATERN52>; ks = mid_sweep_batch_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_f[4], opt_s[4];
    TGP_TRY(opt_f[kernel & 3].ensure(reinterpret_cast<const void *>(kf), c.device, SMALL_FIT_LDS));
    hipLaunchKernelGGL(kf, dim3((unsigned)T), dim3(256), SMALL_FIT_LDS, c.stream, reinterpret_cast<const SmallFitArgs *>(fit_args_dev));
    TGP_TRY(hipGetLastError());
    TGP_TRY(opt_s[kernel & 3].ensure(reinterpret_cast<const void *>(ks), c.device, MidCfg<64>::LDS));
    hipLaunchKernelGGL(ks, dim3((unsigned)((M + NB - 1) / NB), (unsigned)T), dim3(512), MidCfg<64>::LDS, c.stream,
                       reinterpret_cast<const SmallSweepArgs *>(sweep_args_dev));
    return hipGetLastError();
}

hipError_t launch_small_batch(Context &c, int kernel, const void *fit_args_dev, const void *sweep_args_dev,
                              int64_t T, int64_t M, bool fit, bool sweep) {
    void (*kf)(const SmallFitArgs *);
    void (*ks)(const SmallSweepArgs *);
    switch (kernel) {
        case TGP_RBF: kf = small_fit_batch_kernel<TGP_RBF>; ks = small_sweep_batch_kernel<TGP_RBF>; break;
        case TGP_MATERN12: kf = small_fit_batch_kernel<TGP_MATERN12>; ks = small_sweep_batch_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: kf = small_fit_batch_kernel<TGP_MATERN32>; ks = small_sweep_batch_kernel<TGP_MATERN32>; break;
        default: kf = small_fit_batch_kernel<TGP_MATERN52>; ks = small_sweep_batch_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_f[4], opt_s[4];
    if (fit) {
        TGP_TRY(opt_f[kernel & 3].ensure(reinterpret_cast<const void *>(kf), c.device, SMALL_FIT_LDS));
        hipLaunchKernelGGL(kf, dim3((unsigned)T), dim3(256), SMALL_FIT_LDS, c.stream,
                           reinterpret_cast<const SmallFitArgs *>(fit_args_dev));
        TGP_TRY(hipGetLastError());
    }
    if (sweep) {
        TGP_TRY(opt_s[kernel & 3].ensure(reinterpret_cast<const void *>(ks), c.device, SMALL_SWEEP_LDS));
        hipLaunchKernelGGL(ks, dim3((unsigned)((M + NB - 1) / NB), (unsigned)T), dim3(256), SMALL_SWEEP_LDS, c.stream,
                           reinterpret_cast<const SmallSweepArgs *>(sweep_args_dev));
        TGP_TRY(hipGetLastError());
    }
    return hipSuccess;
}

}  // namespace tgp
