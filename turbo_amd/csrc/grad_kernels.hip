// grad_kernels.hip -- gradient of the log marginal likelihood w.r.t. the log hyper-parameters,
// on top of a finished fit (L, Linv, U = Linv^T, alpha resident).  "Next" row SURVEY 8(f)1.
//
//   dLML/dtheta_p = 0.5 * trace((alpha alpha^T - K^-1) dK/dtheta_p)      sklearn _gpr.py:614-650
//   K^-1 = Linv^T Linv = U U^T            (replaces cho_solve(L, I), _gpr.py:624-626; NT MFMA GEMM)
//   dK/dlog c = c*k0, dK/dlog noise = noise*I, dK/dlog l_d = c*g(r)*D_d  (kernels.py Product :968-975,
//   ConstantKernel :1280-1290, WhiteKernel :1403-1410, RBF :1566-1580, Matern :1740-1779)
//
// The pairwise pass produces the scalar sums (constant, isotropic length scale, noise) directly
// from the directly-summed distances.  For ARD length scales it also materialises
// Wt = (alpha alpha^T - K^-1) o g and uses
//   sum_ij Wt_ij (x_id - x_jd)^2 = 2 * sum_i x_id (x_id r_i - (Wt X)_id),   r = Wt 1.
#include <hip/hip_runtime.h>
#include <math.h>

#include "chol64.hpp"
#include "gemm64_glds.hpp"
#include "gemm_nt_glds.hpp"
#include "lds_opt_in.hpp"
#include "mfma_gemm.hpp"
#include "pairwise.hpp"
#include "small_grad.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// lower-triangular 64x64 tiles over the real points.  partial[tile] = [S_c, S_iso, S_diag] and, for
// ARD length scales, gd[0..Dp): the tile's share of 1/2 sum_ij w_ij (x_id - x_jd)^2, taken directly
// in a second pass over the dimensions with the weights w = (alpha alpha^T - K^-1) o g still in
// registers.  (The weights used to be written out, N x N, and contracted with [X | 1] by a skinny
// GEMM: 0.36 ms of a 3.4 ms evaluation at N = 4096, mostly the 256 MiB of writes and a 64-workgroup
// GEMM; this way nothing N x N leaves the kernel.)
template <int KIND>
__global__ __launch_bounds__(256) void lml_weights_kernel(const double *__restrict__ Xs,
                                                          const double *__restrict__ alpha,
                                                          const double *__restrict__ Kinv,
                                                          double *__restrict__ partial, int N,
                                                          int Np, int Dp, int ard) {
    __shared__ double Ct[PwCfg<double>::DC][PwCfg<double>::LD];
    __shared__ double Xt[PwCfg<double>::DC][PwCfg<double>::LD];
    __shared__ double red[3][256];
    int bx = blockIdx.x;
    int tm = (int)((sqrtf(8.0f * (float)bx + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= bx) ++tm;
    while (tm * (tm + 1) / 2 > bx) --tm;
    const int tn = bx - tm * (tm + 1) / 2;
    const int i0 = tm * PW_T, j0 = tn * PW_T;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int ps = ard ? 3 + Dp : 3;                 // doubles per tile in partial
    double d2[4][4];
    pairwise_sqdist<double>(Xs, i0, Np, Xs, j0, Np, Dp, Ct, Xt, d2);
    const double mult = (tm == tn) ? 1.0 : 2.0;
    double sc = 0.0, siso = 0.0, sdiag = 0.0;
    double w[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + 4 * ty + a;
        const double ai = (i < N) ? alpha[i] : 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + 4 * tx + b;
            w[a][b] = 0.0;
            if (i < N && j < N) {
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                const double G = ai * alpha[j] - Kinv[(long)hi * Np + lo];
                if (i == j) {
                    sc += G;          // k0_ii = 1 (np.fill_diagonal(K, 1))
                    sdiag += G;
                } else {
                    const double k0 = kernel_value<double, KIND>(d2[a][b], 1.0);
                    w[a][b] = G * ls_weight<KIND>(d2[a][b]);
                    sc = fma(G, k0, sc);
                    siso = fma(w[a][b], d2[a][b], siso);
                }
            }
        }
    }
    red[0][tid] = sc * mult;
    red[1][tid] = siso * mult;
    red[2][tid] = sdiag;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
            red[2][tid] += red[2][tid + o];
        }
        __syncthreads();
    }
    if (tid < 3) partial[(long)blockIdx.x * ps + tid] = red[tid][0];
    if (!ard) return;
    // ---- ARD: sixteen dimensions per pass, per-dimension sums over the tile (fixed order) ----
    const int lane = tid & 63, wave = tid >> 6;
    PwStage<double> sp, sq;
    for (int d0 = 0; d0 < Dp; d0 += 16) {
        sp.load(Xs, i0, Np, Dp, d0);
        sq.load(Xs, j0, Np, Dp, d0);
        __syncthreads();
        sp.store(Ct);
        sq.store(Xt);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            double cv[4], xv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) cv[a] = Ct[e][4 * ty + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) xv[b] = Xt[e][4 * tx + b];
            double q = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double df = cv[a] - xv[b];
                    q = fma(w[a][b], df * df, q);
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            if (lane == 0) red[0][wave * 16 + e] = q;
        }
        __syncthreads();
        if (tid < 16 && d0 + tid < Dp)
            partial[(long)blockIdx.x * ps + 3 + d0 + tid] =
                (0.5 * mult) * ((red[0][tid] + red[0][16 + tid]) + (red[0][32 + tid] + red[0][48 + tid]));
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void small_grad_kernel(SmallGradArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    small_grad_body<KIND>(p, (int)blockIdx.x, sm);   // grid = 1 (one block) or 3 (the pairs of two blocks)
}

hipError_t launch_small_grad(Context &c, bool ard, double *out) {
    SmallGradArgs a{};
    a.Xs = c.d_Xs; a.alpha = c.d_alpha; a.Linv = c.d_Linv; a.out = out;
    a.N = (int)c.N; a.Np = (int)c.Np; a.Dp = (int)c.Dp; a.ard = ard ? 1 : 0;
    void (*k)(SmallGradArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_grad_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_grad_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_grad_kernel<TGP_MATERN32>; break;
        default: k = small_grad_kernel<TGP_MATERN52>; break;
    }
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, SMALL_GRAD_LDS));
    hipLaunchKernelGGL(k, dim3(c.N > NB ? 3 : 1), dim3(256), SMALL_GRAD_LDS, c.stream, a);
    return hipGetLastError();
}

// out[c] = sum over the tiles of partial[tile][c], c = blockIdx.x < ncols (fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partial,
                                                           int nblk, int ncols, double *__restrict__ out) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(long)b * ncols + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// results land in gout (device memory or device-mapped host memory): [S_c, S_iso, S_diag, gd[0..Dp)]
// timed = false (a polled call, round 6): no event records between the stages (tgp_last_timings then has no stage times)
hipError_t launch_lml_grad(Context &c, bool ard, double *gout, bool timed) {
    hipStream_t s = c.stream;
    const int N = (int)c.N, Np = (int)c.Np, Dp = (int)c.Dp;
    if (timed) {
        for (int i = 0; i < 4; ++i)
            if (!c.evg[i]) TGP_TRY(hipEventCreate(&c.evg[i]));
        TGP_TRY(hipEventRecord(c.evg[0], s));
    }
    const int kinv64 = tuning().kinv64;   // Np up to which the 64-tile template is used (0.44 vs 0.65 ms at N = 4096, 0.031 vs 0.071 at 512: the 128-tile grid is short and very unequal)
    if (Np <= kinv64) {   // K^-1 = U U^T, lower 64-tiles, into W
        GemmArgs g{};
        g.A = c.d_U; g.lda = Np;
        g.B = c.d_U; g.ldb = Np;
        g.C = c.d_W; g.ldc = Np;
        g.ntm = g.ntn = Np / 64; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
        const int nt = Np / 64;
        TGP_TRY((launch_gemm64_glds<KR_UPPER_A, TM_LOWER>(s, c.device, g, nt * (nt + 1) / 2, 1)));
    } else {   // K^-1 = U U^T, lower 128-tiles, into W
        GemmNtArgs g{};
        g.A = c.d_U; g.lda = Np;
        g.B = c.d_U; g.ldb = Np;
        g.C = c.d_W; g.ldc = Np;
        g.Ct = nullptr;
        g.ntm = g.ntn = Np / 128; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
        const int nt = Np / 128;
        TGP_TRY((launch_gemm_nt_glds<double, KN_UPPER_A, TM_LOWER>(s, c.device, g, nt * (nt + 1) / 2, 1)));
    }
    if (timed) TGP_TRY(hipEventRecord(c.evg[1], s));
    const int nt = (N + PW_T - 1) / PW_T;
    const int nblk = nt * (nt + 1) / 2;
    const dim3 grid(nblk);
    const int wr = ard ? 1 : 0;
    switch (c.kernel) {
        case TGP_RBF: hipLaunchKernelGGL(lml_weights_kernel<TGP_RBF>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_gpart, N, Np, Dp, wr); break;
        case TGP_MATERN12: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN12>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_gpart, N, Np, Dp, wr); break;
        case TGP_MATERN32: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN32>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_gpart, N, Np, Dp, wr); break;
        default: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN52>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_gpart, N, Np, Dp, wr); break;
    }
    TGP_TRY(hipGetLastError());
    if (timed) TGP_TRY(hipEventRecord(c.evg[2], s));
    const int ncols = ard ? 3 + Dp : 3;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ncols), dim3(256), 0, s, c.d_gpart, nblk, ncols, gout);
    TGP_TRY(hipGetLastError());
    if (timed) TGP_TRY(hipEventRecord(c.evg[3], s));
    return hipSuccess;
}

}  // namespace tgp
