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

#include "gemm_nt_glds.hpp"
#include "mfma_gemm.hpp"
#include "pairwise.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// unit-amplitude length-scale weight g(r): dk0/dlog l_d = g * D_d
template <int KIND>
__device__ __forceinline__ double ls_weight(double d2) {
    if (KIND == TGP_RBF) {
        return exp(-0.5 * d2);
    } else if (KIND == TGP_MATERN12) {
        const double r = sqrt(d2);
        return r > 0.0 ? exp(-r) / r : 0.0;
    } else if (KIND == TGP_MATERN32) {
        return 3.0 * exp(-sqrt(3.0 * d2));
    } else {
        const double t = sqrt(5.0 * d2);
        return 5.0 / 3.0 * (t + 1.0) * exp(-t);
    }
}

// lower-triangular 64x64 tiles over the real points
template <int KIND>
__global__ __launch_bounds__(256) void lml_weights_kernel(const double *__restrict__ Xs,
                                                          const double *__restrict__ alpha,
                                                          const double *__restrict__ Kinv,
                                                          double *__restrict__ Wt,
                                                          double *__restrict__ partial, int N,
                                                          int Np, int Dp, int write_wt) {
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
    double d2[4][4];
    pairwise_sqdist<double>(Xs, i0, Np, Xs, j0, Np, Dp, Ct, Xt, d2);
    const double mult = (tm == tn) ? 1.0 : 2.0;
    double sc = 0.0, siso = 0.0, sdiag = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + 4 * ty + a;
        const double ai = (i < N) ? alpha[i] : 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + 4 * tx + b;
            double w = 0.0;
            if (i < N && j < N) {
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                const double G = ai * alpha[j] - Kinv[(long)hi * Np + lo];
                if (i == j) {
                    sc += G;          // k0_ii = 1 (np.fill_diagonal(K, 1))
                    sdiag += G;
                } else {
                    const double k0 = kernel_value<double, KIND>(d2[a][b], 1.0);
                    w = G * ls_weight<KIND>(d2[a][b]);
                    sc = fma(G, k0, sc);
                    siso = fma(w, d2[a][b], siso);
                }
            }
            if (write_wt) {
                Wt[(long)i * Np + j] = w;
                Wt[(long)j * Np + i] = w;
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
    if (tid < 3) partial[(long)blockIdx.x * 3 + tid] = red[tid][0];
}

// out[0..2] = column sums of partial (nblk, 3), fixed order, single block
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partial,
                                                           int nblk, double *__restrict__ out) {
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int k = 0; k < 3; ++k) s[k] += partial[(long)b * 3 + k];
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) out[threadIdx.x] = red[threadIdx.x][0];
}

// ARD products on MFMA: Z = Wt * [Xs | 1 | 0], i.e. Z[i][d] = sum_j Wt[i][j] xs[j][d] for d < Dp and
// Z[i][Dp] = sum_j Wt[i][j].  The right-hand side is packed once into (K, Zc) with Zc a multiple of
// 64 so the product runs on the 64 x 64 MFMA template (the naive loop it replaces took 1.25 of the
// 5.8 ms of an evaluation at N = 4096).
__global__ __launch_bounds__(256) void ard_pack_kernel(const double *__restrict__ Xs, double *__restrict__ Xp,
                                                       int N, int K, int Dp, int Zc) {
    const long total = (long)K * Zc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i / Zc), cidx = (int)(i - (long)j * Zc);
        double v = 0.0;
        if (j < N) v = cidx < Dp ? Xs[(long)j * Dp + cidx] : (cidx == Dp ? 1.0 : 0.0);
        Xp[i] = v;
    }
}

// gd[d] = sum_i xs[i][d] * (xs[i][d] * Z[i][Dp] - Z[i][d])   (one block per dimension)
__global__ __launch_bounds__(256) void ard_reduce_kernel(const double *__restrict__ Xs,
                                                         const double *__restrict__ Z,
                                                         double *__restrict__ gd, int N, int Dp, int Zc) {
    __shared__ double red[256];
    const int d = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double x = Xs[(long)i * Dp + d];
        s = fma(x, fma(x, Z[(long)i * Zc + Dp], -Z[(long)i * Zc + d]), s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) gd[d] = red[0];
}

// results land in c.d_gout: [S_c, S_iso, S_diag, gd[0..Dp)]
hipError_t launch_lml_grad(Context &c, bool ard) {
    hipStream_t s = c.stream;
    const int N = (int)c.N, Np = (int)c.Np, Dp = (int)c.Dp;
    for (int i = 0; i < 4; ++i)
        if (!c.evg[i]) TGP_TRY(hipEventCreate(&c.evg[i]));
    TGP_TRY(hipEventRecord(c.evg[0], s));
    {   // K^-1 = U U^T, lower 128-tiles, into W
        GemmNtArgs g{};
        g.A = c.d_U; g.lda = Np;
        g.B = c.d_U; g.ldb = Np;
        g.C = c.d_W; g.ldc = Np;
        g.Ct = nullptr;
        g.ntm = g.ntn = Np / 128; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
        const int nt = Np / 128;
        TGP_TRY((launch_gemm_nt_glds<double, KN_UPPER_A, TM_LOWER>(s, c.device, g, nt * (nt + 1) / 2, 1)));
    }
    TGP_TRY(hipEventRecord(c.evg[1], s));
    const int nt = (N + PW_T - 1) / PW_T;
    const int nblk = nt * (nt + 1) / 2;
    const dim3 grid(nblk);
    const int wr = ard ? 1 : 0;
    switch (c.kernel) {
        case TGP_RBF: hipLaunchKernelGGL(lml_weights_kernel<TGP_RBF>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_U, c.d_gpart, N, Np, Dp, wr); break;
        case TGP_MATERN12: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN12>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_U, c.d_gpart, N, Np, Dp, wr); break;
        case TGP_MATERN32: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN32>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_U, c.d_gpart, N, Np, Dp, wr); break;
        default: hipLaunchKernelGGL(lml_weights_kernel<TGP_MATERN52>, grid, dim3(256), 0, s, c.d_Xs, c.d_alpha, c.d_W, c.d_U, c.d_gpart, N, Np, Dp, wr); break;
    }
    TGP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, c.d_gpart, nblk, c.d_gout);
    TGP_TRY(hipGetLastError());
    TGP_TRY(hipEventRecord(c.evg[2], s));
    if (ard) {
        const int K = nt * PW_T;                                   // rows / columns of Wt that were written
        const int Zc = ((Dp + 1 + 63) / 64) * 64;
        double *Xp = c.d_Z, *Z = c.d_Z + (long)Np * Zc;
        const long pe = (long)K * Zc;
        hipLaunchKernelGGL(ard_pack_kernel, dim3((unsigned)((pe + 255) / 256 < 4096 ? (pe + 255) / 256 : 4096)), dim3(256), 0, s,
                           c.d_Xs, Xp, N, K, Dp, Zc);
        TGP_TRY(hipGetLastError());
        GemmArgs g{};
        g.A = c.d_U; g.lda = Np;
        g.B = Xp; g.ldb = Zc;
        g.C = Z; g.ldc = Zc;
        g.ntm = nt; g.ntn = Zc / 64; g.K = K; g.alpha = 1.0; g.beta = 0.0;
        {
            constexpr int BK = 16;
            auto kern = mfma_gemm_kernel<double, 64, 64, BK, false, KR_FULL, TM_FULL, EP_STORE>;
            constexpr size_t lds = gemm_lds_bytes<double, 64, 64, BK>();
            static LdsOptIn opt_in;
            TGP_TRY(opt_in.ensure(reinterpret_cast<const void *>(kern), c.device, lds));
            hipLaunchKernelGGL(kern, dim3(g.ntm * g.ntn, 1, 1), dim3(256), lds, s, g);
            TGP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(ard_reduce_kernel, dim3(Dp), dim3(256), 0, s, c.d_Xs, Z, c.d_gout + 3, N, Dp, Zc);
        TGP_TRY(hipGetLastError());
    }
    TGP_TRY(hipEventRecord(c.evg[3], s));
    return hipSuccess;
}

}  // namespace tgp
