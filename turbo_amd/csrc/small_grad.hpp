// small_grad.hpp -- the LML gradient of a small fit (N <= 128), one block pair per call: shared by
// small_grad_kernel (grad_kernels.hip: one workgroup per pair behind small_fit_kernel) and the
// one-launch hyper-parameter optimiser (small_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "chol64.hpp"
#include "pairwise.hpp"
#include "tgp_internal.hpp"

namespace tgp {

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

// ------------------------------------------------------------------------------------------
// Small problems (N <= 128, the sizes turbo's own demos run at): the whole gradient in ONE
// workgroup and one launch behind the one-workgroup fit (small_kernels.hip).  The blocks of
// Linv sit transposed in LDS, each 64 x 64 block of K^-1 = Linv^T Linv comes off the MFMA pipe
// into LDS and is consumed at once by the pairwise pass for that block pair; the ARD sums are
// taken directly, sum_ij w_ij (x_id - x_jd)^2, sixteen dimensions per pass.
// out = [S_c, S_iso, S_diag, gd[0..Dp)] as launch_lml_grad leaves them in d_gout.
// With two blocks (64 < N <= 128) the three block pairs (0,0), (1,0), (1,1) are independent: one
// workgroup each, workgroup g leaving its share at out + g * SMALL_GRAD_OUT_STRIDE; the host adds
// the three shares in that order.
// ------------------------------------------------------------------------------------------
struct SmallGradArgs {
    const double *Xs, *alpha, *Linv;
    double *out;            // device memory or device-mapped host memory
    int N, Np, Dp, ard;
};
constexpr int SG_TSZ = NB * CH_LD;
constexpr int SG_STAGE = 2 * PwCfg<double>::DC * PwCfg<double>::LD;
constexpr size_t SMALL_GRAD_LDS = (size_t)(4 * SG_TSZ + SG_STAGE + 128 + 64 + 64) * sizeof(double);

// the share of block pair pr (0: (0,0), 1: (1,0), 2: (1,1)), left at p.out + pr * SMALL_GRAD_OUT_STRIDE;
// sm: SMALL_GRAD_LDS bytes of LDS
template <int KIND>
__device__ __forceinline__ void small_grad_body(const SmallGradArgs &p, const int pr, double *sm) {
    typedef double (*tile_t)[CH_LD];
    // (three named tiles, not an array indexed at run time: with a run-time index the compiler lost
    // the LDS address space of the pointers and emitted GLOBAL stores to the LDS offsets)
    tile_t X11T = reinterpret_cast<tile_t>(sm);
    tile_t X21T = reinterpret_cast<tile_t>(sm + SG_TSZ);
    tile_t X22T = reinterpret_cast<tile_t>(sm + 2 * SG_TSZ);
    tile_t Kt = reinterpret_cast<tile_t>(sm + 3 * SG_TSZ);         // the current block of K^-1
    double (*Ct)[PwCfg<double>::LD] = reinterpret_cast<double (*)[PwCfg<double>::LD]>(sm + 4 * SG_TSZ);
    double (*Xt)[PwCfg<double>::LD] = Ct + PwCfg<double>::DC;
    double *alph = sm + 4 * SG_TSZ + SG_STAGE;    // [128]
    double *red = alph + 128;                     // [4][16]
    double *gtot = red + 64;                      // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tid & 15, ty = tid >> 4;
    const int N = p.N, Np = p.Np, Dp = p.Dp;
    const int nb = (N + NB - 1) / NB;

    auto load_T = [&](tile_t dst, int rb, int cb) {
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int idx = tid + 256 * p8, r = idx >> 5, c2 = (idx & 31) * 2;
            const d2_t v = *reinterpret_cast<const d2_t *>(p.Linv + (long)(rb * NB + r) * Np + cb * NB + c2);
            dst[c2][r] = v[0];
            dst[c2 + 1][r] = v[1];
        }
    };
    if (pr == 0) load_T(X11T, 0, 0);
    if (nb == 2) {
        if (pr < 2) load_T(X21T, 1, 0);
        if (pr > 0) load_T(X22T, 1, 1);
    }
    if (tid < 128) alph[tid] = (tid < N) ? p.alpha[tid] : 0.0;
    if (tid < 64) gtot[tid] = 0.0;
    __syncthreads();

    double sc = 0.0, siso = 0.0, sdiag = 0.0;
    {
        const int bi = pr > 0 ? 1 : 0, bj = pr == 2 ? 1 : 0;
        d4_t acc[2][2];
        acc_zero(acc);
        if (pr == 0) {
            tile_mma64(X11T, X11T, acc);
            if (nb == 2) tile_mma64(X21T, X21T, acc);
        } else if (pr == 1) {
            tile_mma64(X22T, X21T, acc);
        } else {
            tile_mma64(X22T, X22T, acc);
        }
        acc_foreach(acc, [&](int r, int c2, double v) { Kt[r][c2] = v; });
        double d2[4][4];
        pairwise_sqdist<double>(p.Xs, bi * NB, N, p.Xs, bj * NB, N, Dp, Ct, Xt, d2);   // (its barriers publish Kt too)
        const double mult = (bi == bj) ? 1.0 : 2.0;
        double w[4][4];
        double psc = 0.0, piso = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = bi * NB + 4 * ty + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = bj * NB + 4 * tx + b;
                w[a][b] = 0.0;
                if (i < N && j < N) {
                    const double G = alph[i] * alph[j] - Kt[4 * ty + a][4 * tx + b];
                    if (i == j) {
                        psc += G;
                        sdiag += G;
                    } else {
                        const double k0 = kernel_value<double, KIND>(d2[a][b], 1.0);
                        w[a][b] = G * ls_weight<KIND>(d2[a][b]);
                        psc = fma(G, k0, psc);
                        piso = fma(w[a][b], d2[a][b], piso);
                    }
                }
            }
        }
        sc = fma(psc, mult, sc);
        siso = fma(piso, mult, siso);
        if (p.ard) {
            PwStage<double> sp, sq;
            for (int d0 = 0; d0 < Dp; d0 += 16) {
                sp.load(p.Xs, bi * NB, N, Dp, d0);
                sq.load(p.Xs, bj * NB, N, Dp, d0);
                __syncthreads();
                sp.store(Ct);
                sq.store(Xt);
                __syncthreads();
                double pd[16];
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
                    pd[e] = q;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    double q = pd[e];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                    if (lane == 0) red[wave * 16 + e] = q;
                }
                __syncthreads();
                if (tid < 16 && d0 + tid < Dp)
                    gtot[d0 + tid] += (0.5 * mult) * ((red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]));
            }
        }
    }
    // the three scalar sums over the workgroup, fixed order
    {
        double v3[3] = {sc, siso, sdiag};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double q = v3[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            v3[k] = q;
        }
        __syncthreads();
        if (lane == 0) { red[wave * 16] = v3[0]; red[wave * 16 + 1] = v3[1]; red[wave * 16 + 2] = v3[2]; }
        __syncthreads();
        double *out = p.out + (long)pr * SMALL_GRAD_OUT_STRIDE;
        if (tid < 3) out[tid] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
        if (p.ard && tid < Dp && tid < 64) out[3 + tid] = gtot[tid];
    }
}


}  // namespace tgp
