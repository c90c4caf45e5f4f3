// pairwise.hpp -- 64x64 tile of scaled squared distances + the stationary kernel value.
//
// Direct sum of squared differences over the already length-scaled inputs, as scipy's
// pdist/cdist do for sklearn (kernels.py:1556,1562,1711,1715): NOT the |x|^2+|y|^2-2x.y
// expansion, which cancels near observed points.
// 256 threads, thread (tx = tid&15, ty = tid>>4) owns rows 4*ty..4*ty+3 of P against rows
// 4*tx..4*tx+3 of Q.  Point blocks are staged transposed in LDS ([d][point]) so a lane reads its
// four points with one aligned vector load; the 16-byte row pad keeps the transposing writes
// spread over the banks.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/turbogp.h"

namespace tgp {

constexpr int PW_T = 64;     // points per tile side
constexpr int PW_PAD64 = 2;  // 16 bytes of f64
constexpr int PW_PAD32 = 4;  // 16 bytes of f32

// dimensions staged per pass: one 128-byte line per point
template <typename T> struct PwCfg;
template <> struct PwCfg<double> { static constexpr int DC = 16, VEC = 2, LD = PW_T + PW_PAD64; };
template <> struct PwCfg<float> { static constexpr int DC = 32, VEC = 4, LD = PW_T + PW_PAD32; };

__device__ __forceinline__ double tgp_exp(double x) { return exp(x); }
// f32 sweep: v_exp_f32 on x*log2(e) (about 1 ulp of f32 in the result for the arguments seen here)
__device__ __forceinline__ float tgp_exp(float x) { return __expf(x); }

// KIND is a compile-time kernel id so only one formula is instantiated per kernel.
template <typename T, int KIND>
__device__ __forceinline__ T kernel_value(T d2, T constant) {
    // sklearn kernels.py: RBF :1557/:1563, Matern :1717-1724; Product with ConstantKernel :966
    if (KIND == TGP_RBF) {
        return constant * tgp_exp((T)-0.5 * d2);
    } else if (KIND == TGP_MATERN12) {
        const T d = sqrt(d2);
        return constant * tgp_exp(-d);
    } else if (KIND == TGP_MATERN32) {
        const T k = sqrt(d2) * (T)1.7320508075688772;
        return constant * (((T)1.0 + k) * tgp_exp(-k));
    } else {
        const T k = sqrt(d2) * (T)2.23606797749979;
        return constant * (((T)1.0 + k + (k * k) * (T)0.33333333333333333) * tgp_exp(-k));   // (kept in this exact order: the f64 goldens pin it)
    }
}

// Building blocks.  P and Q are row-major with row stride ld (a multiple of 4 elements, zero
// padded beyond D), so the staging loads are 16-byte vectors: 8 consecutive lanes fetch one
// point's 128-byte line.  Rows >= n read as zeros.
template <typename T>
struct PwStage {
    static constexpr int DC = PwCfg<T>::DC, VEC = PwCfg<T>::VEC, LD = PwCfg<T>::LD;
    static constexpr int VPP = DC / VEC;             // vectors per point per pass (8)
    static constexpr int PASSES = PW_T * VPP / 256;  // 2
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    vec_t v[PASSES];

    // global -> registers: rows r0.. of M (n rows, stride ld), dims d0..d0+DC
    __device__ __forceinline__ void load(const T *__restrict__ M, int r0, int n, int ld, int d0) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = (int)threadIdx.x + 256 * p;
            const int r = idx / VPP, dv = (idx % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[p][e] = (T)0;
            if ((d0 + dv) < ld && (r0 + r) < n)
                v[p] = *reinterpret_cast<const vec_t *>(M + (long)(r0 + r) * ld + d0 + dv);
        }
    }
    // registers -> LDS, transposed to [dim][point]
    __device__ __forceinline__ void store(T (*S)[LD]) const {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = (int)threadIdx.x + 256 * p;
            const int r = idx / VPP, dv = (idx % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) S[dv + e][r] = v[p][e];
        }
    }
};

// d2[a][b] += sum over the staged dims of (Ct[d][4ty+a] - Xt[d][4tx+b])^2
template <typename T>
__device__ __forceinline__ void pw_accumulate_at(T (*Ct)[PwCfg<T>::LD], T (*Xt)[PwCfg<T>::LD], int dn,
                                                 T d2[4][4], int tx, int ty) {
    constexpr int DC = PwCfg<T>::DC;
    auto step = [&](int d) {
        T cv[4], xv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) cv[a] = Ct[d][4 * ty + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) xv[b] = Xt[d][4 * tx + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const T df = cv[a] - xv[b];
                d2[a][b] = fma(df, df, d2[a][b]);
            }
    };
    if (dn >= DC) {
#pragma unroll
        for (int d = 0; d < DC; ++d) step(d);
    } else {
        for (int d = 0; d < dn; ++d) step(d);
    }
}
// (256 threads: thread (tx = tid & 15, ty = tid >> 4))
template <typename T>
__device__ __forceinline__ void pw_accumulate(T (*Ct)[PwCfg<T>::LD], T (*Xt)[PwCfg<T>::LD], int dn,
                                              T d2[4][4]) {
    pw_accumulate_at<T>(Ct, Xt, dn, d2, (int)(threadIdx.x & 15), (int)(threadIdx.x >> 4));
}

// one 64x64 tile, no prefetch (the fit's kernel matrix: one tile per workgroup)
template <typename T>
__device__ __forceinline__ void pairwise_sqdist(const T *__restrict__ P, int p0, int nP,
                                                const T *__restrict__ Q, int q0, int nQ, int ld,
                                                T (*Ct)[PwCfg<T>::LD], T (*Xt)[PwCfg<T>::LD],
                                                T d2[4][4]) {
    constexpr int DC = PwCfg<T>::DC;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) d2[a][b] = (T)0;
    PwStage<T> sp, sq;
    for (int d0 = 0; d0 < ld; d0 += DC) {
        sp.load(P, p0, nP, ld, d0);
        sq.load(Q, q0, nQ, ld, d0);
        __syncthreads();
        sp.store(Ct);
        sq.store(Xt);
        __syncthreads();
        pw_accumulate<T>(Ct, Xt, ld - d0, d2);
    }
}

}  // namespace tgp
