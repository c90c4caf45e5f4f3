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
constexpr int PW_DC = 16;    // dimensions staged per pass
constexpr int PW_PAD64 = 2;  // 16 bytes of f64
constexpr int PW_PAD32 = 4;  // 16 bytes of f32

template <typename T> struct PwPad;
template <> struct PwPad<double> { static constexpr int v = PW_PAD64; };
template <> struct PwPad<float> { static constexpr int v = PW_PAD32; };

template <typename T>
__device__ __forceinline__ T kernel_value(int kind, T d2, T constant) {
    // sklearn kernels.py: RBF :1557/:1563, Matern :1717-1724; Product with ConstantKernel :966
    switch (kind) {
        case TGP_RBF:
            return constant * exp((T)-0.5 * d2);
        case TGP_MATERN12: {
            const T d = sqrt(d2);
            return constant * exp(-d);
        }
        case TGP_MATERN32: {
            const T k = sqrt(d2) * (T)1.7320508075688772;
            return constant * (((T)1.0 + k) * exp(-k));
        }
        default: {
            const T k = sqrt(d2) * (T)2.23606797749979;
            return constant * (((T)1.0 + k + k * k / (T)3.0) * exp(-k));
        }
    }
}

// d2[a][b] = sum_d (P[p0+4ty+a][d] - Q[q0+4tx+b][d])^2 ; rows >= nP / nQ read as zeros.
template <typename T, typename TIN, int LD>
__device__ __forceinline__ void pairwise_sqdist(const TIN *__restrict__ P, int p0, int nP,
                                                const TIN *__restrict__ Q, int q0, int nQ, int D,
                                                T (*Ct)[LD], T (*Xt)[LD], T d2[4][4]) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int sd = tid & (PW_DC - 1), sr = tid >> 4;   // staging: dim, row (+16 per pass)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) d2[a][b] = (T)0;

    for (int d0 = 0; d0 < D; d0 += PW_DC) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PW_T / 16; ++p) {
            const int r = sr + 16 * p;
            const bool dok = (d0 + sd) < D;
            T pv = (T)0, qv = (T)0;
            if (dok && (p0 + r) < nP) pv = (T)P[(long)(p0 + r) * D + d0 + sd];
            if (dok && (q0 + r) < nQ) qv = (T)Q[(long)(q0 + r) * D + d0 + sd];
            Ct[sd][r] = pv;
            Xt[sd][r] = qv;
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < PW_DC; ++d) {
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
        }
    }
}

}  // namespace tgp
