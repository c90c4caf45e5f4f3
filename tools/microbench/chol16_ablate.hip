// chol16_ablate -- where the 280 cycles per column of chol16_inv_wave (csrc/chol64.hpp) go: the same instruction
// stream with single pieces taken out (timing only: the ablated variants compute garbage), and candidate
// re-arrangements of the pivot chain with the real arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../turbo_amd/csrc chol16_ablate.hip -o chol16_ablate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define TGP_STAMP(slot)
#include "chol64.hpp"
using namespace tgp;

// ABL: 0 as shipped | 1 no LDS publish / fetch | 2 no v_readlane of t (rows k+1, k+2 from own lane) | 3 rsqrt_newton -> one multiply
// 4 = 1 + 2 | 5 = 1 + 2 + 3 | 6 only the pivot chain (no updates of idx > k + 2) | 7 the own-diagonal variant (real arithmetic)
template <int ABL>
__device__ __forceinline__ int chol16_x(double (*At)[CH_LD], double (*Xt)[CH_LD], double *scratch, int c) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    const bool isX = (lane >> 4) & 1;
    const double sg = isX ? -1.0 : 1.0;
    double *colv = scratch + 16 * (lane >> 4);
    double r[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2 += 2) {
        const d2_t v = *reinterpret_cast<const d2_t *>(&At[c + i][c + k2]);
        r[k2] = isX ? ((k2 == i) ? -1.0 : 0.0) : v[0];
        r[k2 + 1] = isX ? ((k2 + 1 == i) ? -1.0 : 0.0) : v[1];
    }
    constexpr bool NOLDS = ABL == 1 || ABL == 4 || ABL == 5, NORL = ABL == 2 || ABL == 4 || ABL == 5, NORSQ = ABL == 3 || ABL == 5;
    double tprev = 0.0, Sprev[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Sprev[k] = 0.001 * k;
    if (ABL == 7) {
        // own-diagonal form: every lane keeps dg = its own diagonal entry minus the squares of its finished row entries;
        // the NEXT pivot is readlane(dg, k + 1) right behind t, without waiting for the broadcast of t
        double dg = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) dg = (k == i) ? r[k] : dg;
        double piv = readlane_f64(dg, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double y = __builtin_amdgcn_rsq(piv);
            const double tt = piv * y;
            const double e = fma(-tt, y, 1.0);
            const double p = fma(0.375, e, 0.5);
            const double ye = y * e;
            const double rs = fma(ye, p, y);
            const double t = r[k] * rs;
            dg = fma(-t, t, dg);                                  // (meaningful in the lanes below the pivot)
            if (k < 15) piv = readlane_f64(dg, k + 1);            // next pivot: off this step's t by ONE fma
            r[k] = sg * t;
            if (k < 15) r[k + 1] = fma(-readlane_f64(t, k + 1), t, r[k + 1]);
            if (k < 14) r[k + 2] = fma(-readlane_f64(t, k + 2), t, r[k + 2]);
            if (k >= 1) {
#pragma unroll
                for (int idx = k + 2; idx < 16; ++idx) r[idx] = fma(-Sprev[idx], tprev, r[idx]);
            }
            if (k < 13) {
                colv[i] = t;
#pragma unroll
                for (int p2 = ((k + 3) & ~1); p2 < 16; p2 += 2) {
                    const d2_t v = *reinterpret_cast<const d2_t *>(scratch + p2);
                    Sprev[p2] = v[0]; Sprev[p2 + 1] = v[1];
                }
                tprev = t;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double piv = readlane_f64(r[k], k);
            const double rs = NORSQ ? piv * 0.5 : rsqrt_newton(piv);
            const double t = r[k] * rs;
            r[k] = sg * t;
            if (k < 15) r[k + 1] = fma(-(NORL ? t : readlane_f64(t, k + 1)), t, r[k + 1]);
            if (k < 14) r[k + 2] = fma(-(NORL ? t : readlane_f64(t, k + 2)), t, r[k + 2]);
            if (k >= 1 && ABL != 6) {
#pragma unroll
                for (int idx = k + 2; idx < 16; ++idx) r[idx] = fma(-Sprev[idx], tprev, r[idx]);
            }
            if (k < 13) {
                if (!NOLDS && ABL != 6) {
                    colv[i] = t;
#pragma unroll
                    for (int p2 = ((k + 3) & ~1); p2 < 16; p2 += 2) {
                        const d2_t v = *reinterpret_cast<const d2_t *>(scratch + p2);
                        Sprev[p2] = v[0]; Sprev[p2 + 1] = v[1];
                    }
                }
                tprev = t;
            }
        }
    }
    if (lane < 32) {
        if (!isX) {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2 += 2) {
                d2_t v; v[0] = r[k2]; v[1] = r[k2 + 1];
                *reinterpret_cast<d2_t *>(&At[c + i][c + k2]) = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) Xt[c + k][c + i] = r[k];
        }
    }
    return 0;
}

template <int ABL>
__global__ __launch_bounds__(256) void kx(const double *A, long long *out, double *Lout) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + 96];
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += 256) { At[i >> 6][i & 63] = (i >> 6) == (i & 63) ? 70.0 : A[i] * 0.01; Xt[i >> 6][i & 63] = 0.0; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (tid < 64) chol16_x<ABL>(At, Xt, rsbuf, 0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid < 64) chol16_x<ABL>(At, Xt, rsbuf, 16);
    const long long t2 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
    for (int i = tid; i < 32 * 32; i += 256) Lout[i] = At[i >> 5][i & 31] + 1e3 * Xt[i >> 5][i & 31];
}

int main() {
    std::vector<double> A(64 * 64);
    srand(1);
    for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < i; ++j) A[i * 64 + j] = A[j * 64 + i];
    double *dA, *dL; long long *d16;
    (void)hipMalloc(&dA, 64 * 64 * 8); (void)hipMalloc(&dL, 32 * 32 * 8); (void)hipMalloc(&d16, 64);
    (void)hipMemcpy(dA, A.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
    const char *names[8] = {"as shipped", "no LDS publish / fetch", "no v_readlane of t", "rsqrt_newton -> one multiply", "no LDS, no readlane of t",
                            "no LDS, no readlane of t, no rsqrt", "pivot chain only (no updates of idx > k + 2)", "own-diagonal form (real arithmetic)"};
    std::vector<double> ref(32 * 32), got(32 * 32);
    for (int abl = 0; abl < 8; ++abl) {
        for (int rep = 0; rep < 3; ++rep) {
            switch (abl) {
                case 0: hipLaunchKernelGGL(kx<0>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 1: hipLaunchKernelGGL(kx<1>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 2: hipLaunchKernelGGL(kx<2>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 3: hipLaunchKernelGGL(kx<3>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 4: hipLaunchKernelGGL(kx<4>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 5: hipLaunchKernelGGL(kx<5>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                case 6: hipLaunchKernelGGL(kx<6>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
                default: hipLaunchKernelGGL(kx<7>, dim3(1), dim3(256), 0, 0, dA, d16, dL); break;
            }
            (void)hipDeviceSynchronize();
        }
        long long h[2]; (void)hipMemcpy(h, d16, 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(got.data(), dL, 32 * 32 * 8, hipMemcpyDeviceToHost);
        if (abl == 0) ref = got;
        double d = 0; for (int i = 0; i < 32 * 32; ++i) d = fmax(d, fabs(got[i] - ref[i]));
        printf("%-48s %5lld / %5lld cycles per 16 columns (%3lld per column)   max|diff to shipped| %.2e\n", names[abl], h[0], h[1], h[1] / 16, d);
    }
    return 0;
}
