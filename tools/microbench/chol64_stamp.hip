// In-kernel phase timing of the 64 x 64 diagonal-block factorisation (csrc/chol64.hpp): s_memtime
// stamps per wave, step and phase, plus a check of L and X against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -I turbo_amd/csrc tools/microbench/chol64_stamp.hip -o tools/microbench/chol64_stamp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ long long g_stamps[4][16][8];
#define TGP_STAMP(slot)                                                                                   \
    do {                                                                                                  \
        if ((threadIdx.x & 63) == 0) g_stamps[threadIdx.x >> 6][g][slot] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#include "chol64.hpp"

using namespace tgp;

template <int GW>
__global__ __launch_bounds__(256) void k(const double *A, double *Lout, double *Xout, int *flag, long long *total) {
    __shared__ __attribute__((aligned(16))) double lds[2 * NB * (NB + 2)];
    const int tid = threadIdx.x, tc = tid >> 4, tr = tid & 15;
    double a[4][4], x[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = A[(4 * tr + i) * 64 + 4 * tc + j];
            x[i][j] = (tr == tc && i == j) ? 1.0 : 0.0;
        }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (GW == 0) factor64_steps4(a, x, lds, 0, flag, 1e-300);
    else if (GW == 3 || GW == 38) {
        if (GW == 38) factor64_v3<8>(a, lds, 0, Lout, 64, flag, 1e-300);
        else factor64_v3<4>(a, lds, 0, Lout, 64, flag, 1e-300);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) x[i][j] = a[i][j];
    } else factor64_steps<(GW == 0 || GW == 3 || GW == 38 ? 4 : GW)>(a, x, lds, 0, flag, 1e-300);
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) *total = t1 - t0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const bool low = (4 * tc + j) <= (4 * tr + i);
            if (GW != 3 && GW != 38) Lout[(4 * tr + i) * 64 + 4 * tc + j] = low ? a[i][j] : 0.0;
            Xout[(4 * tr + i) * 64 + 4 * tc + j] = low ? x[i][j] : 0.0;
        }
}

__global__ __launch_bounds__(256) void k4(const double *A, double *Lout, double *Xout, int *flag, long long *total) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + 96];
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double (*Tb)[CH_LD] = Xt + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += 256) { At[i >> 6][i & 63] = A[i]; Xt[i >> 6][i & 63] = 0.0; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    factor64_v4(At, Xt, Tb, rsbuf, 0, flag, 1e-300);
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) *total = t1 - t0;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        Lout[i] = c <= r ? At[r][c] : 0.0;
        Xout[i] = Xt[r][c];
    }
}

// cycles of the single-wave 16 x 16 factor + inverse alone, and of one panel + update step
__global__ __launch_bounds__(256) void k16(const double *A, long long *out) {
    __shared__ __attribute__((aligned(16))) double lds[3 * NB * CH_LD + 96];
    double (*At)[CH_LD] = reinterpret_cast<double (*)[CH_LD]>(lds);
    double (*Xt)[CH_LD] = At + NB;
    double *rsbuf = lds + 3 * NB * CH_LD;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += 256) { At[i >> 6][i & 63] = (i >> 6) == (i & 63) ? 70.0 : A[i] * 0.01; Xt[i >> 6][i & 63] = 0.0; }
    __syncthreads();
    int bad = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (tid < 64) bad = chol16_inv_wave(At, Xt, rsbuf, 0, 0, 1e-300, bad);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid < 64) bad = chol16_inv_wave(At, Xt, rsbuf, 16, 0, 1e-300, bad);
    const long long t2 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const long long t3 = __builtin_amdgcn_s_memtime();
    d4_t acc = {0, 0, 0, 0};
    mma16<false>(At, 16 * (tid >> 6), 0, Xt, 0, 0, 16, acc, 1.0);
    acc16_store(At, 16 * (tid >> 6), 0, acc, 1.0);
    const long long t4 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const long long t5 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = bad; }
}

int main() {
    std::vector<double> A(64 * 64), L(64 * 64, 0.0), B(64 * 64);
    srand(1);
    for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = (i == j) ? 8.0 : 0.0;
            for (int k = 0; k < 64; ++k) s += B[i * 64 + k] * B[j * 64 + k];
            A[i * 64 + j] = s;
        }
    for (int j = 0; j < 64; ++j) {
        double d = A[j * 64 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
        L[j * 64 + j] = sqrt(d);
        for (int i = j + 1; i < 64; ++i) {
            double s = A[i * 64 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 64 + k] * L[j * 64 + k];
            L[i * 64 + j] = s / L[j * 64 + j];
        }
    }
    double *dA, *dL, *dX; int *dflag; long long *dtot;
    (void)hipMalloc(&dA, 64 * 64 * 8); (void)hipMalloc(&dL, 64 * 64 * 8); (void)hipMalloc(&dX, 64 * 64 * 8);
    (void)hipMalloc(&dflag, 4); (void)hipMalloc(&dtot, 8);
    (void)hipMemcpy(dA, A.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dflag, 0, 4);
    {
        long long *d16; (void)hipMalloc(&d16, 64);
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k16, dim3(1), dim3(256), 0, 0, dA, d16); (void)hipDeviceSynchronize(); }
        long long h16[6]; (void)hipMemcpy(h16, d16, 48, hipMemcpyDeviceToHost);
        printf("chol16_inv_wave: %lld / %lld cycles; barrier %lld; one mma16 tile + store %lld; barrier %lld\n", h16[0], h16[1], h16[2], h16[3], h16[4]);
    }
    for (int var : {0, 4, 8, 3, 38, 5}) {
        for (int rep = 0; rep < 3; ++rep) {
            if (var == 5) hipLaunchKernelGGL(k4, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            else if (var == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            else if (var == 38) hipLaunchKernelGGL(k<38>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            else if (var == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            else if (var == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            else hipLaunchKernelGGL(k<8>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
            (void)hipDeviceSynchronize();
        }
        std::vector<double> gl(64 * 64), gx(64 * 64);
        long long tot; long long st[4][16][8];
        (void)hipMemcpy(gl.data(), dL, 64 * 64 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(gx.data(), dX, 64 * 64 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&tot, dtot, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
        double el = 0, ex = 0;
        for (int i = 0; i < 64 * 64; ++i) el = fmax(el, fabs(gl[i] - L[i]));
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += gl[i * 64 + k] * gx[k * 64 + j];
                ex = fmax(ex, fabs(s - (i == j ? 1.0 : 0.0)));
            }
        printf("variant %d: total %lld cycles (with stamps), max|L - Lref| %.2e, max|L X - I| %.2e\n", var, tot, el, ex);
        if (var != 0 && var != 5) {
            const int steps = var == 3 ? 16 : (var == 38 ? 8 : 64 / var);
            const char *names[6] = {"publish", "barrier", "factor", "solves", "finalise", "update"};
            for (int w = 0; w < 4; ++w) {
                printf("  wave %d:", w);
                for (int ph = 0; ph < 6; ++ph) {
                    long long sum = 0;
                    for (int g = 0; g < steps; ++g) sum += st[w][g][ph + 1] - st[w][g][ph];
                    printf(" %s %lld", names[ph], sum / steps);
                }
                long long per = (st[w][steps - 1][6] - st[w][0][0]) / steps;
                printf("  | per step %lld\n", per);
            }
        }
    }
    return 0;
}
