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
    for (int var : {0, 4, 8, 3, 38}) {
        for (int rep = 0; rep < 3; ++rep) {
            if (var == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dflag, dtot);
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
        if (var != 0) {
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
