// f64 VALU issue rate from one workgroup: cycles per v_fma_f64 wave-instruction with 1, 2, 4
// waves per SIMD, for independent accumulators (ILP 16) and for one dependent chain.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/f64_issue.hip -o tools/microbench/f64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int ILP>
__global__ void k(double *out, long long *cyc, int iters, double s) {
    double acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], s, 1e-9);
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    double t = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) t += acc[i];
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int ilp : {1, 16}) {
        for (int threads : {64, 256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (ilp == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.999);
                else hipLaunchKernelGGL(k<16>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.999);
                hipDeviceSynchronize();
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double per_wave_instr = (double)c / ((double)iters * 8 * ilp);
            printf("ILP %2d threads %4d (waves/SIMD %.2f): %.2f cycles per fma per wave, %.2f cycles per fma per SIMD-slot\n",
                   ilp, threads, threads / 256.0, per_wave_instr, per_wave_instr / (threads > 256 ? threads / 256.0 : 1.0));
        }
    }
    return 0;
}
