// Does packed-f32 VALU work issued between f32 MFMAs add throughput on MI355X, or do the two
// pipes just share a power budget?  Register-only loops, no memory traffic.
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NV>   // NV packed FMAs per MFMA
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, float seed) {
    f16x acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(threadIdx.x + r);
    f2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f2{seed * i, seed + i};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const f2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = __builtin_elementwise_fma(v[(i * NV + j) & 15], m, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef double d4x __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void k64(float *out, int iters, double seed) {
    d4x acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = seed * (double)(threadIdx.x + r);
    const double a = seed + threadIdx.x, b = seed * 0.5;
    for (int it = 0; it < iters / 16; ++it) {   // 16 rounds per trip: the compiler's AGPR<->VGPR copies at the loop edge amortise
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

static void run64(float *d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, d, iters / 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * (16.0 * 16 * 4 * 2);
    printf("f64 16x16x4 alone  %8.3f ms  MFMA %7.1f TFLOP/s\n", ms, flops / ms / 1e9);
}

template <int NV>
static void run(float *d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters / 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double mfma = waves * iters * 4 * (32.0 * 32 * 2 * 2);
    const double valu = waves * iters * 4 * NV * (64.0 * 2 * 2);
    printf("NV=%2d  %8.3f ms  MFMA %7.1f TFLOP/s  + VALU %7.1f TFLOP/s  = %7.1f\n", NV, ms, mfma / ms / 1e9,
           valu / ms / 1e9, (mfma + valu) / ms / 1e9);
}

int main(int argc, char **argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 4;   // workgroups (of 4 waves) per CU
    const int blocks = 256 * wpc, iters = 200000;
    float *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    printf("%d workgroups of 4 waves per CU\n", wpc);
    run64(d, blocks, iters);
    run<0>(d, blocks, iters);
    run<1>(d, blocks, iters);
    run<2>(d, blocks, iters);
    run<4>(d, blocks, iters);
    run<8>(d, blocks, iters);
    run<12>(d, blocks, iters);
    run<16>(d, blocks, iters);
    return 0;
}
