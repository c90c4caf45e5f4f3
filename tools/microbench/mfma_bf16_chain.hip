// v_mfma_f32_32x32x16_bf16 issue rate as the x3 contraction uses it: NACC accumulators per wave,
// CHAIN consecutive MFMAs on the same accumulator before moving on, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_chain.hip -o mfma_bf16_chain && ./mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned seed) {
    f16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u4_t av = {seed + threadIdx.x, seed * 3u, seed * 5u, seed * 7u}, bv = {seed * 11u, seed, seed + 1u, seed + 2u};
    const bf16x8_t A = __builtin_bit_cast(bf16x8_t, av), B = __builtin_bit_cast(bf16x8_t, bv);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)(t1 - t0);
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, int CHAIN>
static void run(int threads, int iters) {
    float *out;
    (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(256), dim3(threads), 0, 0, out, 10, 1u);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(256), dim3(threads), 0, 0, out, iters, 1u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    float h[2];
    (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * CHAIN;
    const double waves_per_simd = threads / 256.0;
    const double tf = 256.0 * (threads / 64) * n * 32768.0 / (best * 1e-3) / 1e12;
    printf("acc=%d chain=%d waves/SIMD=%.0f: %.1f TFLOP/s (%.3f of 2516.6)  ticks per MFMA and wave %.1f  -> per SIMD %.1f\n", NACC, CHAIN,
           waves_per_simd, tf, tf / 2516.6, h[1] / n, h[1] / n / waves_per_simd);
    (void)hipFree(out);
}

int main() {
    run<8, 1>(256, 20000);
    run<8, 6>(256, 4000);
    run<8, 1>(512, 20000);
    run<8, 6>(512, 4000);
    run<4, 6>(1024, 4000);
    return 0;
}
