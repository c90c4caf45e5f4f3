// trmm_f16x2.hpp -- the sweep's contraction (trmm_sweep.hpp) at f32 ACCURACY from the fp16 matrix
// pipe, second scheme (after trmm_bf16x3.hpp): TWO fp16 planes per f32 operand,
//     a * s_a = a1 + 2^-11 a2,    a1 = fp16(a s_a),  a2 = fp16((a s_a - a1) 2^11),
// s_a a power of two that puts max|a| near 2^14 (so both planes sit in fp16's normal range for
// every element that matters; 22 significand bits in all), and
//     a . b = [ a1 b1  +  2^-11 (a1 b2 + a2 b1) ] / (s_a s_b)        [+ O(2^-22) dropped]
// from THREE v_mfma_f32_32x32x16_f16 per 32x32x16 block (96 cycles; the f32 kernel needs 512, the
// three-bf16-plane scheme 192) and 4 bytes per element (bf16x3: 6).  The leading product and the
// two 2^-11 products have accumulators of their own and meet in the epilogue.
// tools/microbench/fp16x2_split.py: error of ||Linv k*||^2 against f64 equal to plain f32's in
// every conditioning tried (1.4e-6 vs 1.5e-6, 5.0e-6 vs 6.2e-6, 1.2e-3 vs 1.1e-3 at noise 1e-8).
// OPT-IN (dtype TGP_F32H2): BASELINE names fp32 for the f32 configurations.
//
// Geometry: 256 rows x 128 candidates per workgroup, 8 waves (4 x 2) of 64 x 64 (two accumulator
// sets of 4 fragments = 128 VGPRs), k-tile 32 as two pre-tiled 16-k blocks, three LDS buffers of
// (256 + 128) rows x 64 B x 2 planes = 48 KB, two k-tiles in flight.  Operands pre-tiled in HBM as
// trmm_bf16x3.hpp describes (blocks of 32 rows x 16 k x 2 planes, swizzle baked in).
#pragma once
#include <hip/hip_runtime.h>

#include "mfma_gemm.hpp"
#include "trmm_bf16x3.hpp"

namespace tgp {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// byte offset of (row, 16-k block, plane) in a pre-tiled two-plane operand
__device__ __forceinline__ long h2_block_off(long row, long k16, int plane, long nkb) {
    return (((row >> 5) * nkb + k16) * 2 + plane) * 1024 + (row & 31) * 32;
}

// (x0, x1) * s -> packed fp16 pairs of the two planes (x0 in the low half)
__device__ __forceinline__ void split2_f16x2(float x0, float x1, float s, unsigned &p1, unsigned &p2) {
    f2v_t v = {x0 * s, x1 * s};
    const f16x2_t h1 = __builtin_convertvector(v, f16x2_t);          // v_cvt_pk_f16_f32, round to nearest even
    const f2v_t b = __builtin_convertvector(h1, f2v_t);
    f2v_t r = {(v[0] - b[0]) * 2048.0f, (v[1] - b[1]) * 2048.0f};
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

// max |in| over n f32 (n a multiple of 4) -> *out (atomic max on the bit pattern; *out zeroed before)
__global__ __launch_bounds__(256) void maxabs_f32_kernel(const float *__restrict__ in, long n4, unsigned *__restrict__ out) {
    float m = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f4_t v = *reinterpret_cast<const f4_t *>(in + 4 * i);
        m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per workgroup (one per wave took 100 us on 8192 waves)
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));   // non-negative floats order as their bits
    }
}

// in: (rows, K) f32 row-major -> the pre-tiled two-plane operand, scaled by s_a = 2^floor(log2(16384 / max|in|)).
// scal[0] = bits of max|in| (maxabs_f32_kernel); thread 0 leaves scal[1] = 1 / (s_a * s_b) as a float
// for the contraction's epilogue.
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float *__restrict__ in,
                                                          unsigned short *__restrict__ out, long rows, long K,
                                                          unsigned *__restrict__ scal, float s_b) {
    const float mx = __uint_as_float(scal[0]);
    const float s_a = mx > 0.f ? exp2f(floorf(log2f(16384.0f / mx))) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[1] = __float_as_uint(1.0f / (s_a * s_b));
    const long nk8 = K / 8, nkb = K / 16;
    const long total = rows * nk8, stride = (long)gridDim.x * 256;
    char *o = reinterpret_cast<char *>(out);
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += stride) {
        const long row = id / nk8, k8 = id - row * nk8;
        const f4_t v0 = *reinterpret_cast<const f4_t *>(in + row * K + 8 * k8);
        const f4_t v1 = *reinterpret_cast<const f4_t *>(in + row * K + 8 * k8 + 4);
        unsigned h[2][4];
        split2_f16x2(v0[0], v0[1], s_a, h[0][0], h[1][0]);
        split2_f16x2(v0[2], v0[3], s_a, h[0][1], h[1][1]);
        split2_f16x2(v1[0], v1[1], s_a, h[0][2], h[1][2]);
        split2_f16x2(v1[2], v1[3], s_a, h[0][3], h[1][3]);
        const int chunk = (int)(k8 & 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const u4_t w = {h[p][0], h[p][1], h[p][2], h[p][3]};
            *reinterpret_cast<u4_t *>(o + h2_block_off(row, k8 >> 1, p, nkb) + x3_chunk_off((int)(row & 31), chunk)) = w;
        }
    }
}

// g.A / g.B: pre-tiled two-plane operands; g.K_blocks = 16-k blocks per operand row;
// g.part as the other trmm kernels; g.Ct = the device float 1 / (s_a s_b)
__global__ __launch_bounds__(512, 1) void trmm_sumsq_f16x2_kernel(GemmArgs g) {
    using MF = Mfma<float>;                          // same 32 x 32 accumulator layout
    constexpr int BM = 256, BN = 128, BK = 32;
    constexpr int ROWB = 32;                          // bytes per row, 16-k block and plane
    constexpr int A_SUB = 2 * BM * ROWB;              // one 16-k block of A: 2 planes x 256 rows x 32 B = 16 KB
    constexpr int B_SUB = 2 * BN * ROWB;              // ... of B: 8 KB
    constexpr int SUB = A_SUB + B_SUB;                // 24 KB
    constexpr int BUF = 2 * SUB;                      // k-tile of 32 = two 16-k blocks
    constexpr int NFM = 2, NFN = 2;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [3][2 sub][A: 2 planes x 256 x 32 B | B: 2 x 128 x 32 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int tm, tn;
    {
        const int bx = blockIdx.x;
        if ((g.ntn & 7) == 0) {
            const int xcd = bx & 7, q = bx >> 3;
            const int per = g.ntn >> 3;
            tn = xcd * per + (q % per);
            tm = g.ntm - 1 - (q / per);
        } else {
            tm = g.ntm - 1 - bx / g.ntn;
            tn = bx % g.ntn;
        }
    }
    int ke = (tm + 1) * BM;
    ke = ke < g.K ? ke : g.K;
    int ke_wave = tm * BM + 64 * (wm + 1);            // this wave row's last useful k (exclusive)
    ke_wave = ke_wave < ke ? ke_wave : ke;

    // ---- staging: per 16-k block 8 A row-blocks x 2 planes + 4 B row-blocks x 2 planes = 24 pieces of
    // 1 KiB, three per wave: A row-block `wave` (both planes) and plane (wave & 1) of B row-block (wave >> 1)
    const long nkb = g.K_blocks;
    const char *Abase = reinterpret_cast<const char *>(g.A) + ((long)(tm * (BM / 32) + wave) * nkb) * 2048 + lane * 16;
    const char *Bbase = reinterpret_cast<const char *>(g.B) + ((long)(tn * (BN / 32) + (wave >> 1)) * nkb) * 2048 + (wave & 1) * 1024 + lane * 16;
    auto stage = [&](int buf, int k0) {
        char *base = smem_raw + buf * BUF;
        const long koff = (long)(k0 >> 4) * 2048;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            char *s0 = base + sb * SUB;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Abase + koff + sb * 2048), (lds_void_t *)(s0 + wave * 32 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Abase + koff + sb * 2048 + 1024), (lds_void_t *)(s0 + BM * ROWB + wave * 32 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Bbase + koff + sb * 2048),
                                             (lds_void_t *)(s0 + A_SUB + (wave & 1) * BN * ROWB + (wave >> 1) * 32 * ROWB), 16, 0, 0);
        }
    };

    f16_t hi[NFM][NFN], mid[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; mid[i][j][r] = 0.f; }

    const int frow = lane & 31;
    const int kg = lane >> 5;
    const int coff = (kg ^ ((frow >> 3) & 1)) * 16;
    const int a_off = (wm * 64 + frow) * ROWB + coff;                 // + plane * BM * ROWB + i * 32 * ROWB
    const int b_off = A_SUB + (wn * 64 + frow) * ROWB + coff;         // + plane * BN * ROWB + j * 32 * ROWB

    auto mma = [](u4_t a, u4_t b, f16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    };
    // One k-tile (32 k = two 16-k blocks): per wave 16 ds_read_b128, 24 MFMAs and -- for the tile two
    // steps ahead -- 6 LDS-DMA instructions.  An LDS-DMA keeps the wave's issue port for ~100 cycles,
    // so the six are placed BETWEEN the MFMA groups (sched_barrier pins the order): the MFMA pipe
    // works through the queued group while the DMA issues (0.79 -> 0.71 ms per launch at C3).
    // Measured and not kept: the barrier in the middle of a tile with the next tile's operands
    // prefetched across it, three tiles in flight (0.73); a 4 x 8 instead of 2 x 16 window of
    // tiles per XCD (0.716); slabs of 8 192 / 32 768 / 65 536 candidates (17.3 / 18.0 / 18.3 ms per
    // step against 17.0).  DMA alone takes 0.42 ms (62 GB/s per CU, the L2's rate), the MFMA phase
    // alone 0.57.
    auto dma_a = [&](char *s0, long koff, int sb, int pl) {
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Abase + koff + sb * 2048 + pl * 1024),
                                         (lds_void_t *)(s0 + sb * SUB + pl * BM * ROWB + wave * 32 * ROWB), 16, 0, 0);
    };
    auto dma_b = [&](char *s0, long koff, int sb) {
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(Bbase + koff + sb * 2048),
                                         (lds_void_t *)(s0 + sb * SUB + A_SUB + (wave & 1) * BN * ROWB + (wave >> 1) * 32 * ROWB), 16, 0, 0);
    };
    auto load_ab = [&](const char *base, u4_t (&a)[NFM][2], u4_t (&b)[NFN][2]) {
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[i][pl] = *reinterpret_cast<const u4_t *>(base + a_off + i * 32 * ROWB + pl * BM * ROWB);
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[j][pl] = *reinterpret_cast<const u4_t *>(base + b_off + j * 32 * ROWB + pl * BN * ROWB);
    };
    auto group = [&](const u4_t (&a)[NFM][2], const u4_t (&b)[NFN][2], int i) {
#pragma unroll
        for (int j = 0; j < NFN; ++j) {
            mid[i][j] = mma(a[i][0], b[j][1], mid[i][j]);
            mid[i][j] = mma(a[i][1], b[j][0], mid[i][j]);
            hi[i][j] = mma(a[i][0], b[j][0], hi[i][j]);
        }
    };

    // three LDS buffers, two k-tiles in flight (counted vmcnt + raw barrier, as trmm_bf16x3.hpp)
    constexpr int PPW = 6;                            // DMA instructions per wave and stage
    const int ntiles = ke / BK;                       // >= 8
    stage(0, 0);
    stage(1, BK);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
    {
        int buf = 0, k0 = 0;
        for (int it = 0; it < ntiles; ++it, k0 += BK) {
            int nb = buf + 2; nb = nb >= 3 ? nb - 3 : nb;
            // past the end the last tile is staged again into a buffer nobody reads any more: the
            // loop body has no branch around the DMAs
            int kn = k0 + 2 * BK;
            kn = kn < ke ? kn : ke - BK;
            char *s0 = smem_raw + nb * BUF;
            const long koff = (long)(kn >> 4) * 2048;
            const char *base = smem_raw + buf * BUF;
            if (k0 < ke_wave) {
                u4_t a0[NFM][2], b0[NFN][2], a1[NFM][2], b1[NFN][2];
                load_ab(base, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                group(a0, b0, 0);
                __builtin_amdgcn_sched_barrier(0);
                dma_a(s0, koff, 0, 0);
                dma_a(s0, koff, 0, 1);
                load_ab(base + SUB, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                group(a0, b0, 1);
                __builtin_amdgcn_sched_barrier(0);
                dma_b(s0, koff, 0);
                dma_a(s0, koff, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                group(a1, b1, 0);
                __builtin_amdgcn_sched_barrier(0);
                dma_a(s0, koff, 1, 1);
                dma_b(s0, koff, 1);
                __builtin_amdgcn_sched_barrier(0);
                group(a1, b1, 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                stage(nb, kn);
            }
            // (lgkmcnt(0) and one asm statement with the barrier: see gemm64_glds.hpp)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
            buf = buf + 1; buf = buf >= 3 ? 0 : buf;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- v = (hi + 2^-11 mid) / (s_a s_b); per-column sum of squares over the 256 rows: each wave
    // row's 64 rows, then the four groups pairwise, f64, fixed order ----
    const float inv = __uint_as_float(*reinterpret_cast<const unsigned *>(g.Ct));
    double *red = reinterpret_cast<double *>(smem_raw);   // [4][128]
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double v = (double)(fmaf(mid[i][j][r], 1.0f / 2048.0f, hi[i][j][r]) * inv);
                s = fma(v, v, s);
            }
#pragma unroll
        for (int o = MF::COL_LANE_STRIDE; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        if (lane < MF::COL_LANE_STRIDE) red[wm * BN + wn * 64 + j * 32 + lane] = s;
    }
    __syncthreads();
    if (tid < BN)
        g.part[(long)tm * g.prm * g.ldpart + (long)tn * BN + tid] =
            (red[0 * BN + tid] + red[1 * BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
}

constexpr size_t trmm_f16x2_lds_bytes() { return (size_t)3 * 2 * (2 * 256 * 32 + 2 * 128 * 32); }

}  // namespace tgp
