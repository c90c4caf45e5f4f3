// trmm_bf16x3.hpp -- the sweep's contraction (trmm_sweep.hpp) with f32 ACCURACY from the bf16
// matrix pipe: every f32 operand is split into three bf16 planes a = a1 + a2 + a3 (round to
// nearest at each step, so the three planes carry the 24-bit significand exactly) and
//     a . b  =  a1 b1  +  (a1 b2 + a2 b1)  +  (a1 b3 + a2 b2 + a3 b1)   [+ O(2^-24) dropped]
// is accumulated in f32 by six v_mfma_f32_32x32x16_bf16 per 32x32x16 block = 192 cycles, against
// 512 for the eight v_mfma_f32_32x32x2_f32 of the f32 kernel.  The six products of a fragment
// are issued smallest first into one accumulator (a second accumulator for the small terms would
// shave the error a little further -- tools/microbench/bf16x3_split.py: 4e-7 against plain f32's
// 1.5e-6 for ||Linv k*||^2 -- but the registers buy the 256 x 256 tile; measured on the GPU the
// variance error is 0.6-0.9 of the f32 sweep's).  No scaling is involved: bf16 has f32's exponent
// range.  OPT-IN (dtype TGP_F32X3): BASELINE names fp32 for the f32 configurations.
// trmm_f16x2.hpp is the faster sibling (two scaled fp16 planes, three products).
//
// The kernel's geometry (256 x 256 tile, pre-tiled operands) is described at the kernel below.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mfma_gemm.hpp"
#include "trmm_sweep.hpp"

namespace tgp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// f32 -> three bf16 (round to nearest even at every step)
__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void split_bf16x3(float x, unsigned &h1, unsigned &h2, unsigned &h3) {
    h1 = bf16_rne(x);
    const float r1 = x - __uint_as_float(h1 << 16);
    h2 = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(h2 << 16);
    h3 = bf16_rne(r2);
}

// ---- operand layout in HBM: PRE-TILED, so that one LDS-DMA instruction is one contiguous KiB ----
// A (rows x K) bf16x3 operand is stored as blocks of 32 rows x 16 k x 3 planes:
//     byte offset of (row, k, plane) = (((row / 32) * (K / 16) + k / 16) * 3 + plane) * 1024
//                                      + (row % 32) * 32 + (chunk ^ ((row % 32) >> 3 & 1)) * 16 + (k % 8) * 2,
//     chunk = (k % 16) / 8
// i.e. exactly the image the kernel wants in LDS (32-byte rows, the bank swizzle baked in): a
// wave's global_load_lds_dwordx4 copies 1 KiB linearly, eight full 128-byte lines.  (With row-major
// planes a k-tile of 16 is 32 bytes per row: every request used a quarter of its cache line and
// the DMA path alone ran at 10 B/clk/CU, slower than the whole MFMA phase.)
__device__ __forceinline__ long x3_block_off(long row, long k16, int plane, long nkb) {
    return (((row >> 5) * nkb + k16) * 3 + plane) * 1024 + (row & 31) * 32;
}
__device__ __forceinline__ int x3_chunk_off(int row, int chunk) { return ((chunk ^ ((row >> 3) & 1)) & 1) * 16; }

// two values at a time on v_cvt_pk_bf16_f32 (round to nearest even): p1, p2, p3 = the packed
// (x0, x1) pair of each plane, x0 in the low half
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_bf16x3(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3) {
    f2v_t v = {x0, x1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    v[0] -= __uint_as_float(p1 << 16);
    v[1] -= __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    v[0] -= __uint_as_float(p2 << 16);
    v[1] -= __uint_as_float(p2 & 0xFFFF0000u);
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// in: (rows, K) f32 row-major, ld = K  ->  out: the pre-tiled three-plane operand.  One thread per
// 8 consecutive k of a row (one 16-byte chunk per plane).
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ in,
                                                           unsigned short *__restrict__ out, long rows, long K) {
    const long nk8 = K / 8, nkb = K / 16;
    const long total = rows * nk8, stride = (long)gridDim.x * 256;
    char *o = reinterpret_cast<char *>(out);
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += stride) {
        const long row = id / nk8, k8 = id - row * nk8;
        const f4_t v0 = *reinterpret_cast<const f4_t *>(in + row * K + 8 * k8);
        const f4_t v1 = *reinterpret_cast<const f4_t *>(in + row * K + 8 * k8 + 4);
        unsigned h[3][4];
        split2_bf16x3(v0[0], v0[1], h[0][0], h[1][0], h[2][0]);
        split2_bf16x3(v0[2], v0[3], h[0][1], h[1][1], h[2][1]);
        split2_bf16x3(v1[0], v1[1], h[0][2], h[1][2], h[2][2]);
        split2_bf16x3(v1[2], v1[3], h[0][3], h[1][3], h[2][3]);
        const int chunk = (int)(k8 & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const u4_t w = {h[p][0], h[p][1], h[p][2], h[p][3]};
            *reinterpret_cast<u4_t *>(o + x3_block_off(row, k8 >> 1, p, nkb) + x3_chunk_off((int)(row & 31), chunk)) = w;
        }
    }
}

// g.A / g.B: pre-tiled three-plane operands (above); g.K_blocks = 16-k blocks per operand row
//
// 256 x 256 output tile, 8 waves (4 x 2) of 64 rows x 128 candidates, one workgroup per CU, two
// waves per SIMD.  Operand bytes per flop are what bounds this kernel (6 B per element and a third
// of the MFMA time of the f32 kernel): a 128 x 128 version ran at the rate the Infinity Cache
// serves a CU's LDS-DMA (30 GB/s per CU, MI355X_MICROARCH.md "gather into LDS"), 2.5 x over its
// MFMA time.  k-tile = 16 (one 32-byte row per operand row and plane), three LDS buffers of
// 2 x 3 x 256 rows x 32 B = 48 KB, two k-tiles in flight; 16-byte chunk q of row r sits at chunk
// q ^ ((r >> 3) & 1): the 16 rows of a quarter-wave's ds_read_b128 cover all 64 banks once.
__global__ __launch_bounds__(512, 1) void trmm_sumsq_bf16x3_kernel(GemmArgs g) {
    using MF = Mfma<float>;                          // same 32 x 32 accumulator layout
    constexpr int BM = 256, BN = 256, BK = 16;
    constexpr int ROWB = 32;                          // bytes per tile row and plane
    constexpr int PLANE = BM * ROWB;                  // 8 KB
    constexpr int OPER = 3 * PLANE;                   // 24 KB
    constexpr int BUF = 2 * OPER;                     // A | B
    constexpr int NFM = 2, NFN = 4;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [3][A|B][3][256][32 B]

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

    // ---- staging: one wave-instruction copies one pre-tiled block plane (32 rows x 32 B = 1 KiB,
    // contiguous); wave w moves the blocks of rows 32 w .. of each operand.
    const long nkb = g.K_blocks;                      // 16-k blocks per operand row (= ld / 16)
    const char *Abase = reinterpret_cast<const char *>(g.A) + ((long)(tm * (BM / 32) + wave) * nkb) * 3072 + lane * 16;
    const char *Bbase = reinterpret_cast<const char *>(g.B) + ((long)(tn * (BN / 32) + wave) * nkb) * 3072 + lane * 16;
    auto stage = [&](int buf, int k0) {
        char *base = smem_raw + buf * BUF + wave * 32 * ROWB;
        const long koff = (long)(k0 >> 4) * 3072;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Abase + koff + pl * 1024), (lds_void_t *)(base + pl * PLANE), 16, 0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(Bbase + koff + pl * 1024), (lds_void_t *)(base + OPER + pl * PLANE), 16, 0, 0);
    };

    f16_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;                       // row (A) / candidate (B) inside the fragment
    const int kg = lane >> 5;                         // which 8 of the 16 k
    const int coff = (kg ^ ((frow >> 3) & 1)) * 16;   // fragment row offsets are multiples of 32
    const int a_off = (wm * 64 + frow) * ROWB + coff;
    const int b_off = OPER + (wn * 128 + frow) * ROWB + coff;

    auto mma = [](u4_t a, u4_t b, f16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    };
    // One k-tile: 18 ds_read_b128, 48 MFMAs (four column groups of 12) and -- for the tile two steps
    // ahead -- 6 LDS-DMA instructions per wave.  The order is pinned with sched_barrier: the reads
    // of group j+1 and two DMAs are issued behind the MFMAs of group j (an LDS-DMA holds the wave's
    // issue port for ~100 cycles; the matrix pipe works through the queued group meanwhile).
    auto dma = [&](char *s0, const char *src) { __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)s0, 16, 0, 0); };
    auto rd3 = [&](const char *base, int off, u4_t (&v)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) v[pl] = *reinterpret_cast<const u4_t *>(base + off + pl * PLANE);
    };
    auto group = [&](const u4_t (&a)[NFM][3], const u4_t (&b)[3], int j) {
#pragma unroll
        for (int i = 0; i < NFM; ++i) {
            f16_t c = acc[i][j];
            c = mma(a[i][0], b[2], c);       // smallest products first, the leading one last
            c = mma(a[i][1], b[1], c);
            c = mma(a[i][2], b[0], c);
            c = mma(a[i][0], b[1], c);
            c = mma(a[i][1], b[0], c);
            c = mma(a[i][0], b[0], c);
            acc[i][j] = c;
        }
    };
    auto compute = [&](int buf, int nb, int kn) {
        const char *base = smem_raw + buf * BUF;
        char *s0 = smem_raw + nb * BUF + wave * 32 * ROWB;
        const long koff = (long)(kn >> 4) * 3072;
        u4_t a[NFM][3], b0[3], b1[3], b2[3], b3[3];
        rd3(base, a_off, a[0]);
        rd3(base, a_off + 32 * ROWB, a[1]);
        rd3(base, b_off, b0);
        rd3(base, b_off + 32 * ROWB, b1);
        __builtin_amdgcn_sched_barrier(0);
        group(a, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        dma(s0, Abase + koff);
        dma(s0 + PLANE, Abase + koff + 1024);
        rd3(base, b_off + 2 * 32 * ROWB, b2);
        __builtin_amdgcn_sched_barrier(0);
        group(a, b1, 1);
        __builtin_amdgcn_sched_barrier(0);
        dma(s0 + 2 * PLANE, Abase + koff + 2048);
        dma(s0 + OPER, Bbase + koff);
        rd3(base, b_off + 3 * 32 * ROWB, b3);
        __builtin_amdgcn_sched_barrier(0);
        group(a, b2, 2);
        __builtin_amdgcn_sched_barrier(0);
        dma(s0 + OPER + PLANE, Bbase + koff + 1024);
        dma(s0 + OPER + 2 * PLANE, Bbase + koff + 2048);
        __builtin_amdgcn_sched_barrier(0);
        group(a, b3, 3);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Three LDS buffers, two k-tiles in flight (as trmm_sumsq_glds_big_kernel<.., 3>): the barrier
    // that ends iteration i waits for tile i+1 only -- a counted s_waitcnt vmcnt(PPW) leaves this
    // wave's PPW newest DMA instructions (tile i+2) pending -- then a raw s_barrier.
    constexpr int PPW = 6;                            // DMA instructions per wave and stage
    const int ntiles = ke / BK;                       // >= 16
    stage(0, 0);
    stage(1, BK);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
    {
        int buf = 0, k0 = 0;
        for (int it = 0; it < ntiles; ++it, k0 += BK) {
            int nb = buf + 2; nb = nb >= 3 ? nb - 3 : nb;
            // The DMA issue of tile i+2 rides inside the MFMA stream of tile i.  Past the end the last
            // tile is staged again into a buffer nobody reads any more, so the loop body has no
            // branch around the DMAs.
            int kn = k0 + 2 * BK;
            kn = kn < ke ? kn : ke - BK;
            if (k0 < ke_wave) compute(buf, nb, kn);
            else stage(nb, kn);
            // (lgkmcnt(0) and one asm statement with the barrier: see gemm64_glds.hpp)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
            buf = buf + 1; buf = buf >= 3 ? 0 : buf;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the re-staged last tiles land before the buffers are reused below)
    __syncthreads();

    // ---- per-column sum of squares over the tile's 256 rows: each wave row's 64 rows, then the
    // four groups pairwise, f64, fixed order (as trmm_sumsq_glds_big_kernel) ----
    double *red = reinterpret_cast<double *>(smem_raw);   // [4][256]
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double v = (double)acc[i][j][r];
                s = fma(v, v, s);
            }
#pragma unroll
        for (int o = MF::COL_LANE_STRIDE; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        if (lane < MF::COL_LANE_STRIDE) red[wm * BN + wn * 128 + j * 32 + lane] = s;
    }
    __syncthreads();
    if (tid < BN)
        g.part[(long)tm * g.prm * g.ldpart + (long)tn * BN + tid] =
            (red[0 * BN + tid] + red[1 * BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
}

constexpr size_t trmm_bf16x3_lds_bytes() { return (size_t)3 * 2 * 3 * 256 * 32; }

}  // namespace tgp
