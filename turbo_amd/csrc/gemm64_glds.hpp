// gemm64_glds.hpp -- the fit's f64 workhorse (round 4):  C = alpha * A * B^T (+ C)  on 64 x 64 tiles,
// both operands K-contiguous ("NT").  It carries the rank-OB trailing update of the blocked Cholesky
// (LAPACK dpotrf behind scipy.linalg.cholesky, sklearn/gaussian_process/_gpr.py:349) and the products
// of the triangular inverse.  Those problems are lower-triangular tile sets of a few dozen to a few
// thousand 64 x 64 tiles with K = 64 ... 4096: too few tiles for 128-wide ones, and -- in the late
// outer blocks -- too few to fill the chip at all, so a tile's own latency is what the launch costs.
//
// What it does differently from mfma_gemm_kernel<double, 64, 64, 16, ...> (register-staged, 16-byte row pad):
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write); the
//     LDS image is 128-byte k-rows with 16-byte chunk q of row r at q ^ ((r >> 1) & 7), the layout of
//     trmm_sweep.hpp: every ds_read_b128 fragment fetch is conflict-free;
//   * NBUF k-tile buffers, NBUF - 1 tiles of DMA in flight, completion awaited with a counted
//     s_waitcnt vmcnt(n) lgkmcnt(0) and an s_barrier in ONE asm statement (a __syncthreads() would drain to
//     vmcnt(0); a raw barrier intrinsic let the compiler sink the LDS wait below it: see the k-loop);
//   * branch-free loop bodies and a register budget of four waves per SIMD, so the MFMAs are issued in
//     their VGPR form: the old template's loop moved its 32 accumulator registers AGPR -> VGPR -> AGPR on
//     every trip (64 v_accvgpr moves per 16 MFMAs).
// Measured alone on the chip (tools/microbench/gemm64_bench.hip, profiles/r04_gemm64_bench.txt), rank-512
// trailing updates of N = 4096: 1596 tiles 157.6 -> 142 us, 528 tiles 74 -> 64.5, 36 tiles 36 -> 27; the
// inverse's products 20-30 % faster.  What bounds it now is the CU's own fetch path: without any MFMA the
// DMA of a 1596-tile launch takes 70 us (~55 GB/s per CU, L2 hits or not -- every tile fetching tile (0,0)'s
// operands costs the same), without DMA the MFMA loop 128 us; a k-split inside the workgroup (8 or 16
// waves per tile) measured equal or slower -- a tile's 576 KB arrive through ONE CU either way -- and is
// not kept.  Bigger tiles halve the bytes per flop: launches with enough 128 x 128 tiles take
// gemm_nt_glds.hpp (TGP_TRAIL64 / TGP_MERGE64 in fit_kernels.hip).
//
// Requirements: rows of A / B / C in whole 64-tiles; every k-range a multiple of 16 doubles (all callers
// use multiples of 64); lda, ldb multiples of 2 doubles; beta is 0 or 1.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_opt_in.hpp"
#include "mfma_gemm.hpp"
#include "trmm_sweep.hpp"

namespace tgp {

template <int NBUF>
constexpr size_t gemm64_glds_lds_bytes() { return (size_t)NBUF * 16384; }

template <int KR, int TMAP, int NBUF, int DBG = 0>
__global__ __launch_bounds__(256, 4) void gemm64_glds_kernel(GemmArgs g) {
    using MF = Mfma<double>;
    using vec_t = MF::vec_t;
    using acc_t = MF::acc_t;
    constexpr int BT = 64;                            // tile rows = tile columns
    constexpr int BK = 16;                            // one 128-byte row per k-tile
    constexpr int OP_BYTES = BT * 128;                // one operand of one k-tile
    constexpr int TILE_BYTES = 2 * OP_BYTES;          // A + B of one k-tile
    constexpr int PPW = 4;                            // DMA instructions per wave and k-tile (2 A + 2 B)
    static_assert(NBUF >= 2 && NBUF <= 4, "two to four k-tile buffers");
#ifndef TGP_DEBUG_KERNELS
    // the ablation / pre-fix variants exist in debug builds only (make debug, tools/microbench): the shipped library
    // must not contain a loop known to return wrong tiles under load
    static_assert(DBG == 0, "gemm64_glds_kernel<.., DBG != 0> needs -DTGP_DEBUG_KERNELS");
#endif

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [NBUF][A|B][64][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;

    int tm, tn;
    {
        const int bx = blockIdx.x;
        if (TMAP == TM_LOWER) {
            int r = (int)((sqrtf(8.0f * (float)bx + 1.0f) - 1.0f) * 0.5f);
            while ((r + 1) * (r + 2) / 2 <= bx) ++r;
            while (r * (r + 1) / 2 > bx) --r;
            tm = r;
            tn = bx - r * (r + 1) / 2;
        } else {
            tm = bx / g.ntn;
            tn = bx - tm * g.ntn;
        }
    }
    int kb = 0, ke = g.K;
    if (KR == KR_LOWER_A) { const int lim = (tm + 1) * BT; ke = lim < g.K ? lim : g.K; }
    if (KR == KR_UPPER_A) { kb = tm * BT; }
    const int ntiles = (ke - kb) / BK;

    const double *A = reinterpret_cast<const double *>(g.A) + (long)blockIdx.z * g.strideA;
    const double *B = reinterpret_cast<const double *>(g.B) + (long)blockIdx.z * g.strideB;

    // ---- direct-to-LDS staging: this wave fills pieces `wave` and `wave + 4` (8 rows x 128 B each) of both
    // operands; lane (srow, schunk) fetches the source chunk that belongs at
    // LDS chunk `schunk` of its row under the swizzle
    const int srow = lane >> 3, schunk = lane & 7;
    const char *asrc[2];
    const char *bsrc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave + 4 * p) * 8 + srow;
        const int src_chunk = schunk ^ ((row >> 1) & 7);
        // (DBG: microbenchmark ablations only -- 1 = every tile fetches tile (0, 0)'s operands, 2 = no MFMAs, 3 = no DMA)
        const long am = DBG == 1 ? 0 : tm, bn = DBG == 1 ? 0 : tn;
        asrc[p] = reinterpret_cast<const char *>(A + (am * BT + row) * g.lda + kb) + src_chunk * 16;
        bsrc[p] = reinterpret_cast<const char *>(B + (bn * BT + row) * g.ldb + kb) + src_chunk * 16;
    }
    auto stage = [&](int buf, int tile) {
        if (DBG == 3) return;
        const long koff = (long)tile * (BK * 8);
        char *base = smem_raw + buf * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(asrc[p] + koff), (lds_void_t *)(base + (wave + 4 * p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(bsrc[p] + koff), (lds_void_t *)(base + OP_BYTES + (wave + 4 * p) * 1024), 16, 0, 0);
    };

    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (acc_t){0.0, 0.0, 0.0, 0.0};

    const int fidx = MF::ab_idx(lane);                // row of the fragment this lane feeds
    const int grp = MF::ab_kg(lane);                  // lane group along k (0..3)
    const int swz = (fidx >> 1) & 7;                  // == ((row >> 1) & 7): fragment row offsets are multiples of 16
    const int a_off = (wm0 + fidx) * 128;
    const int b_off = OP_BYTES + (wn0 + fidx) * 128;
    auto compute = [&](int buf) {
        if (DBG == 2) return;
        const char *base = smem_raw + buf * TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                 // the lane groups take permuted k, the same for A and B
            const int coff = ((s * 4 + grp) ^ swz) * 16;
            vec_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const vec_t *>(base + a_off + i * 16 * 128 + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const vec_t *>(base + b_off + j * 16 * 128 + coff);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(a[i][e], b[j][e], acc[i][j]);
        }
    };

    // ---- NBUF - 1 k-tiles in flight.  The barrier that ends iteration `it` must only wait for tile
    // it + 1; the NBUF - 2 tiles issued after it may stay pending: s_waitcnt vmcnt(PPW (NBUF - 2)).
    // WAR: the buffer staged in iteration `it` was last read in iteration it - 1, whose barrier every
    // wave has passed.  Both loop bodies are BRANCH-FREE: with a conditional inside, the compiler kept
    // the accumulators in AGPRs and copied all 32 of them to VGPRs and back on every trip (64
    // v_accvgpr moves per 16 MFMAs: the loop ran at 0.65 of the rate it reaches without them).
    if (ntiles > 0) {
        const int npre = ntiles < NBUF - 1 ? ntiles : NBUF - 1;
        for (int i = 0; i < npre; ++i) stage(i, i);
        if (npre == NBUF - 1 && DBG != 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW * (NBUF - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        int buf = 0, it = 0;
        for (; it + NBUF - 1 < ntiles; ++it) {           // a tile to stage in every trip
            int nb = buf + NBUF - 1; nb = nb >= NBUF ? nb - NBUF : nb;
            stage(nb, it + NBUF - 1);
            compute(buf);
            // lgkmcnt(0): the buffer read here is the one the NEXT trip's DMA overwrites, so this wave's ds_reads
            // must have returned before it lets the others past the barrier.  Without it the compiler sank the
            // wait (and the MFMAs it feeds) below the s_barrier -- the intrinsic is not a memory operation to it --
            // and beside the background stream's kernels, with the LDS queue of a shared CU deep enough, a DMA
            // write overtook a pending read: one fit in ten at N = 5000 came out wrong (round 4; DBG == 5 keeps
            // the old form for tools/repeat_fit.py to show it)
            // (the wait and the s_barrier are ONE asm statement: nothing can be scheduled between them)
            if (DBG == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (debug: no counted wait)
            else if (DBG == 5) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW * (NBUF - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW * (NBUF - 2)) : "memory");
            buf = buf + 1; buf = buf >= NBUF ? 0 : buf;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the last NBUF - 1 tiles: everything has been issued
        for (; it < ntiles; ++it) {
            compute(buf);
            buf = buf + 1; buf = buf >= NBUF ? 0 : buf;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    double *C = reinterpret_cast<double *>(g.C) + (long)blockIdx.z * g.strideC;
    double *Ct = g.Ct ? reinterpret_cast<double *>(g.Ct) + (long)blockIdx.z * g.strideCt : nullptr;
    const double alpha = g.alpha;
    const bool use_beta = g.beta != 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = (long)tn * BT + wn0 + j * 16 + MF::c_col(lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = (long)tm * BT + wm0 + i * 16 + MF::c_row(lane, r);
                double v = alpha * acc[i][j][r];
                double *p = C + row * g.ldc + col;
                if (use_beta) v += *p;
                *p = v;
                if (Ct) Ct[col * g.ldct + row] = v;
            }
        }
}

template <int KR, int TMAP, int NBUF = 3, int DBG = 0>
static hipError_t launch_gemm64_glds(hipStream_t s, int device, const GemmArgs &g, int nblocks, int batch) {
    auto kern = gemm64_glds_kernel<KR, TMAP, NBUF, DBG>;
    constexpr size_t lds = gemm64_glds_lds_bytes<NBUF>();
    static LdsOptIn opt_in;
    if (hipError_t e = opt_in.ensure(reinterpret_cast<const void *>(kern), device, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, batch), dim3(256), lds, s, g);
    return hipGetLastError();
}

}  // namespace tgp
