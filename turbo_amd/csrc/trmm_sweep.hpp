// trmm_sweep.hpp -- the dominant kernel of the candidate sweep:
//     part[tm][c] = sum over rows of tile tm of ( Linv[tm rows, 0:ke) * Ks[c, 0:ke)^T )^2
// i.e. the reference's  V = solve_triangular(L, K*^T);  einsum('ij,ji->i', V^T, V)
// (sklearn/gaussian_process/_gpr.py:454,475) as a lower-triangular MFMA contraction with the
// sum of squares fused into the epilogue, so V (N x M) never exists in memory.
//
// gfx950 specifics
//   * 128x128 output tile, 4 waves (2x2), wave tile 64x64 of v_mfma_f32_32x32x2_f32 /
//     v_mfma_f64_16x16x4_f64 fragments; k-tile = 128 bytes per row (32 f32 / 16 f64).
//   * operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no
//     ds_write): one wave-instruction fills 8 rows x 128 B.  The LDS image is lane-linear, so the
//     bank-conflict swizzle is applied to the per-lane SOURCE chunk and to the read address
//     (16-byte chunk q of row r lives at chunk q ^ ((r >> 1) & 7)): every ds_read_b128 fragment
//     fetch is conflict-free for both fragment shapes.
//   * two LDS buffers, the next k-tile is in flight during the whole MFMA phase; two
//     workgroups per CU (64 KB LDS each) so one block's barrier is covered by the other's MFMAs.
//   * blocks walk the row tiles heaviest-first with the candidate tiles grouped per XCD.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mfma_gemm.hpp"

namespace tgp {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <typename T, typename MF = Mfma<T>>
__global__ __launch_bounds__(256, 2) void trmm_sumsq_glds_kernel(GemmArgs g) {
    using vec_t = typename MF::vec_t;
    using acc_t = typename MF::acc_t;
    constexpr int EPL = MF::EPL;
    constexpr int BM = 128, BN = 128;
    constexpr int BK = 128 / (int)sizeof(T);          // one 128-byte row per k-tile
    // Wave tile = ALL 128 rows x 32 candidates (4 waves side by side), not 2 x 2 quadrants: in a
    // diagonal tile the rows 0..63 have nothing but zeros right of their own diagonal block, and
    // with every wave holding both row halves every wave (= every SIMD) drops the same half of
    // its MFMAs there.  With quadrants the two upper waves idled while the two lower ones set the
    // pace -- no time was saved (N = 512: computed / algorithmic 1.25 -> 1.125; C1 0.656 -> 0.715).
    constexpr int WTM = 128, WTN = 32;
    constexpr int NFM = WTM / MF::FM, NFN = WTN / MF::FN;
    constexpr int NFH = NFM / 2;                      // fragments of the upper row half (rows 0..63): the epilogue's pairing
    constexpr int NG = 64 / MF::FM;                   // lane groups along k (2 for f32, 4 for f64)
    constexpr int KSTEPS = 8 / NG;                    // 16-byte chunks per lane per k-tile
    constexpr int TILE_BYTES = BM * 128;              // one operand tile
    constexpr int BUF_BYTES = 2 * TILE_BYTES;         // A + B

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [2][A|B][128][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = 0;
    const int wn0 = wave * WTN;

    int tm, tn;
    sweep_tile(g, (int)blockIdx.x, tm, tn);
    int ke = (tm + 1) * BM;
    ke = ke < g.K ? ke : g.K;

    // ---- direct-to-LDS staging: wave w, piece p covers tile rows (4p + w) * 8 .. + 8 ----------
    const int srow = lane >> 3;                       // row inside the 8-row piece
    const int schunk = lane & 7;                      // LDS chunk this lane fills
    const char *asrc[4];
    const char *bsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (4 * p + wave) * 8 + srow;
        const int src_chunk = schunk ^ ((row >> 1) & 7);
        asrc[p] = reinterpret_cast<const char *>(reinterpret_cast<const T *>(g.A) +
                                                 ((long)tm * BM + row) * g.lda) + src_chunk * 16;
        bsrc[p] = reinterpret_cast<const char *>(reinterpret_cast<const T *>(g.B) +
                                                 ((long)tn * BN + row) * g.ldb) + src_chunk * 16;
    }
    auto stage = [&](int buf, int k0) {
        const long koff = (long)k0 * (long)sizeof(T);
        char *base = smem_raw + buf * BUF_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            char *la = base + (4 * p + wave) * 8 * 128;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(asrc[p] + koff), (lds_void_t *)la, 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            char *lb = base + TILE_BYTES + (4 * p + wave) * 8 * 128;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(bsrc[p] + koff), (lds_void_t *)lb, 16, 0, 0);
        }
    };

    acc_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) acc[i][j][r] = (T)0;

    const int fidx = MF::ab_idx(lane);
    const int grp = MF::ab_kg(lane);
    const int swz = (fidx >> 1) & 7;                  // == ((row >> 1) & 7): row offsets are multiples of 16
    const int a_row_off = (wm0 + fidx) * 128;
    const int b_row_off = TILE_BYTES + (wn0 + fidx) * 128;

    const int ke_lo = (tm * BM + 64) < ke ? (tm * BM + 64) : ke;   // rows 0..63: last useful k (exclusive)
    // the row tile that spans the whole k-range also accumulates the posterior mean's K*.alpha (MeanAcc, mfma_gemm.hpp)
    const bool do_mean = g.mu != nullptr && ke >= g.K;
    const int mrow = tid & (BN - 1);
    const int mgrp = __builtin_amdgcn_readfirstlane(tid / BN);
    MeanAcc<T, BN, 256> macc;
    int buf = 0;
    stage(0, 0);
    __syncthreads();
    for (int k0 = 0; k0 < ke; k0 += BK) {
        typename MeanAcc<T, BN, 256>::Alpha mal;
        if (do_mean) mal.load(g.mu_alpha + k0, mgrp);
        if (k0 + BK < ke) stage(buf ^ 1, k0 + BK);
        const char *base = smem_raw + buf * BUF_BYTES;
        if (k0 < ke_lo) {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int coff = ((s * NG + grp) ^ swz) * 16;
                vec_t a[NFM], b[NFN];
#pragma unroll
                for (int i = 0; i < NFM; ++i)
                    a[i] = *reinterpret_cast<const vec_t *>(base + a_row_off + i * MF::FM * 128 + coff);
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    b[j] = *reinterpret_cast<const vec_t *>(base + b_row_off + j * MF::FN * 128 + coff);
#pragma unroll
                for (int e = 0; e < EPL; ++e)
#pragma unroll
                    for (int i = 0; i < NFM; ++i)
#pragma unroll
                        for (int j = 0; j < NFN; ++j) acc[i][j] = MF::mma(a[i][e], b[j][e], acc[i][j]);
            }
        } else {
            // right half of the diagonal block: only the rows 64..127 still have non-zeros.
            // Finer cuts of the triangle were measured and lost: a per-fragment test at run time
            // serialises the MFMA stream (C1: 0.72 -> 0.25 of the peak), four compile-time
            // instances of the loop body picked per k-tile cost more than they save (0.72 -> 0.61).
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int coff = ((s * NG + grp) ^ swz) * 16;
                vec_t a[NFM], b[NFN];
#pragma unroll
                for (int i = NFH; i < NFM; ++i)
                    a[i] = *reinterpret_cast<const vec_t *>(base + a_row_off + i * MF::FM * 128 + coff);
#pragma unroll
                for (int j = 0; j < NFN; ++j)
                    b[j] = *reinterpret_cast<const vec_t *>(base + b_row_off + j * MF::FN * 128 + coff);
#pragma unroll
                for (int e = 0; e < EPL; ++e)
#pragma unroll
                    for (int i = NFH; i < NFM; ++i)
#pragma unroll
                        for (int j = 0; j < NFN; ++j) acc[i][j] = MF::mma(a[i][e], b[j][e], acc[i][j]);
            }
        }
        if (do_mean) macc.add(base + TILE_BYTES, mal, mrow, mgrp);
        __syncthreads();
        buf ^= 1;
    }
    if (do_mean)   // (the k-loop's last barrier is behind every wave: LDS is free)
        macc.finish(reinterpret_cast<double *>(smem_raw), mrow, mgrp, g.mu + (long)tn * BN);

    // ---- per-column sum of squares over this tile's 128 rows, f64, fixed order: the two 64-row
    // halves separately (each as the quadrant kernel summed its wave row), then their sum ----------
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        double sh[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            double s = 0.0;
#pragma unroll
            for (int i = hh * NFH; i < (hh + 1) * NFH; ++i)
#pragma unroll
                for (int r = 0; r < MF::NACC; ++r) {
                    const double v = (double)acc[i][j][r];
                    s = fma(v, v, s);
                }
#pragma unroll
            for (int o = MF::COL_LANE_STRIDE; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
            sh[hh] = s;
        }
        if (lane < MF::COL_LANE_STRIDE)
            g.part[(long)tm * g.prm * g.ldpart + (long)tn * BN + wn0 + j * MF::FN + lane] = sh[0] + sh[1];
    }
}

constexpr size_t trmm_glds_lds_bytes() { return (size_t)2 * 2 * 128 * 128; }

}  // namespace tgp

namespace tgp {

// ------------------------------------------------------------------------------------------
// Large-tile variant: (64*WM) x (64*WN) output tile, WM x WN waves of 64x64, one workgroup per CU.
// Same staging, swizzle and fragment code as above; what changes is the operand traffic per
// flop ((BM + BN) / (BM * BN): 256x256 halves it against 128x128), the DMA instructions per
// wave (4 instead of 8) and the time a k-tile's loads have to land (16 waves x 64 MFMAs).
// The diagonal tile's zero half is skipped per wave row in a separate tail loop so the main
// loop stays branch-free.
// ------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int NBUF = 2>
__global__ __launch_bounds__(64 * WM * WN) void trmm_sumsq_glds_big_kernel(GemmArgs g) {
    using MF = Mfma<T>;
    using vec_t = typename MF::vec_t;
    using acc_t = typename MF::acc_t;
    constexpr int EPL = MF::EPL;
    constexpr int NW = WM * WN;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int NFM = 64 / MF::FM, NFN = 64 / MF::FN;
    constexpr int NG = 64 / MF::FM;
    constexpr int KSTEPS = 8 / NG;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int PIECES = (BM + BN) / 8;          // 8-row x 128-byte pieces per k-tile
    constexpr int PPW = PIECES / NW;               // pieces per wave
    static_assert(PIECES % NW == 0, "pieces must divide evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [2][A|B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * 64;
    const int wn0 = (wave % WN) * 64;

    int tm, tn;
    sweep_tile(g, (int)blockIdx.x, tm, tn);
    const int kmain = tm * BM;                      // every wave row is dense left of this
    int ke = (tm + 1) * BM;
    ke = ke < g.K ? ke : g.K;
    const int ke_wave = kmain + wm0 + 64;           // this wave row's last useful k (exclusive)

    const int srow = lane >> 3, schunk = lane & 7;
    const char *src[PPW];
    int ldsoff[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int piece = wave + NW * p;            // wave-uniform
        const bool isA = piece < BM / 8;
        const int prow = (isA ? piece : piece - BM / 8) * 8;
        const int row = prow + srow;
        const int src_chunk = schunk ^ ((row >> 1) & 7);
        const T *base = isA ? reinterpret_cast<const T *>(g.A) + ((long)tm * BM + row) * g.lda
                            : reinterpret_cast<const T *>(g.B) + ((long)tn * BN + row) * g.ldb;
        src[p] = reinterpret_cast<const char *>(base) + src_chunk * 16;
        ldsoff[p] = (isA ? 0 : A_BYTES) + prow * 128;
    }
    auto stage = [&](int buf, int k0) {
        const long koff = (long)k0 * (long)sizeof(T);
        char *base = smem_raw + buf * BUF_BYTES;
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src[p] + koff),
                                             (lds_void_t *)(base + __builtin_amdgcn_readfirstlane(ldsoff[p])), 16, 0, 0);
    };

    acc_t acc[NFM][NFN];
#pragma unroll
    for (int i = 0; i < NFM; ++i)
#pragma unroll
        for (int j = 0; j < NFN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) acc[i][j][r] = (T)0;

    const int fidx = MF::ab_idx(lane);
    const int grp = MF::ab_kg(lane);
    const int swz = (fidx >> 1) & 7;
    const int a_row_off = (wm0 + fidx) * 128;
    const int b_row_off = A_BYTES + (wn0 + fidx) * 128;

    auto compute = [&](int buf) {
        const char *base = smem_raw + buf * BUF_BYTES;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int coff = ((s * NG + grp) ^ swz) * 16;
            vec_t a[NFM], b[NFN];
#pragma unroll
            for (int i = 0; i < NFM; ++i)
                a[i] = *reinterpret_cast<const vec_t *>(base + a_row_off + i * MF::FM * 128 + coff);
#pragma unroll
            for (int j = 0; j < NFN; ++j)
                b[j] = *reinterpret_cast<const vec_t *>(base + b_row_off + j * MF::FN * 128 + coff);
#pragma unroll
            for (int e = 0; e < EPL; ++e)
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j) acc[i][j] = MF::mma(a[i][e], b[j][e], acc[i][j]);
        }
    };

    // the row tile that spans the whole k-range also accumulates the posterior mean's K*.alpha (MeanAcc,
    // mfma_gemm.hpp).  Two copies of the k-loop, so that every other workgroup runs the loop it always ran.
    const bool do_mean = g.mu != nullptr && ke >= g.K;
    const int mrow = tid & (BN - 1);
    const int mgrp = __builtin_amdgcn_readfirstlane(tid / BN);
    MeanAcc<T, BN, 64 * NW> macc;
    auto kloop = [&](auto mean_tag) {
        constexpr bool MEAN = decltype(mean_tag)::value;
        int k0 = 0;
        if (NBUF == 2) {
            int buf = 0;
            stage(0, 0);
            __syncthreads();
            for (; k0 < kmain; k0 += BK) {                  // dense part: no conditions
                typename MeanAcc<T, BN, 64 * NW>::Alpha mal;
                if constexpr (MEAN) mal.load(g.mu_alpha + k0, mgrp);
                stage(buf ^ 1, k0 + BK);                    // k0 + BK < ke always holds here
                compute(buf);
                if constexpr (MEAN) macc.add(smem_raw + buf * BUF_BYTES + A_BYTES, mal, mrow, mgrp);
                __syncthreads();
                buf ^= 1;
            }
            for (; k0 < ke; k0 += BK) {                     // diagonal tile: zero half skipped per wave row
                typename MeanAcc<T, BN, 64 * NW>::Alpha mal;
                if constexpr (MEAN) mal.load(g.mu_alpha + k0, mgrp);
                if (k0 + BK < ke) stage(buf ^ 1, k0 + BK);
                if (k0 < ke_wave) compute(buf);
                if constexpr (MEAN) macc.add(smem_raw + buf * BUF_BYTES + A_BYTES, mal, mrow, mgrp);
                __syncthreads();
                buf ^= 1;
            }
        } else {
            // Three LDS buffers, two k-tiles in flight.  The barrier that ends iteration i must only
            // wait for tile i+1 (issued one iteration ago), not for tile i+2 (just issued): a counted
            // s_waitcnt vmcnt(PPW) -- this wave's PPW newest DMA instructions may still be pending,
            // everything older has landed -- and the s_barrier in the same asm statement (a __syncthreads()
            // would drain to vmcnt(0)).  WAR: buffer (i+2)%3 == (i-1)%3 was last read in iteration i-1 and
            // every wave has passed that iteration's barrier.
            // (MEAN: alpha's k-tile is fetched BEFORE the trip's DMAs are issued, so the counted wait -- "all but
            // the PPW newest" -- covers it and the pipeline depth is what it was.)
            const int ntiles = ke / BK;                     // >= BM / BK >= 2
            stage(0, 0);
            stage(1, BK);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
            int buf = 0;
            for (int it = 0; it < ntiles; ++it, k0 += BK) {
                const bool more = it + 2 < ntiles;
                int nb = buf + 2; nb = nb >= 3 ? nb - 3 : nb;
                typename MeanAcc<T, BN, 64 * NW>::Alpha mal;
                if constexpr (MEAN) mal.load(g.mu_alpha + k0, mgrp);
                if (more) stage(nb, k0 + 2 * BK);
                if (k0 < ke_wave) compute(buf);
                if constexpr (MEAN) macc.add(smem_raw + buf * BUF_BYTES + A_BYTES, mal, mrow, mgrp);
                // (lgkmcnt(0): this trip's ds_reads have returned before the barrier lets the next trip's DMA into
                // the buffer they read -- see gemm64_glds.hpp, where the compiler's sinking of that wait below the
                // s_barrier produced wrong tiles under load)
                if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                buf = buf + 1; buf = buf >= 3 ? 0 : buf;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    if (do_mean) kloop(std::true_type{});
    else kloop(std::false_type{});

    // ---- per-column sum of squares over the tile's BM rows, f64, fixed order ---------------
    double *red = reinterpret_cast<double *>(smem_raw);   // [WM][BN]
    double cs[NFN];
#pragma unroll
    for (int j = 0; j < NFN; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) {
                const double v = (double)acc[i][j][r];
                s = fma(v, v, s);
            }
#pragma unroll
        for (int o = MF::COL_LANE_STRIDE; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        cs[j] = s;
    }
    if (lane < MF::COL_LANE_STRIDE) {
#pragma unroll
        for (int j = 0; j < NFN; ++j) red[(wave / WN) * BN + wn0 + j * MF::FN + lane] = cs[j];
    }
    __syncthreads();
    if (tid < BN) {
        // pairs of 64-row groups first: exactly what the 128-row kernel + finalize (which adds
        // its tiles two by two) produce, so the result does not depend on the tile variant
        static_assert(WM == 4, "row-group pairing is written for 4 wave rows");
        const double s = (red[0 * BN + tid] + red[1 * BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
        g.part[(long)tm * g.prm * g.ldpart + (long)tn * BN + tid] = s;
    }
    if (do_mean) {
        __syncthreads();   // `red` has been read
        macc.finish(reinterpret_cast<double *>(smem_raw), mrow, mgrp, g.mu + (long)tn * BN);
    }
}

template <int WM, int WN, int NBUF = 2>
constexpr size_t trmm_big_lds_bytes() { return (size_t)NBUF * (64 * WM + 64 * WN) * 128; }

}  // namespace tgp
