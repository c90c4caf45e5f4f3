// mfma_gemm.hpp -- LDS-tiled MFMA contraction for gfx950 (CDNA4), f64 and f32.
//
// One kernel template serves every dense contraction of the GP hot path:
//   * the candidate sweep  V = Linv * Kstar^T  with a fused sum-of-squares epilogue
//     (replaces solve_triangular + einsum at sklearn/gaussian_process/_gpr.py:454,475),
//   * the Cholesky panel solve, trailing update and the triangular inverse of the fit
//     (replaces LAPACK dpotrf behind scipy.linalg.cholesky at _gpr.py:349).
//
// Tiling is for 64-wide wavefronts: 256 threads = 4 waves in a 2x2 arrangement, each wave
// owning a (BM/2)x(BN/2) block of MFMA fragments
//     f64: v_mfma_f64_16x16x4_f64   (A/B one f64 per lane, C/D 4 f64 per lane)
//     f32: v_mfma_f32_32x32x2_f32   (A/B one f32 per lane, C/D 16 f32 per lane)
// Both operands are staged K-contiguous in LDS with a 16-byte row pad so every lane fetches its
// fragment elements with one ds_read_b128 per 8 k-values.  The k index is permuted between the
// lane groups (lane group g takes k = 8s + g*EPL + e); A and B use the same permutation, so the
// sum over k is complete and only its order differs from the natural one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgp {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;

template <> struct Mfma<double> {
    static constexpr int FM = 16, FN = 16;   // fragment rows / cols
    static constexpr int EPL = 2;            // elements per 16-byte LDS read
    static constexpr int NACC = 4;           // accumulator elements per lane
    typedef d4_t acc_t;
    typedef d2_t vec_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D map (cdna_hip_programming.md section 3): col = lane&15, row = (lane>>4) + 4*r
    static __device__ __forceinline__ int c_row(int lane, int r) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ int c_col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ab_idx(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ab_kg(int lane) { return lane >> 4; }
    static constexpr int COL_LANE_STRIDE = 16;   // lanes l, l+16, l+32, l+48 share a column
    static constexpr int COL_LANE_GROUPS = 4;
};

template <> struct Mfma<float> {
    static constexpr int FM = 32, FN = 32;
    static constexpr int EPL = 4;
    static constexpr int NACC = 16;
    typedef f16_t acc_t;
    typedef f4_t vec_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    static __device__ __forceinline__ int c_row(int lane, int r) {
        return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    static __device__ __forceinline__ int c_col(int lane) { return lane & 31; }
    static __device__ __forceinline__ int ab_idx(int lane) { return lane & 31; }
    static __device__ __forceinline__ int ab_kg(int lane) { return lane >> 5; }
    static constexpr int COL_LANE_STRIDE = 32;   // lanes l and l+32 share a column
    static constexpr int COL_LANE_GROUPS = 2;
};

// alternative f32 shape: v_mfma_f32_16x16x4_f32 (32-cycle issue, 4 accumulator registers)
struct MfmaF32x16 {
    static constexpr int FM = 16, FN = 16;
    static constexpr int EPL = 4;
    static constexpr int NACC = 4;
    typedef f4_t acc_t;
    typedef f4_t vec_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // col = lane&15, row = (lane>>4)*4 + r
    static __device__ __forceinline__ int c_row(int lane, int r) { return (lane >> 4) * 4 + r; }
    static __device__ __forceinline__ int c_col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ab_idx(int lane) { return lane & 15; }
    static __device__ __forceinline__ int ab_kg(int lane) { return lane >> 4; }
    static constexpr int COL_LANE_STRIDE = 16;
    static constexpr int COL_LANE_GROUPS = 4;
};

// k-range of a tile (elements, multiples of BK)
enum KRange { KR_FULL = 0,      // [0, K)
              KR_LOWER_A = 1,   // A rows are lower-triangular: [0, min(K, (tm+1)*BM))
              KR_LOWER_B = 2,   // B (k-major, NN) lower-triangular: [tn*BN, K)
              KR_UPPER_A = 3 }; // A rows are upper-triangular: [tm*BM, K)
// blockIdx.x -> (tm, tn)
enum TileMap { TM_FULL = 0,     // row-major over (ntm, ntn)
               TM_LOWER = 1,    // tm >= tn pairs of a square tile grid
               TM_SWEEP = 2 };  // heaviest-first over tm, XCD-grouped over tn
enum Epilogue { EP_STORE = 0, EP_SUMSQ = 1 };

struct GemmArgs {
    const void *A;  // (M, K) row-major, lda
    const void *B;  // B_KMAJOR ? (N, K) row-major (C = A * B^T) : (K, N) row-major (C = A * B)
    void *C;        // (M, N) row-major, ldc        [EP_STORE]
    void *Ct;       // optional (null): also store C^T here, (N, M) row-major, ldct  [EP_STORE]
    double *part;   // (ntm, ldpart) partial column sums of squares [EP_SUMSQ]
    long lda, ldb, ldc, ldpart, ldct;
    long strideA, strideB, strideC, strideCt;  // blockIdx.z batch strides in elements
    int ntm, ntn;   // tile counts
    int K;          // contraction length, multiple of BK
    double alpha, beta;   // EP_STORE: C = alpha*acc + beta*C  (beta is 0 or 1)
    long K_blocks;        // trmm_bf16x3.hpp: 16-k blocks per row of the pre-tiled operands
    int ntn_group;        // trmm_sweep.hpp: candidate tiles per group of one launch (0 = one group), see sweep_tile()
    // ---- the sweep's contraction (EP_SUMSQ) ----
    int tm0;              // first row tile of this launch (the launch covers row tiles [tm0, tm0 + ntm)); K stays the WHOLE
                          // contraction length, so a launch over a row range computes exactly what the full launch would there
    int prm;              // rows of `part` per row tile: 1 (128-row tiles) or 2 (256-row tiles write row 2 tm), see finalize_kernel
    const double *mu_alpha;  // (K,) f64 or null: with `mu`, the row tile that spans the whole k-range also accumulates
    double *mu;           // (ldpart,) mu[c] = sum_k B[c][k] * alpha[k] in f64 (the posterior mean before scaling), see MeanAcc
};

// blockIdx.x -> (tm, tn) of the sweep's contraction.  A launch covers ntn candidate tiles in GROUPS of
// ntn_group (the slab of one group is what the L2 / Infinity Cache re-serves while its row tiles pass
// over it); inside a group the row tiles go heaviest first (tm descending) and the candidate tiles are
// dealt to the XCDs in contiguous runs (workgroups b, b + 8, ... share an L2).  One launch for the
// whole batch instead of one per group: the light tail of a group is filled by the heavy head of the
// next, so only the last group leaves CUs idle.
__device__ __forceinline__ void sweep_tile(const GemmArgs &g, int bx, int &tm, int &tn) {
    int base = 0, ntn = g.ntn;
    if (g.ntn_group > 0 && g.ntn_group < g.ntn) {
        const int per_group = g.ntm * g.ntn_group;
        const int gi = bx / per_group;
        bx -= gi * per_group;
        base = gi * g.ntn_group;
        ntn = g.ntn - base < g.ntn_group ? g.ntn - base : g.ntn_group;   // (the last group may be short)
    }
    if ((ntn & 7) == 0) {
        const int xcd = bx & 7, q = bx >> 3;
        const int per = ntn >> 3;
        tn = base + xcd * per + (q % per);
        tm = g.tm0 + g.ntm - 1 - (q / per);
    } else {
        tm = g.tm0 + g.ntm - 1 - bx / ntn;
        tn = base + bx % ntn;
    }
}

// The posterior mean's dot product K*[c, :] . alpha inside the contraction (round 5): the workgroups of the row
// tile that spans the whole k-range see every k-tile of their candidates' slab rows pass through LDS anyway, and
// add it up there -- in f64, v_fma_f64 on the f32 / f64 slab values against alpha in f64 exactly as the
// cross-kernel did it before, so the cross-kernel needs nothing of the fit but Xs and can run INSIDE the fit.
// ONE summation order for every tile variant, launch split and schedule: a candidate's sum is kept as FOUR partial
// sums, partial g over the 16-byte chunks {2g, 2g + 1} of every 128-byte k-row (k-tiles ascending, elements
// ascending), combined at the end as (p0 + p1) + (p2 + p3).  Thread t owns candidate row t % BN of the B tile and
// the partials [GPT (t / BN), GPT (t / BN + 1)), GPT = 4 BN / NT (1: 256 x 256 / 256 x 128 tiles, 2: 128 x 128);
// t / BN is wave-uniform, so alpha's addresses are scalar.
template <typename T, int BN, int NT>
struct MeanAcc {
    static constexpr int GPT = 4 * BN / NT;              // partial sums per thread
    static constexpr int EPC = 16 / (int)sizeof(T);      // elements per 16-byte chunk
    static_assert(GPT == 1 || GPT == 2, "MeanAcc: one or two partial sums per thread");
    static_assert(BN % 64 == 0, "MeanAcc: whole waves per thread group");
    typedef T vec_t __attribute__((ext_vector_type(EPC)));
    double s[GPT];
    __device__ __forceinline__ MeanAcc() {
#pragma unroll
        for (int p = 0; p < GPT; ++p) s[p] = 0.0;
    }
    // alpha's share of one k-tile for this thread's partial sums: 2 GPT chunks of EPC doubles, wave-uniform addresses.
    // Fetched at the TOP of a k-loop trip (before the trip's DMAs are issued), consumed after its MFMAs.
    struct Alpha {
        double a[GPT][2 * EPC];
        // (through the CONSTANT address space: alpha was written by an earlier kernel and the addresses are
        // wave-uniform, so these become s_load_dwordx8 / x16 into SGPRs -- as vector loads they cost the 256 x 256
        // kernel 16 VGPRs it does not have: 128 + 5 spilled)
        __device__ __forceinline__ void load(const double *__restrict__ ak, int tgrp) {
            typedef const __attribute__((address_space(4))) double cdouble_t;
            cdouble_t *ck = (cdouble_t *)ak;
#pragma unroll
            for (int p = 0; p < GPT; ++p)
#pragma unroll
                for (int e = 0; e < 2 * EPC; ++e) a[p][e] = ck[(tgrp * GPT + p) * 2 * EPC + e];
        }
    };
    // btile: the B operand of one k-tile in LDS ([BN rows][128 B], chunk q of row r at q ^ ((r >> 1) & 7)); tgrp = t / BN
    __device__ __forceinline__ void add(const char *btile, const Alpha &al, int row, int tgrp) {
        const int swz = (row >> 1) & 7;
#pragma unroll
        for (int p = 0; p < GPT; ++p) {
            const int g = tgrp * GPT + p;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ch = 2 * g + q;
                const vec_t v = *reinterpret_cast<const vec_t *>(btile + row * 128 + ((ch ^ swz) << 4));
#pragma unroll
                for (int e = 0; e < EPC; ++e) s[p] = fma((double)v[e], al.a[p][q * EPC + e], s[p]);
            }
        }
    }
    // after the k-loop (every wave past its last LDS tile read): red = 4 * BN doubles of LDS
    __device__ __forceinline__ void finish(double *red, int row, int tgrp, double *__restrict__ mu_tile) {
#pragma unroll
        for (int p = 0; p < GPT; ++p) red[(tgrp * GPT + p) * BN + row] = s[p];
        __syncthreads();
        if ((int)threadIdx.x < BN) {
            const int c = (int)threadIdx.x;
            mu_tile[c] = (red[c] + red[BN + c]) + (red[2 * BN + c] + red[3 * BN + c]);
        }
    }
};

// (the second launch bound -- at least two waves per SIMD, i.e. at most 256 registers per lane -- is what makes the
// compiler issue the MFMAs in their VGPR form for the small tiles: with 512 registers on offer it parked the
// accumulators in AGPRs and moved all of them to VGPRs and back on EVERY trip of the k-loop, 64 v_accvgpr moves
// per 16 MFMAs in the 64 x 64 f64 instances.  The 128 x 128 instances need the AGPRs and keep one wave per SIMD.)
template <typename T, int BM, int BN, int BK, bool B_KMAJOR, int KR, int TMAP, int EP>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64 ? 2 : 1)) void mfma_gemm_kernel(GemmArgs g) {
    using MF = Mfma<T>;
    using vec_t = typename MF::vec_t;
    using acc_t = typename MF::acc_t;
    constexpr int EPL = MF::EPL;
    constexpr int PAD = EPL;                  // 16 bytes
    constexpr int LDK = BK + PAD;
    constexpr int WTM = BM / 2, WTN = BN / 2; // wave tile
    constexpr int NFM = WTM / MF::FM, NFN = WTN / MF::FN;
    constexpr int VPR = BK / EPL;             // 16-byte vectors per tile row
    constexpr int ROWS_PER_PASS = 256 / VPR;
    constexpr int A_PASSES = BM / ROWS_PER_PASS;
    constexpr int B_PASSES = BN / ROWS_PER_PASS;
    static_assert(BK % 8 == 0, "BK must hold whole 8-wide k steps");
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/pass mismatch");
    static_assert(WTM % MF::FM == 0 && WTN % MF::FN == 0, "wave tile/fragment mismatch");
    // k-major (NN) B staging: BK x BN elements, vectors along n
    constexpr int NVPR = BN / EPL;            // vectors per k-row
    constexpr int KROWS_PER_PASS = 256 / NVPR > 0 ? 256 / NVPR : 1;
    constexpr int BN_PASSES = B_KMAJOR ? 1 : (BK * NVPR + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *As = reinterpret_cast<T *>(smem_raw);             // [2][BM][LDK]
    T *Bs = As + 2 * BM * LDK;                           // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WTM;
    const int wn0 = (wave & 1) * WTN;

    // ---- tile mapping -------------------------------------------------------------------
    int tm, tn;
    {
        const int bx = blockIdx.x;
        if (TMAP == TM_FULL) {
            tm = bx / g.ntn;
            tn = bx - tm * g.ntn;
        } else if (TMAP == TM_LOWER) {
            int r = (int)((sqrtf(8.0f * (float)bx + 1.0f) - 1.0f) * 0.5f);
            while ((r + 1) * (r + 2) / 2 <= bx) ++r;
            while (r * (r + 1) / 2 > bx) --r;
            tm = r;
            tn = bx - r * (r + 1) / 2;
        } else {
            sweep_tile(g, bx, tm, tn);   // heaviest row tiles first, candidate tiles in contiguous runs per XCD
        }
    }
    const T *A = reinterpret_cast<const T *>(g.A) + (long)blockIdx.z * g.strideA;
    const T *B = reinterpret_cast<const T *>(g.B) + (long)blockIdx.z * g.strideB;

    int kb = 0, ke = g.K;
    if (KR == KR_LOWER_A) { int lim = (tm + 1) * BM; ke = lim < g.K ? lim : g.K; }
    if (KR == KR_LOWER_B) { kb = tn * BN; }   // BN is a multiple of BK in every instantiation
    if (KR == KR_UPPER_A) { kb = tm * BM; }

    // ---- global -> register staging ------------------------------------------------------
    vec_t ra[A_PASSES];
    vec_t rb[B_KMAJOR ? B_PASSES : BN_PASSES];
    const int vcol = tid % VPR, vrow = tid / VPR;
    const T *Ag = A + ((long)tm * BM + vrow) * g.lda + vcol * EPL;
    const T *Bg;
    int bn_k = 0, bn_n = 0;
    if (B_KMAJOR) {
        Bg = B + ((long)tn * BN + vrow) * g.ldb + vcol * EPL;
    } else {
        bn_k = tid / NVPR;
        bn_n = (tid % NVPR) * EPL;
        Bg = B + (long)bn_k * g.ldb + (long)tn * BN + bn_n;
    }

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            ra[p] = *reinterpret_cast<const vec_t *>(Ag + (long)p * ROWS_PER_PASS * g.lda + k0);
        if (B_KMAJOR) {
#pragma unroll
            for (int p = 0; p < B_PASSES; ++p)
                rb[p] = *reinterpret_cast<const vec_t *>(Bg + (long)p * ROWS_PER_PASS * g.ldb + k0);
        } else {
#pragma unroll
            for (int p = 0; p < BN_PASSES; ++p)
                rb[p] = *reinterpret_cast<const vec_t *>(Bg + ((long)k0 + p * KROWS_PER_PASS) * g.ldb);
        }
    };
    auto store_tiles = [&](int buf) {
        T *as = As + buf * BM * LDK;
        T *bs = Bs + buf * BN * LDK;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            *reinterpret_cast<vec_t *>(as + (vrow + p * ROWS_PER_PASS) * LDK + vcol * EPL) = ra[p];
        if (B_KMAJOR) {
#pragma unroll
            for (int p = 0; p < B_PASSES; ++p)
                *reinterpret_cast<vec_t *>(bs + (vrow + p * ROWS_PER_PASS) * LDK + vcol * EPL) = rb[p];
        } else {
#pragma unroll
            for (int p = 0; p < BN_PASSES; ++p) {
                const int k = bn_k + p * KROWS_PER_PASS;
#pragma unroll
                for (int e = 0; e < EPL; ++e) bs[(bn_n + e) * LDK + k] = rb[p][e];
            }
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
    const int fkg = MF::ab_kg(lane) * EPL;

    // Software pipeline, prefetch distance 2 with one register set: while tile t is consumed from
    // LDS buffer `buf`, tile t+1 (fetched during the previous iteration) moves from registers to
    // the other buffer behind the first k-step's MFMAs, and the registers are refilled with tile
    // t+2 straight away.  After the barrier a wave therefore only waits for its ds_reads.
    int buf = 0;
    if (kb < ke) {
        load_tiles(kb);
        store_tiles(0);
        if (kb + BK < ke) load_tiles(kb + BK);
    }
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += BK) {
        const bool has1 = (k0 + BK) < ke;
        const bool has2 = (k0 + 2 * BK) < ke;
        const T *as = As + buf * BM * LDK + (wm0 + fidx) * LDK + fkg;
        const T *bs = Bs + buf * BN * LDK + (wn0 + fidx) * LDK + fkg;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 8) {
            vec_t a[NFM], b[NFN];
#pragma unroll
            for (int i = 0; i < NFM; ++i)
                a[i] = *reinterpret_cast<const vec_t *>(as + i * MF::FM * LDK + ks);
#pragma unroll
            for (int j = 0; j < NFN; ++j)
                b[j] = *reinterpret_cast<const vec_t *>(bs + j * MF::FN * LDK + ks);
#pragma unroll
            for (int e = 0; e < EPL; ++e)
#pragma unroll
                for (int i = 0; i < NFM; ++i)
#pragma unroll
                    for (int j = 0; j < NFN; ++j) acc[i][j] = MF::mma(a[i][e], b[j][e], acc[i][j]);
            if (ks == 0) {
                if (has1) store_tiles(buf ^ 1);
                if (has2) load_tiles(k0 + 2 * BK);
            }
        }
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue ------------------------------------------------------------------------
    if (EP == EP_STORE) {
        T *C = reinterpret_cast<T *>(g.C) + (long)blockIdx.z * g.strideC;
        T *Ct = g.Ct ? reinterpret_cast<T *>(g.Ct) + (long)blockIdx.z * g.strideCt : nullptr;
        const T alpha = (T)g.alpha;
        const bool use_beta = g.beta != 0.0;
#pragma unroll
        for (int i = 0; i < NFM; ++i)
#pragma unroll
            for (int j = 0; j < NFN; ++j) {
                const long col = (long)tn * BN + wn0 + j * MF::FN + MF::c_col(lane);
#pragma unroll
                for (int r = 0; r < MF::NACC; ++r) {
                    const long row = (long)tm * BM + wm0 + i * MF::FM + MF::c_row(lane, r);
                    T v = alpha * acc[i][j][r];
                    T *p = C + row * g.ldc + col;
                    if (use_beta) v += *p;
                    *p = v;
                    if (Ct) Ct[col * g.ldct + row] = v;
                }
            }
    } else {
        // per-column sum of squares over this tile's BM rows, in f64, fixed order
        double *red = reinterpret_cast<double *>(smem_raw);   // [2][BN] after the k loop
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
            // lanes sharing a column
#pragma unroll
            for (int o = MF::COL_LANE_STRIDE; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
            cs[j] = s;
        }
        __syncthreads();   // all waves are past their last LDS tile read
        if (lane < MF::COL_LANE_STRIDE) {
#pragma unroll
            for (int j = 0; j < NFN; ++j)
                red[(wave >> 1) * BN + wn0 + j * MF::FN + lane] = cs[j];
        }
        __syncthreads();
        if (tid < BN)
            g.part[(long)tm * g.ldpart + (long)tn * BN + tid] = red[tid] + red[BN + tid];
    }
}

template <typename T, int BM, int BN, int BK>
constexpr size_t gemm_lds_bytes() {
    return (size_t)2 * (BM + BN) * (BK + Mfma<T>::EPL) * sizeof(T);
}

}  // namespace tgp
