// gemm_nt_glds.hpp -- general C = alpha * A * B^T (+ beta * C) on gfx950 with both operands
// K-contiguous ("NT"), f64 or f32, using the same machinery as the sweep kernel
// (trmm_sweep.hpp): 128x128 tile, 4 waves of 64x64 MFMA fragments, operands DMA'd
// global -> LDS (global_load_lds_dwordx4) into a lane-linear, XOR-swizzled image, two LDS
// buffers, two workgroups per CU.  Used by the fit: the rank-256 trailing update of the blocked
// Cholesky and the merges of the triangular inverse (which keeps Linv and its transpose so
// every product is NT).
//
// Requirements: M, N multiples of 128; K and every k-range bound multiples of 128 bytes / sizeof(T);
// lda, ldb multiples of 16 bytes; C row-major.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_opt_in.hpp"
#include "mfma_gemm.hpp"
#include "trmm_sweep.hpp"

namespace tgp {

enum KRangeNt { KN_FULL = 0,      // [0, K)
                KN_LOWER_A = 1,   // A lower-triangular rows: [0, min(K, (tm+1)*128))
                KN_UPPER_A = 2 }; // A upper-triangular rows: [tm*128, K)

struct GemmNtArgs {
    const void *A; const void *B; void *C;
    void *Ct;                 // optional: also store C^T here (row-major, ldct), may be null
    long lda, ldb, ldc, ldct;
    long strideA, strideB, strideC, strideCt;   // blockIdx.z batch strides (elements)
    int ntm, ntn;
    int K;
    double alpha, beta;       // beta is 0 or 1
};

template <typename T, int KR, int TMAP>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(GemmNtArgs g) {
    using MF = Mfma<T>;
    using vec_t = typename MF::vec_t;
    using acc_t = typename MF::acc_t;
    constexpr int EPL = MF::EPL;
    constexpr int BM = 128, BN = 128;
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int WTM = 64, WTN = 64;
    constexpr int NFM = WTM / MF::FM, NFN = WTN / MF::FN;
    constexpr int NG = 64 / MF::FM;
    constexpr int KSTEPS = 8 / NG;
    constexpr int TILE_BYTES = BM * 128;
    constexpr int BUF_BYTES = 2 * TILE_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * WTM;
    const int wn0 = (wave & 1) * WTN;

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
    if (KR == KN_LOWER_A) { const int lim = (tm + 1) * BM; ke = lim < g.K ? lim : g.K; }
    if (KR == KN_UPPER_A) { kb = tm * BM; }

    const T *A = reinterpret_cast<const T *>(g.A) + (long)blockIdx.z * g.strideA;
    const T *B = reinterpret_cast<const T *>(g.B) + (long)blockIdx.z * g.strideB;

    const int srow = lane >> 3, schunk = lane & 7;
    const char *asrc[4];
    const char *bsrc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (4 * p + wave) * 8 + srow;
        const int src_chunk = schunk ^ ((row >> 1) & 7);
        asrc[p] = reinterpret_cast<const char *>(A + ((long)tm * BM + row) * g.lda) + src_chunk * 16;
        bsrc[p] = reinterpret_cast<const char *>(B + ((long)tn * BN + row) * g.ldb) + src_chunk * 16;
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
    const int swz = (fidx >> 1) & 7;
    const int a_row_off = (wm0 + fidx) * 128;
    const int b_row_off = TILE_BYTES + (wn0 + fidx) * 128;

    int buf = 0;
    if (kb < ke) stage(0, kb);
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += BK) {
        if (k0 + BK < ke) stage(buf ^ 1, k0 + BK);
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
        __syncthreads();
        buf ^= 1;
    }

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
}

template <typename T, int KR, int TMAP>
static hipError_t launch_gemm_nt_glds(hipStream_t s, int device, const GemmNtArgs &g, int nblocks, int batch) {
    auto kern = gemm_nt_glds_kernel<T, KR, TMAP>;
    constexpr size_t lds = trmm_glds_lds_bytes();
    static LdsOptIn opt_in;
    if (hipError_t e = opt_in.ensure(reinterpret_cast<const void *>(kern), device, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, batch), dim3(256), lds, s, g);
    return hipGetLastError();
}

}  // namespace tgp
