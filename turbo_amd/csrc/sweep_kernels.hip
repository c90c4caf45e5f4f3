// sweep_kernels.hip -- posterior mean / variance and acquisition over a resident candidate
// batch, in chunks of `chunk` candidates:
//
//   prep     Cs = Xc / length_scale                       (sklearn kernels.py:1562 / 1715)
//   kstar    Ks = c * k(Cs, Xs), mu partials = Ks . alpha  (_gpr.py:443-444)
//   trmm     q_part = column sums of squares of Linv * Ks^T  (replaces _gpr.py:454 + :475;
//            mfma_gemm.hpp, EP_SUMSQ) -- the dominant, MFMA-bound kernel
//   finalize var = (c + noise) - sum q_part, clamp, sigma, mu de-normalise (_gpr.py:446-447,
//            474-494); UCB / PI / EI (turbo/modules/acquisition_functions.py:147-158, 225-247,
//            336-358); per-block arg-max
//   argmax   final (value, lowest index) (turbo/modules/auxiliary_optimisers.py:63-66)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mfma_gemm.hpp"
#include "pairwise.hpp"
#include "tgp_internal.hpp"
#include "trmm_sweep.hpp"
#include "trmm_bf16x3.hpp"
#include "trmm_f16x2.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

template <typename T>
__global__ __launch_bounds__(256) void prep_candidates_kernel(const double *__restrict__ Xc,
                                                              const double *__restrict__ ls,
                                                              T *__restrict__ Cs, long m_valid,
                                                              long rows, int D, int Dp) {
    // Cs is (rows, Dp): padded rows and padded dimensions are zero
    const long total = rows * Dp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / Dp;
        const int d = (int)(i - r * Dp);
        Cs[i] = (r < m_valid && d < D) ? (T)(Xc[r * D + d] / ls[d]) : (T)0;
    }
}

// ---- device-side uniform candidates ---------------------------------------------
// Philox-4x32-10, counter = global element index, key = seed.  One call of the generator yields
// two doubles with numpy's 53-bit construction ((a >> 5) * 2^26 + (b >> 6)) / 2^53; element e of
// the (M, D) batch takes draw e >> 1, half e & 1, so the stream does not depend on how the batch
// is sharded over GPUs.  x = lo + (hi - lo) * u, as numpy.random.uniform
// (turbo/modules/naive_selectors.py:39-46 draws column-wise from the global NumPy RNG instead).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void gen_candidates_kernel(double *__restrict__ Xc, long total,
                                                             unsigned long long first, int D,
                                                             unsigned long long seed,
                                                             const double *__restrict__ lo,
                                                             const double *__restrict__ hi) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned long long e = first + (unsigned long long)i;
        const unsigned long long draw = e >> 1;
        uint32_t r[4];
        philox4x32_10((uint32_t)draw, (uint32_t)(draw >> 32), 0u, 0u, (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        const uint32_t a = (e & 1) ? r[2] : r[0], b = (e & 1) ? r[3] : r[1];
        const double u = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
        const int d = (int)(i % D);
        double t = (hi[d] - lo[d]) * u;
        asm volatile("" : "+v"(t));   // two roundings, as numpy: keep the compiler from fusing into an fma
        Xc[i] = lo[d] + t;
    }
}

// ---- device-side Latin hypercube design (turbo/modules/naive_selectors.py:58-83) -----------------
// Sample i of an n-point design, dimension d:  lo_d + (hi_d - lo_d) * ((pi_d(i) + u_id) / n), where
// pi_d is a pseudo-random permutation of 0..n-1 per dimension (the reference shuffles each column
// with np.random.permutation) and u_id is uniform in [0, 1) (its np.random.rand).  pi_d is a
// 4-round Feistel network on ceil(log2 n) bits (rounded up to an even count) whose round function
// is Philox-4x32-10 keyed by the seed, cycle-walked into [0, n): every sample is computed
// independently of all others, so any shard of a design equals the same rows of the whole design.
__device__ __forceinline__ unsigned long long lhs_perm(unsigned long long i, unsigned long long n, int half,
                                                       uint32_t d, uint32_t k0, uint32_t k1) {
    const unsigned long long mask = (1ull << half) - 1ull;
    unsigned long long x = i;
    do {
        uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
#pragma unroll 1
        for (uint32_t r = 0; r < 4; ++r) {
            uint32_t o[4];
            philox4x32_10(R, r, d, 0x4C485321u, k0, k1, o);          // counter words: (right half, round, dimension, "LHS!")
            const uint32_t F = o[0] & (uint32_t)mask;
            const uint32_t nl = R;
            R = L ^ F;
            L = nl;
        }
        x = ((unsigned long long)L << half) | (unsigned long long)R;
    } while (x >= n);
    return x;
}

__global__ __launch_bounds__(256) void gen_lhs_kernel(double *__restrict__ Xc, long total,
                                                      unsigned long long first, int D,
                                                      unsigned long long n, int half,
                                                      unsigned long long seed,
                                                      const double *__restrict__ lo,
                                                      const double *__restrict__ hi) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned long long e = first + (unsigned long long)i;       // element index of the whole design
        const int d = (int)(e % (unsigned long long)D);
        const unsigned long long smp = e / (unsigned long long)D;
        const unsigned long long pi = lhs_perm(smp, n, half, (uint32_t)d, (uint32_t)seed, (uint32_t)(seed >> 32));
        const unsigned long long draw = e >> 1;
        uint32_t r[4];
        philox4x32_10((uint32_t)draw, (uint32_t)(draw >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);   // stream 1: the jitter
        const uint32_t a = (e & 1) ? r[2] : r[0], b = (e & 1) ? r[3] : r[1];
        const double u = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
        double t = (double)pi + u;
        asm volatile("" : "+v"(t));     // every step rounded separately, as the NumPy restatement does
        t = t / (double)n;
        asm volatile("" : "+v"(t));
        double w = (hi[d] - lo[d]) * t;
        asm volatile("" : "+v"(w));
        Xc[i] = lo[d] + w;
    }
}

hipError_t launch_gen_lhs(Context &c, double *dst, int64_t M, int64_t D, unsigned long long seed,
                          unsigned long long first_sample, unsigned long long n_total,
                          const double *d_lo, const double *d_hi) {
    int bits = 2;
    while ((1ull << bits) < n_total) ++bits;
    if (bits & 1) ++bits;
    const long total = (long)M * D;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gen_lhs_kernel, dim3(blocks), dim3(256), 0, c.stream, dst, total,
                       first_sample * (unsigned long long)D, (int)D, n_total, bits / 2, seed, d_lo, d_hi);
    return hipGetLastError();
}

hipError_t launch_gen_candidates(Context &c, double *dst, int64_t M, unsigned long long seed,
                                 unsigned long long first_candidate, const double *d_lo,
                                 const double *d_hi) {
    const long total = (long)M * c.D;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gen_candidates_kernel, dim3(blocks), dim3(256), 0, c.stream, dst, total,
                       first_candidate * (unsigned long long)c.D, (int)c.D, seed, d_lo, d_hi);
    return hipGetLastError();
}

// The reference's HOST candidate draw finished on the GPU (tgp_set_candidates_mt19937): `words` holds, column after
// column, the 2 M tempered outputs of NumPy's MT19937 stream that column's M draws consume; candidate i of column c is
//     lo[c] + range[c] * (((w[2 i] >> 5) * 2^26 + (w[2 i + 1] >> 6)) / 2^53)
// -- legacy random_sample and random_uniform's  lower + range * u  in two roundings (numpy/random/src/distributions,
// mtrand.pyx uniform) -- written (M, D) row-major.  A workgroup takes 64 rows x 32 columns: the words are read along
// the rows of a column (512 contiguous bytes per wave), the doubles leave through LDS along the columns of a row.
__global__ __launch_bounds__(256) void mt19937_columns_kernel(const uint2 *__restrict__ words, double *__restrict__ out,
                                                              long M, int D, const double *__restrict__ lo,
                                                              const double *__restrict__ range) {
#pragma clang fp contract(off)
    __shared__ double tile[64][33];
    const long i0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int cc = q + 4 * k, c = c0 + cc;
        if (c < D && i0 + r < M) {
            const uint2 w = words[(long)c * M + i0 + r];
            const double u = ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) / 9007199254740992.0;
            const double prod = range[c] * u;
            tile[r][cc] = lo[c] + prod;
        }
    }
    __syncthreads();
    const int cc = threadIdx.x & 31, rr = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int row = rr + 8 * k;
        if (c0 + cc < D && i0 + row < M) out[(i0 + row) * D + c0 + cc] = tile[row][cc];
    }
}

hipError_t launch_mt19937_columns(Context &c, const void *d_words, double *dst, int64_t M, const double *d_lo,
                                  const double *d_range) {
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((c.D + 31) / 32));
    hipLaunchKernelGGL(mt19937_columns_kernel, grid, dim3(256), 0, c.stream, reinterpret_cast<const uint2 *>(d_words), dst,
                       (long)M, (int)c.D, d_lo, d_range);
    return hipGetLastError();
}

// Cross-kernel slab.  A workgroup owns 16 * AR candidates and walks 128-point training tiles with
// the next tile's points prefetched into registers while the current one is reduced from LDS
// (point blocks staged transposed, [dim][point]).  Thread (tx, ty) of a 16 x 16 grid owns
// candidates {4ty..4ty+3 (, 64+4ty..)} against training points {4tx..4tx+3, 64+4tx..}: an
// AR x 8 micro-tile (8 x 8 for f32, 4 x 8 for f64 -- 64 f64 accumulators do not fit), so a staged
// dimension costs (AR + 8) / 4 16-byte LDS reads per 8 * AR (difference, fma) pairs, and operand
// reads and slab stores are 256 contiguous bytes per 16 lanes.  Dimensions are summed in order
// with the DIRECT difference (pairwise.hpp says why).  grid = (rows / (16 AR), splits); split y
// walks the training tiles [y * per, (y + 1) * per) and owns row y of the mean partials.
// Ablations on the GPU (C3, f32): no slab stores -4 us, no LDS reads -10 us, no distance loop
// -32 us of 64 us: VALU-bound (2048 packed issue slots in the loop + ~900 in the epilogue per
// 128 x 128 tile).
// f32 staging rows: 140 floats, and the rows of dimensions 8 g .. 8 g + 7 start 4 g floats further
// right (rot): the transposing stores of each half wave (4 points x 8 four-dimension vectors) then
// fall on 32 distinct banks, where rows of 132 floats put four lanes on each (1.97 M conflict
// cycles per launch in round 1's counters; 0.66 M with a two-step shift, which this model also predicts).
template <typename T> struct KsStage {
    static constexpr int DC = PwCfg<T>::DC, VEC = PwCfg<T>::VEC;
    static constexpr int LD = 128 + (sizeof(T) == 4 ? 12 : 2);
    // (f64: rows of 130 doubles, dimensions 8..15 two doubles further right)
    __device__ static __forceinline__ int rot(int d) { return sizeof(T) == 4 ? ((d >> 3) << 2) : ((d >> 3) << 1); }
    static constexpr int VPP = DC / VEC;              // 16-byte vectors per point per pass (8)
    static constexpr int PASSES = 128 * VPP / 256;    // 4
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    vec_t v[PASSES];
    __device__ __forceinline__ void load(const T *__restrict__ M, int r0, int n, int ld, int d0) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = (int)threadIdx.x + 256 * p;
            const int r = idx / VPP, dv = (idx % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[p][e] = (T)0;
            if ((d0 + dv) < ld && (r0 + r) < n)
                v[p] = *reinterpret_cast<const vec_t *>(M + (long)(r0 + r) * ld + d0 + dv);
        }
    }
    __device__ __forceinline__ void store(T (*S)[LD]) const {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = (int)threadIdx.x + 256 * p;
            const int r = idx / VPP, dv = (idx % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) S[dv + e][r + rot(dv)] = v[p][e];   // (dv is a multiple of VEC: one rot per vector)
        }
    }
};

// X3 (f32 only): the slab is written as the pre-tiled three-plane operand of trmm_bf16x3.hpp;
// pstride = its 16-k blocks per row (Np / 16)
// XP == 2: as two scaled fp16 planes (trmm_f16x2.hpp), values multiplied by xscale first
// MU: also accumulate the partial means Ks . alpha (the split-operand dtypes and the register-staged A/B contraction;
// round 5: the f64 / f32 contraction does it itself -- MeanAcc in mfma_gemm.hpp -- so that this kernel needs nothing
// of a fit but Xs and can run inside it)
template <typename T, int KIND, int AR, int XP = 0, bool MU = true>
__global__ __launch_bounds__(256, 2) void kstar_kernel(const T *__restrict__ Cs,
                                                          const T *__restrict__ Xs,
                                                          const double *__restrict__ alpha,
                                                          T *__restrict__ Ks,
                                                          double *__restrict__ mupart, int rows,
                                                          int N, int Np, int Dp, double constant,
                                                          long ldpart, long pstride = 0, float xscale = 1.f) {
    constexpr bool X3 = XP == 3;
    using St = KsStage<T>;
    constexpr int DC = St::DC, LD = St::LD;
    __shared__ __attribute__((aligned(16))) T Ct[DC][LD];
    __shared__ __attribute__((aligned(16))) T Xt[DC][LD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    constexpr int CT = 16 * AR;                       // candidates per tile: 128 (AR = 8) or 64 (AR = 4)
    const int c0 = blockIdx.x * CT;
    const int njt = Np / 128;
    const int per = (njt + (int)gridDim.y - 1) / (int)gridDim.y;
    const int jt0 = blockIdx.y * per;
    int jt_end = jt0 + per;
    if (jt_end > njt) jt_end = njt;
    const int jt_real = (N + 127) / 128;
    const int jt_live = jt_end < jt_real ? jt_end : jt_real;
    // candidate row / training column of micro-tile index 0..7
    auto crow = [&](int a) { return (a < 4 ? 0 : 60) + 4 * ty + a; };   // AR = 4: 4ty + a
    // f32: {4tx .. 4tx+3, 64+4tx ..}: a lane's 16-byte LDS reads follow its neighbour's.  f64: pairs
    // {2tx, 2tx+1} + 32 m for the same reason (four doubles per lane put lanes tx and tx + 4 on the
    // same banks: 4.2 M conflict cycles per launch at C2)
    auto jcol = [&](int b) { return sizeof(T) == 4 ? (b < 4 ? 0 : 60) + 4 * tx + b : ((b >> 1) << 5) + 2 * tx + (b & 1); };
    // tiles that hold only padding: zero-filled
    for (int jt = (jt_live > jt0 ? jt_live : jt0); jt < jt_end; ++jt) {
#pragma unroll
        for (int a = 0; a < AR; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (XP == 2) {
                    const long row = c0 + crow(a), k = jt * 128 + jcol(b);
                    char *o = reinterpret_cast<char *>(Ks);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        *reinterpret_cast<unsigned short *>(o + h2_block_off(row, k >> 4, pl, pstride) +
                                                            x3_chunk_off((int)(row & 31), (int)((k >> 3) & 1)) + (k & 7) * 2) = 0;
                } else if (X3) {   // pre-tiled three-plane slab (trmm_bf16x3.hpp); pstride = 16-k blocks per row
                    const long row = c0 + crow(a), k = jt * 128 + jcol(b);
                    char *o = reinterpret_cast<char *>(Ks);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        *reinterpret_cast<unsigned short *>(o + x3_block_off(row, k >> 4, pl, pstride) +
                                                            x3_chunk_off((int)(row & 31), (int)((k >> 3) & 1)) + (k & 7) * 2) = 0;
                } else {
                    Ks[(long)(c0 + crow(a)) * Np + jt * 128 + jcol(b)] = (T)0;
                }
            }
    }
    double pm[AR];
#pragma unroll
    for (int a = 0; a < AR; ++a) pm[a] = 0.0;
    const T cst = (T)constant;
    const float log2c = log2f((float)constant);
    const int nch = (Dp + DC - 1) / DC;
    const int nsteps = (jt_live > jt0 ? jt_live - jt0 : 0) * nch;
    const bool one_pass = (nch == 1);

    St sp, sq;
    T d2[AR][8];
    if (nsteps > 0) {
        sp.load(Cs, c0, c0 + CT < rows ? c0 + CT : rows, Dp, 0);
        sq.load(Xs, jt0 * 128, Np, Dp, 0);
        sp.store(Ct);
        sq.store(Xt);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int jt = jt0 + st / nch, ch = st - (st / nch) * nch;
        const int j0 = jt * 128;
        const bool more = (st + 1) < nsteps;
        if (more) {
            const int jn = jt0 + (st + 1) / nch, cn = (st + 1) - ((st + 1) / nch) * nch;
            if (!one_pass) sp.load(Cs, c0, c0 + CT < rows ? c0 + CT : rows, Dp, cn * DC);
            sq.load(Xs, jn * 128, Np, Dp, cn * DC);
        }
        if (ch == 0) {
#pragma unroll
            for (int a = 0; a < AR; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) d2[a][b] = (T)0;
        }
        {
            int dn = Dp - ch * DC;
            if (dn > DC) dn = DC;                      // Dp is a multiple of 4
            // (round 5: unrolling this loop twice, and fetching one dimension ahead into a second register set, measured
            // equal / 9 % slower -- 0.369 / 0.400 against 0.366 ms per C3 launch: the compiler's schedule already overlaps the
            // LDS reads with the packed arithmetic)
#pragma unroll 1
            for (int d4 = 0; d4 < dn; d4 += 4) {
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {   // Dp is a multiple of 4
                    const int d = d4 + dd;
                    T cv[AR], xv[8];
#pragma unroll
                    for (int a = 0; a < AR; ++a) cv[a] = Ct[d][crow(a) + St::rot(d4)];
#pragma unroll
                    for (int b = 0; b < 8; ++b) xv[b] = Xt[d][jcol(b) + St::rot(d4)];
#pragma unroll
                    for (int a = 0; a < AR; ++a)
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const T df = cv[a] - xv[b];
                            d2[a][b] = fma(df, df, d2[a][b]);
                        }
                }
            }
        }
        if (ch == nch - 1) {
            double al[8];
            if (MU) {
#pragma unroll
                for (int b = 0; b < 8; ++b) al[b] = alpha[j0 + jcol(b)];
            }
            const bool edge = j0 + 128 > N;              // only the last real tile holds padding columns
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                T *dst = Ks + (long)(c0 + crow(a)) * Np + j0;
                // two groups of four columns: fewer values live at once
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    T kv[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (KIND == TGP_RBF && sizeof(T) == 4) {
                            // c * exp(-d2 / 2) = 2^(d2 * (-log2(e) / 2) + log2(c)): one fma + v_exp_f32
                            kv[b] = (T)__builtin_amdgcn_exp2f(fmaf((float)d2[a][4 * hb + b], -0.72134752044448170368f, log2c));
                        } else {
                            kv[b] = kernel_value<T, KIND>(d2[a][4 * hb + b], cst);
                        }
                    }
                    if (edge) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) kv[b] = (j0 + jcol(4 * hb + b) < N) ? kv[b] : (T)0;
                    }
                    if (MU) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) pm[a] = fma((double)kv[b], al[4 * hb + b], pm[a]);
                    }
                    if (XP == 2) {
                        unsigned pk[2][2];
                        split2_f16x2((float)kv[0], (float)kv[1], xscale, pk[0][0], pk[1][0]);
                        split2_f16x2((float)kv[2], (float)kv[3], xscale, pk[0][1], pk[1][1]);
                        const long row = c0 + crow(a), k = j0 + 4 * tx + 64 * hb;
                        char *o = reinterpret_cast<char *>(Ks) + x3_chunk_off((int)(row & 31), (int)((k >> 3) & 1)) + (k & 7) * 2;
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            uint2 w;
                            w.x = pk[pl][0];
                            w.y = pk[pl][1];
                            *reinterpret_cast<uint2 *>(o + h2_block_off(row, k >> 4, pl, pstride)) = w;
                        }
                    } else if (X3) {
                        unsigned pk[3][2];
                        split2_bf16x3((float)kv[0], (float)kv[1], pk[0][0], pk[1][0], pk[2][0]);
                        split2_bf16x3((float)kv[2], (float)kv[3], pk[0][1], pk[1][1], pk[2][1]);
                        const long row = c0 + crow(a), k = j0 + 4 * tx + 64 * hb;   // four consecutive k inside one 8-k chunk
                        char *o = reinterpret_cast<char *>(Ks) + x3_chunk_off((int)(row & 31), (int)((k >> 3) & 1)) + (k & 7) * 2;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            uint2 w;
                            w.x = pk[pl][0];
                            w.y = pk[pl][1];
                            *reinterpret_cast<uint2 *>(o + x3_block_off(row, k >> 4, pl, pstride)) = w;
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b) dst[jcol(4 * hb + b)] = kv[b];
                    }
                }
            }
        }
        if (more) {
            __syncthreads();
            if (!one_pass) sp.store(Ct);
            sq.store(Xt);
            __syncthreads();
        }
    }
    if (MU) {
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            double s = pm[a];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (tx == 0) mupart[(long)blockIdx.y * ldpart + c0 + crow(a)] = s;
        }
    }
}

__device__ __forceinline__ double ndtr_dev(double a) {
    // scipy.special.ndtr (cephes ndtr.c) behind scipy.stats.norm.cdf
    // (scipy/stats/_continuous_distns.py:368-369)
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    double y;
    if (z < 0.70710678118654752440) {
        y = 0.5 + 0.5 * erf(x);
    } else {
        y = 0.5 * erfc(z);
        if (x > 0) y = 1.0 - y;
    }
    return y;
}

struct FinArgs {
    const double *part; long ldpart;
    // `part` has one row per 128 rows of Linv.  The 128-row-tile kernels fill every row; the 256-row-tile kernels
    // leave their (already pair-summed) value in row 2 t.  The sum over the row tiles goes by 256-row units t:
    // q += part[2t] + (unit t came from two 128-row tiles ? part[2t + 1] : 0) -- the same additions whichever kernel
    // produced a unit (trmm_sweep.hpp), so a sweep may mix them: candidates below pre_m had their first pre_pairs
    // units contracted by the 128-row-tile kernel inside the fit (tgp_set_overlap), whatever the rest used.
    int nunits;                // 256-row units holding real rows
    int n128;                  // 128-row tiles holding real rows (the second half of the last unit may not exist)
    int base_pairs;            // units [0, base_pairs) are two 128-row tiles for every candidate (nunits: the whole sweep on 128-row tiles; 0: none)
    long pre_m; int pre_pairs; // ... and units [0, pre_pairs) for the candidates [0, pre_m)
    const double *mupart; int njs;
    long off, m;               // global offset of this chunk, valid candidates in it
    double kss;                // constant + noise (kernel_.diag)
    double y_mean, y_std;
    int acq; double sf, incumbent, param;
    double *mu, *sigma, *acqv; // nullable (M,) outputs
    double *bval; long long *bidx; long long *counters;   // counters[1] += clamped
};

__global__ __launch_bounds__(FIN_BLOCK) void finalize_kernel(FinArgs f) {
    __shared__ double sv[FIN_BLOCK];
    __shared__ long long si[FIN_BLOCK];
    __shared__ int sclamp;
    const int tid = threadIdx.x;
    if (tid == 0) sclamp = 0;
    __syncthreads();
    const long c = (long)blockIdx.x * FIN_BLOCK + tid;
    double best = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    if (c < f.m) {
        double q = 0.0;
        const int npair = (c < f.pre_m && f.pre_pairs > f.base_pairs) ? f.pre_pairs : f.base_pairs;
        for (int t = 0; t < f.nunits; ++t) {
            const double a = f.part[(long)(2 * t) * f.ldpart + c];
            const double b = (t < npair && 2 * t + 1 < f.n128) ? f.part[(long)(2 * t + 1) * f.ldpart + c] : 0.0;
            q += a + b;
        }
        double mun = 0.0;
        for (int s = 0; s < f.njs; ++s) mun += f.mupart[(long)s * f.ldpart + c];
        double var = f.kss - q;
        if (var < 0.0) { var = 0.0; atomicAdd(&sclamp, 1); }
        const double mu = f.y_std * mun + f.y_mean;
        const double sigma = sqrt(var * (f.y_std * f.y_std));
        double a = 0.0;
        if (f.acq == TGP_ACQ_UCB) {
            a = f.sf * mu + f.param * sigma;
        } else if (f.acq == TGP_ACQ_SIGMA) {
            a = sigma;
        } else if (f.acq == TGP_ACQ_PI || f.acq == TGP_ACQ_EI) {
            if (sigma != 0.0) {
                const double diff = f.sf * (mu - f.incumbent) - f.param;
                const double Z = diff / sigma;
                if (f.acq == TGP_ACQ_PI) {
                    a = ndtr_dev(Z);
                } else {
                    const double pdf = exp(-(Z * Z) / 2.0) / 2.5066282746310002;
                    a = diff * ndtr_dev(Z) + sigma * pdf;
                }
            }
        }
        const long gc = f.off + c;
        if (f.mu) f.mu[gc] = mu;
        if (f.sigma) f.sigma[gc] = sigma;
        if (f.acqv) f.acqv[gc] = a;
        if (f.acq != TGP_ACQ_NONE && !isnan(a)) { best = a; bi = gc; }
        else if (f.acq != TGP_ACQ_NONE) { bi = gc; }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int o = FIN_BLOCK / 2; o > 0; o >>= 1) {
        if (tid < o) {
            const double v2 = sv[tid + o];
            const long long i2 = si[tid + o];
            if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const long gb = f.off / FIN_BLOCK + blockIdx.x;
        f.bval[gb] = sv[0];
        f.bidx[gb] = si[0];
        if (sclamp) atomicAdd((unsigned long long *)&f.counters[1], (unsigned long long)sclamp);
    }
}

// Final (value, lowest index) of the per-block partials.  When a winner record is attached
// (tgp_set_winner_out: the sharded arg-max of SURVEY 8e) the same block also packs
// [value, (double)(global_offset + index), candidate row] into it, so the exchange between GPUs
// starts from device memory and the row never visits the host.
__global__ __launch_bounds__(256) void argmax_final_kernel(const double *__restrict__ bval,
                                                           const long long *__restrict__ bidx,
                                                           long nblk, double *__restrict__ best,
                                                           long long *__restrict__ besti,
                                                           double *__restrict__ winner,
                                                           const double *__restrict__ cand, int D,
                                                           long M, long long global_offset,
                                                           double *__restrict__ res_host, Bell bell) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    double v = -INFINITY;
    long long i = 0x7fffffffffffffffLL;
    for (long b = threadIdx.x; b < nblk; b += 256) {
        const double v2 = bval[b];
        const long long i2 = bidx[b];
        if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
    }
    sv[threadIdx.x] = v;
    si[threadIdx.x] = i;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const double v2 = sv[threadIdx.x + o];
            const long long i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) {
                sv[threadIdx.x] = v2;
                si[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best[0] = sv[0]; besti[0] = si[0]; }
    if (res_host && threadIdx.x == 0) {
        // zero-copy result record (small-problem path): [best value, best index, clamp count];
        // the clamp counter is handed back at zero
        res_host[0] = sv[0];
        res_host[1] = (double)((si[0] >= M) ? 0 : si[0]);
        res_host[2] = (double)besti[1];
        besti[1] = 0;
    }
    if (winner) {
        const long long wi = (si[0] >= M) ? 0 : si[0];      // all-NaN batch: index 0, as tgp_sweep reports
        if (threadIdx.x == 0) { winner[0] = sv[0]; winner[1] = (double)(global_offset + wi); }
        for (int d = threadIdx.x; d < D; d += 256) winner[2 + d] = cand[wi * D + d];
    }
    // (round 6) a polled small-problem call: this is the call's last kernel and a kernel of its own, so everything the
    // sweep kernel before it wrote -- means, deviations, acquisition values in mapped host memory -- is out
    bell_ring(bell, 1);
}

// ------------------------------------------------------------------------------------------
// What one sweep of the resident batch runs: decided once per call -- and once per fit for the part of it that
// starts INSIDE the fit (tgp_set_overlap, presweep_* below).
struct TrmmVariant {
    void (*kern)(GemmArgs) = nullptr;
    size_t lds = 0;
    int tile_m = SW_BM, tile_n = SW_BN, threads = 256;
};

template <typename T>
struct SweepPlan {
    TrmmVariant main;          // the contraction of this sweep
    TrmmVariant early;         // 128 x 128 direct-to-LDS tiles, one workgroup per CU: the row tiles contracted inside the fit
    bool glds = false, x3 = false, h2 = false;
    bool mean_in_trmm = false; // K*.alpha inside the contraction's full-k row tile (else: in the cross-kernel, njs partial rows)
    float h2_sb = 1.f;
    int njs = 1;               // splits of the training points over the cross-kernel grid (= rows of mupart in use when !mean_in_trmm)
    int n128 = 0, nunits = 0;  // 128-row tiles / 256-row units that hold real rows of Linv
};

static hipError_t lds_opt_in(Context &c, void (*kern)(GemmArgs), size_t lds) {
    // one opt-in record per kernel variant this call site can select
    static LdsOptIn opt_in[12];
    static std::atomic<const void *> owner[12];
    const void *fp = reinterpret_cast<const void *>(kern);
    int slot = -1;
    for (int i = 0; i < 12 && slot < 0; ++i) {
        const void *cur = owner[i].load(std::memory_order_acquire);
        if (cur == fp) slot = i;
        else if (cur == nullptr) {
            const void *expect = nullptr;
            if (owner[i].compare_exchange_strong(expect, fp, std::memory_order_acq_rel) || expect == fp) slot = i;
        }
    }
    if (slot >= 0) return opt_in[slot].ensure(fp, c.device, lds);
    return hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T, int BK>
static hipError_t make_plan(Context &c, SweepPlan<T> &p) {
    const Tuning &tu = tuning();
    const int N = (int)c.N, Np = (int)c.Np;
    // dominant kernel: direct-to-LDS variant by default, the register-staged template as an
    // A/B reference (TGP_TRMM=reg)
    auto trmm_reg = mfma_gemm_kernel<T, SW_BM, SW_BN, BK, true, KR_LOWER_A, TM_SWEEP, EP_SUMSQ>;
    void (*trmm_glds)(GemmArgs) = trmm_sumsq_glds_kernel<T>;
    if (sizeof(T) == 4 && tu.mfma16) trmm_glds = reinterpret_cast<void (*)(GemmArgs)>(trmm_sumsq_glds_kernel<float, MfmaF32x16>);
    p.glds = !tu.trmm_reg && (BK * sizeof(T) == 128);
    // Tile choice.  The large-tile variants (one workgroup per CU: 256x256 with 16 waves for f32,
    // 256x128 with 8 waves for f64) halve / cut the operand traffic per flop and are faster once
    // a launch has enough tiles to fill the chip twice; small launches keep the 128x128 kernel.
    // TGP_TILE = 128 | 256x128 | 256x256 overrides.
    TrmmVariant v;
    v.kern = p.glds ? trmm_glds : trmm_reg;
    v.lds = p.glds ? trmm_glds_lds_bytes() : gemm_lds_bytes<T, SW_BM, SW_BN, BK>();
    int want = 0;   // 0: 128x128, 1: 256x128, 2: 256x256
    if (p.glds) {
        if (!tu.tile.empty()) {
            want = tu.tile == "256x256" ? 2 : (tu.tile == "256x128" ? 1 : 0);
        } else {
            const int64_t mfirst = c.M < c.launch_rows ? c.M : c.launch_rows;
            const int64_t big_m = (N + 255) / 256;
            if (big_m >= 4) {   // N > 768: enough k per tile for the big tiles to pay
                if (sizeof(T) == 4 && big_m * ((mfirst + 255) / 256) >= 512) want = 2;
                else if (big_m * ((mfirst + 127) / 128) >= 512) want = 1;
            }
        }
        if (want == 2 && sizeof(T) != 4) want = 1;   // f64 accumulators need 2 waves / SIMD at most
    }
    if (want == 2) {
        v.kern = reinterpret_cast<void (*)(GemmArgs)>(trmm_sumsq_glds_big_kernel<float, 4, 4>);
        v.lds = trmm_big_lds_bytes<4, 4>(); v.tile_m = 256; v.tile_n = 256; v.threads = 1024;
    } else if (want == 1) {
        // three LDS buffers / two k-tiles in flight pay for f64 (C2 0.776 -> 0.829), not for f32;
        // TGP_NBUF=2|3 overrides
        const bool three = tu.nbuf ? (tu.nbuf == 3) : (sizeof(T) == 8);
        if (three) {
            v.kern = trmm_sumsq_glds_big_kernel<T, 4, 2, 3>;
            v.lds = trmm_big_lds_bytes<4, 2, 3>();
        } else {
            v.kern = trmm_sumsq_glds_big_kernel<T, 4, 2>;
            v.lds = trmm_big_lds_bytes<4, 2>();
        }
        v.tile_m = 256; v.tile_n = 128; v.threads = 512;
    }
    // f32 accuracy from three bf16 planes (opt-in dtype, trmm_bf16x3.hpp): 256 x 256 tiles
    p.x3 = sizeof(T) == 4 && c.dtype == TGP_F32X3 && p.glds;
    p.h2 = sizeof(T) == 4 && c.dtype == TGP_F32H2 && p.glds;   // two scaled fp16 planes (trmm_f16x2.hpp): 256 x 128 tiles
    if (p.x3) {
        v.kern = trmm_sumsq_bf16x3_kernel;
        v.lds = trmm_bf16x3_lds_bytes(); v.tile_m = 256; v.tile_n = 256; v.threads = 512;
    }
    if (p.h2) {
        v.kern = trmm_sumsq_f16x2_kernel;
        v.lds = trmm_f16x2_lds_bytes(); v.tile_m = 256; v.tile_n = 128; v.threads = 512;
    }
    p.h2_sb = p.h2 ? exp2f(floorf(log2f(16384.0f / (float)c.constant))) : 1.f;   // K* <= constant
    p.main = v;
    // the mean inside the contraction: the direct-to-LDS f64 / f32 kernels (the split-operand dtypes read pre-tiled
    // planes, the register-staged A/B template has another LDS image: they keep it in the cross-kernel)
    p.mean_in_trmm = tu.mean_in_trmm != 0 && p.glds && !p.x3 && !p.h2 && !(sizeof(T) == 4 && tu.mfma16);
    // the early row tiles (inside the fit): the 128 x 128 kernel with its LDS request padded past half a CU's, so
    // that ONE workgroup sits on a CU and the fit's own workgroups (48 KB per f64 GEMM tile) still find room beside it
    p.early.kern = trmm_sumsq_glds_kernel<T>;
    p.early.lds = (size_t)(tuning().pre_lds_kb < 64 ? 64 : tuning().pre_lds_kb > 160 ? 160 : tuning().pre_lds_kb) * 1024;
    p.early.tile_m = 128; p.early.tile_n = 128; p.early.threads = 256;
    // (the opt-in is set once per kernel and device: the 128 x 128 kernel gets the larger of its two requests)
    TGP_TRY(lds_opt_in(c, v.kern, v.kern == p.early.kern && p.early.lds > v.lds ? p.early.lds : v.lds));
    // splits of the training points over the cross-kernel grid
    int njs = Np / 128 < KS_JS ? Np / 128 : KS_JS;
    if (tu.ks_js >= 1 && tu.ks_js <= KS_JS && tu.ks_js <= Np / 128) njs = tu.ks_js;
    p.njs = njs;
    p.n128 = (N + 127) / 128;
    p.nunits = (N + 255) / 256;
    return hipSuccess;
}

// rows of launch pair n: offset, valid candidates, rows padded to the candidate tile
struct PairRows { int64_t off, m, rows; };
static PairRows pair_rows(const Context &c, int64_t n, int tile_n) {
    PairRows r;
    r.off = n * c.launch_rows;
    r.m = (c.M - r.off) < c.launch_rows ? (c.M - r.off) : c.launch_rows;
    r.rows = ((r.m + tile_n - 1) / tile_n) * tile_n;   // <= launch_rows, off + rows <= Mpad
    return r;
}

template <typename T>
static hipError_t issue_prep(Context &c, hipStream_t st) {
    const long pe = (long)c.ws_Mpad * c.Dp;
    hipLaunchKernelGGL(prep_candidates_kernel<T>, dim3((unsigned)((pe + 255) / 256 < 8192 ? (pe + 255) / 256 : 8192)),
                       dim3(256), 0, st, c.d_cand, c.d_ls, reinterpret_cast<T *>(c.d_Cs), (long)c.M, (long)c.ws_Mpad, (int)c.D, (int)c.Dp);
    return hipGetLastError();
}

// the cross-kernel slab of launch pair n (rows padded to `tile_n`, the widest candidate tile any contraction of this
// pair uses)
template <typename T>
static hipError_t issue_kstar(Context &c, const SweepPlan<T> &p, int64_t n, int sl, int tile_n, hipStream_t st) {
    const int N = (int)c.N, Np = (int)c.Np, Dp = (int)c.Dp;
    const T *Xs = reinterpret_cast<const T *>(sizeof(T) == 8 ? (const void *)c.d_Xs : (const void *)c.d_Xs32);
    const PairRows r = pair_rows(c, n, tile_n);
    void (*kst)(const T *, const T *, const double *, T *, double *, int, int, int, int, double, long, long, float);
    constexpr int KAR = sizeof(T) == 4 ? 8 : 4;
    constexpr int P3 = sizeof(T) == 4 ? 3 : 0, P2 = sizeof(T) == 4 ? 2 : 0;
    const dim3 kgrid((unsigned)(r.rows / (16 * KAR)), (unsigned)p.njs);
    const bool h2 = p.h2, x3 = p.x3, nomu = p.mean_in_trmm;
    switch (c.kernel) {
        case TGP_RBF: kst = h2 ? kstar_kernel<T, TGP_RBF, KAR, P2> : x3 ? kstar_kernel<T, TGP_RBF, KAR, P3> : nomu ? kstar_kernel<T, TGP_RBF, KAR, 0, false> : kstar_kernel<T, TGP_RBF, KAR>; break;
        case TGP_MATERN12: kst = h2 ? kstar_kernel<T, TGP_MATERN12, KAR, P2> : x3 ? kstar_kernel<T, TGP_MATERN12, KAR, P3> : nomu ? kstar_kernel<T, TGP_MATERN12, KAR, 0, false> : kstar_kernel<T, TGP_MATERN12, KAR>; break;
        case TGP_MATERN32: kst = h2 ? kstar_kernel<T, TGP_MATERN32, KAR, P2> : x3 ? kstar_kernel<T, TGP_MATERN32, KAR, P3> : nomu ? kstar_kernel<T, TGP_MATERN32, KAR, 0, false> : kstar_kernel<T, TGP_MATERN32, KAR>; break;
        default: kst = h2 ? kstar_kernel<T, TGP_MATERN52, KAR, P2> : x3 ? kstar_kernel<T, TGP_MATERN52, KAR, P3> : nomu ? kstar_kernel<T, TGP_MATERN52, KAR, 0, false> : kstar_kernel<T, TGP_MATERN52, KAR>; break;
    }
    hipLaunchKernelGGL(kst, kgrid, dim3(256), 0, st, reinterpret_cast<const T *>(c.d_Cs) + r.off * Dp, Xs, c.d_alpha,
                       reinterpret_cast<T *>(c.d_Ks[sl]), c.d_mupart + r.off, (int)r.rows, N, Np, Dp, c.constant,
                       (long)c.ws_Mpad, (long)Np / 16, p.h2_sb);   // (Np / 16: 16-k blocks per row of the pre-tiled split-operand slabs)
    return hipGetLastError();
}

// the contraction of launch pair n over the row tiles [tm0, tm0 + ntm) of variant v
template <typename T>
static hipError_t issue_trmm(Context &c, const SweepPlan<T> &p, const TrmmVariant &v, int64_t n, int sl, int tile_n_rows,
                             int tm0, int ntm, hipStream_t st) {
    if (ntm <= 0) return hipSuccess;
    const int N = (int)c.N, Np = (int)c.Np;
    const PairRows r = pair_rows(c, n, tile_n_rows);
    GemmArgs g{};
    g.A = sizeof(T) == 8 ? (const void *)c.d_Linv : (const void *)c.d_Linv32; g.lda = Np;
    g.B = c.d_Ks[sl]; g.ldb = Np;
    if (p.h2) { g.A = c.d_Linv16; g.K_blocks = (long)Np / 16; g.Ct = c.d_x2scal + 1; }
    if (p.x3) { g.A = c.d_Linv16; g.K_blocks = (long)Np / 16; }
    g.part = c.d_part + r.off; g.ldpart = c.ws_Mpad;
    g.prm = v.tile_m / 128;
    g.tm0 = tm0; g.ntm = ntm; g.ntn = (int)(r.rows / v.tile_n);
    g.ntn_group = (c.chunk % v.tile_n == 0 && c.chunk < r.rows) ? (int)(c.chunk / v.tile_n) : 0;
    g.K = ((N + v.tile_m - 1) / v.tile_m) * v.tile_m;
    if (p.mean_in_trmm) { g.mu_alpha = c.d_alpha; g.mu = c.d_mupart + r.off; }
    hipLaunchKernelGGL(v.kern, dim3((unsigned)(g.ntm * g.ntn)), dim3(v.threads), v.lds, st, g);
    return hipGetLastError();
}

// ---- the front of a sweep INSIDE the fit (tgp_set_overlap; north_star's step is fit -> sweep over a batch that is
// already resident, turbo/optimiser.py:336-340 + modules/auxiliary_optimisers.py:59-66) ------------------------
// The scaling of the candidates and the first launch pair's cross-kernel need nothing of the fit but Xs and the
// length scales; the contraction's row tile t needs rows [128 t, 128 (t + 1)) of Linv, final as soon as the outer
// block holding them has its share of the inverse.  Both run on the device's third stream beside the panel chain;
// tgp_sweep then skips what is done.  Same kernels, same arithmetic per value, same order of every sum: the
// results are those of the serial schedule bit for bit (tests/test_gpu_round5.py).
template <typename T, int BK>
static hipError_t presweep_front_t(Context &c, hipStream_t st) {
    SweepPlan<T> p;
    TGP_TRY((make_plan<T, BK>(c, p)));
    if (!p.mean_in_trmm) return hipSuccess;          // the cross-kernel needs alpha: nothing to start early
    TGP_TRY(lds_opt_in(c, p.early.kern, p.early.lds));
    TGP_TRY(issue_prep<T>(c, st));
    const int tn = p.main.tile_n > 128 ? p.main.tile_n : 128;
    TGP_TRY(issue_kstar<T>(c, p, 0, 0, tn, st));
    c.pre.front = true;
    c.pre.rows128 = 0;
    return hipSuccess;
}
template <typename T, int BK>
static hipError_t presweep_rows_t(Context &c, hipStream_t st, int rows_final, int budget128) {
    if (!c.pre.front) return hipSuccess;
    SweepPlan<T> p;
    TGP_TRY((make_plan<T, BK>(c, p)));
    int upto = rows_final / 128;                      // 128-row tiles whose rows of Linv are final
    if (upto > budget128) upto = budget128;
    upto &= ~1;                                       // whole 256-row units (finalize_kernel's pairing)
    if (upto > 2 * (p.nunits - 1)) upto = 2 * (p.nunits - 1);   // never the last unit: it spans the whole k-range and carries the mean
    if (upto <= c.pre.rows128) return hipSuccess;
    const int tn = p.main.tile_n > 128 ? p.main.tile_n : 128;
    TGP_TRY(issue_trmm<T>(c, p, p.early, 0, 0, tn, c.pre.rows128, upto - c.pre.rows128, st));
    c.pre.rows128 = upto;
    return hipSuccess;
}

template <typename T, int BK>
static hipError_t sweep_chunks(Context &c, int acq, double sf, double incumbent, double param,
                               bool want_mu, bool want_sigma, bool want_acq) {
    const int Np = (int)c.Np, D = (int)c.D;
    SweepPlan<T> p;
    TGP_TRY((make_plan<T, BK>(c, p)));
    const TrmmVariant &v = p.main;
    const bool x3 = p.x3, h2 = p.h2;

    // One pass over the batch: scale all candidates once, then per chunk the cross-kernel slab and
    // its contraction, then ONE finalize + arg-max over all M.  The partial sums live in (Np / 128, Mpad)
    // / (KS_JS, Mpad) arrays, so nothing but the slab is per-chunk.
    // (Running kstar / finalize on a second stream beside the contraction was measured slower twice
    // -- their workgroups take CUs from the MFMA kernel instead of sharing them -- and was removed.)
    hipStream_t sa = c.stream;
    const int64_t Mpad = c.ws_Mpad;
    // what the last fit already started for this very batch (tgp_set_overlap): the scaling, launch pair 0's
    // cross-kernel and the first pre128 row tiles of its contraction
    const bool pre = c.pre.usable && p.mean_in_trmm;
    const int pre128 = pre ? c.pre.rows128 : 0;
    if (!pre) TGP_TRY(issue_prep<T>(c, sa));
    if (h2 && (c.linv16_gen != c.fit_gen || c.linv16_sb != p.h2_sb)) {
        TGP_TRY(hipMemsetAsync(c.d_x2scal, 0, 2 * sizeof(unsigned), sa));
        hipLaunchKernelGGL(maxabs_f32_kernel, dim3(1024), dim3(256), 0, sa, c.d_Linv32, (long)Np * Np / 4, c.d_x2scal);
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(split_f16x2_kernel, dim3(4096), dim3(256), 0, sa, c.d_Linv32, c.d_Linv16, (long)Np, (long)Np, c.d_x2scal, p.h2_sb);
        TGP_TRY(hipGetLastError());
        c.linv16_gen = c.fit_gen;
        c.linv16_sb = p.h2_sb;
    }
    if (x3 && c.linv16_gen != c.fit_gen) {   // the factor changed since its planes were cut
        hipLaunchKernelGGL(split_bf16x3_kernel, dim3(4096), dim3(256), 0, sa, c.d_Linv32, c.d_Linv16, (long)Np, (long)Np);
        TGP_TRY(hipGetLastError());
        c.linv16_gen = c.fit_gen;
    }
    // one launch pair (cross-kernel, contraction) per launch_rows candidates -- normally the whole batch;
    // inside the contraction the tiles walk the slab in groups of c.chunk candidates (sweep_tile())
    const bool two_slots = c.d_Ks[1] != nullptr && c.launch_rows == c.chunk && (x3 || h2);
    const int64_t nchunks = (c.M + c.launch_rows - 1) / c.launch_rows;
    const int tn_rows = v.tile_n > 128 ? v.tile_n : 128;   // (the early contraction's tiles are 128 candidates wide: pad to the wider of the two, always)
    int mark = -1;   // profiling: the event that closed the previous launch opens the next
    for (int64_t n = 0; n < nchunks; ++n) {
        const int sl = two_slots ? (int)(n & 1) : 0;
        if (mark < 0) mark = prof_mark(c, sa);
        if (!(pre && n == 0)) {
            TGP_TRY(issue_kstar<T>(c, p, n, sl, p.mean_in_trmm ? tn_rows : v.tile_n, sa));
            const int m2 = prof_mark(c, sa);
            prof_seg(c, mark, m2, 1);
            mark = m2;
        }
        // row tiles of this variant from the first one the fit has not contracted already
        const int tm_first = (pre && n == 0) ? pre128 * 128 / v.tile_m : 0;
        const int ntm_all = ((int)c.N + v.tile_m - 1) / v.tile_m;
        TGP_TRY(issue_trmm<T>(c, p, v, n, sl, p.mean_in_trmm ? tn_rows : v.tile_n, tm_first, ntm_all - tm_first, sa));
        {
            const int m2 = prof_mark(c, sa);
            // algorithmic flops of this launch: rows^2 per candidate over the rows it covers (the solve's
            // N (N + 1) / 2 FMAs) -- a launch the fit took row tiles of has N^2 - r0^2
            const double r0 = (double)tm_first * v.tile_m;
            prof_seg(c, mark, m2, 0, (double)pair_rows(c, n, v.tile_n).m * ((double)c.N * (double)c.N - r0 * r0));
            mark = m2;
        }
    }
    {
        FinArgs f{};
        f.part = c.d_part; f.ldpart = Mpad;
        f.nunits = p.nunits; f.n128 = p.n128;
        f.base_pairs = (v.tile_m == 128) ? p.nunits : 0;
        f.pre_m = pre ? pair_rows(c, 0, tn_rows).m : 0;
        f.pre_pairs = pre128 / 2;
        f.mupart = c.d_mupart; f.njs = p.mean_in_trmm ? 1 : p.njs;
        f.off = 0; f.m = c.M;
        f.kss = c.constant + c.noise;
        f.y_mean = c.y_mean; f.y_std = c.y_std;
        f.acq = acq; f.sf = sf; f.incumbent = incumbent; f.param = param;
        f.mu = want_mu ? c.d_mu : nullptr;
        f.sigma = want_sigma ? c.d_sigma : nullptr;
        f.acqv = want_acq ? c.d_acq : nullptr;
        f.bval = c.d_bval; f.bidx = c.d_bidx; f.counters = c.d_besti;
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((c.M + FIN_BLOCK - 1) / FIN_BLOCK)),
                           dim3(FIN_BLOCK), 0, sa, f);
        TGP_TRY(hipGetLastError());
    }
    if (acq != TGP_ACQ_NONE || c.sweep_res_host) {
        // (without an acquisition the launch only hands the clamp count over and zeroes the counter)
        const long nblk = acq != TGP_ACQ_NONE ? (long)((c.M + FIN_BLOCK - 1) / FIN_BLOCK) : 0L;
        hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, sa, c.d_bval, c.d_bidx, nblk,
                           c.d_best, c.d_besti, acq != TGP_ACQ_NONE ? c.d_winner : nullptr, c.d_cand, D, (long)c.M,
                           (long long)c.winner_offset, c.sweep_res_host, Bell{nullptr, 0, nullptr});
        TGP_TRY(hipGetLastError());
    }
    return hipSuccess;
}

hipError_t launch_argmax_final(Context &c, long nblk, double *res_host, const Bell &bell) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, c.stream, c.d_bval, c.d_bidx, nblk,
                       c.d_best, c.d_besti, c.d_winner, c.d_cand, (int)c.D, (long)c.M,
                       (long long)c.winner_offset, res_host, bell);
    return hipGetLastError();
}

// dispatch on (dtype, TGP_BK)
#define TGP_SWEEP_DISPATCH(fn, ...)                                                  \
    do {                                                                             \
        const int bk_env = tuning().bk;                                              \
        if (c.dtype != TGP_F64) {                                                    \
            if (bk_env == 16) return fn<float, 16>(__VA_ARGS__);                     \
            if (bk_env == 64) return fn<float, 64>(__VA_ARGS__);                     \
            return fn<float, 32>(__VA_ARGS__);                                       \
        }                                                                            \
        if (bk_env == 8) return fn<double, 8>(__VA_ARGS__);                          \
        if (bk_env == 32) return fn<double, 32>(__VA_ARGS__);                        \
        return fn<double, 16>(__VA_ARGS__);                                          \
    } while (0)

hipError_t launch_sweep(Context &c, int acq, double sf, double incumbent, double param,
                        bool want_mu, bool want_sigma, bool want_acq) {
    TGP_SWEEP_DISPATCH(sweep_chunks, c, acq, sf, incumbent, param, want_mu, want_sigma, want_acq);
}

hipError_t presweep_front(Context &c, hipStream_t st) { TGP_SWEEP_DISPATCH(presweep_front_t, c, st); }
hipError_t presweep_rows(Context &c, hipStream_t st, int rows_final, int budget128) {
    TGP_SWEEP_DISPATCH(presweep_rows_t, c, st, rows_final, budget128);
}

}  // namespace tgp
