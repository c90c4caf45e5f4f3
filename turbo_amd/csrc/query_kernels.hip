// query_kernels.hip -- acquisition value AND its gradient w.r.t. the query point, for a small
// batch of points (the gradient-refinement stage of the auxiliary optimiser, "next" row
// turbo/modules/auxiliary_optimisers.py:69-112 runs L-BFGS-B from the best random
// candidates with finite-difference gradients over 1-point acq calls).  All f64.
//
//   k_j = c k0(r_j),  dk_j/dx_d = -c h(r_j) (u_d - xs_jd) / l_d      (u = x / l;  h as in the LML gradient)
//   mu  = s_y k.alpha + ybar            dmu/dx_d  = -s_y / l_d * sum_j alpha_j c h_j (u_d - xs_jd)
//   v = Linv k, w = Linv^T v = K^-1 k,  var = (c + s2) - v.v
//                                       dvar/dx_d = +2 / l_d * sum_j w_j c h_j (u_d - xs_jd)
//   sigma = s_y sqrt(var);  UCB / PI / EI and their gradients follow in closed form.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mfma_gemm.hpp"
#include "pairwise.hpp"
#include "query_math.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// ks[q][j] = c k0, hw[q][j] = c h  (0 for j >= N);  uq[q][d] = x / l
template <int KIND>
__global__ __launch_bounds__(256) void q_kvec_kernel(const double *__restrict__ Xq,
                                                     const double *__restrict__ ls,
                                                     const double *__restrict__ Xs,
                                                     double *__restrict__ uq, double *__restrict__ ks,
                                                     double *__restrict__ hw, int N, int Np, int D,
                                                     int Dp, double constant, unsigned long long *stamp) {
    extern __shared__ double u[];
    const int q = blockIdx.y;
    if (stamp && blockIdx.x == 0 && q == 0 && threadIdx.x == 0) *stamp = wall_clock64();   // the polled call's start tick (doorbell.hpp)
    for (int d = threadIdx.x; d < Dp; d += 256) {
        const double v = d < D ? Xq[(long)q * D + d] / ls[d] : 0.0;
        u[d] = v;
        if (blockIdx.x == 0) uq[(long)q * Dp + d] = v;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    double k = 0.0, h = 0.0;
    if (j < N) {
        const double *xj = Xs + (long)j * Dp;
        double d2 = 0.0;
        for (int d = 0; d < Dp; ++d) {
            const double df = u[d] - xj[d];
            d2 = fma(df, df, d2);
        }
        k = kernel_value<double, KIND>(d2, constant);
        h = constant * h_weight<KIND>(d2);
    }
    ks[(long)q * Np + j] = k;
    hw[(long)q * Np + j] = h;
}

// z[q][i] = sum_{j<=i} Linv[i][j] v[q][j]: one wave per row i for up to QROWS_QB query points at a
// time (blockIdx.y walks groups of query points), every element of the row read once for all of
// them, two row segments in flight per lane
// (QB = 8, or 16 for batches of more than eight points: the default gradient stage's ten restarts used to take TWO
// groups of eight, i.e. two passes over Linv in each gemv; a point's sums do not depend on the group it sits in)
constexpr int QROWS_QB = 8;
__device__ __forceinline__ double q_wave_sum(double s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
template <int QB>
__global__ __launch_bounds__(256) void q_gemv_rows_kernel(const double *__restrict__ Linv,
                                                          const double *__restrict__ v,
                                                          double *__restrict__ z, int Np, int m) {
    // (round 6: 16-byte loads, 256 columns of the row per trip -- with 8-byte loads and 128 columns a row of N = 900
    // took eight dependent trips to L2; the pairs of a lane are summed in the order below, fixed)
    typedef double rd2_t __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Np) return;
    const int q0 = blockIdx.y * QB;
    const int mq = min(QB, m - q0);
    const double *row = Linv + (long)i * Np;
    const double *vq[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) vq[qq] = v + (long)(q0 + min(qq, mq - 1)) * Np;   // (absent points repeat the last: no branch)
    double acc[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) acc[qq] = 0.0;
    // columns [0, i] in pairs (2 lane, 2 lane + 1); the half of a pair right of the diagonal is masked (the whole row
    // is addressable: Np is a multiple of 256)
    for (int jj = 2 * lane; jj <= i; jj += 256) {
        const int j1 = jj + 128;
        const bool two = j1 <= i;
        const int j1c = two ? j1 : jj;
        rd2_t l0 = *reinterpret_cast<const rd2_t *>(row + jj);
        rd2_t l1 = *reinterpret_cast<const rd2_t *>(row + j1c);
        if (jj + 1 > i) l0[1] = 0.0;
        if (!two) { l1[0] = 0.0; l1[1] = 0.0; }
        else if (j1 + 1 > i) l1[1] = 0.0;
        rd2_t a0[QB], a1[QB];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
            a0[qq] = *reinterpret_cast<const rd2_t *>(vq[qq] + jj);
            a1[qq] = *reinterpret_cast<const rd2_t *>(vq[qq] + j1c);
        }
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
            acc[qq] = fma(l1[1], a1[qq][1], fma(l1[0], a1[qq][0], fma(l0[1], a0[qq][1], fma(l0[0], a0[qq][0], acc[qq]))));
    }
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) {
        const double s = q_wave_sum(acc[qq]);
        if (lane == 0 && qq < mq) z[(long)(q0 + qq) * Np + i] = s;
    }
}

// wp[s][q][j] = sum over the rows i >= j of row split s of Linv[i][j] v[q][i]: 64 columns per block,
// every element of Linv read ONCE for up to QCOLS_QB query points (blockIdx.z walks groups of
// query points).  The rows below the block's columns go in chunks of QCOLS_ROWS, dealt round-robin
// to QCOLS_SPLIT blocks (blockIdx.y) so that N / 64 column blocks still fill the chip; a chunk's
// slice of v sits in LDS ([row][query point], read back as broadcasts), and every thread keeps
// QCOLS_U loads of Linv in flight.  w = the sum of the splits' shares, added by the reader in a
// fixed order.  (One block per (query point, column block) walking all rows one load at a time
// was latency-bound: 139 us at N = 2048 with 10 points, rocprofv3.)
typedef double qd2_t __attribute__((ext_vector_type(2)));
// (round 6: chunks of 64 rows over 16 splits, was 256 over 8 -- at N = 900 only 40 of the 128 workgroups had a chunk,
// each walking it in eight dependent trips of eight loads: 11 us, the longest kernel of a 25 us call)
constexpr int QCOLS_QB = 8, QCOLS_SPLIT = 16, QCOLS_U = 8, QCOLS_ROWS = 64;
template <int QB>
__global__ __launch_bounds__(256) void q_gemv_cols_kernel(const double *__restrict__ Linv,
                                                          const double *__restrict__ v,
                                                          double *__restrict__ wp, int N, int Np, int m) {
    __shared__ __attribute__((aligned(16))) double vs[QCOLS_ROWS][QB];
    __shared__ double red[4][QB][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, j = j0 + c;
    const int q0 = blockIdx.z * QB;
    const int mq = min(QB, m - q0);
    double acc[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) acc[qq] = 0.0;
    for (int r0 = j0 + (int)blockIdx.y * QCOLS_ROWS; r0 < N; r0 += QCOLS_SPLIT * QCOLS_ROWS) {
        __syncthreads();
        if (threadIdx.x < QCOLS_ROWS) {
            const int i = r0 + threadIdx.x, ic = min(i, Np - 1);
            double t[QB];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) t[qq] = v[(long)(q0 + min(qq, mq - 1)) * Np + ic];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) vs[threadIdx.x][qq] = (qq < mq && i < N) ? t[qq] : 0.0;   // (rows >= N and absent points: 0)
        }
        __syncthreads();
#pragma unroll 1
        for (int u0 = 0; u0 < QCOLS_ROWS / 4; u0 += QCOLS_U) {
            double l[QCOLS_U];
#pragma unroll
            for (int u = 0; u < QCOLS_U; ++u) {
                const int i = min(r0 + 4 * (u0 + u) + rg, Np - 1);    // (always addressable; the products below use 0 past N)
                l[u] = Linv[(long)i * Np + j];
            }
#pragma unroll
            for (int u = 0; u < QCOLS_U; ++u) {
                const int rr = 4 * (u0 + u) + rg;
                const double lu = (r0 + rr >= j) ? l[u] : 0.0;        // the strictly upper part of the diagonal block is not Linv
                const qd2_t *vr = reinterpret_cast<const qd2_t *>(vs[rr]);
#pragma unroll
                for (int q2 = 0; q2 < QB / 2; ++q2) {
                    const qd2_t vv = vr[q2];
                    acc[2 * q2] = fma(lu, vv[0], acc[2 * q2]);
                    acc[2 * q2 + 1] = fma(lu, vv[1], acc[2 * q2 + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) red[rg][qq][c] = acc[qq];
    __syncthreads();
    for (int e = threadIdx.x; e < mq * 64; e += 256) {
        const int qq = e >> 6, cc = e & 63;
        wp[((long)blockIdx.y * m + q0 + qq) * Np + j0 + cc] =
            (red[0][qq][cc] + red[1][qq][cc]) + (red[2][qq][cc] + red[3][qq][cc]);
    }
}

// ---- the two products on the matrix cores (round 6, late) ----------------------------------------------------------
// z = Linv v and w = Linv^T z for up to 16 query points are (N x N triangular) x (N x 16) products: 2 N^2 x 16 flops --
// nothing for v_mfma_f64_16x16x4 -- over ONE pass through Linv each.  The wave-per-row / split-column kernels above
// spent 19.5 + 11.0 us on them at N = 2048 with ten points (0.9 TB/s: thirty-four loads per 256 columns of a row, sixteen
// shares of w for the reduction to add up again for every dimension); these: a workgroup per block of 16 rows
// (columns), the longest first, its k-range in chunks of 16 dealt to the four waves, a chunk = two 32-byte loads per
// lane + four MFMAs, the waves' accumulators added in a fixed order through LDS.  A point's column of the B operand
// meets nobody else's: its sums do not depend on the batch it travels in (absent points repeat the last one).
// Linv is zero above the diagonal and v, z are zero from N on (q_kvec_kernel), so the diagonal chunks need no mask.
constexpr int QM_PTS = 16;
// NW waves per workgroup share a block's chunks (wave w takes chunks w, w + NW, ...): the longest block of N = 2048 is
// 128 chunks -- thirty-two dependent load -> MFMA trips for each of four waves (14 us measured), eight for each of
// sixteen.  (Cutting the k-range over several WORKGROUPS instead made the consumers add the shares up again: rows
// 14.0 -> 8.9 us but columns 11.5 -> 14.7 and the reduction 15.3 -> 20.9.)  NW depends on the size class only.
__host__ __device__ inline int query_waves(int Np) { return Np <= 512 ? 4 : (Np <= 1024 ? 8 : 16); }

// the NW waves' accumulators, added in a fixed order: pairs, pairs of pairs, ...
template <int NW>
__device__ __forceinline__ double q_sum_waves(const double (*part)[16][17], int row, int col) {
    double t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = part[w][row][col];
#pragma unroll
    for (int h = 1; h < NW; h *= 2)
#pragma unroll
        for (int w = 0; w < NW; w += 2 * h) t[w] += t[w + h];
    return t[0];
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void q_rows_mfma_kernel(const double *__restrict__ Linv, const double *__restrict__ v,
                                                              double *__restrict__ z, int Np, int m) {
    __shared__ double part[NW][16][17];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rb = (int)gridDim.x - 1 - (int)blockIdx.x;      // block of rows [16 rb, 16 rb + 16): the longest first
    const int q0 = blockIdx.y * QM_PTS;
    const int r = lane & 15, kq = lane >> 4;
    const int qn = min(q0 + r, m - 1);
    const double *arow = Linv + (long)(16 * rb + r) * Np + 4 * kq;     // A[i = r][k slot kq] of MFMA e: column 16 c + 4 kq + e
    const double *bvec = v + (long)qn * Np + 4 * kq;                   // B[k slot kq][n = r]: the same column of point q0 + r
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    int c = wave;
    for (; c + 3 * NW <= rb; c += 4 * NW) {       // four of the wave's chunks per trip: eight 32-byte loads in flight
        d4_t a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const d4_t *>(arow + 16 * (c + u * NW));
            b[u] = *reinterpret_cast<const d4_t *>(bvec + 16 * (c + u * NW));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    for (; c <= rb; c += NW) {
        const d4_t a = *reinterpret_cast<const d4_t *>(arow + 16 * c);
        const d4_t b = *reinterpret_cast<const d4_t *>(bvec + 16 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) part[wave][kq + 4 * t][r] = acc[t];    // C: row = (lane >> 4) + 4 t, column = lane & 15
    __syncthreads();
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        if (q0 + col < m) z[(long)(q0 + col) * Np + 16 * rb + row] = q_sum_waves<NW>(part, row, col);
    }
}

// w[q][j] = sum_{i >= j} Linv[i][j] z[q][i]: block of columns [16 jb, 16 jb + 16), rows from its diagonal block down to
// the last real one; A[j = r][k slot kq] of MFMA e is Linv[i0 + 4 kq + e][16 jb + r] (sixteen lanes read 128 contiguous bytes)
template <int NW>
__global__ __launch_bounds__(64 * NW) void q_cols_mfma_kernel(const double *__restrict__ Linv, const double *__restrict__ zv,
                                                              double *__restrict__ w, int N, int Np, int m) {
    __shared__ double part[NW][16][17];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int jb = blockIdx.x;                                   // (block 0 is the longest)
    const int q0 = blockIdx.y * QM_PTS;
    const int r = lane & 15, kq = lane >> 4;
    const int qn = min(q0 + r, m - 1);
    const int nchunks = (N + 15) / 16 - jb;                      // row chunks [16 (jb + c), ...): z is zero from N on
    const double *acol = Linv + (long)(16 * jb + 4 * kq) * Np + 16 * jb + r;
    const double *bvec = zv + (long)qn * Np + 16 * jb + 4 * kq;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    int c = wave;
    for (; c + 3 * NW < nchunks; c += 4 * NW) {   // four of the wave's chunks per trip
        double a[4][4];
        d4_t b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *ac = acol + (long)(16 * (c + u * NW)) * Np;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = ac[(long)e * Np];
            b[u] = *reinterpret_cast<const d4_t *>(bvec + 16 * (c + u * NW));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    for (; c < nchunks; c += NW) {
        const double *ac = acol + (long)(16 * c) * Np;
        double a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = ac[(long)e * Np];
        const d4_t b = *reinterpret_cast<const d4_t *>(bvec + 16 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) part[wave][kq + 4 * t][r] = acc[t];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        if (q0 + col < m) w[(long)(q0 + col) * Np + 16 * jb + row] = q_sum_waves<NW>(part, row, col);
    }
}

// value and gradient of the acquisition at query point q from its sums r = [k.alpha, v.v, gm[0..D), gv[0..D)]
// (LD: how r is read -- plainly behind a kernel boundary, past the L1 inside the kernel that wrote it)
struct QFinal {
    const double *ls;
    double *val, *grad;      // (m), (m, D): device memory or device-mapped host memory
    int m, D, acq;
    double kss, y_mean, y_std, sf, incumbent, param;
    Bell bell;               // word != null: q_reduce_kernel's last workgroup forms value + gradient and rings
};
template <typename LD>
__device__ __forceinline__ void q_finalize_point(const QFinal &f, const double *r, int q, LD ld) {
    const int D = f.D;
    const double mu = f.y_std * ld(r) + f.y_mean;
    double var = f.kss - ld(r + 1);
    const bool pos = var > 0.0;
    if (!pos) var = 0.0;
    const double sn = sqrt(var);
    const double sigma = f.y_std * sn;
    const AcqCoef ac = acq_coef(f.acq, mu, sigma, f.sf, f.incumbent, f.param);
    const double a = ac.a, cm = ac.cm, cs = ac.cs;
    f.val[q] = a;
    for (int d = 0; d < D; ++d) {
        const double dmu = -f.y_std * ld(r + 2 + d) / f.ls[d];
        const double dvar = 2.0 * ld(r + 2 + D + d) / f.ls[d];
        const double dsig = pos && sn > 0.0 ? f.y_std * dvar / (2.0 * sn) : 0.0;
        f.grad[(long)q * D + d] = cm * dmu + cs * dsig;
    }
}

// per (q, d): gmu = sum_j alpha_j hw_j (u_d - xs_jd), gv = sum_j w_j hw_j (u_d - xs_jd);
// the d == 0 block also reduces mun = ks.alpha and qv = v.v.
// Round 6 (fin.bell.word != null, tgp_acq_grad's polled call): the workgroup that draws the last ticket turns the
// sums of ALL points into value + gradient -- straight into device-mapped host memory -- and rings the call's
// doorbell: no finalize launch, no D2H copies, no stream synchronisation.  Nobody waits for anybody.
__global__ __launch_bounds__(256) void q_reduce_kernel(const double *__restrict__ Xs,
                                                       const double *__restrict__ alpha,
                                                       const double *__restrict__ uq,
                                                       const double *__restrict__ ks,
                                                       const double *__restrict__ hw,
                                                       const double *__restrict__ v,
                                                       const double *__restrict__ w,
                                                       double *__restrict__ out, int N, int Np, int D,
                                                       int Dp, int m, int nshare, QFinal fin) {
    __shared__ double red[4][256];
    __shared__ int is_last;
    const int q = blockIdx.y, d = blockIdx.x;
    const double ud = uq[(long)q * Dp + d];
    const double *hq = hw + (long)q * Np, *wq = w + (long)q * Np, *kq = ks + (long)q * Np, *vq = v + (long)q * Np;
    const long ws = (long)m * Np;     // between the row splits' shares of w (q_gemv_cols_kernel)
    double gm = 0.0, gv = 0.0, mun = 0.0, qv = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        const double t = hq[j] * (ud - Xs[(long)j * Dp + d]);
        double wj;
        if (nshare == 1) {               // (uniform over the launch) w itself: q_cols_mfma_kernel
            wj = wq[j];
        } else {                         // the row splits' shares of q_gemv_cols_kernel, added in a fixed order
            wj = 0.0;
#pragma unroll
            for (int sp = QCOLS_SPLIT - 1; sp >= 0; --sp) wj += wq[j + sp * ws];
        }
        gm = fma(alpha[j], t, gm);
        gv = fma(wj, t, gv);
        if (d == 0) {
            mun = fma(kq[j], alpha[j], mun);
            qv = fma(vq[j], vq[j], qv);
        }
    }
    red[0][threadIdx.x] = gm; red[1][threadIdx.x] = gv; red[2][threadIdx.x] = mun; red[3][threadIdx.x] = qv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    // out[q] = [mun, qv, gm[0..D), gv[0..D)]
    double *oq = out + (long)q * (2 + 2 * D);
    if (threadIdx.x == 0) {
        oq[2 + d] = red[0][0];
        oq[2 + D + d] = red[1][0];
        if (d == 0) { oq[0] = red[2][0]; oq[1] = red[3][0]; }
    }
    if (!fin.bell.word) return;       // (uniform over the launch)
    if (threadIdx.x == 0) {
        __threadfence();              // this workgroup's sums are out, device-wide
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned t = __hip_atomic_fetch_add(fin.bell.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t == total - 1;
        if (is_last) __hip_atomic_store(fin.bell.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // q_finalize_point's arithmetic, spread over the workgroup: a thread per point forms the value and the point's
    // coefficients (into red, free now), then a thread per (point, dimension) forms the gradient entries -- one thread
    // walking a point's D entries (atomic loads in, PCIe stores out) made the 64-point call slower than round 5's
    auto ld = [](const double *a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int q0 = 0; q0 < m; q0 += 256) {
        const int qq = q0 + threadIdx.x;
        if (qq < m) {
            const double *r = out + (long)qq * (2 + 2 * D);
            const double mu = fin.y_std * ld(r) + fin.y_mean;
            double var = fin.kss - ld(r + 1);
            const bool pos = var > 0.0;
            if (!pos) var = 0.0;
            const double sn = sqrt(var);
            const AcqCoef ac = acq_coef(fin.acq, mu, fin.y_std * sn, fin.sf, fin.incumbent, fin.param);
            fin.val[qq] = ac.a;
            red[0][threadIdx.x] = ac.cm; red[1][threadIdx.x] = ac.cs; red[2][threadIdx.x] = sn;
            red[3][threadIdx.x] = (pos && sn > 0.0) ? 1.0 : 0.0;
        }
        __syncthreads();
        const int nq = min(256, m - q0);
        for (int e = threadIdx.x; e < nq * D; e += 256) {
            const int ql = e / D, dd = e - ql * D;
            const double *r = out + (long)(q0 + ql) * (2 + 2 * D);
            const double dmu = -fin.y_std * ld(r + 2 + dd) / fin.ls[dd];
            const double dvar = 2.0 * ld(r + 2 + D + dd) / fin.ls[dd];
            const double dsig = red[3][ql] != 0.0 ? fin.y_std * dvar / (2.0 * red[2][ql]) : 0.0;
            fin.grad[(long)(q0 + ql) * D + dd] = red[0][ql] * dmu + red[1][ql] * dsig;
        }
        __syncthreads();
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        fin.bell.word[2] = wall_clock64();
        __hip_atomic_store(fin.bell.word, fin.bell.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one thread per query point: value and gradient of the acquisition
__global__ void q_finalize_kernel(const double *__restrict__ red, QFinal fin) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= fin.m) return;
    q_finalize_point(fin, red + (long)q * (2 + 2 * fin.D), q, [](const double *a) { return *a; });
}

// doubles of launch_query's workspace per query point
int64_t query_ws_doubles(const Context &c) { return c.Dp + (int64_t)(3 + QCOLS_SPLIT) * c.Np + 2 + 2 * c.D; }

// where launch_query leaves, per query point q, [k.alpha, v.v, gm[0..D), gv[0..D)] (stride 2 + 2 D)
double *query_red(const Context &c, double *d_ws, int m) {
    return d_ws + (long)m * c.Dp + (long)(3 + QCOLS_SPLIT) * m * c.Np;
}

// workspace per query point: uq (Dp) | ks, hw, v (3 Np) | w's QCOLS_SPLIT shares (16 Np) | red (2 + 2 D)  [query_ws_doubles].
// d_val == nullptr: stop after the sums (the caller turns query_red() into value + gradient itself).
hipError_t launch_query(Context &c, const double *d_Xq, int m, int acq, double sf, double incumbent,
                        double param, double *d_ws, double *d_val, double *d_grad, const Bell &bell) {
    hipStream_t s = c.stream;
    const int N = (int)c.N, Np = (int)c.Np, D = (int)c.D, Dp = (int)c.Dp;
    double *uq = d_ws;
    double *ks = uq + (long)m * Dp;
    double *hw = ks + (long)m * Np;
    double *v = hw + (long)m * Np;
    double *w = v + (long)m * Np;
    double *red = w + (long)QCOLS_SPLIT * m * Np;
    unsigned long long *stamp = (d_val && bell.word) ? bell.word + 1 : nullptr;
    const dim3 g1((Np + 255) / 256, m);
    const size_t sh = (size_t)Dp * sizeof(double);
    switch (c.kernel) {
        case TGP_RBF: hipLaunchKernelGGL(q_kvec_kernel<TGP_RBF>, g1, dim3(256), sh, s, d_Xq, c.d_ls, c.d_Xs, uq, ks, hw, N, Np, D, Dp, c.constant, stamp); break;
        case TGP_MATERN12: hipLaunchKernelGGL(q_kvec_kernel<TGP_MATERN12>, g1, dim3(256), sh, s, d_Xq, c.d_ls, c.d_Xs, uq, ks, hw, N, Np, D, Dp, c.constant, stamp); break;
        case TGP_MATERN32: hipLaunchKernelGGL(q_kvec_kernel<TGP_MATERN32>, g1, dim3(256), sh, s, d_Xq, c.d_ls, c.d_Xs, uq, ks, hw, N, Np, D, Dp, c.constant, stamp); break;
        default: hipLaunchKernelGGL(q_kvec_kernel<TGP_MATERN52>, g1, dim3(256), sh, s, d_Xq, c.d_ls, c.d_Xs, uq, ks, hw, N, Np, D, Dp, c.constant, stamp); break;
    }
    TGP_TRY(hipGetLastError());
    const bool mfma = tuning().query_mfma != 0;      // A/B: 0 = round 6's earlier wave-per-row / split-column kernels
    if (mfma) {
        const unsigned groups = (unsigned)((m + QM_PTS - 1) / QM_PTS);
        const dim3 gr(Np / 16, groups), gc((N + 15) / 16, groups);
        switch (query_waves(Np)) {
            case 4:
                hipLaunchKernelGGL(q_rows_mfma_kernel<4>, gr, dim3(256), 0, s, c.d_Linv, ks, v, Np, m);
                TGP_TRY(hipGetLastError());
                hipLaunchKernelGGL(q_cols_mfma_kernel<4>, gc, dim3(256), 0, s, c.d_Linv, v, w, N, Np, m);
                break;
            case 8:
                hipLaunchKernelGGL(q_rows_mfma_kernel<8>, gr, dim3(512), 0, s, c.d_Linv, ks, v, Np, m);
                TGP_TRY(hipGetLastError());
                hipLaunchKernelGGL(q_cols_mfma_kernel<8>, gc, dim3(512), 0, s, c.d_Linv, v, w, N, Np, m);
                break;
            default:
                hipLaunchKernelGGL(q_rows_mfma_kernel<16>, gr, dim3(1024), 0, s, c.d_Linv, ks, v, Np, m);
                TGP_TRY(hipGetLastError());
                hipLaunchKernelGGL(q_cols_mfma_kernel<16>, gc, dim3(1024), 0, s, c.d_Linv, v, w, N, Np, m);
                break;
        }
    } else if (m > QROWS_QB) {
        hipLaunchKernelGGL(q_gemv_rows_kernel<2 * QROWS_QB>, dim3((Np + 3) / 4, (m + 2 * QROWS_QB - 1) / (2 * QROWS_QB)), dim3(256), 0, s, c.d_Linv, ks, v, Np, m);
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(q_gemv_cols_kernel<2 * QCOLS_QB>, dim3(Np / 64, QCOLS_SPLIT, (m + 2 * QCOLS_QB - 1) / (2 * QCOLS_QB)), dim3(256), 0, s,
                           c.d_Linv, v, w, N, Np, m);
    } else {
        hipLaunchKernelGGL(q_gemv_rows_kernel<QROWS_QB>, dim3((Np + 3) / 4, 1), dim3(256), 0, s, c.d_Linv, ks, v, Np, m);
        TGP_TRY(hipGetLastError());
        hipLaunchKernelGGL(q_gemv_cols_kernel<QCOLS_QB>, dim3(Np / 64, QCOLS_SPLIT, 1), dim3(256), 0, s, c.d_Linv, v, w, N, Np, m);
    }
    TGP_TRY(hipGetLastError());
    QFinal fin{};
    fin.ls = c.d_ls; fin.val = d_val; fin.grad = d_grad; fin.m = m; fin.D = D; fin.acq = acq;
    fin.kss = c.constant + c.noise; fin.y_mean = c.y_mean; fin.y_std = c.y_std;
    fin.sf = sf; fin.incumbent = incumbent; fin.param = param;
    fin.bell = d_val ? bell : Bell{nullptr, 0, nullptr};
    hipLaunchKernelGGL(q_reduce_kernel, dim3(D, m), dim3(256), 0, s, c.d_Xs, c.d_alpha, uq, ks, hw, v, w, red, N, Np, D, Dp, m,
                       mfma ? 1 : QCOLS_SPLIT, fin);
    TGP_TRY(hipGetLastError());
    if (!d_val || fin.bell.word) return hipSuccess;
    hipLaunchKernelGGL(q_finalize_kernel, dim3((m + 63) / 64), dim3(64), 0, s, red, fin);
    return hipGetLastError();
}

}  // namespace tgp
