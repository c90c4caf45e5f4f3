// refine_kernels.hip -- the gradient stage of the auxiliary optimiser on the device
// (turbo/modules/auxiliary_optimisers.py:69-112: L-BFGS-B from the best random
// candidates plus fresh random starts).
//
//   topk_kernel          the k best candidates of a sweep (value descending, lowest index on ties,
//                        NaN last): what `best_ids = argsort(random_y)[:start_from_best]` selects
//                        (auxiliary_optimisers.py:63-66, :77-79), without the (M,) vector leaving
//                        the GPU.  Each block keeps 16 values per thread in registers and extracts
//                        its k best by k block-wide arg-max rounds; levels repeat until one block.
//   refine_step_wave_kernel   one iteration of the projected L-BFGS of lbfgs_wave.hpp (memory 8, a
//                        line search with L-BFGS-B's two conditions) for EVERY restart at once, one
//                        wave per restart (a lane owns 1 coordinate up to D = 64, 4 up to D = 256);
//                        the objective and its gradient come from the
//                        batched closed-form kernels of query_kernels.hip (launch_query: the sums;
//                        value + gradient are formed here), so an iteration is one fixed launch
//                        sequence whatever the number of restarts.  Replaces SciPy's L-BFGS-B runs
//                        on the host (one Python thread per restart in round 1).
//   refine_step_kernel   the first form, one thread per restart with plain backtracking: D > 256 only.
//   small_refine_kernel  N <= 128, D <= 64: the WHOLE stage in one launch, a workgroup per restart
//                        with the model in LDS.
#include <hip/hip_runtime.h>
#include <math.h>

#include "lbfgs_wave.hpp"
#include "lds_opt_in.hpp"
#include "query_math.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#define TGP_TRY(x)                         \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

// ---- top-k ---------------------------------------------------------------------------------
constexpr int TOPK_PER_THREAD = 16;
constexpr int TOPK_SLICE = 256 * TOPK_PER_THREAD;     // 4096 entries per block
constexpr long long IDX_NONE = 0x7fffffffffffffffLL;

__device__ __forceinline__ bool topk_better(double v2, long long i2, double v, long long i) {
    return v2 > v || (v2 == v && i2 < i);
}

// vals: n entries; idx_in: their global indices (null: entry p of this level is candidate p).
// Block b writes its k best of entries [b * SLICE, (b + 1) * SLICE) to out_v / out_i [b * k ..],
// padded with (-inf, IDX_NONE) when the slice holds fewer.
__global__ __launch_bounds__(256) void topk_kernel(const double *__restrict__ vals,
                                                   const long long *__restrict__ idx_in, long n, int k,
                                                   double *__restrict__ out_v,
                                                   long long *__restrict__ out_i, double *__restrict__ done_host,
                                                   long long *__restrict__ besti, Bell bell) {
    __shared__ double sv[4];
    __shared__ long long si[4];
    __shared__ int sown[4], sslot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long base = (long)blockIdx.x * TOPK_SLICE;
    double v[TOPK_PER_THREAD];
    long long ix[TOPK_PER_THREAD];
#pragma unroll
    for (int r = 0; r < TOPK_PER_THREAD; ++r) {
        const long p = base + tid + 256L * r;
        double x = -INFINITY;
        long long id = IDX_NONE;
        if (p < n) {
            x = vals[p];
            id = idx_in ? idx_in[p] : (long long)p;
            if (isnan(x)) x = -INFINITY;            // NaN never beats a number
            if (id == IDX_NONE) x = -INFINITY;      // padding of the previous level
        }
        v[r] = x; ix[r] = id;
    }
    for (int round = 0; round < k; ++round) {
        double bv = -INFINITY;
        long long bi = IDX_NONE;
        int bs = 0;
#pragma unroll
        for (int r = 0; r < TOPK_PER_THREAD; ++r)
            if (topk_better(v[r], ix[r], bv, bi)) { bv = v[r]; bi = ix[r]; bs = r; }
        int own = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(bv, o, 64);
            const long long i2 = __shfl_xor(bi, o, 64);
            const int o2 = __shfl_xor(own, o, 64);
            const int s2 = __shfl_xor(bs, o, 64);
            if (topk_better(v2, i2, bv, bi)) { bv = v2; bi = i2; own = o2; bs = s2; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; sown[wave] = own; sslot[wave] = bs; }
        __syncthreads();
        double wv = sv[0];
        long long wi = si[0];
        int wo = sown[0], ws = sslot[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (topk_better(sv[w], si[w], wv, wi)) { wv = sv[w]; wi = si[w]; wo = sown[w]; ws = sslot[w]; }
        if (tid == 0) {
            out_v[(long)blockIdx.x * k + round] = wv;
            out_i[(long)blockIdx.x * k + round] = wi;
            if (done_host) { done_host[round] = wv; done_host[k + round] = (wi == IDX_NONE) ? -1.0 : (double)wi; }
        }
        if (tid == wo) {                             // the winner leaves the pool
#pragma unroll
            for (int r = 0; r < TOPK_PER_THREAD; ++r)
                if (r == ws) { v[r] = -INFINITY; ix[r] = IDX_NONE; }
        }
        __syncthreads();
    }
    if (done_host) {      // the final pass of a polled call (one workgroup): clamp count out, counters back at zero, ring
        if (tid == 0) {
            done_host[2 * k] = (double)besti[1];
            besti[0] = 0; besti[1] = 0; besti[3] = 0;
        }
        bell_ring(bell, 1);
    }
}

// On return the k best are at ws_v / ws_i [final_off ..]: the caller copies them out.
// ws_v / ws_i must hold 2 * ceil(M / SLICE) * k entries.
hipError_t launch_topk(Context &c, const double *d_vals, long M, int k, double *ws_v, long long *ws_i,
                       long *final_off, double *done_host, const Bell &bell) {
    const double *src_v = d_vals;
    const long long *src_i = nullptr;
    long n = M;
    long off = 0;
    const long cap_half = ((M + TOPK_SLICE - 1) / TOPK_SLICE) * (long)k;
    int level = 0;
    for (;;) {
        const long nb = (n + TOPK_SLICE - 1) / TOPK_SLICE;
        off = (level & 1) ? cap_half : 0;            // ping-pong halves of the workspace
        const bool last = nb == 1;
        hipLaunchKernelGGL(topk_kernel, dim3((unsigned)nb), dim3(256), 0, c.stream, src_v, src_i, n, k,
                           ws_v + off, ws_i + off, last ? done_host : nullptr, c.d_besti,
                           (last && done_host) ? bell : Bell{nullptr, 0, nullptr});
        TGP_TRY(hipGetLastError());
        if (nb == 1) break;
        src_v = ws_v + off; src_i = ws_i + off; n = nb * k;
        ++level;
    }
    *final_off = off;
    return hipSuccess;
}

// ---- projected L-BFGS, all restarts in lock-step ---------------------------------------------

// per-restart state, doubles:  [ x (D) | g (D) | d (D) | S (MEM x D) | Y (MEM x D) | rho (MEM) | scalars (8) |
//                                 x_lo (D) | g_lo (D) | line-search scalars (16) ]   (the tail: wave kernel only)
__host__ __device__ inline long rf_stride(int D) { return 5L * D + 2L * RF_MEM * D + RF_MEM + 24; }
// scalars: 0 phi (objective being MINIMISED = -acq), 1 t, 2 hist count, 3 hist head, 4 status
// (0 running, 1 converged, 2 line search failed), 5 iterations, 6 last accepted |delta phi|, 7 spare

struct RefineArgs {
    double *state; double *xt;          // (R, stride), (R, D) trial points = the next evaluation batch
    const double *val, *grad;           // acquisition value / gradient at xt (launch_query outputs)
    const double *lo, *hi;              // (D,) bounds
    int R, D, first;
    double pgtol, ftol;
    int *active;                        // number of restarts still running after this step
    int *active_next;                   // the next step's counter, zeroed here (the two alternate)
    // wave kernel only, red != nullptr: val / grad are not read; value and gradient come from
    // launch_query's sums [k.alpha, v.v, gm (D), gv (D)] per restart (what q_finalize_kernel would do)
    const double *red, *ls;
    double kss, y_mean, y_std, sf, incumbent, param;
    int acq;
    double *hist;                       // D > 1024 (a team of eight waves per restart): (R, 2, RF_MEM, RF_TEAM_LDH) history pairs in global memory
};


// One optimiser step for every restart, ONE WAVE per restart up to D = 1024: lane i owns coordinates i, i + 64,
// ... of every vector (x, g, d, the trial point, the history pairs), the scalars are computed redundantly by
// every lane, dot products are wave reductions.  DK coordinates per lane: 1 (D <= 64, four restarts per
// workgroup), 4 (D <= 256) or 16 (D <= 1024; the history pairs then fill 128 KB of LDS), one restart per
// workgroup.  Beyond D = 1024 (NW = 8, DK = 8: up to the library's limit D = 4096) a TEAM of eight waves shares a
// restart: 512 threads, reductions meet in LDS (RfTeam), the history lives in global memory (a.hist).
// (The first, one-thread-per-restart form of round 1 -- sufficient decrease only, 67 us per step at D = 16
// against 5 us here -- is gone: every D the library accepts now gets the line search with the curvature condition.)
constexpr int RF_TEAM_NW = 8, RF_TEAM_DK = 8;                    // the team of D > 1024: waves, coordinates per lane
constexpr int RF_TEAM_LDH = 64 * RF_TEAM_NW * RF_TEAM_DK;        // = its history row length (LDH below) = the largest D, 4096
template <int DK, int NW = 1>
__global__ __launch_bounds__(DK == 1 ? 256 : 64 * NW) void refine_step_wave_kernel(RefineArgs a) {
    constexpr int WPB = DK == 1 ? 4 : 1;               // restarts per workgroup
    constexpr int TS = 64 * NW;                        // threads per restart
    constexpr int LDH = TS * DK;                       // row length of the history pairs
    const int lane = NW == 1 ? (threadIdx.x & 63) : (int)threadIdx.x;   // index inside the restart's team
    const int wv = NW == 1 ? (threadIdx.x >> 6) : 0;
    const int r = blockIdx.x * WPB + wv;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.active_next = 0;
    if (r >= a.R) return;
    const int D = a.D;
    bool on[DK];
    int li[DK];
    RF_EACH(k) { on[k] = lane + TS * k < D; li[k] = on[k] ? lane + TS * k : 0; }
    double *st = a.state + (long)r * rf_stride(D);
    double *x = st, *g = st + D, *d = st + 2 * D, *S = st + 3 * D, *Y = S + (long)RF_MEM * D;
    double *rho = Y + (long)RF_MEM * D, *sc = rho + RF_MEM;
    double *xlo = sc + 8, *glo = xlo + D, *sl = glo + D;
    double *xt = a.xt + (long)r * D;
    double lo_i[DK], hi_i[DK], xt_i[DK], gt_i[DK];
    RF_EACH(k) { lo_i[k] = a.lo[li[k]]; hi_i[k] = a.hi[li[k]]; xt_i[k] = on[k] ? xt[li[k]] : 0.0; }
    double phit;                                        // phi = -acq and its gradient at the trial point
    if (a.red) {
        const double *rq = a.red + (long)r * (2 + 2 * D);
        const double mu = a.y_std * rq[0] + a.y_mean;
        double var = a.kss - rq[1];
        const bool pos = var > 0.0;
        if (!pos) var = 0.0;
        const double sn = sqrt(var);
        const AcqCoef ac = acq_coef(a.acq, mu, a.y_std * sn, a.sf, a.incumbent, a.param);
        RF_EACH(k) {
            const double ls_i = a.ls[li[k]];
            const double dmu = -a.y_std * rq[2 + li[k]] / ls_i;
            const double dvar = 2.0 * rq[2 + D + li[k]] / ls_i;
            const double dsig = pos && sn > 0.0 ? a.y_std * dvar / (2.0 * sn) : 0.0;
            gt_i[k] = on[k] ? -(ac.cm * dmu + ac.cs * dsig) : 0.0;
        }
        phit = -ac.a;
    } else {
        RF_EACH(k) gt_i[k] = on[k] ? -a.grad[(long)r * D + li[k]] : 0.0;
        phit = -a.val[r];
    }
    // the history pairs: in LDS (one wave per restart), each lane reading back only what it wrote itself (no
    // barrier); an eight-wave team keeps them in global memory, where they also persist between the launches
    __shared__ double hist[NW == 1 ? WPB : 1][2][NW == 1 ? RF_MEM : 1][NW == 1 ? LDH : 1];
    double (*Sv)[LDH], (*Yv)[LDH];
    if (NW == 1) {
        Sv = reinterpret_cast<double (*)[LDH]>(&hist[wv][0][0][0]);
        Yv = reinterpret_cast<double (*)[LDH]>(&hist[wv][1][0][0]);
    } else {
        double *hb = a.hist + (long)r * 2 * RF_MEM * LDH;
        Sv = reinterpret_cast<double (*)[LDH]>(hb);
        Yv = reinterpret_cast<double (*)[LDH]>(hb + (long)RF_MEM * LDH);
    }
    __shared__ double rhs[WPB][RF_MEM];                // (every lane writes the same value before it reads it)
    __shared__ double team_red[2 * NW];
    double *rh = rhs[wv];
    RfTeam<NW> tm{team_red, (int)(threadIdx.x >> 6), 0};
    RfWaveT<DK> w{};
    if (!a.first) {
        RF_EACH(k) {
            w.x_i[k] = on[k] ? x[li[k]] : 0.0; w.g_i[k] = on[k] ? g[li[k]] : 0.0; w.d_i[k] = on[k] ? d[li[k]] : 0.0;
            w.xlo_i[k] = on[k] ? xlo[li[k]] : 0.0; w.glo_i[k] = on[k] ? glo[li[k]] : 0.0;
        }
        w.phi = sc[0]; w.t = sc[1]; w.cnt = (int)sc[2]; w.head = (int)sc[3]; w.status = (int)sc[4];
        w.iters = (int)sc[5]; w.last = sc[6];
        w.dphi0 = sl[0]; w.t_cap = sl[1]; w.t_lo = sl[2]; w.phi_lo = sl[3]; w.dphi_lo = sl[4]; w.t_hi = sl[5];
        w.phi_hi = sl[6]; w.stage = (int)sl[7]; w.n_ls = (int)sl[8];
#pragma unroll
        for (int m = 0; m < RF_MEM; ++m) {
            if (NW == 1) {
                RF_EACH(k) {
                    Sv[m][lane + TS * k] = on[k] ? S[(long)m * D + li[k]] : 0.0;
                    Yv[m][lane + TS * k] = on[k] ? Y[(long)m * D + li[k]] : 0.0;
                }
            }
            if (lane < 64 || NW == 1) rh[m] = rho[m];
        }
    } else {
#pragma unroll
        for (int m = 0; m < RF_MEM; ++m) rh[m] = 0.0;
    }
    if (NW > 1) __syncthreads();                       // rh is shared by the team's waves
    const int stored = rf_wave_step<DK, NW>(w, xt_i, gt_i, phit, a.first != 0, on, lane, lo_i, hi_i, a.pgtol, a.ftol, Sv, Yv, rh, tm);
    if (stored >= 0 && NW == 1) {
        RF_EACH(k) if (on[k]) {
            S[(long)stored * D + li[k]] = Sv[stored][lane + TS * k];
            Y[(long)stored * D + li[k]] = Yv[stored][lane + TS * k];
        }
    }
    RF_EACH(k) if (on[k]) {
        x[li[k]] = w.x_i[k]; g[li[k]] = w.g_i[k]; d[li[k]] = w.d_i[k]; xt[li[k]] = xt_i[k];
        xlo[li[k]] = w.xlo_i[k]; glo[li[k]] = w.glo_i[k];
    }
    if (NW > 1) __syncthreads();                       // every wave has read rh for the last time
    if (lane == 0) {
        sl[0] = w.dphi0; sl[1] = w.t_cap; sl[2] = w.t_lo; sl[3] = w.phi_lo; sl[4] = w.dphi_lo; sl[5] = w.t_hi;
        sl[6] = w.phi_hi; sl[7] = (double)w.stage; sl[8] = (double)w.n_ls;
        sc[0] = w.phi; sc[1] = w.t; sc[2] = (double)w.cnt; sc[3] = (double)w.head; sc[4] = (double)w.status;
        sc[5] = (double)w.iters; sc[6] = w.last;
#pragma unroll
        for (int m = 0; m < RF_MEM; ++m) rho[m] = rh[m];
        if (w.status == 0) atomicAdd(a.active, 1);
    }
}

// ------------------------------------------------------------------------------------------
// Small problems (N <= 128, D <= 64): the WHOLE gradient stage in one launch.  One workgroup per
// restart keeps Linv in LDS (rows padded by one element), evaluates the acquisition and its
// gradient at the trial point with the closed forms of query_kernels.hip, and wave 0 takes the
// optimiser step above; the loop ends when the restart has finished or after max_iter
// evaluations.  No host round trips, no lock-step between restarts.
// ------------------------------------------------------------------------------------------
struct SmallRefineArgs {
    const double *Xs, *alpha, *Linv, *ls;   // scaled training points (N, Dp), alpha, Linv (row stride ldl), length scales (D)
    const double *x0, *lo, *hi;             // (R, D) starts, (D) bounds
    double *x_out, *v_out, *info;           // (R, D), (R), (3 R): status, accepted steps, evaluations
    int N, Np, ldl, D, Dp, max_iter, acq;   // Np: N rounded up to 64 or 128
    double constant, kss, y_mean, y_std, sf, incumbent, param, pgtol, ftol;
};

__host__ __device__ inline size_t small_refine_lds_doubles(int Np, int N, int Dp) {
    // Lt (packed lower triangle) | Xs (N rows of Dp + 1) | al, ks, hw, vs, wv (5 Np) | u (64) | pm, pv (2 x 4 x 64) |
    // ps (8) | hist (2 x RF_MEM x 64) | rh (RF_MEM) | flag (2)
    return (size_t)Np * (Np + 1) / 2 + (size_t)N * (Dp + 1) + 5 * (size_t)Np + 64 + 512 + 8 + 2 * RF_MEM * 64 + RF_MEM + 2;
}

// The fitted state of a small problem in LDS and ONE evaluation of the posterior sums at the point u = x / l
// (shared by the one-launch optimiser below and by small_query_kernel, the one-launch tgp_acq_grad of round 6):
//   stage()  Linv (packed lower triangle), the scaled training points, alpha -> LDS, once per workgroup
//   sums()   k, hw -> v = Linv k -> w = Linv^T v -> per-wave shares of gm, gv (pm, pv) and of k.alpha, v.v (ps);
//            starts and ends with a barrier; u is read, everything else written
struct SmallEval {
    double *Lt, *Xl, *al, *ks, *hw, *vs, *wv, *u, *pm, *pv, *ps;
    int N, Np, Dp, LDX;
    double constant;

    // sm: small_refine_lds_doubles() doubles; returns the first double behind ps (the optimiser's history lives there)
    __device__ __forceinline__ double *carve(double *sm, int N_, int Np_, int Dp_, double constant_) {
        N = N_; Np = Np_; Dp = Dp_; LDX = Dp_ + 1; constant = constant_;
        Lt = sm;                                   // row i at i (i + 1) / 2
        Xl = Lt + (size_t)Np * (Np + 1) / 2;
        al = Xl + (size_t)N * LDX; ks = al + Np; hw = ks + Np; vs = hw + Np; wv = vs + Np;
        u = wv + Np;
        pm = u + 64; pv = pm + 256; ps = pv + 256;
        return ps + 8;
    }
    __device__ __forceinline__ void stage(const double *__restrict__ Linv, int ldl, const double *__restrict__ Xs,
                                          const double *__restrict__ alpha) const {
        const int tid = threadIdx.x;
        // (eight loads in flight per thread: one at a time, each followed by its LDS store, the 16-64 dependent round
        // trips to L2 were most of a one-evaluation kernel)
        const int sh = Np == 64 ? 6 : 7, total = Np * Np;     // Np is 64 or 128
        for (int e0 = tid; e0 < total; e0 += 256 * 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + 256 * k, i = e >> sh, j = e & (Np - 1);
                v[k] = (e < total && j <= i && i < N) ? Linv[(long)i * ldl + j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + 256 * k, i = e >> sh, j = e & (Np - 1);
                if (e < total && j <= i) Lt[i * (i + 1) / 2 + j] = v[k];
            }
        }
        for (int e = tid; e < N * Dp; e += 256) {
            const int j = e / Dp, d = e - j * Dp;
            Xl[j * LDX + d] = Xs[e];
        }
        for (int j = tid; j < Np; j += 256) al[j] = (j < N) ? alpha[j] : 0.0;
    }
    template <int KIND>
    __device__ __forceinline__ void sums() const {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        // TPR threads per training point / row / column: 2 (Np = 128) or 4 (Np = 64), neighbours in a quad
        const int tpr = 256 / Np, sub = tid & (tpr - 1), jrow = tid / tpr;
        const int jw = Np / 4;                            // training points per wave in the gradient sums
        auto quad_sum = [&](double v) {
            v += rf_dpp<0xB1>(v);
            if (tpr == 4) v += rf_dpp<0x4E>(v);
            return v;
        };
        __syncthreads();
        // k_j = c k0(r_j), hw_j = c h(r_j)
        {
            double d2 = 0.0;
            if (jrow < N) {
                const double *xj = Xl + jrow * LDX;
                for (int d = sub; d < Dp; d += tpr) {
                    const double df = u[d] - xj[d];
                    d2 = fma(df, df, d2);
                }
            }
            d2 = quad_sum(d2);
            if (sub == 0) {
                ks[jrow] = (jrow < N) ? kernel_value<double, KIND>(d2, constant) : 0.0;
                hw[jrow] = (jrow < N) ? constant * h_weight<KIND>(d2) : 0.0;
            }
        }
        __syncthreads();
        // v = Linv k
        {
            const double *row = Lt + jrow * (jrow + 1) / 2;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = sub;
            for (; j + 3 * tpr <= jrow; j += 4 * tpr) {
                s0 = fma(row[j], ks[j], s0);
                s1 = fma(row[j + tpr], ks[j + tpr], s1);
                s2 = fma(row[j + 2 * tpr], ks[j + 2 * tpr], s2);
                s3 = fma(row[j + 3 * tpr], ks[j + 3 * tpr], s3);
            }
            for (; j <= jrow; j += tpr) s0 = fma(row[j], ks[j], s0);
            const double s = quad_sum((s0 + s1) + (s2 + s3));
            if (sub == 0) vs[jrow] = s;
        }
        __syncthreads();
        // w = Linv^T v = K^-1 k
        {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int i = jrow + sub;
            for (; i + 3 * tpr < N; i += 4 * tpr) {
                s0 = fma(Lt[i * (i + 1) / 2 + jrow], vs[i], s0);
                s1 = fma(Lt[(i + tpr) * (i + tpr + 1) / 2 + jrow], vs[i + tpr], s1);
                s2 = fma(Lt[(i + 2 * tpr) * (i + 2 * tpr + 1) / 2 + jrow], vs[i + 2 * tpr], s2);
                s3 = fma(Lt[(i + 3 * tpr) * (i + 3 * tpr + 1) / 2 + jrow], vs[i + 3 * tpr], s3);
            }
            for (; i < N; i += tpr) s0 = fma(Lt[i * (i + 1) / 2 + jrow], vs[i], s0);
            const double s = quad_sum((s0 + s1) + (s2 + s3));
            if (sub == 0) wv[jrow] = s;
        }
        __syncthreads();
        // the wave's share of: gm_d = sum_j alpha_j hw_j (u_d - xs_jd), gv_d = sum_j w_j hw_j (u_d - xs_jd),
        // mun = k.alpha, qv = v.v
        {
            double gm = 0.0, gv = 0.0;
            const int ld = lane < Dp ? lane : 0;
            const double ud = u[ld];
            const int j1 = min(N, (wave + 1) * jw);
#pragma unroll 4
            for (int j = wave * jw; j < j1; ++j) {
                const double t = hw[j] * (ud - Xl[j * LDX + ld]);
                gm = fma(al[j], t, gm);
                gv = fma(wv[j], t, gv);
            }
            pm[wave * 64 + lane] = gm;
            pv[wave * 64 + lane] = gv;
            const int j = wave * jw + lane;
            const bool in = lane < jw && j < N;
            const double mun = rf_wsum(in ? ks[j] * al[j] : 0.0);
            const double qv = rf_wsum(in ? vs[j] * vs[j] : 0.0);
            if (lane == 0) { ps[2 * wave] = mun; ps[2 * wave + 1] = qv; }
        }
        __syncthreads();
    }
    // wave 0, lane = dimension: value, and the gradient entry of this lane (0 beyond D), from the shares sums() left
    __device__ __forceinline__ AcqCoef finish(int lane, bool on, double ls_i, double kss, double y_mean, double y_std,
                                              int acq, double sf, double incumbent, double param, double &grad_i) const {
        const double mun = (ps[0] + ps[2]) + (ps[4] + ps[6]);
        const double qv = (ps[1] + ps[3]) + (ps[5] + ps[7]);
        const double gm = (pm[lane] + pm[64 + lane]) + (pm[128 + lane] + pm[192 + lane]);
        const double gv = (pv[lane] + pv[64 + lane]) + (pv[128 + lane] + pv[192 + lane]);
        const double mu = y_std * mun + y_mean;
        double var = kss - qv;
        const bool pos = var > 0.0;
        if (!pos) var = 0.0;
        const double sn = sqrt(var);
        const double sigma = y_std * sn;
        const AcqCoef ac = acq_coef(acq, mu, sigma, sf, incumbent, param);
        const double dmu = -y_std * gm / ls_i;
        const double dvar = 2.0 * gv / ls_i;
        const double dsig = pos && sn > 0.0 ? y_std * dvar / (2.0 * sn) : 0.0;
        grad_i = on ? (ac.cm * dmu + ac.cs * dsig) : 0.0;
        return ac;
    }
};

template <int KIND>
__global__ __launch_bounds__(256) void small_refine_kernel(SmallRefineArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int N = p.N, Np = p.Np, D = p.D;
    SmallEval ev;
    double *behind = ev.carve(sm, N, Np, p.Dp, p.constant);
    double *u = ev.u;
    double (*Sv)[64] = reinterpret_cast<double (*)[64]>(behind);
    double (*Yv)[64] = Sv + RF_MEM;
    double *rh = reinterpret_cast<double *>(Yv + RF_MEM);
    double *flag = rh + RF_MEM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    ev.stage(p.Linv, p.ldl, p.Xs, p.alpha);
    if (tid < RF_MEM) rh[tid] = 0.0;
    const bool on = lane < D;
    const int li = on ? lane : 0;
    const double lo_i = p.lo[li], hi_i = p.hi[li], ls_i = p.ls[li];
    double xt_i = on ? rf_clip(p.x0[(long)r * D + li], lo_i, hi_i) : 0.0;   // (wave 0's copy is the one that counts)
    if (wave == 0) u[lane] = on ? xt_i / ls_i : 0.0;
    RfWave w{};
    int evals = 0;
    for (int it = 0; it <= p.max_iter; ++it) {
        ev.sums<KIND>();
        if (wave == 0) {
            double g_i;
            const AcqCoef ac = ev.finish(lane, on, ls_i, p.kss, p.y_mean, p.y_std, p.acq, p.sf, p.incumbent, p.param, g_i);
            const double gt_i = on ? -g_i : 0.0;      // gradient of phi = -acq
            (void)rf_wave_step(w, xt_i, gt_i, -ac.a, it == 0, on, lane, lo_i, hi_i, p.pgtol, p.ftol, Sv, Yv, rh);
            ++evals;
            u[lane] = on ? xt_i / ls_i : 0.0;
            if (lane == 0) flag[0] = (w.status != 0) ? 1.0 : 0.0;
        }
        __syncthreads();
        if (flag[0] != 0.0) break;
    }
    if (wave == 0) {
        if (on) p.x_out[(long)r * D + li] = w.x_i[0];
        if (lane == 0) {
            p.v_out[r] = -w.phi;
            p.info[3 * r] = (double)w.status;
            p.info[3 * r + 1] = (double)w.iters;
            p.info[3 * r + 2] = (double)evals;
        }
    }
}

// ------------------------------------------------------------------------------------------
// small_query_kernel (round 6): tgp_acq_grad for a small problem in ONE launch -- acquisition value and gradient at
// m points, one workgroup per point, the evaluation of the optimiser above (SmallEval).  The points are read from, the
// results written to, device-mapped host memory, and the last workgroup rings the call's doorbell (doorbell.hpp): no
// memcpy, no stream synchronisation.  What round 5 ran here -- five launches of the general query kernels between a
// pageable H2D copy and two D2H copies -- cost 75-83 us a call whatever the batch (profiles/r05_gradient_stage.jsonl),
// and the default gradient stage (tgp_acq_lbfgsb; turbo/modules/auxiliary_optimisers.py:69-112) is tens of such calls.
// A point's value does not depend on the batch it travels in (its workgroup sees nothing of the others).
// ------------------------------------------------------------------------------------------
struct SmallQueryArgs {
    const double *Xs, *alpha, *Linv, *ls;
    const double *xq;          // (m, D)
    double *val, *grad;        // (m), (m, D)
    int N, Np, ldl, D, Dp, acq;
    double constant, kss, y_mean, y_std, sf, incumbent, param;
    Bell bell;
};

template <int KIND>
__global__ __launch_bounds__(256) void small_query_kernel(SmallQueryArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    bell_start(p.bell);
    SmallEval ev;
    (void)ev.carve(sm, p.N, p.Np, p.Dp, p.constant);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x, D = p.D;
    const bool on = lane < D;
    const int li = on ? lane : 0;
    const double x_i = on ? p.xq[(long)q * D + li] : 0.0;      // (a PCIe round trip: issued in front of the staging)
    const double ls_i = p.ls[li];
    ev.stage(p.Linv, p.ldl, p.Xs, p.alpha);
    if (wave == 0) ev.u[lane] = on ? x_i / ls_i : 0.0;
    ev.sums<KIND>();
    if (wave == 0) {
        double g_i;
        const AcqCoef ac = ev.finish(lane, on, ls_i, p.kss, p.y_mean, p.y_std, p.acq, p.sf, p.incumbent, p.param, g_i);
        if (on) p.grad[(long)q * D + li] = g_i;
        if (lane == 0) p.val[q] = ac.a;
    }
    bell_ring(p.bell, gridDim.x);
}

hipError_t launch_small_query(Context &c, const double *d_xq, int m, int acq, double sf, double incumbent, double param,
                              double *d_val, double *d_grad, const Bell &bell) {
    SmallQueryArgs a{};
    a.Xs = c.d_Xs; a.alpha = c.d_alpha; a.Linv = c.d_Linv; a.ls = c.d_ls;
    a.xq = d_xq; a.val = d_val; a.grad = d_grad;
    a.N = (int)c.N; a.Np = (int)((c.N + NB - 1) / NB) * NB; a.ldl = (int)c.Np;
    a.D = (int)c.D; a.Dp = (int)c.Dp; a.acq = acq;
    a.constant = c.constant; a.kss = c.constant + c.noise; a.y_mean = c.y_mean; a.y_std = c.y_std;
    a.sf = sf; a.incumbent = incumbent; a.param = param;
    a.bell = bell;
    void (*k)(SmallQueryArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_query_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_query_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_query_kernel<TGP_MATERN32>; break;
        default: k = small_query_kernel<TGP_MATERN52>; break;
    }
    const size_t lds = small_refine_lds_doubles(a.Np, (int)c.N, (int)c.Dp) * sizeof(double);
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, 160 * 1024));   // (once per device: the most)
    hipLaunchKernelGGL(k, dim3((unsigned)m), dim3(256), lds, c.stream, a);
    return hipGetLastError();
}

// x0 clipped into the bounds -> the first evaluation batch
__global__ void refine_clip_kernel(double *xt, const double *lo, const double *hi, long n, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xt[i] = rf_clip(xt[i], lo[i % D], hi[i % D]);
}

// results: x (R, D), value = -phi, status, iterations
__global__ void refine_collect_kernel(const double *state, int R, int D, double *x_out, double *v_out,
                                      double *info_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double *st = state + (long)r * rf_stride(D);
    const double *sc = st + 3L * D + 2L * RF_MEM * D + RF_MEM;
    for (int i = 0; i < D; ++i) x_out[(long)r * D + i] = st[i];
    v_out[r] = -sc[0];
    info_out[2 * r] = sc[4];
    info_out[2 * r + 1] = sc[5];
}

hipError_t launch_refine_clip(Context &c, double *d_xt, const double *d_lo, const double *d_hi, int R) {
    const long n = (long)R * c.D;
    hipLaunchKernelGGL(refine_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_xt, d_lo, d_hi, n, (int)c.D);
    return hipGetLastError();
}

hipError_t launch_refine_step(Context &c, double *d_state, double *d_xt, const double *d_val,
                              const double *d_grad, const double *d_lo, const double *d_hi, int R,
                              int it, double pgtol, double ftol, int *d_active, const double *d_red, int acq,
                              double sf, double incumbent, double param) {
    RefineArgs a{};
    if (d_red) {
        a.red = d_red; a.ls = c.d_ls; a.kss = c.constant + c.noise; a.y_mean = c.y_mean; a.y_std = c.y_std;
        a.sf = sf; a.incumbent = incumbent; a.param = param; a.acq = acq;
    }
    a.state = d_state; a.xt = d_xt; a.val = d_val; a.grad = d_grad; a.lo = d_lo; a.hi = d_hi;
    a.R = R; a.D = (int)c.D; a.first = it == 0 ? 1 : 0; a.pgtol = pgtol; a.ftol = ftol;
    a.hist = d_state + (long)R * rf_stride((int)c.D);   // behind the per-restart states (only D > 1024 uses it)
    // two counters of running restarts, used in turn: step it counts into d_active[it & 1] and zeroes the other
    a.active = d_active + (it & 1);
    a.active_next = d_active + ((it + 1) & 1);
    if (it == 0) TGP_TRY(hipMemsetAsync(d_active, 0, 2 * sizeof(int), c.stream));
    if (c.D <= 64)
        hipLaunchKernelGGL(refine_step_wave_kernel<1>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, c.stream, a);
    else if (c.D <= 256)
        hipLaunchKernelGGL(refine_step_wave_kernel<4>, dim3((unsigned)R), dim3(64), 0, c.stream, a);
    else if (c.D <= 1024)
        hipLaunchKernelGGL(refine_step_wave_kernel<16>, dim3((unsigned)R), dim3(64), 0, c.stream, a);
    else   // D <= 4096 (tgp_fit s limit): eight waves per restart
        hipLaunchKernelGGL((refine_step_wave_kernel<RF_TEAM_DK, RF_TEAM_NW>), dim3((unsigned)R), dim3(64 * RF_TEAM_NW), 0, c.stream, a);
    return hipGetLastError();
}

hipError_t launch_refine_collect(Context &c, const double *d_state, int R, double *d_x, double *d_v, double *d_info) {
    hipLaunchKernelGGL(refine_collect_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, c.stream, d_state, R, (int)c.D, d_x, d_v, d_info);
    return hipGetLastError();
}

long refine_state_stride(int D) { return rf_stride(D); }
// doubles behind the R states: the history pairs of the eight-wave teams (D > 1024)
long refine_hist_doubles(int D, int R) { return D > 1024 ? (long)R * 2 * RF_MEM * RF_TEAM_LDH : 0; }

// the whole stage for a small problem: d_x0 (R, D) starts -> d_x (R, D), d_v (R), d_info (3 R)
bool small_refine_fits(const Context &c) {
    const int np = (int)((c.N + NB - 1) / NB) * NB;
    return c.N <= 2 * NB && c.D <= 64 && small_refine_lds_doubles(np, (int)c.N, (int)c.Dp) * sizeof(double) <= 160 * 1024;
}

hipError_t launch_small_refine(Context &c, const double *d_x0, const double *d_lo, const double *d_hi, int R,
                               int acq, double sf, double incumbent, double param, int max_iter,
                               double pgtol, double ftol, double *d_x, double *d_v, double *d_info) {
    SmallRefineArgs a{};
    a.Xs = c.d_Xs; a.alpha = c.d_alpha; a.Linv = c.d_Linv; a.ls = c.d_ls;
    a.x0 = d_x0; a.lo = d_lo; a.hi = d_hi; a.x_out = d_x; a.v_out = d_v; a.info = d_info;
    a.N = (int)c.N; a.Np = (int)((c.N + NB - 1) / NB) * NB; a.ldl = (int)c.Np;
    a.D = (int)c.D; a.Dp = (int)c.Dp; a.max_iter = max_iter; a.acq = acq;
    a.constant = c.constant; a.kss = c.constant + c.noise; a.y_mean = c.y_mean; a.y_std = c.y_std;
    a.sf = sf; a.incumbent = incumbent; a.param = param; a.pgtol = pgtol; a.ftol = ftol;
    void (*k)(SmallRefineArgs);
    switch (c.kernel) {
        case TGP_RBF: k = small_refine_kernel<TGP_RBF>; break;
        case TGP_MATERN12: k = small_refine_kernel<TGP_MATERN12>; break;
        case TGP_MATERN32: k = small_refine_kernel<TGP_MATERN32>; break;
        default: k = small_refine_kernel<TGP_MATERN52>; break;
    }
    const size_t lds = small_refine_lds_doubles(a.Np, (int)c.N, (int)c.Dp) * sizeof(double);
    static LdsOptIn opt_in[4];
    TGP_TRY(opt_in[c.kernel & 3].ensure(reinterpret_cast<const void *>(k), c.device, 160 * 1024));   // (once per device: the most)
    hipLaunchKernelGGL(k, dim3((unsigned)R), dim3(256), lds, c.stream, a);
    return hipGetLastError();
}

}  // namespace tgp
