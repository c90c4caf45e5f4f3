// refine_kernels.hip -- the gradient stage of the auxiliary optimiser on the device
// (turbo/modules/auxiliary_optimisers.py:69-112: L-BFGS-B from the best random
// candidates plus fresh random starts).
//
//   topk_kernel          the k best candidates of a sweep (value descending, lowest index on ties,
//                        NaN last): what `best_ids = argsort(random_y)[:start_from_best]` selects
//                        (auxiliary_optimisers.py:63-66, :77-79), without the (M,) vector leaving
//                        the GPU.  Each block keeps 16 values per thread in registers and extracts
//                        its k best by k block-wide arg-max rounds; levels repeat until one block.
//   refine_step_kernel   one iteration of a projected L-BFGS (memory 8, Armijo backtracking) for
//                        EVERY restart at once, one thread per restart; the objective and its
//                        gradient come from the batched closed-form kernels of query_kernels.hip
//                        (launch_query), so an iteration is one fixed launch sequence whatever
//                        the number of restarts.  Replaces SciPy's L-BFGS-B runs on the host
//                        (one Python thread per restart in round 1).
#include <hip/hip_runtime.h>
#include <math.h>

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
                                                   long long *__restrict__ out_i) {
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
        }
        if (tid == wo) {                             // the winner leaves the pool
#pragma unroll
            for (int r = 0; r < TOPK_PER_THREAD; ++r)
                if (r == ws) { v[r] = -INFINITY; ix[r] = IDX_NONE; }
        }
        __syncthreads();
    }
}

// On return the k best are at ws_v / ws_i [final_off ..]: the caller copies them out.
// ws_v / ws_i must hold 2 * ceil(M / SLICE) * k entries.
hipError_t launch_topk(Context &c, const double *d_vals, long M, int k, double *ws_v, long long *ws_i,
                       long *final_off) {
    const double *src_v = d_vals;
    const long long *src_i = nullptr;
    long n = M;
    long off = 0;
    const long cap_half = ((M + TOPK_SLICE - 1) / TOPK_SLICE) * (long)k;
    int level = 0;
    for (;;) {
        const long nb = (n + TOPK_SLICE - 1) / TOPK_SLICE;
        off = (level & 1) ? cap_half : 0;            // ping-pong halves of the workspace
        hipLaunchKernelGGL(topk_kernel, dim3((unsigned)nb), dim3(256), 0, c.stream, src_v, src_i, n, k,
                           ws_v + off, ws_i + off);
        TGP_TRY(hipGetLastError());
        if (nb == 1) break;
        src_v = ws_v + off; src_i = ws_i + off; n = nb * k;
        ++level;
    }
    *final_off = off;
    return hipSuccess;
}

// ---- projected L-BFGS, all restarts in lock-step ---------------------------------------------
constexpr int RF_MEM = 8;                         // history pairs

// per-restart state, doubles:  [ x (D) | g (D) | d (D) | S (MEM x D) | Y (MEM x D) | rho (MEM) | scalars (8) ]
__host__ __device__ inline long rf_stride(int D) { return 3L * D + 2L * RF_MEM * D + RF_MEM + 8; }
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
};

__device__ inline double rf_clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(64) void refine_step_kernel(RefineArgs a) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r == 0) *a.active_next = 0;
    if (r >= a.R) return;
    const int D = a.D;
    double *st = a.state + (long)r * rf_stride(D);
    double *x = st, *g = st + D, *d = st + 2 * D, *S = st + 3 * D, *Y = S + (long)RF_MEM * D;
    double *rho = Y + (long)RF_MEM * D, *sc = rho + RF_MEM;
    double *xt = a.xt + (long)r * D;
    const double *gt_acq = a.grad + (long)r * D;
    const double phit = -a.val[r];
    bool new_dir = false;
    if (a.first) {
        for (int i = 0; i < D; ++i) { x[i] = xt[i]; g[i] = -gt_acq[i]; }
        sc[0] = phit; sc[2] = 0.0; sc[3] = 0.0; sc[4] = isfinite(phit) ? 0.0 : 2.0; sc[5] = 0.0; sc[6] = INFINITY;
        new_dir = true;
    } else if (sc[4] == 0.0) {
        // sufficient decrease along the PROJECTED step s = xt - x
        double slope = 0.0;
        for (int i = 0; i < D; ++i) slope = fma(g[i], xt[i] - x[i], slope);
        if (isfinite(phit) && phit <= sc[0] + 1e-4 * slope) {
            // accept: curvature pair, new iterate
            double sy = 0.0, yy = 0.0, ss = 0.0;
            const int head = (int)sc[3];
            double *Sh = S + (long)head * D, *Yh = Y + (long)head * D;
            for (int i = 0; i < D; ++i) {
                const double s_i = xt[i] - x[i], y_i = -gt_acq[i] - g[i];
                Sh[i] = s_i; Yh[i] = y_i;
                sy = fma(s_i, y_i, sy); yy = fma(y_i, y_i, yy); ss = fma(s_i, s_i, ss);
            }
            if (sy > 2.2e-16 * yy && sy > 0.0) {             // keep the pair (as L-BFGS-B's curvature test)
                rho[head] = 1.0 / sy;
                sc[3] = (double)((head + 1) % RF_MEM);
                sc[2] = fmin(sc[2] + 1.0, (double)RF_MEM);
            }
            const double dphi = sc[0] - phit;
            sc[6] = dphi;
            for (int i = 0; i < D; ++i) { x[i] = xt[i]; g[i] = -gt_acq[i]; }
            const double scale = fmax(fmax(fabs(sc[0]), fabs(phit)), 1.0);
            sc[0] = phit;
            sc[5] += 1.0;
            if (dphi <= a.ftol * scale) sc[4] = 1.0;         // relative reduction below factr * eps
            new_dir = true;
        } else {
            // backtrack: minimiser of the quadratic through phi(x), its slope and phi(xt), kept
            // inside [0.1, 0.5] of the failed step
            double theta = 0.5;
            const double denom = 2.0 * (phit - sc[0] - slope);
            if (isfinite(phit) && denom > 0.0 && slope < 0.0) theta = fmin(0.5, fmax(0.1, -slope / denom));
            sc[1] *= theta;
            if (sc[1] < 1e-12) {
                sc[4] = (sc[5] > 0.0) ? 1.0 : 2.0;           // no further progress possible from here
            } else {
                for (int i = 0; i < D; ++i) xt[i] = rf_clip(fma(sc[1], d[i], x[i]), a.lo[i], a.hi[i]);
            }
        }
    }
    if (new_dir && sc[4] == 0.0) {
        // projected gradient: zero when x is a constrained stationary point
        double pg = 0.0;
        for (int i = 0; i < D; ++i) pg = fmax(pg, fabs(x[i] - rf_clip(x[i] - g[i], a.lo[i], a.hi[i])));
        if (pg <= a.pgtol) {
            sc[4] = 1.0;
        } else {
            // two-loop recursion on the free variables (bound variables whose gradient pushes
            // outward stay put)
            double al[RF_MEM];
            const int cnt = (int)sc[2], head = (int)sc[3];
            double gn = 0.0;
            for (int i = 0; i < D; ++i) {
                const bool fixed = (x[i] <= a.lo[i] && g[i] > 0.0) || (x[i] >= a.hi[i] && g[i] < 0.0);
                d[i] = fixed ? 0.0 : g[i];
                gn = fma(d[i], d[i], gn);
            }
            for (int k = 0; k < cnt; ++k) {
                const int j = (head - 1 - k + 2 * RF_MEM) % RF_MEM;
                double sq = 0.0;
                for (int i = 0; i < D; ++i) sq = fma(S[(long)j * D + i], d[i], sq);
                al[k] = rho[j] * sq;
                for (int i = 0; i < D; ++i) d[i] = fma(-al[k], Y[(long)j * D + i], d[i]);
            }
            double gamma = 1.0;
            if (cnt > 0) {
                const int j = (head - 1 + RF_MEM) % RF_MEM;
                double yy = 0.0;
                for (int i = 0; i < D; ++i) yy = fma(Y[(long)j * D + i], Y[(long)j * D + i], yy);
                gamma = 1.0 / (rho[j] * yy);
            }
            for (int i = 0; i < D; ++i) d[i] *= gamma;
            for (int k = cnt - 1; k >= 0; --k) {
                const int j = (head - 1 - k + 2 * RF_MEM) % RF_MEM;
                double yq = 0.0;
                for (int i = 0; i < D; ++i) yq = fma(Y[(long)j * D + i], d[i], yq);
                const double be = rho[j] * yq;
                for (int i = 0; i < D; ++i) d[i] = fma(al[k] - be, S[(long)j * D + i], d[i]);
            }
            double gd = 0.0;
            for (int i = 0; i < D; ++i) {
                const bool fixed = (x[i] <= a.lo[i] && g[i] > 0.0) || (x[i] >= a.hi[i] && g[i] < 0.0);
                d[i] = fixed ? 0.0 : -d[i];
                gd = fma(g[i], d[i], gd);
            }
            if (!(gd < 0.0) || !isfinite(gd)) {              // not a descent direction: steepest descent, history dropped
                for (int i = 0; i < D; ++i) {
                    const bool fixed = (x[i] <= a.lo[i] && g[i] > 0.0) || (x[i] >= a.hi[i] && g[i] < 0.0);
                    d[i] = fixed ? 0.0 : -g[i];
                }
                sc[2] = 0.0;
            }
            // first step like L-BFGS-B: 1 / |g| without curvature information, 1 afterwards
            sc[1] = ((int)sc[2] == 0) ? fmin(1.0, 1.0 / sqrt(fmax(gn, 1e-300))) : 1.0;
            for (int i = 0; i < D; ++i) xt[i] = rf_clip(fma(sc[1], d[i], x[i]), a.lo[i], a.hi[i]);
        }
    }
    if (sc[4] != 0.0) {
        for (int i = 0; i < D; ++i) xt[i] = x[i];            // finished restarts keep evaluating their optimum
    } else {
        atomicAdd(a.active, 1);
    }
}


// The same step for D <= 64 with ONE WAVE per restart: lane i owns coordinate i of every vector
// (x, g, d, the trial point, the history pairs, all in registers), the scalars are computed
// redundantly by every lane, dot products are wave reductions.  (The one-thread version above
// walks its state through global memory coordinate by coordinate: 67 us per step at D = 16
// against 5 us here, rocprofv3.)
__device__ __forceinline__ double rf_wsum(double s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
__device__ __forceinline__ double rf_wmax(double s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
    return s;
}
__global__ __launch_bounds__(256) void refine_step_wave_kernel(RefineArgs a) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.active_next = 0;
    if (r >= a.R) return;
    const int D = a.D;
    const bool on = lane < D;
    const int li = on ? lane : 0;
    double *st = a.state + (long)r * rf_stride(D);
    double *x = st, *g = st + D, *d = st + 2 * D, *S = st + 3 * D, *Y = S + (long)RF_MEM * D;
    double *rho = Y + (long)RF_MEM * D, *sc = rho + RF_MEM;
    double *xt = a.xt + (long)r * D;
    const double lo_i = a.lo[li], hi_i = a.hi[li];
    double xt_i = on ? xt[li] : 0.0;
    const double gt_i = on ? -a.grad[(long)r * D + li] : 0.0;   // gradient of phi = -acq at the trial point
    const double phit = -a.val[r];
    double x_i = 0.0, g_i = 0.0, d_i = 0.0;
    double phi = 0.0, t = 0.0, last = INFINITY;
    int cnt = 0, head = 0, status = 0, iters = 0;
    // the history pairs: in LDS, each lane reading back only what it wrote itself (no barrier)
    __shared__ double hist[4][2][RF_MEM][64];
    double (*Sv)[64] = hist[threadIdx.x >> 6][0], (*Yv)[64] = hist[threadIdx.x >> 6][1];
    __shared__ double rhs[4][RF_MEM];                 // (every lane writes the same value before it reads it)
    double *rh = rhs[threadIdx.x >> 6];
#pragma unroll
    for (int k = 0; k < RF_MEM; ++k) rh[k] = 0.0;
    bool new_dir = false;
    if (a.first) {
        x_i = xt_i; g_i = gt_i; phi = phit;
        status = isfinite(phit) ? 0 : 2;
        new_dir = true;
    } else {
        x_i = on ? x[li] : 0.0; g_i = on ? g[li] : 0.0; d_i = on ? d[li] : 0.0;
        phi = sc[0]; t = sc[1]; cnt = (int)sc[2]; head = (int)sc[3]; status = (int)sc[4]; iters = (int)sc[5]; last = sc[6];
#pragma unroll
        for (int k = 0; k < RF_MEM; ++k) {
            Sv[k][lane] = on ? S[(long)k * D + li] : 0.0;
            Yv[k][lane] = on ? Y[(long)k * D + li] : 0.0;
            rh[k] = rho[k];
        }
        if (status == 0) {
            const double slope = rf_wsum(g_i * (xt_i - x_i));
            if (isfinite(phit) && phit <= phi + 1e-4 * slope) {
                const double s_i = xt_i - x_i, y_i = gt_i - g_i;
                const double sy = rf_wsum(s_i * y_i), yy = rf_wsum(y_i * y_i);
                if (sy > 2.2e-16 * yy && sy > 0.0) {
                    Sv[head][lane] = s_i;
                    Yv[head][lane] = y_i;
                    rh[head] = 1.0 / sy;
                    if (on) { S[(long)head * D + li] = s_i; Y[(long)head * D + li] = y_i; }
                    head = (head + 1) % RF_MEM;
                    cnt = min(cnt + 1, RF_MEM);
                }
                const double dphi = phi - phit;
                last = dphi;
                x_i = xt_i; g_i = gt_i;
                const double scale = fmax(fmax(fabs(phi), fabs(phit)), 1.0);
                phi = phit;
                iters += 1;
                if (dphi <= a.ftol * scale) status = 1;
                new_dir = true;
            } else {
                double theta = 0.5;
                const double denom = 2.0 * (phit - phi - slope);
                if (isfinite(phit) && denom > 0.0 && slope < 0.0) theta = fmin(0.5, fmax(0.1, -slope / denom));
                t *= theta;
                if (t < 1e-12) {
                    status = (iters > 0) ? 1 : 2;
                } else {
                    xt_i = rf_clip(fma(t, d_i, x_i), lo_i, hi_i);
                }
            }
        }
    }
    if (new_dir && status == 0) {
        const double pg = rf_wmax(on ? fabs(x_i - rf_clip(x_i - g_i, lo_i, hi_i)) : 0.0);
        if (pg <= a.pgtol) {
            status = 1;
        } else {
            const bool fixed = !on || (x_i <= lo_i && g_i > 0.0) || (x_i >= hi_i && g_i < 0.0);
            double q_i = fixed ? 0.0 : g_i;
            const double gn = rf_wsum(q_i * q_i);
            double al[RF_MEM];
#pragma unroll
            for (int k = 0; k < RF_MEM; ++k) {
                al[k] = 0.0;
                if (k < cnt) {
                    const int j = (head - 1 - k + 2 * RF_MEM) % RF_MEM;
                    al[k] = rh[j] * rf_wsum(Sv[j][lane] * q_i);
                    q_i = fma(-al[k], Yv[j][lane], q_i);
                }
            }
            if (cnt > 0) {
                const int j = (head - 1 + RF_MEM) % RF_MEM;
                const double yj = Yv[j][lane];
                q_i *= 1.0 / (rh[j] * rf_wsum(yj * yj));
            }
#pragma unroll
            for (int k = RF_MEM - 1; k >= 0; --k) {
                if (k < cnt) {
                    const int j = (head - 1 - k + 2 * RF_MEM) % RF_MEM;
                    const double be = rh[j] * rf_wsum(Yv[j][lane] * q_i);
                    q_i = fma(al[k] - be, Sv[j][lane], q_i);
                }
            }
            d_i = fixed ? 0.0 : -q_i;
            const double gd = rf_wsum(g_i * d_i);
            if (!(gd < 0.0) || !isfinite(gd)) {
                d_i = fixed ? 0.0 : -g_i;
                cnt = 0;
            }
            t = (cnt == 0) ? fmin(1.0, 1.0 / sqrt(fmax(gn, 1e-300))) : 1.0;
            xt_i = rf_clip(fma(t, d_i, x_i), lo_i, hi_i);
        }
    }
    if (status != 0) xt_i = x_i;
    if (on) { x[li] = x_i; g[li] = g_i; d[li] = d_i; xt[li] = xt_i; }
    if (lane == 0) {
        sc[0] = phi; sc[1] = t; sc[2] = (double)cnt; sc[3] = (double)head; sc[4] = (double)status;
        sc[5] = (double)iters; sc[6] = last;
#pragma unroll
        for (int k = 0; k < RF_MEM; ++k) rho[k] = rh[k];
        if (status == 0) atomicAdd(a.active, 1);
    }
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
                              int it, double pgtol, double ftol, int *d_active) {
    RefineArgs a{};
    a.state = d_state; a.xt = d_xt; a.val = d_val; a.grad = d_grad; a.lo = d_lo; a.hi = d_hi;
    a.R = R; a.D = (int)c.D; a.first = it == 0 ? 1 : 0; a.pgtol = pgtol; a.ftol = ftol;
    // two counters of running restarts, used in turn: step it counts into d_active[it & 1] and zeroes the other
    a.active = d_active + (it & 1);
    a.active_next = d_active + ((it + 1) & 1);
    if (it == 0) TGP_TRY(hipMemsetAsync(d_active, 0, 2 * sizeof(int), c.stream));
    if (c.D <= 64)
        hipLaunchKernelGGL(refine_step_wave_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, c.stream, a);
    else
        hipLaunchKernelGGL(refine_step_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, c.stream, a);
    return hipGetLastError();
}

hipError_t launch_refine_collect(Context &c, const double *d_state, int R, double *d_x, double *d_v, double *d_info) {
    hipLaunchKernelGGL(refine_collect_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, c.stream, d_state, R, (int)c.D, d_x, d_v, d_info);
    return hipGetLastError();
}

long refine_state_stride(int D) { return rf_stride(D); }

}  // namespace tgp
