// lbfgs_wave.hpp -- one wave's (or, for D > 1024, one eight-wave team's) projected L-BFGS step with an L-BFGS-B
// style line search (thread i of the restart owns coordinates i, i + T, i + 2 T, ... with T = 64 NW threads):
// shared by the acquisition's gradient stage
// (refine_kernels.hip) and the one-launch hyper-parameter optimiser (small_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace tgp {

constexpr int RF_MEM = 8;                         // history pairs
__device__ inline double rf_clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// wave reductions on the DPP path (every lane ends with the result): within each row of 16 lanes
// by quad permutes and the two row mirrors, across the four rows through readlane.  (The
// __shfl_xor ladder compiles to 12 dependent ds_bpermute per reduction, and a step takes 25
// reductions one after another.)
template <int CTRL>
__device__ __forceinline__ double rf_dpp(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double rf_lane(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double rf_wsum(double s) {
    s += rf_dpp<0xB1>(s);      // quad_perm [1,0,3,2]
    s += rf_dpp<0x4E>(s);      // quad_perm [2,3,0,1]
    s += rf_dpp<0x141>(s);     // row_half_mirror
    s += rf_dpp<0x140>(s);     // row_mirror: every lane holds its row's sum
    return (rf_lane(s, 0) + rf_lane(s, 16)) + (rf_lane(s, 32) + rf_lane(s, 48));
}
__device__ __forceinline__ double rf_wmax(double s) {
    s = fmax(s, rf_dpp<0xB1>(s));
    s = fmax(s, rf_dpp<0x4E>(s));
    s = fmax(s, rf_dpp<0x141>(s));
    s = fmax(s, rf_dpp<0x140>(s));
    return fmax(fmax(rf_lane(s, 0), rf_lane(s, 16)), fmax(rf_lane(s, 32), rf_lane(s, 48)));
}
// what one wave carries for its restart between steps; a lane owns the DK coordinates lane + 64 k
template <int DK>
struct RfWaveT {
    double x_i[DK], g_i[DK], d_i[DK];      // iterate, gradient of phi = -acq there, search direction
    double xlo_i[DK], glo_i[DK];           // the line search's best point so far and the gradient there
    double phi, t, last;           // phi(x), current step length, last accepted decrease
    double dphi0, t_cap;           // slope of phi along d at x; the step beyond which every moving coordinate is clipped
    double t_lo, phi_lo, dphi_lo;  // line search: best step satisfying the decrease condition (0: the iterate itself) ...
    double t_hi, phi_hi;           // ... and the other end of the bracket once there is one
    int cnt, head, status, iters;  // history pairs held, ring head, 0 running / 1 converged / 2 failed, accepted steps
    int stage, n_ls;               // line search: 0 lengthening the step, 1 inside a bracket; evaluations so far
};
using RfWave = RfWaveT<1>;
constexpr int RF_LS_MAX = 12;      // evaluations per line search (L-BFGS-B allows 20)

#define RF_EACH(k) _Pragma("unroll") for (int k = 0; k < DK; ++k)
// The threads that share a restart: one wave (NW = 1: wave reductions only) or NW waves of one workgroup that
// meet in LDS.  `red` holds 2 NW doubles used in turn (one barrier per reduction: a thread can only overwrite a
// slot two reductions later, i.e. after every thread has passed the barrier behind its last reader).  Every
// thread of the team takes the same branches (all decisions hang on reduced scalars), so the barriers inside
// the reductions are reached by all of them.
template <int NW>
struct RfTeam {
    double *red;
    int wave;
    int flip;
};
template <int NW>
__device__ __forceinline__ double rf_team_sum(double s, RfTeam<NW> &tm) {
    s = rf_wsum(s);
    if (NW == 1) return s;
    tm.flip ^= 1;
    double *slot = tm.red + tm.flip * NW;
    if ((threadIdx.x & 63) == 0) slot[tm.wave] = s;
    __syncthreads();
    double t = slot[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += slot[w];
    return t;
}
template <int NW>
__device__ __forceinline__ double rf_team_max(double s, RfTeam<NW> &tm) {
    s = rf_wmax(s);
    if (NW == 1) return s;
    tm.flip ^= 1;
    double *slot = tm.red + tm.flip * NW;
    if ((threadIdx.x & 63) == 0) slot[tm.wave] = s;
    __syncthreads();
    double t = slot[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = fmax(t, slot[w]);
    return t;
}
// sum / max over the thread's own coordinates, then over the team (fixed order)
template <int DK, int NW, typename F>
__device__ __forceinline__ double rf_sum_each(F f, RfTeam<NW> &tm) {
    double s = f(0);
#pragma unroll
    for (int k = 1; k < DK; ++k) s += f(k);
    return rf_team_sum<NW>(s, tm);
}
template <int DK, int NW, typename F>
__device__ __forceinline__ double rf_max_each(F f, RfTeam<NW> &tm) {
    double s = f(0);
#pragma unroll
    for (int k = 1; k < DK; ++k) s = fmax(s, f(k));
    return rf_team_max<NW>(s, tm);
}

// One step of the projected L-BFGS for the wave's restart: (phit, gt_i) = phi and its gradient at
// the trial point xt_i = P(x + t d).  The line search asks for what L-BFGS-B's dcsrch asks
// (sufficient decrease 1e-4 AND |phi'(t)| <= 0.9 |phi'(0)|, phi' taken over the coordinates the
// projection leaves moving): a step whose slope is still steep is LENGTHENED (secant on phi',
// growth between 1.1 and 4 times the last increment, up to t_cap), one that overshoots is bracketed
// and bisected by quadratic interpolation; without the second condition restarts that begin on
// the flat part of EI crept along at unit quasi-Newton steps for thousands of evaluations.
// Leaves the next trial point in xt_i (the iterate itself once the restart has finished).
// Sv / Yv [RF_MEM][64 NW DK] and rh [RF_MEM] are the restart's history (LDS, or global memory for a team of
// several waves: each thread reads back only what it wrote itself); `lane` = the thread's index inside the
// team (0 .. 64 NW - 1), coordinate k of a thread = lane + 64 NW k; returns the ring slot a new pair went
// into, or -1.
template <int DK, int NW = 1>
__device__ __forceinline__ int rf_wave_step(RfWaveT<DK> &w, double (&xt_i)[DK], const double (&gt_in)[DK], double phit,
                                            bool first, const bool (&on)[DK], int lane, const double (&lo_i)[DK],
                                            const double (&hi_i)[DK], double pgtol, double ftol,
                                            double (*Sv)[64 * NW * DK], double (*Yv)[64 * NW * DK], double *rh,
                                            RfTeam<NW> tm = RfTeam<NW>{nullptr, 0, 0}) {
    constexpr int TS = 64 * NW;                 // threads of the team = stride between a thread's coordinates
    bool new_dir = false;
    int stored = -1;
    double gt_i[DK];
    RF_EACH(k) gt_i[k] = gt_in[k];
    if (first) {
        RF_EACH(k) { w.x_i[k] = xt_i[k]; w.g_i[k] = gt_i[k]; w.d_i[k] = 0.0; w.xlo_i[k] = xt_i[k]; w.glo_i[k] = gt_i[k]; }
        w.phi = phit; w.t = 0.0; w.last = INFINITY;
        w.cnt = 0; w.head = 0; w.iters = 0; w.stage = 0; w.n_ls = 0;
        w.status = isfinite(phit) ? 0 : 2;
        new_dir = true;
    } else if (w.status == 0) {
        const double dphit = rf_sum_each<DK, NW>([&](int k) {
            const bool moving = on[k] && w.d_i[k] != 0.0 && xt_i[k] > lo_i[k] && xt_i[k] < hi_i[k];
            return moving ? gt_i[k] * w.d_i[k] : 0.0;
        }, tm);
        // sufficient decrease along the PROJECTED step s = xt - x
        const double slope = rf_sum_each<DK, NW>([&](int k) { return w.g_i[k] * (xt_i[k] - w.x_i[k]); }, tm);
        const bool armijo = isfinite(phit) && isfinite(dphit) && phit <= w.phi + 1e-4 * slope;
        const bool curv = fabs(dphit) <= 0.9 * fabs(w.dphi0);
        w.n_ls += 1;
        bool accept = false, from_lo = false, fail = false, take_lo = false;
        double t_new = w.t;
        if (w.stage == 0) {
            if (!armijo || (w.t_lo > 0.0 && phit >= w.phi_lo)) {
                w.t_hi = w.t; w.phi_hi = phit; w.stage = 1;
            } else if (curv) {
                accept = true;
            } else if (dphit >= 0.0) {                  // went past the minimiser: it lies between the last good step and this one
                w.t_hi = w.t_lo; w.phi_hi = w.phi_lo; w.stage = 1;
                take_lo = true;
            } else if (w.t >= w.t_cap * (1.0 - 1e-12) || w.n_ls >= RF_LS_MAX) {
                accept = true;                          // still descending, nowhere further to go
            } else {
                const double dt = w.t - w.t_lo;
                double inc = 4.0 * dt;
                if (dphit > w.dphi_lo) inc = dt * dphit / (w.dphi_lo - dphit);   // secant on phi' (slope flattening)
                inc = fmin(4.0 * dt, fmax(1.1 * dt, inc));
                t_new = fmin(w.t_cap, w.t + inc);
                take_lo = true;
            }
        } else {
            if (!armijo || phit >= w.phi_lo) {
                w.t_hi = w.t; w.phi_hi = phit;
            } else if (curv) {
                accept = true;
            } else {
                if (dphit * (w.t_hi - w.t_lo) >= 0.0) { w.t_hi = w.t_lo; w.phi_hi = w.phi_lo; }
                take_lo = true;
            }
        }
        if (take_lo) {
            w.t_lo = w.t; w.phi_lo = phit; w.dphi_lo = dphit;
            RF_EACH(k) { w.xlo_i[k] = xt_i[k]; w.glo_i[k] = gt_i[k]; }
        }
        if (!accept && w.stage == 1) {
            const double dl = w.t_hi - w.t_lo;
            if (w.n_ls >= RF_LS_MAX || fabs(dl) <= 1e-13 * fmax(fmax(w.t_hi, w.t_lo), 1e-300) || fmax(w.t_hi, w.t_lo) < 1e-12) {
                if (w.t_lo > 0.0) { accept = true; from_lo = true; }
                else fail = true;
            } else {
                // minimiser of the quadratic through phi(t_lo), phi'(t_lo), phi(t_hi), kept inside the bracket
                const double denom = 2.0 * (w.phi_hi - w.phi_lo - w.dphi_lo * dl);
                double frac = 0.5;
                if (isfinite(w.phi_hi) && denom > 0.0 && w.dphi_lo * dl < 0.0) frac = -w.dphi_lo * dl / denom;
                frac = fmin(w.t_lo > 0.0 ? 0.9 : 0.5, fmax(0.1, frac));
                t_new = fma(frac, dl, w.t_lo);
            }
        }
        if (accept) {
            if (from_lo) {
                RF_EACH(k) { xt_i[k] = w.xlo_i[k]; gt_i[k] = w.glo_i[k]; }
                phit = w.phi_lo;
            }
            // curvature pair (kept as L-BFGS-B's curvature test keeps it), new iterate
            double s_i[DK], y_i[DK];
            RF_EACH(k) { s_i[k] = xt_i[k] - w.x_i[k]; y_i[k] = gt_i[k] - w.g_i[k]; }
            const double sy = rf_sum_each<DK, NW>([&](int k) { return s_i[k] * y_i[k]; }, tm);
            const double yy = rf_sum_each<DK, NW>([&](int k) { return y_i[k] * y_i[k]; }, tm);
            if (sy > 2.2e-16 * yy && sy > 0.0) {
                RF_EACH(k) { Sv[w.head][lane + TS * k] = s_i[k]; Yv[w.head][lane + TS * k] = y_i[k]; }
                rh[w.head] = 1.0 / sy;
                stored = w.head;
                w.head = (w.head + 1) % RF_MEM;
                w.cnt = min(w.cnt + 1, RF_MEM);
            }
            const double dphi = w.phi - phit;
            w.last = dphi;
            RF_EACH(k) { w.x_i[k] = xt_i[k]; w.g_i[k] = gt_i[k]; }
            const double scale = fmax(fmax(fabs(w.phi), fabs(phit)), 1.0);
            w.phi = phit;
            w.iters += 1;
            if (dphi <= ftol * scale) w.status = 1;     // relative reduction below factr * eps
            new_dir = true;
        } else if (fail) {
            w.status = (w.iters > 0) ? 1 : 2;           // no further progress possible from here
        } else {
            w.t = t_new;
            RF_EACH(k) xt_i[k] = rf_clip(fma(w.t, w.d_i[k], w.x_i[k]), lo_i[k], hi_i[k]);
        }
    }
    if (new_dir && w.status == 0) {
        // projected gradient: zero when x is a constrained stationary point
        const double pg = rf_max_each<DK, NW>([&](int k) {
            return on[k] ? fabs(w.x_i[k] - rf_clip(w.x_i[k] - w.g_i[k], lo_i[k], hi_i[k])) : 0.0;
        }, tm);
        if (pg <= pgtol) {
            w.status = 1;
        } else {
            // two-loop recursion on the free variables (bound variables whose gradient pushes outward stay put)
            bool fixed[DK];
            double q_i[DK];
            RF_EACH(k) {
                fixed[k] = !on[k] || (w.x_i[k] <= lo_i[k] && w.g_i[k] > 0.0) || (w.x_i[k] >= hi_i[k] && w.g_i[k] < 0.0);
                q_i[k] = fixed[k] ? 0.0 : w.g_i[k];
            }
            const double gn = rf_sum_each<DK, NW>([&](int k) { return q_i[k] * q_i[k]; }, tm);
            double al[RF_MEM];
#pragma unroll
            for (int m = 0; m < RF_MEM; ++m) {
                al[m] = 0.0;
                if (m < w.cnt) {
                    const int j = (w.head - 1 - m + 2 * RF_MEM) % RF_MEM;
                    al[m] = rh[j] * rf_sum_each<DK, NW>([&](int k) { return Sv[j][lane + TS * k] * q_i[k]; }, tm);
                    RF_EACH(k) q_i[k] = fma(-al[m], Yv[j][lane + TS * k], q_i[k]);
                }
            }
            if (w.cnt > 0) {
                const int j = (w.head - 1 + RF_MEM) % RF_MEM;
                const double sc = 1.0 / (rh[j] * rf_sum_each<DK, NW>([&](int k) { const double yj = Yv[j][lane + TS * k]; return yj * yj; }, tm));
                RF_EACH(k) q_i[k] *= sc;
            }
#pragma unroll
            for (int m = RF_MEM - 1; m >= 0; --m) {
                if (m < w.cnt) {
                    const int j = (w.head - 1 - m + 2 * RF_MEM) % RF_MEM;
                    const double be = rh[j] * rf_sum_each<DK, NW>([&](int k) { return Yv[j][lane + TS * k] * q_i[k]; }, tm);
                    RF_EACH(k) q_i[k] = fma(al[m] - be, Sv[j][lane + TS * k], q_i[k]);
                }
            }
            // (a coordinate sitting on a bound does not move outward either)
            RF_EACH(k) {
                const bool out = (w.x_i[k] <= lo_i[k] && q_i[k] > 0.0) || (w.x_i[k] >= hi_i[k] && q_i[k] < 0.0);
                w.d_i[k] = (fixed[k] || out) ? 0.0 : -q_i[k];
            }
            double gd = rf_sum_each<DK, NW>([&](int k) { return w.g_i[k] * w.d_i[k]; }, tm);
            if (!(gd < 0.0) || !isfinite(gd)) {         // not a descent direction: steepest descent, history dropped
                RF_EACH(k) w.d_i[k] = fixed[k] ? 0.0 : -w.g_i[k];
                w.cnt = 0;
                gd = -gn;
            }
            w.dphi0 = gd;
            // beyond t_cap the projection holds every moving coordinate on its bound
            w.t_cap = rf_max_each<DK, NW>([&](int k) {
                return w.d_i[k] > 0.0 ? (hi_i[k] - w.x_i[k]) / w.d_i[k] : (w.d_i[k] < 0.0 ? (lo_i[k] - w.x_i[k]) / w.d_i[k] : 0.0);
            }, tm);
            w.t_lo = 0.0; w.phi_lo = w.phi; w.dphi_lo = gd; w.t_hi = 0.0; w.phi_hi = w.phi;
            RF_EACH(k) { w.xlo_i[k] = w.x_i[k]; w.glo_i[k] = w.g_i[k]; }
            w.stage = 0; w.n_ls = 0;
            // first step like L-BFGS-B: 1 / |g| without curvature information, 1 afterwards
            w.t = fmin(w.t_cap, (w.cnt == 0) ? fmin(1.0, 1.0 / sqrt(fmax(gn, 1e-300))) : 1.0);
            RF_EACH(k) xt_i[k] = rf_clip(fma(w.t, w.d_i[k], w.x_i[k]), lo_i[k], hi_i[k]);
        }
    }
    if (w.status != 0) {                                // finished restarts keep evaluating their optimum
        RF_EACH(k) xt_i[k] = w.x_i[k];
    }
    return stored;
}

// the one-coordinate-per-lane form (D <= 64) the small-problem kernels use
__device__ __forceinline__ int rf_wave_step(RfWave &w, double &xt_i, double gt_i, double phit, bool first, bool on,
                                            int lane, double lo_i, double hi_i, double pgtol, double ftol,
                                            double (*Sv)[64], double (*Yv)[64], double *rh) {
    double xt_a[1] = {xt_i};
    const double gt_a[1] = {gt_i}, lo_a[1] = {lo_i}, hi_a[1] = {hi_i};
    const bool on_a[1] = {on};
    const int stored = rf_wave_step<1>(w, xt_a, gt_a, phit, first, on_a, lane, lo_a, hi_a, pgtol, ftol, Sv, Yv, rh);
    xt_i = xt_a[0];
    return stored;
}

}  // namespace tgp
