// host_lbfgs.hpp -- the projected L-BFGS of lbfgs_wave.hpp (rf_wave_step) as plain host C++, one object per
// start.  Used by tgp_fit_optimise above the one-launch sizes (N > 128): the hyper-parameter vector has at
// most 66 entries, so the optimiser's arithmetic is nothing -- what costs is the evaluation (a fit + the LML
// gradient on the GPU) and whatever sits between two evaluations.  With SciPy in that place every callback
// pays 60-100 us of interpreter time under the GIL and the starts' threads queue for it; here the loop that
// drives the GPU objective is a C++ thread per start inside the library.
// Same decisions as the device version -- sufficient decrease 1e-4 and the curvature condition 0.9 on the
// slope over the coordinates the projection leaves moving, steps lengthened by a secant on the slope,
// brackets cut by quadratic interpolation, SciPy's L-BFGS-B stopping rules (pgtol 1e-5, factr 1e7) -- with
// the sums taken in coordinate order (the device version reduces in a wave-shaped tree: results agree to
// rounding, not to the bit).  Replaces scipy.optimize.minimize(method='L-BFGS-B') inside
// GaussianProcessRegressor._constrained_optimization (sklearn _gpr.py:654-670).
#pragma once
#include <math.h>

#include <vector>

namespace tgp {

struct HostLbfgs {
    static constexpr int MEM = 8;          // history pairs (RF_MEM)
    static constexpr int LS_MAX = 12;      // evaluations per line search (RF_LS_MAX)
    int P = 0;
    std::vector<double> x, g, d, xlo, glo, lo, hi;
    std::vector<double> S[MEM], Y[MEM];
    double rh[MEM];
    double phi = 0, t = 0, last = INFINITY, dphi0 = 0, t_cap = 0;
    double t_lo = 0, phi_lo = 0, dphi_lo = 0, t_hi = 0, phi_hi = 0;
    int cnt = 0, head = 0, status = 0, iters = 0, stage = 0, n_ls = 0;   // status: 0 running / 1 converged / 2 failed

    HostLbfgs(const double *lo_, const double *hi_, int P_) : P(P_), x(P_), g(P_), d(P_), xlo(P_), glo(P_), lo(lo_, lo_ + P_), hi(hi_, hi_ + P_) {
        for (int m = 0; m < MEM; ++m) { S[m].assign((size_t)P, 0.0); Y[m].assign((size_t)P, 0.0); rh[m] = 0.0; }
    }
    static double clip(double v, double a, double b) { return v < a ? a : (v > b ? b : v); }

    // (phit, gt) = phi and its gradient at the trial point xt = P(x + t d); leaves the next trial point in xt
    // (the iterate itself once the start has finished)
    void step(std::vector<double> &xt, std::vector<double> gt, double phit, bool first, double pgtol, double ftol) {
        bool new_dir = false;
        if (first) {
            x = xt; g = gt; xlo = xt; glo = gt;
            for (int k = 0; k < P; ++k) d[k] = 0.0;
            phi = phit; t = 0.0; last = INFINITY;
            cnt = head = iters = stage = n_ls = 0;
            status = isfinite(phit) ? 0 : 2;
            new_dir = true;
        } else if (status == 0) {
            double dphit = 0.0, slope = 0.0;
            for (int k = 0; k < P; ++k) {
                const bool moving = d[k] != 0.0 && xt[k] > lo[k] && xt[k] < hi[k];
                if (moving) dphit += gt[k] * d[k];
                slope += g[k] * (xt[k] - x[k]);
            }
            const bool armijo = isfinite(phit) && isfinite(dphit) && phit <= phi + 1e-4 * slope;
            const bool curv = fabs(dphit) <= 0.9 * fabs(dphi0);
            n_ls += 1;
            bool accept = false, from_lo = false, fail = false, take_lo = false;
            double t_new = t;
            if (stage == 0) {
                if (!armijo || (t_lo > 0.0 && phit >= phi_lo)) {
                    t_hi = t; phi_hi = phit; stage = 1;
                } else if (curv) {
                    accept = true;
                } else if (dphit >= 0.0) {                  // past the minimiser: it lies between the last good step and this one
                    t_hi = t_lo; phi_hi = phi_lo; stage = 1;
                    take_lo = true;
                } else if (t >= t_cap * (1.0 - 1e-12) || n_ls >= LS_MAX) {
                    accept = true;                          // still descending, nowhere further to go
                } else {
                    const double dt = t - t_lo;
                    double inc = 4.0 * dt;
                    if (dphit > dphi_lo) inc = dt * dphit / (dphi_lo - dphit);   // secant on phi'
                    inc = fmin(4.0 * dt, fmax(1.1 * dt, inc));
                    t_new = fmin(t_cap, t + inc);
                    take_lo = true;
                }
            } else {
                if (!armijo || phit >= phi_lo) {
                    t_hi = t; phi_hi = phit;
                } else if (curv) {
                    accept = true;
                } else {
                    if (dphit * (t_hi - t_lo) >= 0.0) { t_hi = t_lo; phi_hi = phi_lo; }
                    take_lo = true;
                }
            }
            if (take_lo) { t_lo = t; phi_lo = phit; dphi_lo = dphit; xlo = xt; glo = gt; }
            if (!accept && stage == 1) {
                const double dl = t_hi - t_lo;
                if (n_ls >= LS_MAX || fabs(dl) <= 1e-13 * fmax(fmax(t_hi, t_lo), 1e-300) || fmax(t_hi, t_lo) < 1e-12) {
                    if (t_lo > 0.0) { accept = true; from_lo = true; }
                    else fail = true;
                } else {
                    const double denom = 2.0 * (phi_hi - phi_lo - dphi_lo * dl);
                    double frac = 0.5;
                    if (isfinite(phi_hi) && denom > 0.0 && dphi_lo * dl < 0.0) frac = -dphi_lo * dl / denom;
                    frac = fmin(t_lo > 0.0 ? 0.9 : 0.5, fmax(0.1, frac));
                    t_new = fma(frac, dl, t_lo);
                }
            }
            if (accept) {
                if (from_lo) { xt = xlo; gt = glo; phit = phi_lo; }
                double sy = 0.0, yy = 0.0;
                for (int k = 0; k < P; ++k) {
                    const double s = xt[k] - x[k], y = gt[k] - g[k];
                    sy += s * y; yy += y * y;
                }
                if (sy > 2.2e-16 * yy && sy > 0.0) {
                    for (int k = 0; k < P; ++k) { S[head][k] = xt[k] - x[k]; Y[head][k] = gt[k] - g[k]; }
                    rh[head] = 1.0 / sy;
                    head = (head + 1) % MEM;
                    cnt = cnt + 1 < MEM ? cnt + 1 : MEM;
                }
                const double dphi = phi - phit;
                last = dphi;
                x = xt; g = gt;
                const double scale = fmax(fmax(fabs(phi), fabs(phit)), 1.0);
                phi = phit;
                iters += 1;
                if (dphi <= ftol * scale) status = 1;       // relative reduction below factr * eps
                new_dir = true;
            } else if (fail) {
                status = (iters > 0) ? 1 : 2;               // no further progress possible from here
            } else {
                t = t_new;
                for (int k = 0; k < P; ++k) xt[k] = clip(fma(t, d[k], x[k]), lo[k], hi[k]);
            }
        }
        if (new_dir && status == 0) {
            double pg = 0.0;
            for (int k = 0; k < P; ++k) pg = fmax(pg, fabs(x[k] - clip(x[k] - g[k], lo[k], hi[k])));
            if (pg <= pgtol) {
                status = 1;
            } else {
                std::vector<char> fixed((size_t)P);
                std::vector<double> q((size_t)P);
                double gn = 0.0;
                for (int k = 0; k < P; ++k) {
                    fixed[k] = (x[k] <= lo[k] && g[k] > 0.0) || (x[k] >= hi[k] && g[k] < 0.0);
                    q[k] = fixed[k] ? 0.0 : g[k];
                    gn += q[k] * q[k];
                }
                double al[MEM];
                for (int m = 0; m < MEM; ++m) {
                    al[m] = 0.0;
                    if (m < cnt) {
                        const int j = (head - 1 - m + 2 * MEM) % MEM;
                        double sq = 0.0;
                        for (int k = 0; k < P; ++k) sq += S[j][k] * q[k];
                        al[m] = rh[j] * sq;
                        for (int k = 0; k < P; ++k) q[k] = fma(-al[m], Y[j][k], q[k]);
                    }
                }
                if (cnt > 0) {
                    const int j = (head - 1 + MEM) % MEM;
                    double yy = 0.0;
                    for (int k = 0; k < P; ++k) yy += Y[j][k] * Y[j][k];
                    const double sc = 1.0 / (rh[j] * yy);
                    for (int k = 0; k < P; ++k) q[k] *= sc;
                }
                for (int m = MEM - 1; m >= 0; --m) {
                    if (m < cnt) {
                        const int j = (head - 1 - m + 2 * MEM) % MEM;
                        double yq = 0.0;
                        for (int k = 0; k < P; ++k) yq += Y[j][k] * q[k];
                        const double be = rh[j] * yq;
                        for (int k = 0; k < P; ++k) q[k] = fma(al[m] - be, S[j][k], q[k]);
                    }
                }
                double gd = 0.0;
                for (int k = 0; k < P; ++k) {
                    const bool out = (x[k] <= lo[k] && q[k] > 0.0) || (x[k] >= hi[k] && q[k] < 0.0);
                    d[k] = (fixed[k] || out) ? 0.0 : -q[k];
                    gd += g[k] * d[k];
                }
                if (!(gd < 0.0) || !isfinite(gd)) {         // not a descent direction: steepest descent, history dropped
                    for (int k = 0; k < P; ++k) d[k] = fixed[k] ? 0.0 : -g[k];
                    cnt = 0;
                    gd = -gn;
                }
                dphi0 = gd;
                t_cap = 0.0;
                for (int k = 0; k < P; ++k)
                    t_cap = fmax(t_cap, d[k] > 0.0 ? (hi[k] - x[k]) / d[k] : (d[k] < 0.0 ? (lo[k] - x[k]) / d[k] : 0.0));
                t_lo = 0.0; phi_lo = phi; dphi_lo = gd; t_hi = 0.0; phi_hi = phi;
                xlo = x; glo = g;
                stage = 0; n_ls = 0;
                t = fmin(t_cap, (cnt == 0) ? fmin(1.0, 1.0 / sqrt(fmax(gn, 1e-300))) : 1.0);
                for (int k = 0; k < P; ++k) xt[k] = clip(fma(t, d[k], x[k]), lo[k], hi[k]);
            }
        }
        if (status != 0) xt = x;
    }
};

}  // namespace tgp
