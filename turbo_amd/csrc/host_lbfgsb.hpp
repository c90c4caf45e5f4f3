// host_lbfgsb.hpp -- L-BFGS-B (Byrd, Lu, Nocedal, Zhu 1995; the 3.0 revision of Morales & Nocedal 2011 with the
// projected subspace step) as plain host C++ in reverse communication, one object per start.  It is the optimiser
// scikit-learn's GaussianProcessRegressor hands the negative log marginal likelihood to
// (scipy.optimize.minimize(method='L-BFGS-B'), sklearn _gpr.py:654-670, reached from
// turbo/modules/surrogates.py:313-318), restated from the published algorithm so that tgp_fit_optimise above the
// one-launch sizes (N > 128) walks the iterates SciPy walks -- same generalised Cauchy point, same subspace
// minimisation, same More-Thuente line search (dcsrch / dcstep of MINPACK-2: ftol 1e-3, gtol 0.9, xtol 0.1), same
// memory (10 pairs), same update-skipping and memory-refresh rules, same stopping tests -- with no interpreter
// between two evaluations of the GPU objective.  The hyper-parameter vector has at most 66 entries: the optimiser's
// own arithmetic is nothing, and the small dense solves are written for clarity (an LU of the 2m x 2m middle
// matrix where the Fortran keeps an LEL^T factorisation up to date): same mathematics, rounding-level differences.
// tests/test_host_lbfgs.py holds it against SciPy on the CPU: equal iterates, evaluation and iteration counts.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

namespace tgp {

struct HostLbfgsb {
    static constexpr int M = 10;           // pairs kept (SciPy's maxcor)
    static constexpr int MAXLS = 20;       // evaluations per line search (SciPy's maxls)
    static constexpr double EPS = 2.220446049250313e-16;
    int P = 0;
    std::vector<double> lo, hi;
    std::vector<double> x, g;              // the iterate and its gradient
    double phi = 0;                        // ... and its value
    int status = 0;                        // 0 running / 1 converged / 2 the line search found no acceptable step
    int iters = 0;                         // accepted iterations
    // ---- limited memory: pairs 0 (oldest) .. col-1 (newest) ----
    int col = 0;
    double theta = 1.0;
    std::vector<double> S[M], Y[M];
    double sy[M][M], ss[M][M];             // sy[i][j] = s_i . y_j, ss[i][j] = s_i . s_j
    double J[M][M];                        // lower Cholesky factor of T = theta SS + L D^-1 L^T
    // ---- the iteration in progress ----
    std::vector<double> z, d, gt_prev;
    std::vector<int> where;                // after the Cauchy point: 1 at the lower bound, 2 at the upper, 3 lo == hi, -1 no bounds, <= 0 free
    double c[2 * M];                       // W^T (x_cauchy - x)
    bool cnstnd = false, boxed = true;
    // ---- line search (dcsrch's saved state) ----
    double stp = 0, stpmx = 0, gd = 0, gdold = 0, fold = 0;
    int ifun = 0;
    bool brackt = false;
    int stage = 1;
    double finit = 0, ginit = 0, gtest = 0, width = 0, width1 = 0, stx = 0, fx = 0, gx = 0, sty = 0, fy = 0, gy = 0, stmin = 0, stmax = 0;

    // ---- the last point handed to step(): a search that ends on "no further progress" asks for its best point once
    // more; SciPy answers that from its cache of the last evaluation (ScalarFunction), and so can the caller
    std::vector<double> x_eval, g_eval;
    double f_eval = 0;
    bool evaluated(const std::vector<double> &xt) const { return !x_eval.empty() && xt == x_eval; }

    HostLbfgsb(const double *lo_, const double *hi_, int P_)
        : P(P_), lo(lo_, lo_ + P_), hi(hi_, hi_ + P_), x(P_), g(P_), z(P_), d(P_), gt_prev(P_), where(P_) {
        for (int m = 0; m < M; ++m) { S[m].assign((size_t)P, 0.0); Y[m].assign((size_t)P, 0.0); }
    }
    static double clip(double v, double a, double b) { return v < a ? a : (v > b ? b : v); }
    bool has_lo(int i) const { return lo[i] > -INFINITY; }
    bool has_hi(int i) const { return hi[i] < INFINITY; }

    // infinity norm of the projected gradient (projgr)
    double projected_gradient_norm() const {
        double n = 0.0;
        for (int i = 0; i < P; ++i) {
            double gi = g[i];
            if (gi < 0.0) { if (has_hi(i)) gi = fmax(x[i] - hi[i], gi); }
            else if (has_lo(i)) gi = fmin(x[i] - lo[i], gi);
            n = fmax(n, fabs(gi));
        }
        return n;
    }
    void refresh_memory() { col = 0; theta = 1.0; }

    // out = M v, M = [[-D, L^T], [L, theta SS]]^-1 (bmv): T p2 = v2 + L D^-1 v1, p1 = D^-1 (L^T p2 - v1)
    bool middle_times(const double *v, double *out) const {
        if (col == 0) return true;
        double p2[M];
        for (int i = 0; i < col; ++i) {
            double sum = 0.0;
            for (int k = 0; k < i; ++k) sum += sy[i][k] * v[k] / sy[k][k];
            p2[i] = v[col + i] + sum;
        }
        for (int i = 0; i < col; ++i) {
            double a = p2[i];
            for (int k = 0; k < i; ++k) a -= J[i][k] * p2[k];
            if (!(J[i][i] != 0.0)) return false;
            p2[i] = a / J[i][i];
        }
        for (int i = col - 1; i >= 0; --i) {
            double a = p2[i];
            for (int k = i + 1; k < col; ++k) a -= J[k][i] * p2[k];
            p2[i] = a / J[i][i];
        }
        for (int i = 0; i < col; ++i) {
            double sum = 0.0;
            for (int k = i + 1; k < col; ++k) sum += sy[k][i] * p2[k];
            out[i] = (sum - v[i]) / sy[i][i];
            out[col + i] = p2[i];
        }
        return true;
    }
    // T = theta SS + L D^-1 L^T and its Cholesky factor (formt); false if T is not positive definite
    bool factor_T() {
        double T[M][M];
        for (int i = 0; i < col; ++i)
            for (int j = 0; j <= i; ++j) {
                double sum = 0.0;
                for (int k = 0; k < j; ++k) sum += sy[i][k] * sy[j][k] / sy[k][k];
                T[i][j] = theta * ss[i][j] + sum;
            }
        for (int j = 0; j < col; ++j) {
            double a = T[j][j];
            for (int k = 0; k < j; ++k) a -= J[j][k] * J[j][k];
            if (!(a > 0.0) || !isfinite(a)) return false;
            J[j][j] = sqrt(a);
            for (int i = j + 1; i < col; ++i) {
                double b = T[i][j];
                for (int k = 0; k < j; ++k) b -= J[i][k] * J[j][k];
                J[i][j] = b / J[j][j];
            }
        }
        return true;
    }
    // row a of W^T = [Y theta S]^T at coordinate k
    double W(int a, int k) const { return a < col ? Y[a][k] : theta * S[a - col][k]; }

    // ---- generalised Cauchy point: the first local minimiser of the quadratic model along the projected
    // steepest-descent path (cauchy); leaves it in z, W^T (z - x) in c, the bound each coordinate sits on in `where`
    bool cauchy(double sbgnrm) {
        z = x;
        for (int a = 0; a < 2 * M; ++a) c[a] = 0.0;
        if (sbgnrm <= 0.0) return true;
        const int c2 = 2 * col;
        double p[2 * M] = {0}, v[2 * M] = {0}, wbp[2 * M];
        double f1 = 0.0;
        bool bnded = true;
        int nfree_dir = 0;
        std::vector<std::pair<double, int>> bps;
        for (int i = 0; i < P; ++i) {
            const double neggi = -g[i];
            double tl = 0.0, tu = 0.0;
            if (where[i] != 3 && where[i] != -1) {
                if (has_lo(i)) tl = x[i] - lo[i];
                if (has_hi(i)) tu = hi[i] - x[i];
                const bool xlower = has_lo(i) && tl <= 0.0, xupper = has_hi(i) && tu <= 0.0;
                where[i] = 0;
                if (xlower) { if (neggi <= 0.0) where[i] = 1; }
                else if (xupper) { if (neggi >= 0.0) where[i] = 2; }
                else if (fabs(neggi) <= 0.0) where[i] = -3;
            }
            if (where[i] != 0 && where[i] != -1) {
                d[i] = 0.0;
            } else {
                d[i] = neggi;
                f1 -= neggi * neggi;
                for (int j = 0; j < col; ++j) { p[j] += Y[j][i] * neggi; p[col + j] += S[j][i] * neggi; }
                if (has_lo(i) && neggi < 0.0) bps.emplace_back(tl / (-neggi), i);
                else if (has_hi(i) && neggi > 0.0) bps.emplace_back(tu / neggi, i);
                else { ++nfree_dir; if (fabs(neggi) > 0.0) bnded = false; }
            }
        }
        if (theta != 1.0)
            for (int j = 0; j < col; ++j) p[col + j] *= theta;
        if (bps.empty() && nfree_dir == 0) return true;             // the path does not move
        double f2 = -theta * f1;
        const double f2_org = f2;
        if (col > 0) {
            if (!middle_times(p, v)) return false;
            for (int a = 0; a < c2; ++a) f2 -= v[a] * p[a];
        }
        double dtm = -f1 / f2, tsum = 0.0, tj = 0.0;
        std::stable_sort(bps.begin(), bps.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
        size_t nleft = bps.size();
        bool all_fixed = false;
        for (size_t k = 0; k < bps.size(); ++k) {
            const double tj0 = tj;
            tj = bps[k].first;
            const int ibp = bps[k].second;
            const double dt = tj - tj0;
            if (dtm < dt) break;                                     // the minimiser lies inside this segment
            tsum += dt;
            --nleft;
            const double dibp = d[ibp];
            d[ibp] = 0.0;
            double zibp;
            if (dibp > 0.0) { zibp = hi[ibp] - x[ibp]; z[ibp] = hi[ibp]; where[ibp] = 2; }
            else { zibp = lo[ibp] - x[ibp]; z[ibp] = lo[ibp]; where[ibp] = 1; }
            if (nleft == 0 && (int)bps.size() == P) { dtm = dt; all_fixed = true; break; }
            const double dibp2 = dibp * dibp;
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
            f2 = f2 - theta * dibp2;
            if (col > 0) {
                for (int a = 0; a < c2; ++a) c[a] += dt * p[a];
                for (int j = 0; j < col; ++j) { wbp[j] = Y[j][ibp]; wbp[col + j] = theta * S[j][ibp]; }
                if (!middle_times(wbp, v)) return false;
                double wmc = 0.0, wmp = 0.0, wmw = 0.0;
                for (int a = 0; a < c2; ++a) { wmc += c[a] * v[a]; wmp += p[a] * v[a]; wmw += wbp[a] * v[a]; }
                for (int a = 0; a < c2; ++a) p[a] -= dibp * wbp[a];
                f1 += dibp * wmc;
                f2 += 2.0 * dibp * wmp - dibp2 * wmw;
            }
            f2 = fmax(EPS * f2_org, f2);
            if (nleft > 0) dtm = -f1 / f2;
            else if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
            else dtm = -f1 / f2;
        }
        if (!all_fixed) {
            if (dtm <= 0.0) dtm = 0.0;
            tsum += dtm;
            for (int i = 0; i < P; ++i) z[i] += tsum * d[i];
        }
        for (int a = 0; a < c2; ++a) c[a] += dtm * p[a];
        return true;
    }

    // ---- subspace minimisation over the coordinates free at the Cauchy point (cmprlb + formk + subsm), then the
    // projection of the 3.0 revision.  Direct primal method: with Z the free coordinates and U = Z^T W,
    //     r = -Z^T (theta (z - x) + g - W M c),   d = r / theta + U (M^-1 - U^T U / theta)^-1 U^T r / theta^2
    bool subspace_min(const std::vector<int> &free_idx) {
        const int ns = (int)free_idx.size(), c2 = 2 * col;
        double mc[2 * M] = {0};
        if (!middle_times(c, mc)) return false;
        std::vector<double> r((size_t)ns);
        for (int q = 0; q < ns; ++q) {
            const int k = free_idx[(size_t)q];
            double a = -theta * (z[k] - x[k]) - g[k];
            for (int j = 0; j < col; ++j) a += Y[j][k] * mc[j] + theta * S[j][k] * mc[col + j];
            r[(size_t)q] = a;
        }
        double K[2 * M][2 * M + 1];
        for (int a = 0; a < c2; ++a) {
            for (int b = 0; b < c2; ++b) {
                double minv;
                if (a < col && b < col) minv = a == b ? -sy[a][a] : 0.0;
                else if (a >= col && b >= col) minv = theta * ss[a - col][b - col];
                else {
                    const int i = a >= col ? a - col : b - col, j = a >= col ? b : a;     // L[i][j] = s_i . y_j, i > j
                    minv = i > j ? sy[i][j] : 0.0;
                }
                double gab = 0.0;
                for (int q = 0; q < ns; ++q) gab += W(a, free_idx[(size_t)q]) * W(b, free_idx[(size_t)q]);
                K[a][b] = minv - gab / theta;
            }
            double rhs = 0.0;
            for (int q = 0; q < ns; ++q) rhs += W(a, free_idx[(size_t)q]) * r[(size_t)q];
            K[a][c2] = rhs;
        }
        for (int a = 0; a < c2; ++a) {                               // Gaussian elimination, partial pivoting
            int piv = a;
            for (int b = a + 1; b < c2; ++b)
                if (fabs(K[b][a]) > fabs(K[piv][a])) piv = b;
            if (!(fabs(K[piv][a]) > 0.0) || !isfinite(K[piv][a])) return false;
            if (piv != a)
                for (int b = 0; b <= c2; ++b) std::swap(K[a][b], K[piv][b]);
            for (int b = a + 1; b < c2; ++b) {
                const double m = K[b][a] / K[a][a];
                if (m != 0.0)
                    for (int e = a; e <= c2; ++e) K[b][e] -= m * K[a][e];
            }
        }
        double w[2 * M];
        for (int a = c2 - 1; a >= 0; --a) {
            double s = K[a][c2];
            for (int b = a + 1; b < c2; ++b) s -= K[a][b] * w[b];
            w[a] = s / K[a][a];
        }
        std::vector<double> dn((size_t)ns);
        for (int q = 0; q < ns; ++q) {
            const int k = free_idx[(size_t)q];
            double a = 0.0;
            for (int e = 0; e < c2; ++e) a += W(e, k) * w[e];
            dn[(size_t)q] = r[(size_t)q] / theta + a / (theta * theta);
            if (!isfinite(dn[(size_t)q])) return false;
        }
        // the Newton point projected onto the box; if that is not a descent step from x, back along dn to the first bound
        const std::vector<double> zc = z;
        bool touched = false;
        for (int q = 0; q < ns; ++q) {
            const int k = free_idx[(size_t)q];
            double xk = zc[k] + dn[(size_t)q];
            if (has_lo(k)) xk = fmax(lo[k], xk);
            if (has_hi(k)) xk = fmin(hi[k], xk);
            if ((has_lo(k) && xk == lo[k]) || (has_hi(k) && xk == hi[k])) touched = true;
            z[k] = xk;
        }
        if (!touched) return true;
        double dd_p = 0.0;
        for (int i = 0; i < P; ++i) dd_p += (z[i] - x[i]) * g[i];
        if (dd_p > 0.0) {
            z = zc;
            double alpha = 1.0, t1 = 1.0;
            int ibd = -1;
            for (int q = 0; q < ns; ++q) {
                const int k = free_idx[(size_t)q];
                const double dk = dn[(size_t)q];
                if (!has_lo(k) && !has_hi(k)) continue;
                if (dk < 0.0 && has_lo(k)) {
                    const double t2 = lo[k] - z[k];
                    if (t2 >= 0.0) t1 = 0.0;
                    else if (dk * alpha < t2) t1 = t2 / dk;
                } else if (dk > 0.0 && has_hi(k)) {
                    const double t2 = hi[k] - z[k];
                    if (t2 <= 0.0) t1 = 0.0;
                    else if (dk * alpha > t2) t1 = t2 / dk;
                }
                if (t1 < alpha) { alpha = t1; ibd = q; }
            }
            if (alpha < 1.0 && ibd >= 0) {
                const int k = free_idx[(size_t)ibd];
                if (dn[(size_t)ibd] > 0.0) { z[k] = hi[k]; dn[(size_t)ibd] = 0.0; }
                else if (dn[(size_t)ibd] < 0.0) { z[k] = lo[k]; dn[(size_t)ibd] = 0.0; }
            }
            for (int q = 0; q < ns; ++q) z[free_idx[(size_t)q]] += alpha * dn[(size_t)q];
        }
        return true;
    }

    // ---- More-Thuente: the safeguarded step and the interval update (dcstep) ----
    static void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp, double dp,
                       bool &brackt, double stpmin, double stpmax) {
        const double sgnd = dp * (dx / fabs(dx));
        double stpf;
        if (fp > fx) {                                               // higher value: the minimiser is bracketed
            const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
            double gamma = s * sqrt((th / s) * (th / s) - (dx / s) * (dp / s));
            if (stp < stx) gamma = -gamma;
            const double p = (gamma - dx) + th, q = ((gamma - dx) + gamma) + dp, r = p / q;
            const double stpc = stx + r * (stp - stx);
            const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
            stpf = fabs(stpc - stx) < fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
            brackt = true;
        } else if (sgnd < 0.0) {                                     // lower value, slopes of opposite sign
            const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
            double gamma = s * sqrt((th / s) * (th / s) - (dx / s) * (dp / s));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + th, q = ((gamma - dp) + gamma) + dx, r = p / q;
            const double stpc = stp + r * (stx - stp);
            const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
            stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
            brackt = true;
        } else if (fabs(dp) < fabs(dx)) {                            // lower value, same sign, the slope shrinks
            const double th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = fmax(fabs(th), fmax(fabs(dx), fabs(dp)));
            double gamma = s * sqrt(fmax(0.0, (th / s) * (th / s) - (dx / s) * (dp / s)));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + th, q = (gamma + (dx - dp)) + gamma, r = p / q;
            double stpc;
            if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
            else if (stp > stx) stpc = stpmax;
            else stpc = stpmin;
            const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
            if (brackt) {
                stpf = fabs(stpc - stp) < fabs(stpq - stp) ? stpc : stpq;
                if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
                else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
            } else {
                stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
                stpf = fmin(stpmax, stpf);
                stpf = fmax(stpmin, stpf);
            }
        } else {                                                     // lower value, same sign, the slope does not shrink
            if (brackt) {
                const double th = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
                const double s = fmax(fabs(th), fmax(fabs(dy), fabs(dp)));
                double gamma = s * sqrt((th / s) * (th / s) - (dy / s) * (dp / s));
                if (stp > sty) gamma = -gamma;
                const double p = (gamma - dp) + th, q = ((gamma - dp) + gamma) + dy, r = p / q;
                stpf = stp + r * (sty - stp);
            } else if (stp > stx) stpf = stpmax;
            else stpf = stpmin;
        }
        if (fp > fx) { sty = stp; fy = fp; dy = dp; }
        else {
            if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
            stx = stp; fx = fp; dx = dp;
        }
        stp = stpf;
    }
    // dcsrch after the first evaluation of a search: true when the search is over (converged or one of its warnings)
    bool dcsrch_next(double f, double gdir) {
        const double ftest = finit + stp * gtest;
        if (stage == 1 && f <= ftest && gdir >= 0.0) stage = 2;
        bool done = false;
        if (brackt && (stp <= stmin || stp >= stmax)) done = true;                 // rounding errors prevent progress
        if (brackt && stmax - stmin <= 0.1 * stmax) done = true;                   // xtol test satisfied
        if (stp == stpmx && f <= ftest && gdir <= gtest) done = true;              // stp = stpmax
        if (stp == 0.0 && (f > ftest || gdir >= gtest)) done = true;               // stp = stpmin
        if (f <= ftest && fabs(gdir) <= 0.9 * (-ginit)) done = true;               // the strong Wolfe conditions hold
        if (done) return true;
        if (stage == 1 && f <= fx && f > ftest) {                                  // the modified function of the first stage
            double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
            double gm = gdir - gtest, gxm = gx - gtest, gym = gy - gtest;
            dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
            fx = fxm + stx * gtest; fy = fym + sty * gtest; gx = gxm + gtest; gy = gym + gtest;
        } else {
            dcstep(stx, fx, gx, sty, fy, gy, stp, f, gdir, brackt, stmin, stmax);
        }
        if (brackt) {
            if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = fabs(sty - stx);
        }
        if (brackt) { stmin = fmin(stx, sty); stmax = fmax(stx, sty); }
        else { stmin = stp + 1.1 * (stp - stx); stmax = stp + 4.0 * (stp - stx); }
        stp = fmax(stp, 0.0);
        stp = fmin(stp, stpmx);
        if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= 0.1 * stmax)) stp = stx;
        return false;
    }

    void trial_point(std::vector<double> &xt) const {
        if (stp == 1.0) xt = z;
        else
            for (int i = 0; i < P; ++i) xt[i] = stp * d[i] + x[i];
    }

    // a new iteration from the iterate (x, phi, g): Cauchy point, subspace step, the first trial point of the line
    // search in xt.  Breakdowns refresh the memory and start over from steepest descent, as mainlb does.
    void begin_iteration(std::vector<double> &xt) {
        for (;;) {
            if (!cnstnd && col > 0) {                                // no bounds at all: the Newton step from x itself
                z = x;
                for (int a = 0; a < 2 * M; ++a) c[a] = 0.0;
            } else if (!cauchy(projected_gradient_norm())) { refresh_memory(); continue; }
            if (col > 0) {
                std::vector<int> free_idx;
                for (int i = 0; i < P; ++i)
                    if (where[i] <= 0) free_idx.push_back(i);
                if (!free_idx.empty() && !subspace_min(free_idx)) { refresh_memory(); continue; }
            }
            double dtd = 0.0;
            gd = 0.0;
            for (int i = 0; i < P; ++i) { d[i] = z[i] - x[i]; dtd += d[i] * d[i]; gd += g[i] * d[i]; }
            const double dnorm = sqrt(dtd);
            stpmx = 1e10;
            if (cnstnd) {
                if (iters == 0) stpmx = 1.0;
                else
                    for (int i = 0; i < P; ++i) {
                        const double a1 = d[i];
                        if (a1 < 0.0 && has_lo(i)) {
                            const double a2 = lo[i] - x[i];
                            if (a2 >= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx < a2) stpmx = a2 / a1;
                        } else if (a1 > 0.0 && has_hi(i)) {
                            const double a2 = hi[i] - x[i];
                            if (a2 <= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx > a2) stpmx = a2 / a1;
                        }
                    }
            }
            stp = (iters == 0 && !boxed) ? fmin(1.0 / dnorm, stpmx) : 1.0;
            if (!(gd < 0.0)) {                                       // not a descent direction
                if (col == 0) { status = 2; xt = x; return; }
                refresh_memory();
                continue;
            }
            fold = phi;
            gdold = gd;
            ifun = 0;
            // dcsrch, task START
            brackt = false; stage = 1;
            finit = phi; ginit = gd; gtest = 1e-3 * ginit;
            width = stpmx; width1 = 2.0 * width;
            stx = 0.0; fx = finit; gx = ginit; sty = 0.0; fy = finit; gy = ginit;
            stmin = 0.0; stmax = stp + 4.0 * stp;
            ifun = 1;
            trial_point(xt);
            return;
        }
    }

    // (phit, gt) = the objective and its gradient at xt; leaves the next point to evaluate in xt (the iterate itself
    // once the start has finished).  pgtol and ftol are SciPy's gtol and ftol (= factr x machine epsilon).
    void step(std::vector<double> &xt, const std::vector<double> &gt, double phit, bool first, double pgtol, double ftol) {
        x_eval = xt; g_eval = gt; f_eval = phit;
        if (first) {
            cnstnd = false; boxed = true;
            for (int i = 0; i < P; ++i) {
                x[i] = clip(xt[i], lo[i], hi[i]);
                cnstnd = cnstnd || has_lo(i) || has_hi(i);
                boxed = boxed && has_lo(i) && has_hi(i);
                where[i] = (has_lo(i) && has_hi(i) && hi[i] - lo[i] <= 0.0) ? 3 : ((has_lo(i) || has_hi(i)) ? 0 : -1);
            }
            g = gt; phi = phit;
            iters = 0; status = 0;
            refresh_memory();
            // (a start where K is not positive definite: +inf with a zero gradient, which SciPy -- and so scikit-learn --
            // takes for a stationary point: "converged", f = inf; the caller's arg-min never picks it)
            if (projected_gradient_norm() <= pgtol) { status = 1; xt = x; return; }
            if (!isfinite(phit)) { status = 2; xt = x; return; }
            begin_iteration(xt);
            return;
        }
        if (status != 0) { xt = x; return; }
        // ---- an evaluation inside the line search ----
        // (a trial point where K is not positive definite comes back as +inf with a zero gradient, _gpr.py:586-589.
        // Nothing special is done about it, because SciPy does nothing special: the inf goes through dcstep's
        // interpolation as IEEE arithmetic takes it -- the cubic step becomes NaN, fmax(NaN, stpmin) makes the next
        // trial the best step of the search so far, and the search ends there with "rounding errors prevent
        // progress"; when the FIRST trial of a search is such a point that best step is 0 and the start stops at its
        // current iterate with "relative reduction of f <= factr x epsmch".  scikit-learn's restarts end that way, so
        // these do too.)
        const double f = phit;
        double gdir = 0.0;
        for (int i = 0; i < P; ++i) gdir += gt[i] * d[i];
        if (!dcsrch_next(f, gdir)) {
            if (ifun >= MAXLS) {                                     // twenty trials and no acceptable step
                if (col == 0) { status = 2; xt = x; return; }
                refresh_memory();
                begin_iteration(xt);
                return;
            }
            ++ifun;
            trial_point(xt);
            return;
        }
        // ---- the step is accepted: xt is the new iterate ----
        iters += 1;
        gt_prev = g;
        const std::vector<double> xold = x;
        x = xt; g = gt; phi = phit;
        gd = gdir;
        if (!isfinite(phit)) { x = xold; g = gt_prev; phi = fold; status = 2; xt = x; return; }
        if (projected_gradient_norm() <= pgtol) { status = 1; xt = x; return; }
        if (fold - phi <= ftol * fmax(fmax(fabs(fold), fabs(phi)), 1.0)) { status = 1; xt = x; return; }
        // ---- the new pair ----
        std::vector<double> yv((size_t)P);
        double rr = 0.0, dr, ddum;
        for (int i = 0; i < P; ++i) { yv[(size_t)i] = g[i] - gt_prev[i]; rr += yv[(size_t)i] * yv[(size_t)i]; }
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else {
            dr = (gd - gdold) * stp;
            for (int i = 0; i < P; ++i) d[i] *= stp;
            ddum = -gdold * stp;
        }
        if (!(dr <= EPS * ddum)) {
            if (col == M) {                                          // drop the oldest pair
                for (int m = 0; m + 1 < M; ++m) { S[m].swap(S[m + 1]); Y[m].swap(Y[m + 1]); }
                for (int i = 0; i + 1 < M; ++i)
                    for (int j = 0; j + 1 < M; ++j) { sy[i][j] = sy[i + 1][j + 1]; ss[i][j] = ss[i + 1][j + 1]; }
                col = M - 1;
            }
            S[col] = d;
            Y[col] = yv;
            for (int j = 0; j <= col; ++j) {
                double a = 0.0, b = 0.0, e = 0.0;
                for (int i = 0; i < P; ++i) { a += S[col][i] * Y[j][i]; b += S[j][i] * Y[col][i]; e += S[col][i] * S[j][i]; }
                sy[col][j] = a; sy[j][col] = b; ss[col][j] = e; ss[j][col] = e;
            }
            sy[col][col] = dr;
            col += 1;
            theta = rr / dr;
            if (!factor_T()) refresh_memory();
        }
        begin_iteration(xt);
    }
};

}  // namespace tgp
