// Driver of tests/test_host_lbfgs.py::test_lbfgsb_under_sanitizers: csrc/host_lbfgsb.hpp compiled with
// -fsanitize=address,undefined and walked over objectives that reach every branch -- memory wrap-around (more than 10
// pairs), active bounds at the Cauchy point, the subspace step's projection and back-tracking, a wall of +inf, a start
// inside the wall, lo == hi, no bounds at all, the iteration limit (GPU sanitizers are not available on this pool: CPU
// build only).
#include <cmath>
#include <cstdio>
#include <vector>

#include "host_lbfgsb.hpp"

static double rosen(const std::vector<double> &x, std::vector<double> &g, double wall) {
    const size_t P = x.size();
    if (x[0] > wall) { for (auto &v : g) v = 0.0; return INFINITY; }
    double f = 0.0;
    for (auto &v : g) v = 0.0;
    for (size_t i = 0; i + 1 < P; ++i) {
        const double a = x[i + 1] - x[i] * x[i], b = 1.0 - x[i];
        f += 100.0 * a * a + b * b;
        g[i] += -400.0 * a * x[i] - 2.0 * b;
        g[i + 1] += 200.0 * a;
    }
    return f;
}

int main() {
    struct Case { int P; double lo, hi, x0, wall; int max_iter; };
    const Case cases[] = {{2, -5, 5, -1.2, 1e9, 15000}, {66, -5, 5, -1.2, 1e9, 15000}, {20, -0.5, 0.8, 0.0, 1e9, 15000},
                          {5, -2, 2, -1.5, 0.3, 15000}, {3, -2, 2, 1.0, 0.5, 15000}, {8, -INFINITY, INFINITY, -1.2, 1e9, 15000},
                          {12, -5, 5, -1.2, 1e9, 7}, {4, 0.5, 0.5, 0.5, 1e9, 15000}};
    for (const Case &c : cases) {
        std::vector<double> lo((size_t)c.P, c.lo), hi((size_t)c.P, c.hi), xt((size_t)c.P, c.x0), gt((size_t)c.P);
        if (c.P == 20) { hi[3] = INFINITY; lo[7] = -INFINITY; }            // half-open coordinates among boxed ones
        tgp::HostLbfgsb opt(lo.data(), hi.data(), c.P);
        int evals = 0, it = 0;
        for (; opt.iters < c.max_iter && it < 15000; ++it) {
            double f;
            if (it > 0 && opt.evaluated(xt)) { f = opt.f_eval; gt = opt.g_eval; }
            else { f = rosen(xt, gt, c.wall); ++evals; }
            opt.step(xt, gt, f, it == 0, 1e-5, 2.220446049250313e-09);
            if (opt.status != 0) break;
        }
        bool inside = true;
        for (int k = 0; k < c.P; ++k) inside = inside && opt.x[(size_t)k] >= lo[(size_t)k] && opt.x[(size_t)k] <= hi[(size_t)k];
        printf("P=%d status=%d iters=%d evals=%d f=%.6g inside=%d\n", c.P, opt.status, opt.iters, evals, opt.phi, (int)inside);
    }
    return 0;
}
