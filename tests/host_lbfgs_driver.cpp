// Test driver (CPU): csrc/host_lbfgsb.hpp behind a C entry so the optimiser tgp_fit_optimise runs above the
// one-launch sizes can be exercised without a GPU, on objectives supplied by the test (ctypes callback).
//   g++ -O2 -std=c++17 -shared -fPIC -I turbo_amd/csrc tests/host_lbfgs_driver.cpp -o <tmp>/libhost_lbfgs_test.so
#include "host_lbfgsb.hpp"

extern "C" int host_lbfgs_minimise(double (*fun)(const double *x, double *grad, void *ctx), void *ctx, int P, double *x,
                                   const double *lo, const double *hi, int max_iter, double pgtol, double ftol,
                                   double *f_out, int *evals_out, int *iters_out) {
    tgp::HostLbfgsb opt(lo, hi, P);
    std::vector<double> xt((size_t)P), gt((size_t)P);
    for (int k = 0; k < P; ++k) xt[(size_t)k] = tgp::HostLbfgsb::clip(x[k], lo[k], hi[k]);
    int it = 0, evals = 0;
    for (; opt.iters < max_iter && it < 15000; ++it) {
        double phit;
        if (it > 0 && opt.evaluated(xt)) { phit = opt.f_eval; gt = opt.g_eval; }
        else { phit = fun(xt.data(), gt.data(), ctx); ++evals; }
        opt.step(xt, gt, phit, it == 0, pgtol, ftol);
        if (opt.status != 0) break;
    }
    for (int k = 0; k < P; ++k) x[k] = opt.x[(size_t)k];
    *f_out = opt.phi;
    *evals_out = evals;
    *iters_out = opt.iters;
    return opt.status;
}
