// query_math.hpp -- the closed forms shared by the batched query kernels (query_kernels.hip) and the
// one-launch optimiser for small problems (refine_kernels.hip): the kernel's radial derivative
// weight, the normal CDF, and the acquisition as a function of (mu, sigma) with its two partials.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "pairwise.hpp"
#include "tgp_internal.hpp"

namespace tgp {

// h(r) with dk/dx_d = -c h(r) (u_d - xs_d) / l_d   (as in the LML gradient)
template <int KIND>
__device__ __forceinline__ double h_weight(double d2) {
    if (KIND == TGP_RBF) {
        return exp(-0.5 * d2);
    } else if (KIND == TGP_MATERN12) {
        const double r = sqrt(d2);
        return r > 0.0 ? exp(-r) / r : 0.0;
    } else if (KIND == TGP_MATERN32) {
        return 3.0 * exp(-sqrt(3.0 * d2));
    } else {
        const double t = sqrt(5.0 * d2);
        return 5.0 / 3.0 * (t + 1.0) * exp(-t);
    }
}

__device__ __forceinline__ double ndtr_q(double a) {
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    const double y = 0.5 * erfc(z);
    return x > 0 ? 1.0 - y : y;
}

// acq = f(mu, sigma) and its partials: d acq = cm dmu + cs dsigma
// (turbo/modules/acquisition_functions.py:147-158 UCB, :225-247 PI, :336-358 EI)
struct AcqCoef { double a, cm, cs; };
__device__ __forceinline__ AcqCoef acq_coef(int acq, double mu, double sigma, double sf, double incumbent, double param) {
    AcqCoef r{0.0, 0.0, 0.0};
    if (acq == TGP_ACQ_NONE) {
        r.a = mu; r.cm = 1.0;
    } else if (acq == TGP_ACQ_UCB) {
        r.a = sf * mu + param * sigma; r.cm = sf; r.cs = param;
    } else if (acq == TGP_ACQ_SIGMA) {
        r.a = sigma; r.cs = 1.0;
    } else if (sigma != 0.0) {
        const double diff = sf * (mu - incumbent) - param;
        const double Z = diff / sigma;
        const double pdf = exp(-(Z * Z) / 2.0) / 2.5066282746310002;
        const double cdf = ndtr_q(Z);
        if (acq == TGP_ACQ_PI) {
            r.a = cdf; r.cm = pdf * sf / sigma; r.cs = -pdf * Z / sigma;
        } else {
            r.a = diff * cdf + sigma * pdf; r.cm = sf * cdf; r.cs = pdf;
        }
    }
    return r;
}

}  // namespace tgp
