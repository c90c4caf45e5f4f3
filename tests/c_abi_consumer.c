/* A plain-C consumer of include/turbogp.h: proves the boundary needs neither C++ nor Python.
 * Built and run by tests (gcc, linked against turbo_amd/csrc/libturbogp.so).
 *   c_abi_consumer            -> symbol / error-path checks that need no GPU, exit 0
 *   c_abi_consumer --gpu      -> a tiny fit + sweep on device 0, checked against closed forms */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/turbogp.h"

#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv) {
    int gpu = argc > 1 && strcmp(argv[1], "--gpu") == 0;
    CHECK(strstr(tgp_version(), "gfx950") != NULL);
    CHECK(tgp_destroy(NULL) == TGP_OK);
    CHECK(tgp_fit(NULL, NULL, 0, 0, NULL, 0, 1.0, NULL, 0, 0.0, 0.0, 1, NULL, NULL, NULL) == TGP_BAD_ARG);
    if (!gpu) {
        printf("c-abi ok (no gpu)\n");
        return 0;
    }
    tgp_handle h = NULL;
    CHECK(tgp_create(0, TGP_F64, &h) == TGP_OK);
    /* two observations in 1-D: everything has a closed form */
    double X[2] = {0.0, 1.0}, y[2] = {1.0, 3.0}, ls = 0.5, lml = 0, ym = 0, ys = 0;
    CHECK(tgp_fit(h, X, 2, 1, y, TGP_RBF, 1.0, &ls, 1, 0.0, 1e-10, 1, &lml, &ym, &ys) == TGP_OK);
    CHECK(fabs(ym - 2.0) < 1e-15 && fabs(ys - 1.0) < 1e-15);
    {
        /* K = [[1+a, k],[k, 1+a]], yn = (-1, 1):  alpha = (-1, 1) / (1 + a - k) */
        const double a = 1e-10, k = exp(-0.5 * 4.0), e = 1.0 + a - k;
        const double want = -0.5 * (2.0 / e) - 0.5 * log((1.0 + a) * (1.0 + a) - k * k) - log(2.0 * 3.14159265358979323846);
        CHECK(fabs(lml - want) < 1e-12);
        double Xc[3] = {0.0, 0.5, 1.0}, mu[3], sg[3], acq[3], best = 0;
        int64_t idx = -1, clamped = -1;
        CHECK(tgp_set_candidates(h, Xc, 3) == TGP_OK);
        CHECK(tgp_sweep(h, TGP_ACQ_UCB, 1.0, 0.0, 0.0, mu, sg, acq, &best, &idx, &clamped) == TGP_OK);
        CHECK(fabs(mu[0] - 1.0) < 1e-8 && fabs(mu[2] - 3.0) < 1e-8 && fabs(mu[1] - 2.0) < 1e-12);
        CHECK(sg[0] < 1e-4 && sg[2] < 1e-4 && sg[1] > 0.1);
        CHECK(idx == 2 && fabs(best - acq[2]) == 0.0);   /* beta = 0: the largest mean */
        {
            int64_t need = 0;
            char blob[256];
            CHECK(tgp_export_state(h, NULL, 0, &need) == TGP_OK && need == (8 + 1 + 2 + 2) * 8);
            CHECK(tgp_export_state(h, blob, sizeof blob, &need) == TGP_OK);
            double lml2 = 0;
            CHECK(tgp_import_state(h, blob, need, &lml2) == TGP_OK && lml2 == lml);
        }
    }
    CHECK(tgp_destroy(h) == TGP_OK);
    printf("c-abi ok (gpu)\n");
    return 0;
}
