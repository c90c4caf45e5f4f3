/* A plain-C consumer of include/turbogp.h: proves the boundary needs neither C++ nor Python.
 * Built and run by tests (gcc, linked against turbo_amd/csrc/libturbogp.so).
 *   c_abi_consumer            -> symbol / error-path checks and the HOST backend (tgp_create(TGP_DEVICE_HOST):
 *                                the reload path of include/turbogp.h), no GPU needed, exit 0
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
        /* the host backend: the same two-point closed forms the --gpu run checks on the device */
        tgp_handle hh = NULL;
        double HX[2] = {0.0, 1.0}, Hy[2] = {1.0, 3.0}, hls = 0.5, hlml = 0, hym = 0, hys = 0;
        CHECK(tgp_create(TGP_DEVICE_HOST, TGP_F64, &hh) == TGP_OK && hh != NULL);
        CHECK(tgp_fit(hh, HX, 2, 1, Hy, TGP_RBF, 1.0, &hls, 1, 0.0, 1e-10, 1, &hlml, &hym, &hys) == TGP_OK);
        CHECK(fabs(hym - 2.0) < 1e-15 && fabs(hys - 1.0) < 1e-15);
        {
            const double a = 1e-10, k = exp(-0.5 * 4.0), e = 1.0 + a - k;
            const double want = -0.5 * (2.0 / e) - 0.5 * log((1.0 + a) * (1.0 + a) - k * k) - log(2.0 * 3.14159265358979323846);
            double Xc[3] = {0.0, 0.5, 1.0}, mu[3], sg[3], acq[3], best = 0, row = -1, lml2 = 0;
            int64_t idx = -1, clamped = -1, need = 0;
            char blob[256];
            CHECK(fabs(hlml - want) < 1e-12);
            CHECK(tgp_evaluate(hh, Xc, 3, TGP_ACQ_UCB, 1.0, 0.0, 0.0, mu, sg, acq, &best, &idx, &clamped) == TGP_OK);
            CHECK(fabs(mu[0] - 1.0) < 1e-8 && fabs(mu[2] - 3.0) < 1e-8 && fabs(mu[1] - 2.0) < 1e-12);
            CHECK(sg[0] < 1e-4 && sg[2] < 1e-4 && sg[1] > 0.1);
            CHECK(idx == 2 && best == acq[2] && clamped >= 0);
            CHECK(tgp_get_candidate(hh, 1, &row) == TGP_OK && row == 0.5);
            CHECK(tgp_predict(hh, Xc, 3, mu, NULL) == TGP_OK && fabs(mu[1] - 2.0) < 1e-12);
            CHECK(tgp_export_state(hh, NULL, 0, &need) == TGP_OK && need == (8 + 1 + 2 + 2) * 8);
            CHECK(tgp_export_state(hh, blob, sizeof blob, &need) == TGP_OK);
            CHECK(tgp_import_state(hh, blob, need, &lml2) == TGP_OK && lml2 == hlml);
            /* what only exists on the GPU says so instead of touching HIP */
            CHECK(tgp_sweep_topk(hh, TGP_ACQ_UCB, 1.0, 0.0, 0.0, 1, mu, &idx, NULL) == TGP_BAD_ARG);
            CHECK(strstr(tgp_last_error(hh), "host backend") != NULL);
            /* a singular kernel matrix is reported, not factored */
            {
                double SX[2] = {0.25, 0.25};
                CHECK(tgp_fit(hh, SX, 2, 1, Hy, TGP_RBF, 1.0, &hls, 1, 0.0, 0.0, 1, NULL, NULL, NULL) == TGP_NOT_PD);
                CHECK(tgp_predict(hh, Xc, 3, mu, NULL) == TGP_NOT_FITTED);
            }
        }
        CHECK(tgp_destroy(hh) == TGP_OK);
        {   /* the reference's host candidate draw from C: MT19937 seeded by init_genrand(5489) -- mt19937ar.c's default --
               gives 0.8147236863931789 as its first 53-bit double; column 0 is [0, 1), column 1 is 2 + 4 u */
            uint32_t key[624], s0 = 5489u;
            int32_t pos = 624;
            int i;
            double lo[2] = {0.0, 2.0}, hi[2] = {1.0, 6.0}, out[6];
            for (i = 0; i < 624; ++i) { key[i] = s0; s0 = 1812433253u * (s0 ^ (s0 >> 30)) + (uint32_t)i + 1u; }
            CHECK(tgp_mt19937_uniform_columns(key, &pos, 3, 2, lo, hi, out) == TGP_OK);
            CHECK(pos == 12 && out[0] == 0.8147236863931789 && out[1] >= 2.0 && out[1] < 6.0 && out[5] >= 2.0 && out[5] < 6.0);
            pos = 625;
            CHECK(tgp_mt19937_uniform_columns(key, &pos, 3, 2, lo, hi, out) == TGP_BAD_ARG && pos == 625);
        }
        printf("c-abi ok (no gpu, host backend)\n");
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
    {
        /* round 5: the switches' table, the overlap switch, the device's worker pool -- from plain C */
        char tbuf[64];
        tgp_handle w[4] = {NULL, NULL, NULL, NULL}, w2[2] = {NULL, NULL};
        int64_t need = tgp_tuning(NULL, 0);
        CHECK(need > 1000 && tgp_tuning(tbuf, sizeof tbuf) == need && strncmp(tbuf, "TGP_", 4) == 0 && strlen(tbuf) == sizeof tbuf - 1);
        CHECK(tgp_set_overlap(h, 3) == TGP_BAD_ARG && tgp_set_overlap(h, 2) == TGP_OK && tgp_set_overlap(h, 0) == TGP_OK);
        CHECK(tgp_workers_release(h) == TGP_BAD_ARG);                       /* not acquired by this thread */
        CHECK(tgp_workers_acquire(h, 5, w) == TGP_BAD_ARG && tgp_workers_acquire(h, 3, w) == TGP_OK && w[0] && w[1] && w[2] && w[0] != w[1]);
        CHECK(tgp_fit(w[1], X, 2, 1, y, TGP_RBF, 1.0, &ls, 1, 0.0, 1e-10, 1, &lml, &ym, &ys) == TGP_OK);
        CHECK(tgp_destroy(w[0]) == TGP_BAD_ARG && tgp_workers_acquire(w[0], 1, w2) == TGP_BAD_ARG);   /* the library's, and no borrowing through one */
        CHECK(tgp_workers_acquire(h, 2, w2) == TGP_BAD_ARG);                /* already held by this thread: refused, no deadlock */
        CHECK(tgp_workers_release(h) == TGP_OK);
        CHECK(tgp_workers_acquire(h, 2, w2) == TGP_OK && w2[0] == w[0] && w2[1] == w[1] && tgp_workers_release(h) == TGP_OK);   /* ONE pool */
    }
    CHECK(tgp_destroy(h) == TGP_OK);
    {
        /* one process, several contexts (here three on device 0): the sharded sweep equals the
         * single-handle sweep, value, GLOBAL index and row */
        enum { NT = 40, ND = 3, NM = 1000 };
        static double TX[NT * ND], Ty[NT], TC[NM * ND], acq1[NM], acqm[NM];
        unsigned s = 12345u;
        int i;
        for (i = 0; i < NT * ND; ++i) { s = s * 1664525u + 1013904223u; TX[i] = (s >> 8) / 16777216.0; }
        for (i = 0; i < NT; ++i) Ty[i] = sin(3.0 * TX[i * ND]) + TX[i * ND + 1] * TX[i * ND + 2];
        for (i = 0; i < NM * ND; ++i) { s = s * 1664525u + 1013904223u; TC[i] = (s >> 8) / 16777216.0; }
        double l1 = 0, lm = 0, b1 = 0, bm = 0, row[ND], ls3 = 0.7;
        int64_t i1 = -1, im = -1;
        int ids[3] = {0, 0, 0};
        tgp_handle one = NULL;
        tgp_multi m = NULL;
        CHECK(tgp_create(0, TGP_F64, &one) == TGP_OK);
        CHECK(tgp_fit(one, TX, NT, ND, Ty, TGP_MATERN52, 1.3, &ls3, 1, 1e-3, 1e-10, 1, &l1, NULL, NULL) == TGP_OK);
        CHECK(tgp_set_candidates(one, TC, NM) == TGP_OK);
        CHECK(tgp_sweep(one, TGP_ACQ_EI, -1.0, -0.5, 0.01, NULL, NULL, acq1, &b1, &i1, NULL) == TGP_OK);
        CHECK(tgp_multi_create(3, ids, TGP_F64, &m) == TGP_OK && tgp_multi_size(m) == 3);
        CHECK(tgp_multi_sweep(m, TGP_ACQ_EI, -1.0, -0.5, 0.01, &bm, &im, row, NULL) == TGP_BAD_ARG);   /* nothing set yet */
        CHECK(tgp_multi_fit(m, TX, NT, ND, Ty, TGP_MATERN52, 1.3, &ls3, 1, 1e-3, 1e-10, 1, &lm, NULL, NULL) == TGP_OK);
        CHECK(lm == l1);
        CHECK(tgp_multi_set_candidates(m, TC, NM) == TGP_OK);
        CHECK(tgp_multi_sweep(m, TGP_ACQ_EI, -1.0, -0.5, 0.01, &bm, &im, row, acqm) == TGP_OK);
        CHECK(bm == b1 && im == i1);
        CHECK(memcmp(acq1, acqm, sizeof acq1) == 0);
        CHECK(row[0] == TC[im * ND] && row[1] == TC[im * ND + 1] && row[2] == TC[im * ND + 2]);
        CHECK(tgp_multi_set_candidates(m, TC, 2) == TGP_OK);               /* fewer rows than devices */
        CHECK(tgp_multi_sweep(m, TGP_ACQ_UCB, 1.0, 0.0, 1.0, &bm, &im, row, NULL) == TGP_OK && im >= 0 && im < 2);
        CHECK(tgp_multi_destroy(m) == TGP_OK);
        {
            /* the hyper-parameter fit from plain C: two starts walked by L-BFGS-B inside the library; the better one's
             * -LML is what a fit at its theta reports, and it is not worse than where it started */
            const double th0[6] = {0.0, -0.3, -4.0, 1.0, 0.5, -2.0}, lo[3] = {-11.5, -11.5, -11.5}, hi[3] = {11.5, 11.5, 11.5};
            double th[6], f[2], lbest = 0, lsb;
            int64_t st[2] = {-1, -1}, ev = 0;
            int b;
            CHECK(tgp_fit_lbfgsb(one, TX, NT, ND, Ty, TGP_MATERN52, th0, 2, 1, lo, hi, 1e-10, 1, 15000, th, f, st, &ev) == TGP_OK);
            CHECK(ev >= 4 && st[0] >= 0 && st[0] <= 2 && st[1] >= 0 && st[1] <= 2);
            b = f[1] < f[0] ? 1 : 0;
            lsb = exp(th[3 * b + 1]);
            CHECK(tgp_fit(one, TX, NT, ND, Ty, TGP_MATERN52, exp(th[3 * b]), &lsb, 1, exp(th[3 * b + 2]), 1e-10, 1, &lbest, NULL, NULL) == TGP_OK);
            CHECK(fabs(lbest + f[b]) <= 1e-9 * fabs(lbest) + 1e-9);
            {
                double l_start = 0, ls0 = exp(th0[1]);
                CHECK(tgp_fit(one, TX, NT, ND, Ty, TGP_MATERN52, exp(th0[0]), &ls0, 1, exp(th0[2]), 1e-10, 1, &l_start, NULL, NULL) == TGP_OK);
                CHECK(lbest >= l_start - 1e-9);
            }
            CHECK(tgp_fit_lbfgsb(one, TX, NT, ND, Ty, TGP_MATERN52, th0, 2, 1, hi, lo, 1e-10, 1, 15000, th, f, st, &ev) == TGP_BAD_ARG);   /* lo > hi */
        }
        {
            /* ... and the gradient stage: three restarts walked by L-BFGS-B in lock-step on the fitted model; every end
             * point lies in the box, its value is the acquisition there and not below the start's */
            const double x0[3 * ND] = {0.2, 0.3, 0.4, 0.9, 0.1, 0.5, 0.5, 0.5, 0.5}, blo[ND] = {0.0, 0.0, 0.0}, bhi[ND] = {1.0, 1.0, 1.0};
            double xr[3 * ND], vr[3], v0[3], g0[3 * ND], v1[3], g1[3 * ND];
            int64_t rs[3] = {-1, -1, -1}, nev = 0;
            int r, d;
            CHECK(tgp_acq_lbfgsb(one, x0, 3, blo, bhi, TGP_ACQ_EI, -1.0, -0.5, 0.01, 15000, xr, vr, rs, &nev) == TGP_OK);
            CHECK(tgp_acq_grad(one, x0, 3, TGP_ACQ_EI, -1.0, -0.5, 0.01, v0, g0) == TGP_OK);
            CHECK(tgp_acq_grad(one, xr, 3, TGP_ACQ_EI, -1.0, -0.5, 0.01, v1, g1) == TGP_OK);
            CHECK(nev >= 3);
            for (r = 0; r < 3; ++r) {
                CHECK(rs[r] >= 0 && rs[r] <= 2 && vr[r] >= v0[r] - 1e-15 && fabs(vr[r] - v1[r]) <= 1e-12 * (1.0 + fabs(vr[r])));
                for (d = 0; d < ND; ++d) CHECK(xr[r * ND + d] >= 0.0 && xr[r * ND + d] <= 1.0);
            }
            CHECK(tgp_acq_lbfgsb(one, x0, 3, bhi, blo, TGP_ACQ_EI, -1.0, -0.5, 0.01, 15000, xr, vr, rs, &nev) == TGP_BAD_ARG);   /* lo > hi */
        }
        {   /* the same draw finished on the GPU: rows 1..2 of a 3-row batch of the model's ND columns, the stream
               passed over for the rest; what comes back is what the host entry forms */
            uint32_t key[624], key2[624], s0 = 5489u;
            int32_t pos = 624, pos2 = 624;
            int i, d;
            double blo[ND], bhi[ND], host[3 * ND], dev[2 * ND];
            for (i = 0; i < 624; ++i) { key[i] = s0; s0 = 1812433253u * (s0 ^ (s0 >> 30)) + (uint32_t)i + 1u; }
            memcpy(key2, key, sizeof key);
            for (d = 0; d < ND; ++d) { blo[d] = -1.0 - d; bhi[d] = 2.5 + d; }
            CHECK(tgp_mt19937_uniform_columns(key, &pos, 3, ND, blo, bhi, host) == TGP_OK);
            CHECK(tgp_set_candidates_mt19937(one, key2, &pos2, 3, 1, 2, blo, bhi) == TGP_OK);
            CHECK(pos2 == pos && memcmp(key, key2, sizeof key) == 0);
            CHECK(tgp_read_candidates(one, 0, 2, dev) == TGP_OK);
            CHECK(memcmp(dev, host + ND, sizeof dev) == 0);
        }
        CHECK(tgp_destroy(one) == TGP_OK);
    }
    printf("c-abi ok (gpu)\n");
    return 0;
}
