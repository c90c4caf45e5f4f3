// host_api.cpp -- libturbogp_host.so: the HOST backend of libturbogp.so (host_backend.cpp) behind the same
// C-ABI names, for machines that have no ROCm installation at all -- the process that loads a recorder and
// plots it (turbo/recorder.py:157-163, turbo/plotting/trials.py:192-195, :574-577) is often a laptop.
// libturbogp.so links libamdhip64.so and cannot even be loaded there; this library is plain C++ (g++, no
// HIP header) and exports exactly the entries a TGP_DEVICE_HOST handle serves in the full library, with the
// same semantics; tgp_create answers TGP_NO_DEVICE for any other device.  turbo_amd/_lib.py falls back to
// it when the full library cannot be loaded.
#include <new>
#include <string>

#include "../../include/turbogp.h"
#include "host_backend.hpp"
#include "tuning.hpp"
#include <string.h>
#include <algorithm>

struct tgp_handle_s {
    tgp_host::HostGP g;
};

static thread_local std::string g_create_err;

#define HOST_TRY(expr)                                                                        \
    try {                                                                                     \
        return (expr);                                                                        \
    } catch (const std::bad_alloc &) {                                                        \
        if (h) h->g.err = "out of host memory";                                               \
        return TGP_NO_MEMORY;                                                                 \
    } catch (...) {                                                                           \
        if (h) h->g.err = "unexpected C++ exception";                                         \
        return TGP_HIP_ERROR;                                                                 \
    }

extern "C" {

const char *tgp_version(void) { return "turbogp 0.1 host-only (no gfx950 code: reload / plot path)"; }
const char *tgp_last_error(tgp_handle h) { return h ? h->g.err.c_str() : g_create_err.c_str(); }

int tgp_create(int device, int dtype, tgp_handle *out) {
    if (!out) { g_create_err = "tgp_create: out is NULL"; return TGP_BAD_ARG; }
    *out = nullptr;
    if (dtype < TGP_F64 || dtype > TGP_F32H2) { g_create_err = "tgp_create: unknown dtype"; return TGP_BAD_ARG; }
    if (device != TGP_DEVICE_HOST) {
        g_create_err = "tgp_create: no HIP device (this is libturbogp_host.so, the host-only build: TGP_DEVICE_HOST is the only device)";
        return TGP_NO_DEVICE;
    }
    tgp_handle h = new (std::nothrow) tgp_handle_s();
    if (!h) { g_create_err = "tgp_create: out of host memory"; return TGP_NO_MEMORY; }
    *out = h;
    return TGP_OK;
}

int tgp_destroy(tgp_handle h) {
    delete h;
    return TGP_OK;
}

int tgp_fit(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel, double constant,
            const double *ls, int64_t n_ls, double noise, double jitter, int normalize_y, double *lml,
            double *y_mean, double *y_std) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.fit(X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std))
}

int tgp_fit_append(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel, double constant,
                   const double *ls, int64_t n_ls, double noise, double jitter, int normalize_y, double *lml,
                   double *y_mean, double *y_std, int *appended) {
    if (!h) return TGP_BAD_ARG;
    if (appended) *appended = 0;
    HOST_TRY(h->g.fit(X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std))
}

int tgp_export_state(tgp_handle h, void *buf, int64_t cap, int64_t *size) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.export_state(buf, cap, size))
}

int tgp_import_state(tgp_handle h, const void *buf, int64_t size, double *lml) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.import_state(buf, size, lml))
}

int tgp_debug_read(tgp_handle h, int which, double *out) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.debug_read(which, out))
}

int tgp_set_candidates(tgp_handle h, const double *Xc, int64_t M) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.set_candidates(Xc, M))
}

int tgp_read_candidates(tgp_handle h, int64_t first, int64_t count, double *out) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.read_candidates(first, count, out))
}

int tgp_get_candidate(tgp_handle h, int64_t idx, double *out_row) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.read_candidates(idx, 1, out_row))
}

int tgp_sweep(tgp_handle h, int acq, double sf, double incumbent, double param, double *mu, double *sigma,
              double *acq_out, double *best_val, int64_t *best_idx, int64_t *n_clamped) {
    if (!h) return TGP_BAD_ARG;
    HOST_TRY(h->g.sweep(acq, sf, incumbent, param, mu, sigma, acq_out, best_val, best_idx, n_clamped))
}

int tgp_evaluate(tgp_handle h, const double *Xc, int64_t M, int acq, double sf, double incumbent, double param,
                 double *mu, double *sigma, double *acq_out, double *best_val, int64_t *best_idx,
                 int64_t *n_clamped) {
    if (!h) return TGP_BAD_ARG;
    try {
        const int rc = h->g.set_candidates(Xc, M);
        return rc != TGP_OK ? rc : h->g.sweep(acq, sf, incumbent, param, mu, sigma, acq_out, best_val, best_idx, n_clamped);
    } catch (const std::bad_alloc &) {
        h->g.err = "out of host memory";
        return TGP_NO_MEMORY;
    } catch (...) {
        h->g.err = "unexpected C++ exception";
        return TGP_HIP_ERROR;
    }
}

int tgp_predict(tgp_handle h, const double *Xc, int64_t M, double *mu, double *sigma) {
    return tgp_evaluate(h, Xc, M, TGP_ACQ_NONE, 1.0, 0.0, 0.0, mu, sigma, nullptr, nullptr, nullptr, nullptr);
}

int tgp_mt19937_uniform_columns(uint32_t *key624, int32_t *pos, int64_t M, int64_t D, const double *lo, const double *hi,
                                double *out) {
    try {
        return tgp_host::mt19937_uniform_columns(key624, pos, M, D, lo, hi, out);
    } catch (const std::bad_alloc &) {
        return TGP_NO_MEMORY;
    } catch (...) {
        return TGP_HIP_ERROR;
    }
}

int64_t tgp_tuning(char *buf, int64_t cap) {
    try {
        const std::string t = tgp::tuning().dump();
        if (buf && cap > 0) {
            const size_t n = std::min<size_t>((size_t)cap - 1, t.size());
            memcpy(buf, t.data(), n);
            buf[n] = 0;
        }
        return (int64_t)t.size() + 1;
    } catch (...) {
        return -1;
    }
}

int tgp_profile_enable(tgp_handle h, int on) { (void)on; return h ? TGP_OK : TGP_BAD_ARG; }
int tgp_profile_reset(tgp_handle h) { return h ? TGP_OK : TGP_BAD_ARG; }

int tgp_profile_read(tgp_handle h, int64_t *trmm_launches, double *trmm_ms, int64_t *kstar_launches, double *kstar_ms,
                     double *last_fit_ms, double *last_sweep_ms) {
    if (!h) return TGP_BAD_ARG;
    if (trmm_launches) *trmm_launches = 0;
    if (trmm_ms) *trmm_ms = 0.0;
    if (kstar_launches) *kstar_launches = 0;
    if (kstar_ms) *kstar_ms = 0.0;
    if (last_fit_ms) *last_fit_ms = h->g.last_fit_ms;
    if (last_sweep_ms) *last_sweep_ms = h->g.last_sweep_ms;
    return TGP_OK;
}

int tgp_last_timings(tgp_handle h, double *out, int64_t n) {
    if (!h) return TGP_BAD_ARG;
    if (!out || n < 1) { h->g.err = "tgp_last_timings: need out and n >= 1"; return TGP_BAD_ARG; }
    for (int64_t i = 0; i < n; ++i) out[i] = i == 0 ? h->g.last_fit_ms : (i == 1 ? h->g.last_sweep_ms : (i == 6 ? 1.0 : 0.0));   // [6]: the host backend is float64 throughout
    return TGP_OK;
}

int tgp_sweep_geometry(tgp_handle h, int64_t *chunk, int64_t *n_padded) {
    if (!h) return TGP_BAD_ARG;
    if (chunk) *chunk = 16;
    if (n_padded) *n_padded = h->g.N;
    return TGP_OK;
}

}  // extern "C"
