// multi_api.hip -- one process, several GPUs: the sharded candidate sweep of SURVEY.md 8e behind the
// C-ABI, for consumers that are not Python (turbo_amd/distributed.py does the same across
// processes with torch.distributed / RCCL).  A tgp_multi owns one tgp_handle per device:
//   fit        replicated on every device (identical inputs -> identical factor; no exchange)
//   candidates contiguous shards of ceil(M / n) rows, uploaded or drawn per device
//   sweep      every device sweeps its shard on its own stream, driven by its own host thread;
//              the n winners (value, global index) land in host memory with each device's own
//              synchronisation and are reduced there with the library-wide rule (largest value,
//              lowest global index, NaN never wins).  The whole exchange is 16 bytes per device that
//              are already on the host, so no collective library is linked.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/turbogp.h"

struct tgp_multi_s {
    std::vector<tgp_handle> h;
    std::vector<int64_t> off, cnt;     // shard of each device in the resident batch
    int64_t M = 0, D = 0;
    std::string err;
};

static thread_local std::string g_multi_create_err;

namespace {

// run f(i) for every device on its own host thread; returns the first non-OK status
template <typename F>
int for_each_device(tgp_multi m, F f, bool skip_empty = false) {
    const size_t n = m->h.size();
    std::vector<int> rc(n, TGP_OK);
    std::vector<std::thread> th;
    th.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (skip_empty && m->cnt[i] == 0) continue;
        th.emplace_back([&, i]() { rc[i] = f((int)i); });
    }
    for (auto &t : th) t.join();
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != TGP_OK) {
            m->err = "device " + std::to_string(i) + ": " + tgp_last_error(m->h[i]);
            return rc[i];
        }
    return TGP_OK;
}

void plan_shards(tgp_multi m, int64_t M) {
    const int64_t n = (int64_t)m->h.size(), per = (M + n - 1) / n;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t lo = std::min(i * per, M), hi = std::min(lo + per, M);
        m->off[(size_t)i] = lo;
        m->cnt[(size_t)i] = hi - lo;
    }
    m->M = M;
}

}  // namespace

#define MULTI_CATCH                                                                              \
    catch (const std::bad_alloc &) { if (m) m->err = "out of host memory"; return TGP_NO_MEMORY; } \
    catch (const std::exception &ex_) { if (m) m->err = ex_.what(); return TGP_HIP_ERROR; }        \
    catch (...) { if (m) m->err = "unknown C++ exception"; return TGP_HIP_ERROR; }

extern "C" {

const char *tgp_multi_last_error(tgp_multi m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

int tgp_multi_create(int n, const int *device_ids, int dtype, tgp_multi *out) {
    if (!out) { g_multi_create_err = "tgp_multi_create: out is NULL"; return TGP_BAD_ARG; }
    *out = nullptr;
    if (n < 1 || n > 64 || !device_ids) { g_multi_create_err = "tgp_multi_create: need 1 <= n <= 64 device ids"; return TGP_BAD_ARG; }
    tgp_multi m = new (std::nothrow) tgp_multi_s();
    if (!m) { g_multi_create_err = "tgp_multi_create: out of host memory"; return TGP_NO_MEMORY; }
    for (int i = 0; i < n; ++i)
        if (device_ids[i] < 0) { delete m; g_multi_create_err = "tgp_multi_create: device ids must be HIP devices (the host backend is single-handle)"; return TGP_BAD_ARG; }
    for (int i = 0; i < n; ++i) {
        tgp_handle h = nullptr;
        const int rc = tgp_create(device_ids[i], dtype, &h);
        if (rc != TGP_OK) {
            g_multi_create_err = std::string("tgp_multi_create: device ") + std::to_string(device_ids[i]) + ": " + tgp_last_error(nullptr);
            for (auto hh : m->h) (void)tgp_destroy(hh);
            delete m;
            return rc;
        }
        m->h.push_back(h);
    }
    m->off.assign((size_t)n, 0);
    m->cnt.assign((size_t)n, 0);
    *out = m;
    return TGP_OK;
}

int tgp_multi_destroy(tgp_multi m) {
    if (!m) return TGP_OK;
    for (auto h : m->h) (void)tgp_destroy(h);
    delete m;
    return TGP_OK;
}

int tgp_multi_size(tgp_multi m) { return m ? (int)m->h.size() : 0; }

int tgp_multi_handle(tgp_multi m, int i, tgp_handle *out) {
    if (!m || !out || i < 0 || i >= (int)m->h.size()) return TGP_BAD_ARG;
    *out = m->h[(size_t)i];
    return TGP_OK;
}

int tgp_multi_fit(tgp_multi m, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                  double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                  int normalize_y, double *lml, double *y_mean, double *y_std) try {
    if (!m) return TGP_BAD_ARG;
    const size_t n = m->h.size();
    std::vector<double> l(n), ym(n), ys(n);
    const int rc = for_each_device(m, [&](int i) {
        return tgp_fit(m->h[(size_t)i], X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y,
                       &l[(size_t)i], &ym[(size_t)i], &ys[(size_t)i]);
    });
    if (rc != TGP_OK) return rc;
    for (size_t i = 1; i < n; ++i)      // same code, same inputs: the replicas must agree bit for bit
        if (l[i] != l[0] || ym[i] != ym[0] || ys[i] != ys[0]) { m->err = "tgp_multi_fit: replicas disagree"; return TGP_HIP_ERROR; }
    if (D != m->D) { m->M = 0; std::fill(m->cnt.begin(), m->cnt.end(), 0); }
    m->D = D;
    if (lml) *lml = l[0];
    if (y_mean) *y_mean = ym[0];
    if (y_std) *y_std = ys[0];
    return TGP_OK;
} MULTI_CATCH

int tgp_multi_set_candidates(tgp_multi m, const double *Xc, int64_t M) try {
    if (!m) return TGP_BAD_ARG;
    if (!Xc || M < 1 || m->D < 1) { m->err = "tgp_multi_set_candidates: fit first, then Xc and M >= 1"; return TGP_BAD_ARG; }
    plan_shards(m, M);
    return for_each_device(m, [&](int i) {
        return tgp_set_candidates(m->h[(size_t)i], Xc + m->off[(size_t)i] * m->D, m->cnt[(size_t)i]);
    }, true);
} MULTI_CATCH

int tgp_multi_gen_candidates(tgp_multi m, uint64_t seed, int64_t M, const double *lo, const double *hi) try {
    if (!m) return TGP_BAD_ARG;
    if (!lo || !hi || M < 1 || m->D < 1) { m->err = "tgp_multi_gen_candidates: fit first, then lo, hi and M >= 1"; return TGP_BAD_ARG; }
    plan_shards(m, M);
    return for_each_device(m, [&](int i) {      // every device draws ITS rows of the one stream
        return tgp_gen_candidates(m->h[(size_t)i], seed, (uint64_t)m->off[(size_t)i], m->cnt[(size_t)i], lo, hi);
    }, true);
} MULTI_CATCH

int tgp_multi_sweep(tgp_multi m, int acq, double sf, double incumbent, double param, double *best_val,
                    int64_t *best_idx, double *best_row, double *acq_out) try {
    if (!m) return TGP_BAD_ARG;
    if (m->M < 1) { m->err = "tgp_multi_sweep: no candidates set"; return TGP_BAD_ARG; }
    if (acq == TGP_ACQ_NONE) { m->err = "tgp_multi_sweep: needs an acquisition"; return TGP_BAD_ARG; }
    const size_t n = m->h.size();
    std::vector<double> bv(n, -INFINITY);
    std::vector<int64_t> bi(n, -1);
    const int rc = for_each_device(m, [&](int i) {
        const size_t s = (size_t)i;
        return tgp_sweep(m->h[s], acq, sf, incumbent, param, nullptr, nullptr,
                         acq_out ? acq_out + m->off[s] : nullptr, &bv[s], &bi[s], nullptr);
    }, true);
    if (rc != TGP_OK) return rc;
    // reduce: largest value, then lowest global index; NaN never wins (as argmax_final_kernel)
    int w = -1;
    for (size_t i = 0; i < n; ++i) {
        if (m->cnt[i] == 0) continue;
        if (w < 0) { w = (int)i; continue; }
        const double v = bv[i], b = bv[(size_t)w];
        const int64_t gi = m->off[i] + bi[i], gb = m->off[(size_t)w] + bi[(size_t)w];
        if (v > b || (isnan(b) && !isnan(v)) || (v == b && gi < gb)) w = (int)i;
    }
    if (best_val) *best_val = bv[(size_t)w];
    if (best_idx) *best_idx = m->off[(size_t)w] + bi[(size_t)w];
    if (best_row) return tgp_get_candidate(m->h[(size_t)w], bi[(size_t)w], best_row);
    return TGP_OK;
} MULTI_CATCH

}  // extern "C"
