// host_backend.cpp -- see host_backend.hpp.  Plain C++17: no HIP header, no HIP call.
#include "host_backend.hpp"
#include "tuning.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>

#include "../../include/turbogp.h"

namespace tgp_host {

namespace {

constexpr int64_t PB = 64;   // Cholesky panel width
constexpr int64_t CB = 16;   // candidates per forward-substitution tile
constexpr int64_t HOST_MAX_CANDIDATE_DOUBLES = (int64_t)1 << 34;   // 128 GiB of candidates: a plot grid is 1e2 .. 1e6 rows

int threads_for(double work) {
    const int env = tgp::tuning().host_threads;
    int n = env > 0 ? env : (int)std::min<unsigned>(std::max<unsigned>(std::thread::hardware_concurrency(), 1u), 16u);
    if (work < 4e6) n = 1;   // not worth a thread start
    return std::max(n, 1);
}

// fn(begin, end) over [0, n) in contiguous shares, one per thread.  No exception leaves a worker thread
// (that would be std::terminate): the first one is kept and rethrown here after every thread has been
// joined, so a std::bad_alloc inside a share still reaches the C-ABI's catch as TGP_NO_MEMORY.  A thread
// that cannot be started (std::system_error) is not an error: its share and the rest run on the caller.
void parallel_for(int64_t n, int nthreads, const std::function<void(int64_t, int64_t)> &fn) {
    if (n <= 0) return;
    if (nthreads <= 1 || n == 1) { fn(0, n); return; }
    const int64_t t = std::min<int64_t>(nthreads, n);
    std::vector<std::thread> pool;
    pool.reserve((size_t)t);
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](int64_t b, int64_t e) {
        try {
            fn(b, e);
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu);
            if (!first) first = std::current_exception();
        }
    };
    int64_t started = 0;
    for (; started < t - 1; ++started) {        // the last share is the caller's own
        const int64_t b = n * started / t, e = n * (started + 1) / t;
        try {
            pool.emplace_back(guarded, b, e);
        } catch (const std::system_error &) {
            break;
        }
    }
    guarded(n * started / t, n);
    for (auto &th : pool) th.join();
    if (first) std::rethrow_exception(first);
}

inline double dot(const double *a, const double *b, int64_t n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t k = 0;
    for (; k + 4 <= n; k += 4) {
        s0 += a[k] * b[k];
        s1 += a[k + 1] * b[k + 1];
        s2 += a[k + 2] * b[k + 2];
        s3 += a[k + 3] * b[k + 3];
    }
    for (; k < n; ++k) s0 += a[k] * b[k];
    return (s0 + s1) + (s2 + s3);
}

// sklearn kernels.py: RBF :1557/:1563, Matern :1717-1724, times the ConstantKernel (:966)
inline double kernel_value(int kind, double d2, double c) {
    switch (kind) {
        case TGP_RBF: return c * exp(-0.5 * d2);
        case TGP_MATERN12: return c * exp(-sqrt(d2));
        case TGP_MATERN32: { const double k = sqrt(d2) * 1.7320508075688772; return c * ((1.0 + k) * exp(-k)); }
        default: { const double k = sqrt(d2) * 2.23606797749979; return c * ((1.0 + k + (k * k) * 0.33333333333333333) * exp(-k)); }
    }
}

// scipy.special.ndtr (cephes ndtr.c) behind scipy.stats.norm.cdf
inline double ndtr(double a) {
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    const double y = 0.5 * erfc(z);
    return x > 0 ? 1.0 - y : y;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

const char STATE_MAGIC[8] = {'T', 'G', 'P', 'S', 'T', 'A', 'T', '1'};   // the blob of tgp_export_state (tgp_api.hip)

}  // namespace

int HostGP::fit(const double *Xin, int64_t n, int64_t d, const double *yin, int kern, double c,
                const double *lsin, int64_t n_ls, double nz, double jit, int norm,
                double *lml_out, double *ym_out, double *ys_out) {
    fitted = false;
    if (!Xin || !yin || !lsin) { err = "tgp_fit: X, y and ls must not be NULL"; return TGP_BAD_ARG; }
    if (n < 1 || d < 1) { err = "tgp_fit: need N >= 1 and D >= 1"; return TGP_BAD_ARG; }
    if (n > 65536 || d > 4096) { err = "tgp_fit: N <= 65536 and D <= 4096 supported"; return TGP_BAD_ARG; }
    if (n_ls != 1 && n_ls != d) { err = "tgp_fit: n_ls must be 1 or D"; return TGP_BAD_ARG; }
    if (kern < TGP_RBF || kern > TGP_MATERN52) { err = "tgp_fit: unknown kernel"; return TGP_BAD_ARG; }
    if (!(c > 0.0) || !(nz >= 0.0) || !(jit >= 0.0)) { err = "tgp_fit: constant > 0, noise >= 0, jitter >= 0 required"; return TGP_BAD_ARG; }
    for (int64_t k = 0; k < n_ls; ++k)
        if (!(lsin[k] > 0.0)) { err = "tgp_fit: length scales must be > 0"; return TGP_BAD_ARG; }
    const double t0 = now_ms();
    if (D != d) { cand.clear(); M = 0; }   // resident candidates belong to the old D
    N = n; D = d; kernel = kern; constant = c; noise = nz; jitter = jit; normalize_y = norm ? 1 : 0;
    ls.assign((size_t)d, 0.0);
    for (int64_t k = 0; k < d; ++k) ls[(size_t)k] = lsin[n_ls == 1 ? 0 : k];
    X.assign(Xin, Xin + (size_t)n * d);
    y.assign(yin, yin + (size_t)n);

    // _gpr.py:272-282: mean, population std, an exactly zero std -> 1
    std::vector<double> yn((size_t)n);
    y_mean = 0.0; y_std = 1.0;
    if (normalize_y) {
        long double s = 0.0L;
        for (int64_t i = 0; i < n; ++i) s += y[(size_t)i];
        y_mean = (double)(s / (long double)n);
        long double v = 0.0L;
        for (int64_t i = 0; i < n; ++i) { const long double t = (long double)y[(size_t)i] - y_mean; v += t * t; }
        y_std = sqrt((double)(v / (long double)n));
        if (y_std == 0.0) y_std = 1.0;
        for (int64_t i = 0; i < n; ++i) yn[(size_t)i] = (y[(size_t)i] - y_mean) / y_std;
    } else {
        yn = y;
    }
    Xs.resize((size_t)n * d);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = 0; k < d; ++k) Xs[(size_t)(i * d + k)] = X[(size_t)(i * d + k)] / ls[(size_t)k];   // X / length_scale

    // lower triangle of K, diagonal forced to c * 1 + noise, + jitter (kernels.py:1560, _gpr.py:347)
    L.assign((size_t)n * n, 0.0);
    const int nt = threads_for((double)n * n * d);
    const double *xs = Xs.data();
    double *Lp = L.data();
    const double diag = (c * 1.0 + nz) + jit;
    parallel_for(n, nt, [=](int64_t b, int64_t e) {
        // (shares of equal row counts are unequal in work; the Cholesky below dominates anyway)
        for (int64_t i = b; i < e; ++i) {
            const double *xi = xs + i * d;
            for (int64_t j = 0; j < i; ++j) {
                const double *xj = xs + j * d;
                double d2 = 0.0;
                for (int64_t k = 0; k < d; ++k) { const double df = xi[k] - xj[k]; d2 += df * df; }
                Lp[i * n + j] = kernel_value(kern, d2, c);
            }
            Lp[i * n + i] = diag;
        }
    });

    // Cholesky, left-looking by panels of PB columns: the panel's own rows one after the other,
    // the rows below it in parallel (each L[i][j] is one dot product of two contiguous rows).
    // LAPACK dpotrf stops at a pivot <= 0 (scipy.linalg.cholesky -> LinAlgError, _gpr.py:348-358);
    // like the HIP path a pivot without a significant digit left (< 8 eps of the diagonal) counts too.
    const double tiny = 8.0 * 2.220446049250313e-16 * diag;
    int64_t bad = 0;
    for (int64_t c0 = 0; c0 < n && !bad; c0 += PB) {
        const int64_t c1 = std::min(c0 + PB, n);
        for (int64_t i = c0; i < c1 && !bad; ++i) {
            double *ri = Lp + i * n;
            for (int64_t j = c0; j < i; ++j) ri[j] = (ri[j] - dot(ri, Lp + j * n, j)) / Lp[j * n + j];
            const double piv = ri[i] - dot(ri, ri, i);
            if (!(piv > tiny) || !isfinite(piv)) { bad = i + 1; break; }
            ri[i] = sqrt(piv);
        }
        if (bad) break;
        const int pt = threads_for((double)(n - c1) * (double)(c1 - c0) * (double)c1);
        parallel_for(n - c1, pt, [=](int64_t b, int64_t e) {
            for (int64_t i = c1 + b; i < c1 + e; ++i) {
                double *ri = Lp + i * n;
                for (int64_t j = c0; j < c1; ++j) ri[j] = (ri[j] - dot(ri, Lp + j * n, j)) / Lp[j * n + j];
            }
        });
    }
    if (bad) {
        char buf[160];
        snprintf(buf, sizeof buf, "kernel matrix is not positive definite (pivot %lld of %lld <= 0)", (long long)(bad - 1), (long long)n);
        err = buf;
        return TGP_NOT_PD;
    }
    // alpha = L^-T (L^-1 yn) (_gpr.py:360-364)
    std::vector<double> z((size_t)n);
    for (int64_t i = 0; i < n; ++i) z[(size_t)i] = (yn[(size_t)i] - dot(Lp + i * n, z.data(), i)) / Lp[i * n + i];
    alpha = z;
    for (int64_t i = n - 1; i >= 0; --i) {
        const double a = alpha[(size_t)i] / Lp[i * n + i];
        alpha[(size_t)i] = a;
        const double *ri = Lp + i * n;
        for (int64_t j = 0; j < i; ++j) alpha[(size_t)j] -= ri[j] * a;
    }
    // _gpr.py:609-611: -0.5 y.alpha - sum(log(diag L)) - n/2 log(2 pi)
    double ya = 0.0;
    sumlog = 0.0;
    for (int64_t i = 0; i < n; ++i) { ya += yn[(size_t)i] * alpha[(size_t)i]; sumlog += log(Lp[i * n + i]); }
    lml = -0.5 * ya - sumlog - (double)n / 2.0 * log(2.0 * M_PI);
    if (lml_out) *lml_out = lml;
    if (ym_out) *ym_out = y_mean;
    if (ys_out) *ys_out = y_std;
    fitted = true;
    last_fit_ms = now_ms() - t0;
    return TGP_OK;
}

int HostGP::export_state(void *buf, int64_t cap, int64_t *size) {
    if (!fitted) { err = "tgp_export_state: no fitted model"; return TGP_NOT_FITTED; }
    const int64_t need = (8 + D + N * D + N) * 8;
    if (size) *size = need;
    if (!buf) {
        if (size) return TGP_OK;
        err = "tgp_export_state: buf and size both NULL";
        return TGP_BAD_ARG;
    }
    if (cap < need) { err = "tgp_export_state: buffer too small"; return TGP_BAD_ARG; }
    char *p = static_cast<char *>(buf);
    memcpy(p, STATE_MAGIC, 8); p += 8;
    const int64_t ints[4] = {N, D, (int64_t)kernel, (int64_t)normalize_y};
    memcpy(p, ints, sizeof ints); p += sizeof ints;
    const double reals[3] = {constant, noise, jitter};
    memcpy(p, reals, sizeof reals); p += sizeof reals;
    memcpy(p, ls.data(), (size_t)D * 8); p += D * 8;
    memcpy(p, X.data(), (size_t)(N * D) * 8); p += N * D * 8;
    memcpy(p, y.data(), (size_t)N * 8);
    return TGP_OK;
}

int HostGP::import_state(const void *buf, int64_t size, double *lml_out) {
    if (!buf || size < 64) { err = "tgp_import_state: blob too short"; return TGP_BAD_ARG; }
    const char *p = static_cast<const char *>(buf);
    if (memcmp(p, STATE_MAGIC, 8) != 0) { err = "tgp_import_state: bad magic"; return TGP_BAD_ARG; }
    int64_t ints[4];
    double reals[3];
    memcpy(ints, p + 8, sizeof ints);
    memcpy(reals, p + 40, sizeof reals);
    const int64_t n = ints[0], d = ints[1];
    if (n < 1 || d < 1 || n > 65536 || d > 4096) { err = "tgp_import_state: bad shape"; return TGP_BAD_ARG; }
    if (size != (8 + d + n * d + n) * 8) { err = "tgp_import_state: size does not match the header"; return TGP_BAD_ARG; }
    std::vector<double> l((size_t)d), Xb((size_t)(n * d)), yb((size_t)n);
    p += 64;
    memcpy(l.data(), p, (size_t)d * 8); p += d * 8;
    memcpy(Xb.data(), p, (size_t)(n * d) * 8); p += n * d * 8;
    memcpy(yb.data(), p, (size_t)n * 8);
    return fit(Xb.data(), n, d, yb.data(), (int)ints[2], reals[0], l.data(), d, reals[1], reals[2], (int)ints[3],
               lml_out, nullptr, nullptr);
}

int HostGP::debug_read(int which, double *out) {
    if (!out) { err = "tgp_debug_read: out is NULL"; return TGP_BAD_ARG; }
    if (!fitted) { err = "tgp_debug_read: no fitted model"; return TGP_NOT_FITTED; }
    if (which == TGP_BUF_ALPHA) { memcpy(out, alpha.data(), (size_t)N * 8); return TGP_OK; }
    if (which == TGP_BUF_L) { memcpy(out, L.data(), (size_t)(N * N) * 8); return TGP_OK; }
    err = "tgp_debug_read: the host backend keeps L and alpha only";
    return TGP_BAD_ARG;
}

int HostGP::set_candidates(const double *Xc, int64_t m) {
    if (!fitted) { err = "tgp_set_candidates: fit first (D is taken from the model)"; return TGP_NOT_FITTED; }
    if (!Xc || m < 1) { err = "tgp_set_candidates: need Xc and M >= 1"; return TGP_BAD_ARG; }
    if (m > HOST_MAX_CANDIDATE_DOUBLES / D) { err = "tgp_set_candidates: M * D beyond the host backend's limit of 2^34 values"; return TGP_BAD_ARG; }
    cand.assign(Xc, Xc + (size_t)(m * D));
    M = m;
    return TGP_OK;
}

int HostGP::read_candidates(int64_t first, int64_t count, double *out) {
    if (M < 1 || !out || first < 0 || count < 1 || first + count > M) { err = "tgp_read_candidates: bad range or no candidates"; return TGP_BAD_ARG; }
    memcpy(out, cand.data() + first * D, (size_t)(count * D) * 8);
    return TGP_OK;
}

int HostGP::sweep(int acq, double sf, double incumbent, double param, double *mu, double *sigma,
                  double *acq_out, double *best_val, int64_t *best_idx, int64_t *n_clamped) {
    if (!fitted) { err = "tgp_sweep: no fitted model"; return TGP_NOT_FITTED; }
    if (M < 1) { err = "tgp_sweep: no candidates set"; return TGP_BAD_ARG; }
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) { err = "tgp_sweep: unknown acquisition"; return TGP_BAD_ARG; }
    if (sf != 1.0 && sf != -1.0) { err = "tgp_sweep: sf must be +1 or -1"; return TGP_BAD_ARG; }
    const double t0 = now_ms();
    const int64_t n = N, d = D, m = M;
    const int64_t ntiles = (m + CB - 1) / CB;
    const int nt = threads_for((double)m * (double)n * (double)(n / 2 + d + 8));
    std::vector<double> vbest((size_t)ntiles, -INFINITY);
    std::vector<int64_t> ibest((size_t)ntiles, INT64_MAX), nclamp((size_t)ntiles, 0);
    const double *xs = Xs.data(), *Lp = L.data(), *al = alpha.data(), *cd = cand.data(), *lsp = ls.data();
    const double kss = constant + noise;   // kernel_.diag(X*): the white noise is in the predictive variance
    const double c = constant, ym = y_mean, ys = y_std;
    const int kern = kernel;
    double *vb = vbest.data();
    int64_t *ib = ibest.data(), *nc = nclamp.data();
    parallel_for(ntiles, nt, [=](int64_t tb, int64_t te) {
        std::vector<double> V((size_t)(n * CB)), cs((size_t)(CB * d));
        for (int64_t t = tb; t < te; ++t) {
            const int64_t j0 = t * CB, w = std::min<int64_t>(CB, m - j0);
            for (int64_t q = 0; q < CB; ++q)
                for (int64_t k = 0; k < d; ++k)
                    cs[(size_t)(q * d + k)] = q < w ? cd[(j0 + q) * d + k] / lsp[k] : 0.0;   // X / length_scale
            double mun[CB], qq[CB];
            for (int64_t q = 0; q < CB; ++q) { mun[q] = 0.0; qq[q] = 0.0; }
            // K* tile (the WhiteKernel's cross term is 0: kernels.py:1413-1414) and the mean K* alpha
            for (int64_t i = 0; i < n; ++i) {
                const double *xi = xs + i * d;
                double *vi = V.data() + i * CB;
                for (int64_t q = 0; q < CB; ++q) {
                    const double *cq = cs.data() + q * d;
                    double d2 = 0.0;
                    for (int64_t k = 0; k < d; ++k) { const double df = cq[k] - xi[k]; d2 += df * df; }
                    const double kv = kernel_value(kern, d2, c);
                    vi[q] = kv;
                    mun[q] += kv * al[i];
                }
            }
            // V = L^-1 K*^T in place, 16 right-hand sides at once (L is read once per tile)
            for (int64_t i = 0; i < n; ++i) {
                const double *ri = Lp + i * n;
                double s[CB];
                double *vi = V.data() + i * CB;
                for (int64_t q = 0; q < CB; ++q) s[q] = vi[q];
                for (int64_t k = 0; k < i; ++k) {
                    const double l = ri[k];
                    const double *vk = V.data() + k * CB;
                    for (int64_t q = 0; q < CB; ++q) s[q] -= l * vk[q];
                }
                const double dinv = ri[i];
                for (int64_t q = 0; q < CB; ++q) { const double v = s[q] / dinv; vi[q] = v; qq[q] += v * v; }
            }
            double best = -INFINITY;
            int64_t bi = INT64_MAX, clamped = 0;
            for (int64_t q = 0; q < w; ++q) {
                double var = kss - qq[q];
                if (var < 0.0) { var = 0.0; ++clamped; }   // _gpr.py:479-485
                const double muq = ys * mun[q] + ym;
                const double sg = sqrt(var * (ys * ys));
                double a = 0.0;
                if (acq == TGP_ACQ_UCB) {
                    a = sf * muq + param * sg;
                } else if (acq == TGP_ACQ_SIGMA) {
                    a = sg;
                } else if ((acq == TGP_ACQ_PI || acq == TGP_ACQ_EI) && sg != 0.0) {
                    const double diff = sf * (muq - incumbent) - param;
                    const double Z = diff / sg;
                    if (acq == TGP_ACQ_PI) a = ndtr(Z);
                    else a = diff * ndtr(Z) + sg * (exp(-(Z * Z) / 2.0) / 2.5066282746310002);
                }
                const int64_t g = j0 + q;
                if (mu) mu[g] = muq;
                if (sigma) sigma[g] = sg;
                if (acq_out) acq_out[g] = a;
                const double v2 = isnan(a) ? -INFINITY : a;   // NaN never wins
                if (v2 > best || (v2 == best && g < bi)) { best = v2; bi = g; }
            }
            vb[t] = best; ib[t] = bi; nc[t] = clamped;
        }
    });
    double best = -INFINITY;
    int64_t bi = INT64_MAX, clamped = 0;
    for (int64_t t = 0; t < ntiles; ++t) {
        if (vb[t] > best || (vb[t] == best && ib[t] < bi)) { best = vb[t]; bi = ib[t]; }
        clamped += nc[t];
    }
    if (acq != TGP_ACQ_NONE) {
        if (best_val) *best_val = best;
        if (best_idx) *best_idx = bi >= m ? 0 : bi;
    }
    if (n_clamped) *n_clamped = clamped;
    last_sweep_ms = now_ms() - t0;
    return TGP_OK;
}

// ---- NumPy's legacy global RNG, continued outside the interpreter (host_backend.hpp) --------------------------------
namespace {

// the next 624 words of the state (numpy/random/src/mt19937/mt19937.c mt19937_gen; Matsumoto & Nishimura's reference
// recurrence).  Both loops vectorise: a word depends on words at distance 1 (still old) and 227 / 397.
__attribute__((always_inline)) inline void mt_regen(uint32_t *mt) {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
    int k;
    for (k = 0; k < 227; ++k) {
        const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
        mt[k] = mt[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    for (; k < 623; ++k) {
        const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
        mt[k] = mt[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    const uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);
    mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
}

__attribute__((always_inline)) inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// n outputs of the stream into dst; (key, pos) advanced as n calls of mt19937_next would.
// (compiled three times -- baseline x86-64, AVX2, AVX-512 -- and chosen at load time: the recurrence and the tempering
// are 32-bit integer loops that widen with the vector registers; integer arithmetic, so the words are the same)
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("default", "avx2", "avx512f")))
#endif
void mt_fill(uint32_t *key, int32_t *pos, uint32_t *dst, int64_t n) {
    int32_t p = *pos;
    int64_t i = 0;
    while (i < n) {
        if (p >= 624) { mt_regen(key); p = 0; }
        const int64_t take = std::min<int64_t>(624 - p, n - i);
        for (int64_t j = 0; j < take; ++j) dst[i + j] = mt_temper(key[p + j]);
        i += take;
        p += (int32_t)take;
    }
    *pos = p;
}

}  // namespace

void mt19937_fill(uint32_t *key, int32_t *pos, uint32_t *dst, int64_t n) { mt_fill(key, pos, dst, n); }

void mt19937_skip(uint32_t *key, int32_t *pos, int64_t n) {
    int32_t p = *pos;
    while (n > 0) {
        if (p >= 624) { mt_regen(key); p = 0; }
        const int64_t take = std::min<int64_t>(624 - p, n);
        p += (int32_t)take;
        n -= take;
    }
    *pos = p;
}

int mt19937_uniform_columns(uint32_t *key, int32_t *pos, int64_t M, int64_t D, const double *lo, const double *hi,
                            double *out) {
    if (!key || !pos || !lo || !hi || !out || M < 1 || D < 1 || *pos < 0 || *pos > 624) return TGP_BAD_ARG;
    // groups of G columns: their outputs are generated one column after the other (the stream is sequential), then
    // the rows of the group are formed and written by several threads -- G doubles of a row at a time, so the
    // (M, D) row-major result is written in whole cache lines instead of one 8-byte store per line and column
    constexpr int64_t G = 16;
    const int64_t gcols = std::min(G, D);
    std::unique_ptr<uint32_t[]> w(new uint32_t[(size_t)(gcols * 2 * M)]);   // (not zero-filled: every word is written before it is read)
    const int nt = threads_for((double)M * (double)D * 4.0);
    for (int64_t c0 = 0; c0 < D; c0 += G) {
        const int64_t g = std::min(G, D - c0);
        for (int64_t c = 0; c < g; ++c) mt_fill(key, pos, w.get() + (size_t)(c * 2 * M), 2 * M);
        double low[G], range[G];
        for (int64_t c = 0; c < g; ++c) { low[c] = lo[c0 + c]; range[c] = hi[c0 + c] - lo[c0 + c]; }
        const uint32_t *wp = w.get();
        parallel_for(M, nt, [=](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                double *row = out + i * D + c0;
                for (int64_t c = 0; c < g; ++c) {
                    const uint32_t *wc = wp + (size_t)(c * 2 * M);
                    const uint32_t a = wc[2 * i] >> 5, bb = wc[2 * i + 1] >> 6;
                    // legacy random_sample, then random_uniform's  lower + range * u  (two roundings: this file is
                    // compiled with -ffp-contract=off)
                    const double u = ((double)a * 67108864.0 + (double)bb) / 9007199254740992.0;
                    row[c] = low[c] + range[c] * u;
                }
            }
        });
    }
    return TGP_OK;
}

}  // namespace tgp_host
