// tgp_api.hip -- the C-ABI of libturbogp.so (see include/turbogp.h for the reference call
// sites each entry replaces).  Host orchestration only: buffers, copies, launch order, status.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "host_backend.hpp"
#include "host_lbfgsb.hpp"
#include "tgp_internal.hpp"

using namespace tgp;

struct tgp_handle_s {
    Context c;
    tgp_host::HostGP *host = nullptr;   // tgp_create(TGP_DEVICE_HOST): this handle never touches HIP
    bool worker = false;                 // a pooled worker handle (tgp_workers_acquire): owned by the library, tgp_destroy refuses it
};

// ---- the device's pool of WORKER handles (round 5) -------------------------------------------------------------
// A worker is a handle on a private stream: what runs the concurrent starts of a hyper-parameter fit above the
// one-launch sizes, in C++ threads (tgp_fit_optimise) or in the host's threads (HipGPSurrogate drives tgp_fit_grad on
// them with SciPy).  Round 4 gave every factory / every caller handle three or four of its own and kept them alive:
// with two factories in a process that is the library's shared streams + 6-8 private ones on the runtime's hardware
// queues (4 by default, 8 asked for by turbo_amd/_lib.py), two streams share a queue, run one after the other, and a
// hyper-parameter fit took twice as long (N = 500: 11.6 -> 19.3 ms).  Now ONE pool per device serves every caller, one
// hyper-parameter fit at a time (the pool's mutex is held from acquire to release): at most MAX_WORKERS private
// streams per device whatever the number of factories.  The workers live until the last ordinary handle on the device
// is destroyed.
namespace {
constexpr int MAX_WORKERS = 4;
struct WorkerPool {
    std::mutex mu;                       // held from tgp_workers_acquire to tgp_workers_release
    std::atomic<std::thread::id> owner{std::thread::id()};   // ... by this thread (valid while held; read without the mutex by acquire / release)
    std::atomic<bool> held{false};
    std::vector<tgp_handle> workers;
    int users = 0;                       // ordinary (non-worker) GPU handles alive on the device
    std::mutex count_mu;                 // guards users / the workers' destruction against a concurrent create
};
WorkerPool g_pools[64];
}  // namespace

// ---- the library's START THREADS (round 6) ----------------------------------------------------------------------
// tgp_fit_lbfgsb runs the starts of a hyper-parameter fit side by side, a host thread each.  Round 5 created and joined
// those threads in every call: at the small sizes of the reference's everyday regime (N <= 128: a whole fit is 1-2 ms)
// creating three threads -- and the HIP runtime's per-thread set-up at each one's first call -- was a tenth of the call.
// Now up to MAX_WORKERS - 1 threads live for the process (detached, parked on a condition variable; the state is
// never destroyed, so nothing runs at exit), and a call hands them its shares: helper t runs fn(t), the caller fn(0).
// Only the holder of a device's worker pool dispatches (tgp_workers_acquire serialises the fits of a device), and the
// helpers are shared by all devices: a second device's fit waits at `busy` for the first one's shares to finish.
namespace {
struct StartThreads {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> fn;
    int pending = 0;              // shares handed out and not finished
    unsigned long long epoch = 0; // bumped per dispatch
    int want = 0;                 // shares 1 .. want of this epoch are the helpers'
    int taken = 0;
    int threads = 0;
    bool busy = false;
};
StartThreads &start_threads() {
    static StartThreads *st = new StartThreads();      // (leaked on purpose: parked detached threads must outlive every static)
    return *st;
}
void start_thread_main() {
    StartThreads &st = start_threads();
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(st.mu);
    for (;;) {
        st.cv_work.wait(lk, [&] { return st.epoch != seen && st.taken < st.want; });
        const int share = ++st.taken;
        if (st.taken == st.want) seen = st.epoch;      // (this epoch has no share left for this thread)
        std::function<void(int)> fn = st.fn;
        lk.unlock();
        fn(share);
        lk.lock();
        if (st.taken >= st.want) seen = st.epoch;
        if (--st.pending == 0) st.cv_done.notify_all();
    }
}
// fn(0) on the caller, fn(1) .. fn(T - 1) on the parked threads (created on demand); returns how many shares ran on
// helpers -- the caller runs the rest itself when threads could not be had
int run_on_start_threads(int T, const std::function<void(int)> &fn) {
    StartThreads &st = start_threads();
    int helpers = 0;
    {
        std::unique_lock<std::mutex> lk(st.mu);
        st.cv_done.wait(lk, [&] { return !st.busy; });
        while (st.threads < T - 1) {
            try {
                std::thread(start_thread_main).detach();
                ++st.threads;
            } catch (const std::system_error &) {
                break;
            }
        }
        helpers = std::min(T - 1, st.threads);
        if (helpers > 0) {
            st.busy = true;
            st.fn = fn;
            st.want = helpers; st.taken = 0; st.pending = helpers;
            ++st.epoch;
            st.cv_work.notify_all();
        }
    }
    fn(0);
    for (int t = helpers + 1; t < T; ++t) fn(t);
    if (helpers > 0) {
        std::unique_lock<std::mutex> lk(st.mu);
        st.cv_done.wait(lk, [&] { return st.pending == 0; });
        st.fn = nullptr;
        st.busy = false;
        st.cv_done.notify_all();
    }
    return helpers;
}
}  // namespace

// entries that only exist on the GPU
#define HOST_NA(name)                                                                                   \
    do {                                                                                                \
        if (h->host) {                                                                                  \
            h->host->err = name ": not available on the host backend (a TGP_DEVICE_HOST handle keeps a " \
                                "reloaded model queryable: fit, predict, acquisition; the rest needs the GPU)"; \
            return TGP_BAD_ARG;                                                                         \
        }                                                                                               \
    } while (0)

static thread_local std::string g_create_err;

namespace tgp {
int prof_mark(Context &c, hipStream_t s) {
    if (!c.profiling) return -1;
    if (c.ev_used == c.ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return -1; }
        c.ev_pool.push_back(e);
    }
    if (hipEventRecord(c.ev_pool[c.ev_used], s) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (int)c.ev_used++;
}
void prof_seg(Context &c, int a, int b, int kind, double flops) {
    if (a >= 0 && b >= 0) c.segs.push_back(ProfSeg{a, b, kind, flops});
}
}  // namespace tgp

static void prof_collect(Context &c) {
    for (const auto &g : c.segs) {
        float ms = 0.f;
        if (hipEventSynchronize(c.ev_pool[g.b]) == hipSuccess &&
            hipEventElapsedTime(&ms, c.ev_pool[g.a], c.ev_pool[g.b]) == hipSuccess) {
            if (g.kind == 0) { c.trmm_ms += ms; c.trmm_launches++; c.trmm_flops += g.flops; }
            else { c.kstar_ms += ms; c.kstar_launches++; }
        } else {
            (void)hipGetLastError();
        }
    }
    c.segs.clear();
    c.ev_used = 0;
}

static int fail(Context &c, int code, const std::string &msg) {
    c.err = msg;
    return code;
}
static int hip_fail(Context &c, hipError_t e, const char *where) {
    c.err = std::string(where) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return TGP_HIP_ERROR;
}

// The front of a sweep a fit started on the device's third stream (tgp_set_overlap): EVERY entry that uses the
// handle's buffers first makes its own stream wait for it -- one hipStreamWaitEvent, and only when something is
// pending -- so that nothing a later call writes, frees or re-uses is still being read or written over there.
// ... and the front is DISCARDED: only tgp_sweep, which looks at c.pre.front before it joins, can use one -- whatever
// else is called between the fit and the sweep may have changed the candidates (in place, same pointer and count),
// the factor or the workspace.
static hipError_t pre_join(Context &c) {
    c.pre.front = false;
    if (!c.pre.pending) return hipSuccess;
    c.pre.pending = false;
    return hipStreamWaitEvent(c.stream, c.pre.ev, 0);
}

#define API_HIP(call, where)                                    \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(c, e_, where);    \
    } while (0)

// ---- polled completion of the short calls (doorbell.hpp) ----
// the doorbell of the polled call about to be launched on this handle, or a null one (TGP_POLL_US=0)
static Bell bell_next(Context &c) {
    Bell b{nullptr, 0, nullptr};
    if (tuning().poll_us > 0 && c.d_bell) { b.word = c.d_bell; b.seq = ++c.bell_seq; b.ticket = c.d_ticket; }
    return b;
}
// wait for that call: spin on the mapped word, after TGP_POLL_US (or without a bell) synchronise the stream
static int bell_wait(Context &c, const Bell &b, const char *where) {
    if (b.word) {
        const volatile unsigned long long *w = c.h_bell;
        const auto t0 = std::chrono::steady_clock::now();
        const auto limit = std::chrono::microseconds(tuning().poll_us);
        for (unsigned spin = 1;; ++spin) {
            if (*w == b.seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return TGP_OK;
            }
            __builtin_ia32_pause();
            if ((spin & 255u) == 0 && std::chrono::steady_clock::now() - t0 > limit) break;
        }
    }
    API_HIP(hipStreamSynchronize(c.stream), where);
    return TGP_OK;
}
// device time of the polled call that just finished, from the kernels' own wall_clock64() stamps (100 MHz)
static double bell_ms(const Context &c) {
    const unsigned long long t0 = c.h_bell[1], t1 = c.h_bell[2];
    return t1 >= t0 ? (double)(t1 - t0) * 1e-5 : 0.0;
}

// No C++ exception may cross the C boundary: every entry is a function-try-block.
static int exception_status(tgp_handle h, const char *fn, const char *what, int code) {
    try {
        if (h) (h->host ? h->host->err : h->c.err) = std::string(fn) + ": " + what;
    } catch (...) {
    }
    return code;
}
#define TGP_CATCH                                                                               \
    catch (const std::bad_alloc &) { return exception_status(h, __func__, "out of host memory", TGP_NO_MEMORY); } \
    catch (const std::exception &ex_) { return exception_status(h, __func__, ex_.what(), TGP_HIP_ERROR); }       \
    catch (...) { return exception_status(h, __func__, "unknown C++ exception", TGP_HIP_ERROR); }

template <typename P>
static void dfree(P *&p) {
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

static void free_fit(Context &c) {
    dfree(c.d_Xs); dfree(c.d_ls); dfree(c.d_K); dfree(c.d_Linv); dfree(c.d_W); dfree(c.d_U); dfree(c.d_Dinv); dfree(c.d_Apan); dfree(c.d_apart);
    dfree(c.d_yn); dfree(c.d_z); dfree(c.d_alpha); dfree(c.d_Xs32); dfree(c.d_Linv32); dfree(c.d_Linv16); dfree(c.d_x2scal);
    dfree(c.d_t1); dfree(c.d_t2);
    dfree(c.d_gpart); dfree(c.d_gout); dfree(c.d_qws); dfree(c.d_rf); dfree(c.d_stamp);
    c.qws_cap = 0; c.cap_rf = 0;
    c.cap_Np = c.cap_D = 0;
    c.cap_full = false;
    c.g_cap_Np = c.g_cap_Dp = 0;
}
static void free_ws(Context &c) {
    dfree(c.d_Cs); dfree(c.d_Ks[0]); dfree(c.d_Ks[1]); dfree(c.d_part); dfree(c.d_mupart);
    dfree(c.d_topv); dfree(c.d_topi); c.cap_topv = c.cap_topi = 0;
    dfree(c.d_batch); c.cap_batch = 0;
    c.cap_Cs = c.cap_Ks[0] = c.cap_Ks[1] = c.cap_part = c.cap_mupart = 0;
    c.ws_Mpad = 0;
}

template <typename P>
static int grow(Context &c, P *&buf, size_t &cap, size_t need, const char *what) {
    if (need <= cap) return TGP_OK;
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    dfree(buf);
    cap = 0;
    API_HIP(hipMalloc((void **)&buf, need), what);
    cap = need;
    return TGP_OK;
}

// y normalisation (sklearn _gpr.py:272-282): mean, population std, exact-zero std -> 1
static void normalise_targets(const double *y, int64_t N, int normalize_y, std::vector<double> &yn,
                              double &mean, double &sd) {
    mean = 0.0; sd = 1.0;
    if (normalize_y) {
        long double s = 0.0L;
        for (int64_t i = 0; i < N; ++i) s += y[i];
        mean = (double)(s / (long double)N);
        long double v = 0.0L;
        for (int64_t i = 0; i < N; ++i) { const long double t = (long double)y[i] - mean; v += t * t; }
        sd = sqrt((double)(v / (long double)N));
        if (sd == 0.0) sd = 1.0;
        for (int64_t i = 0; i < N; ++i) yn[i] = (y[i] - mean) / sd;
    } else {
        for (int64_t i = 0; i < N; ++i) yn[i] = y[i];
    }
}

extern "C" {

const char *tgp_version(void) { return "turbogp 0.1 gfx950"; }

const char *tgp_last_error(tgp_handle h) { return h ? (h->host ? h->host->err.c_str() : h->c.err.c_str()) : g_create_err.c_str(); }

int tgp_create(int device, int dtype, tgp_handle *out) {
    if (!out) { g_create_err = "tgp_create: out is NULL"; return TGP_BAD_ARG; }
    *out = nullptr;
    if (dtype != TGP_F64 && dtype != TGP_F32 && dtype != TGP_F32X3 && dtype != TGP_F32H2) { g_create_err = "tgp_create: dtype must be TGP_F64, TGP_F32, TGP_F32X3 or TGP_F32H2"; return TGP_BAD_ARG; }
    if (device == TGP_DEVICE_HOST) {   // no HIP call on this path (always f64, whatever dtype says)
        tgp_handle hh = new (std::nothrow) tgp_handle_s();
        if (hh) hh->host = new (std::nothrow) tgp_host::HostGP();
        if (!hh || !hh->host) { delete hh; g_create_err = "tgp_create: out of host memory"; return TGP_NO_MEMORY; }
        *out = hh;
        return TGP_OK;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_err = std::string("tgp_create: no HIP device (") + hipGetErrorString(e) + ")";
        (void)hipGetLastError();
        return TGP_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_err = "tgp_create: device index out of range"; return TGP_BAD_ARG; }
    tgp_handle h = new (std::nothrow) tgp_handle_s();
    if (!h) { g_create_err = "tgp_create: out of host memory"; return TGP_HIP_ERROR; }
    Context &c = h->c;
    c.device = device;
    c.dtype = dtype;
    auto bail = [&](hipError_t er, const char *w) {
        g_create_err = std::string(w) + ": " + hipGetErrorString(er);
        (void)hipGetLastError();
        dfree(c.d_scal); dfree(c.d_flag); dfree(c.d_best); dfree(c.d_besti); dfree(c.d_ticket);
        if (c.h_bell) (void)hipHostFree(c.h_bell);
        if (c.ev0) (void)hipEventDestroy(c.ev0);
        if (c.ev1) (void)hipEventDestroy(c.ev1);
        if (c.pre.ev) (void)hipEventDestroy(c.pre.ev);
        if (c.pre.ev_in) (void)hipEventDestroy(c.pre.ev_in);
        const bool had = c.stream != nullptr;
        delete h;
        if (had) device_streams_release(device);
        return (int)TGP_HIP_ERROR;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = device_streams(device, &c.stream, nullptr)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&c.ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c.ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c.pre.ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c.pre.ev_in, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc((void **)&c.d_scal, 4 * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void **)&c.d_flag, sizeof(int))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void **)&c.d_best, sizeof(double))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc((void **)&c.d_besti, 4 * sizeof(long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c.d_besti, 0, 4 * sizeof(long long))) != hipSuccess) return bail(e, "hipMemset");   // [1] = clamp counter, kept at zero between calls
    // the doorbell of the short calls: one cache line of coherent device-mapped host memory + ticket counters
    if ((e = hipHostMalloc((void **)&c.h_bell, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e, "hipHostMalloc");
    memset(c.h_bell, 0, 64);
    if ((e = hipHostGetDevicePointer((void **)&c.d_bell, c.h_bell, 0)) != hipSuccess) return bail(e, "hipHostGetDevicePointer");
    if ((e = hipMalloc((void **)&c.d_ticket, 64)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c.d_ticket, 0, 64)) != hipSuccess) return bail(e, "hipMemset");
    {
        WorkerPool &wp = g_pools[device & 63];
        std::lock_guard<std::mutex> lk(wp.count_mu);
        ++wp.users;
    }
    *out = h;
    return TGP_OK;
}

static int destroy_handle(tgp_handle h);

int tgp_destroy(tgp_handle h) try {
    if (!h) return TGP_OK;
    if (h->host) { delete h->host; delete h; return TGP_OK; }
    if (h->worker) return fail(h->c, TGP_BAD_ARG, "tgp_destroy: a pooled worker handle belongs to the library (tgp_workers_acquire)");
    return destroy_handle(h);
} TGP_CATCH

static int destroy_handle(tgp_handle h) {
    Context &c = h->c;
    if (!h->worker) {   // the last ordinary handle on the device takes the worker pool with it
        WorkerPool &wp = g_pools[c.device & 63];
        std::vector<tgp_handle> doomed;
        {
            std::lock_guard<std::mutex> lk(wp.count_mu);
            if (--wp.users == 0) {
                std::lock_guard<std::mutex> lk2(wp.mu);
                doomed.swap(wp.workers);
            }
        }
        for (tgp_handle w : doomed) (void)destroy_handle(w);
    }
    (void)hipSetDevice(c.device);
    if (c.pre.pending && c.stream_pre) (void)hipStreamSynchronize(c.stream_pre);   // the front of a sweep nobody came for
    c.pre.pending = false;
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    prof_collect(c);
    free_fit(c);
    free_ws(c);
    if (c.h_pin_in) (void)hipHostFree(c.h_pin_in);
    if (c.h_pin_out) (void)hipHostFree(c.h_pin_out);
    if (c.h_pin_cand) (void)hipHostFree(c.h_pin_cand);
    c.h_pin_in = c.d_pin_in = c.h_pin_out = c.d_pin_out = c.h_pin_cand = c.d_pin_cand = nullptr;
    if (c.h_mt_words) (void)hipHostFree(c.h_mt_words);
    c.h_mt_words = nullptr; c.mt_words_cap = 0;
    for (int b = 0; b < 2; ++b)
        if (c.ev_mt[b]) { (void)hipEventDestroy(c.ev_mt[b]); c.ev_mt[b] = nullptr; }
    if (c.h_bell) (void)hipHostFree(c.h_bell);
    c.h_bell = c.d_bell = nullptr;
    dfree(c.d_ticket); dfree(c.d_sfg); c.cap_sfg = 0;
    dfree(c.d_cand_owned); dfree(c.d_mu); dfree(c.d_sigma); dfree(c.d_acq);
    dfree(c.d_bval); dfree(c.d_bidx); c.cap_bval = c.cap_bidx = 0; dfree(c.d_scal); dfree(c.d_flag); dfree(c.d_best); dfree(c.d_besti);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    if (c.pre.ev) (void)hipEventDestroy(c.pre.ev);
    if (c.pre.ev_in) (void)hipEventDestroy(c.pre.ev_in);
    if (c.ev_winner) (void)hipEventDestroy(c.ev_winner);
    for (int i = 0; i < 4; ++i)
        if (c.evg[i]) (void)hipEventDestroy(c.evg[i]);
    for (hipEvent_t e : c.ev_la) (void)hipEventDestroy(e);
    for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
    if (c.stream_own) { (void)hipStreamSynchronize(c.stream_own); (void)hipStreamDestroy(c.stream_own); c.stream_own = nullptr; }
    const int dev = c.device;
    delete h;
    device_streams_release(dev);
    return TGP_OK;
}

int tgp_set_private_stream(tgp_handle h, int on) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_set_private_stream");
    Context &c = h->c;
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    if (on && !c.stream_own) {
        API_HIP(hipStreamCreateWithFlags(&c.stream_own, hipStreamNonBlocking), "hipStreamCreate");
        c.stream = c.stream_own;
    } else if (!on && c.stream_own) {
        hipStream_t shared = nullptr;
        hipError_t e = device_streams(c.device, &shared, nullptr);   // (takes a reference ...)
        if (e != hipSuccess) return hip_fail(c, e, "device_streams");
        device_streams_release(c.device);                            // (... which this handle already holds)
        (void)hipStreamDestroy(c.stream_own);
        c.stream_own = nullptr;
        c.stream = shared;
    }
    return TGP_OK;
} TGP_CATCH

// n worker handles of h's device (each on a private stream), the pool locked for the caller until tgp_workers_release
int tgp_workers_acquire(tgp_handle h, int n, tgp_handle *out) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_workers_acquire");
    Context &c = h->c;
    if (!out || n < 1 || n > MAX_WORKERS) return fail(c, TGP_BAD_ARG, "tgp_workers_acquire: need out and 1 <= n <= 4");
    if (h->worker) return fail(c, TGP_BAD_ARG, "tgp_workers_acquire: a worker handle cannot borrow workers");
    WorkerPool &wp = g_pools[c.device & 63];
    if (wp.held.load(std::memory_order_acquire) && wp.owner.load(std::memory_order_acquire) == std::this_thread::get_id())
        return fail(c, TGP_BAD_ARG, "tgp_workers_acquire: this thread already holds the device's pool (release it first)");
    wp.mu.lock();
    while ((int)wp.workers.size() < n) {
        tgp_handle w = nullptr;
        int rc = tgp_create(c.device, TGP_F64, &w);
        if (rc == TGP_OK) {
            {   // (a worker is not a user of the device: the pool goes when the last ORDINARY handle does)
                std::lock_guard<std::mutex> lk(wp.count_mu);
                --wp.users;
            }
            w->worker = true;
            rc = tgp_set_private_stream(w, 1);
            if (rc != TGP_OK) (void)destroy_handle(w);
        }
        if (rc != TGP_OK) {
            wp.mu.unlock();
            return fail(c, rc, "tgp_workers_acquire: could not create a worker handle on a stream of its own");
        }
        wp.workers.push_back(w);
    }
    for (int i = 0; i < n; ++i) out[i] = wp.workers[(size_t)i];
    wp.owner.store(std::this_thread::get_id(), std::memory_order_release);
    wp.held.store(true, std::memory_order_release);
    return TGP_OK;
} TGP_CATCH

int tgp_workers_release(tgp_handle h) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_workers_release");
    WorkerPool &wp = g_pools[h->c.device & 63];
    // (only the thread that holds the pool reads `held` as true with its own id: the fields are written under the mutex)
    if (!wp.held.load(std::memory_order_acquire) || wp.owner.load(std::memory_order_acquire) != std::this_thread::get_id())
        return fail(h->c, TGP_BAD_ARG, "tgp_workers_release: the pool is not held by this thread");
    // a worker that grew to a large problem does not keep its four N^2 f64 buffers until the device's last handle goes
    // (2 GiB each at N = 8192, three workers): above 0.5 GiB they are given back here; the next large fit allocates
    // again (well under a millisecond against a fit of tens of milliseconds)
    for (tgp_handle w : wp.workers) {
        if (w->c.cap_Np > 4096) {
            (void)hipSetDevice(w->c.device);
            (void)hipStreamSynchronize(w->c.stream);
            w->c.fitted = false;
            free_fit(w->c);
            w->c.linv_ld = 0; w->c.linv_extent = 0;
        }
    }
    wp.owner.store(std::thread::id(), std::memory_order_release);
    wp.held.store(false, std::memory_order_release);
    wp.mu.unlock();
    return TGP_OK;
} TGP_CATCH

// The next fits start the sweep of the RESIDENT candidate batch inside themselves (include/turbogp.h).
int tgp_set_overlap(tgp_handle h, int mode) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_set_overlap");
    Context &c = h->c;
    if (mode < 0 || mode > 2) return fail(c, TGP_BAD_ARG, "tgp_set_overlap: mode must be 0, 1 or 2");
    c.pre.mode = mode;
    return TGP_OK;
} TGP_CATCH

int tgp_stream_status(tgp_handle h, int *out3) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_stream_status");
    Context &c = h->c;
    if (!out3) return fail(c, TGP_BAD_ARG, "tgp_stream_status: out3 is NULL");
    device_stream_status(c.device, &out3[0], &out3[1]);
    out3[2] = runtime_hw_queues_env();
    return TGP_OK;
} TGP_CATCH

// the reference's host candidate draw (NumPy's global MT19937 stream, continued in C++: host_backend.hpp)
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
        const std::string t = tuning().dump();
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

// pinned, device-mapped host memory for the small-problem path: the kernels read their inputs
// from it and write their scalars / small outputs to it, so a call needs no memcpy at all
static int ensure_pinned(Context &c, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > c.pin_in_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        if (c.h_pin_in) (void)hipHostFree(c.h_pin_in);
        c.h_pin_in = nullptr; c.d_pin_in = nullptr; c.pin_in_cap = 0;
        const size_t cap = std::max<size_t>(in_bytes, 1u << 20);
        API_HIP(hipHostMalloc((void **)&c.h_pin_in, cap, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
        API_HIP(hipHostGetDevicePointer((void **)&c.d_pin_in, c.h_pin_in, 0), "hipHostGetDevicePointer");
        c.pin_in_cap = cap;
    }
    if (out_bytes > c.pin_out_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        if (c.h_pin_out) (void)hipHostFree(c.h_pin_out);
        c.h_pin_out = nullptr; c.d_pin_out = nullptr; c.pin_out_cap = 0;
        const size_t cap = std::max<size_t>(out_bytes, 1u << 20);
        API_HIP(hipHostMalloc((void **)&c.h_pin_out, cap, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
        API_HIP(hipHostGetDevicePointer((void **)&c.d_pin_out, c.h_pin_out, 0), "hipHostGetDevicePointer");
        c.pin_out_cap = cap;
    }
    return TGP_OK;
}

static bool small_path_enabled() {
    return tuning().small != 0;   // A/B switch
}

static int fit_impl(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                    double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                    int normalize_y, double *lml, double *y_mean, double *y_std, bool allow_small,
                    int grad_mode = 0);   // grad_mode: 1 / 2 = the LML gradient (iso / ARD) is launched behind the fit, before the one synchronisation

static int ensure_workspace(Context &c);

// LML-gradient workspace of the blocked path (allocated on first use, grown with the fit)
static int ensure_grad_workspace(Context &c) {
    if (c.Np > c.g_cap_Np || c.Dp > c.g_cap_Dp) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        dfree(c.d_gpart); dfree(c.d_gout);
        const size_t nt = (size_t)(c.Np / 64);
        API_HIP(hipMalloc((void **)&c.d_gpart, nt * (nt + 1) / 2 * (size_t)(3 + c.Dp) * sizeof(double)), "hipMalloc gpart");
        API_HIP(hipMalloc((void **)&c.d_gout, (size_t)(3 + c.Dp) * sizeof(double)), "hipMalloc gout");
        c.g_cap_Np = c.Np; c.g_cap_Dp = c.Dp;
    }
    return TGP_OK;
}   // small_grad: 1 / 2 = also launch the one-workgroup LML gradient (iso / ARD) when the small path is taken

int tgp_fit(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
            double constant, const double *ls, int64_t n_ls, double noise, double jitter,
            int normalize_y, double *lml, double *y_mean, double *y_std) try {
    if (h && h->host) return h->host->fit(X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std);
    return fit_impl(h, X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std, true);
} TGP_CATCH

// the fit state's device buffers for Np padded rows and D dimensions.  full: everything a fit needs; otherwise only what
// a handle that RECEIVES a factor needs to sweep with it (tgp_import_factor_dev: no K, no inverse workspaces -- 4 N^2
// doubles less); a later fit on such a handle allocates the rest
static int ensure_fit_buffers(Context &c, int64_t Np, int64_t D, bool full) {
    if (Np <= c.cap_Np && D <= c.cap_D && (c.cap_full || !full)) return TGP_OK;
    const int64_t Dp = ((D + 3) / 4) * 4;
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    free_fit(c);
    const size_t nn = (size_t)Np * Np;
    API_HIP(hipMalloc((void **)&c.d_Xs, (size_t)Np * Dp * sizeof(double)), "hipMalloc Xs");
    API_HIP(hipMalloc((void **)&c.d_ls, (size_t)D * sizeof(double)), "hipMalloc ls");
    API_HIP(hipMalloc((void **)&c.d_Linv, nn * sizeof(double)), "hipMalloc Linv");
    if (full) {
        API_HIP(hipMalloc((void **)&c.d_K, nn * sizeof(double)), "hipMalloc K");
        API_HIP(hipMalloc((void **)&c.d_W, nn * sizeof(double)), "hipMalloc W");
        API_HIP(hipMalloc((void **)&c.d_U, nn * sizeof(double)), "hipMalloc U");
        API_HIP(hipMalloc((void **)&c.d_Dinv, (size_t)2 * (Np / NB) * NB * NB * sizeof(double)), "hipMalloc Dinv");
        API_HIP(hipMalloc((void **)&c.d_Apan, (size_t)2 * (Np / NB) * NB * NB * sizeof(double)), "hipMalloc Apan");
        API_HIP(hipMalloc((void **)&c.d_apart, ((size_t)(Np / 128) * Np + Np / 128) * sizeof(double)), "hipMalloc alpha shares");
        API_HIP(hipMalloc((void **)&c.d_t1, (size_t)Np * sizeof(double)), "hipMalloc t1");
        API_HIP(hipMalloc((void **)&c.d_t2, (size_t)Np * sizeof(double)), "hipMalloc t2");
    }
    API_HIP(hipMalloc((void **)&c.d_yn, (size_t)Np * sizeof(double)), "hipMalloc yn");
    API_HIP(hipMalloc((void **)&c.d_z, (size_t)Np * sizeof(double)), "hipMalloc z");
    API_HIP(hipMalloc((void **)&c.d_alpha, (size_t)Np * sizeof(double)), "hipMalloc alpha");
    if (c.dtype != TGP_F64) {
        API_HIP(hipMalloc((void **)&c.d_Xs32, (size_t)Np * Dp * sizeof(float)), "hipMalloc Xs32");
        API_HIP(hipMalloc((void **)&c.d_Linv32, nn * sizeof(float)), "hipMalloc Linv32");
    }
    if (c.dtype == TGP_F32X3 || c.dtype == TGP_F32H2) {
        API_HIP(hipMalloc((void **)&c.d_Linv16, (c.dtype == TGP_F32X3 ? 3 : 2) * nn * sizeof(unsigned short)), "hipMalloc Linv16");
        if (!c.d_x2scal) API_HIP(hipMalloc((void **)&c.d_x2scal, 2 * sizeof(unsigned)), "hipMalloc x2scal");
        c.linv16_gen = -1;
    }
    c.cap_Np = Np;
    c.cap_D = D;
    c.cap_full = full;
    c.linv_ld = 0;       // fresh allocation: contents unknown
    return TGP_OK;
}

static int fit_impl(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                    double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                    int normalize_y, double *lml, double *y_mean, double *y_std, bool allow_small,
                    int grad_mode) {
    if (!h) return TGP_BAD_ARG;
    Context &c = h->c;
    c.fitted = false;
    if (!X || !y || !ls) return fail(c, TGP_BAD_ARG, "tgp_fit: X, y and ls must not be NULL");
    if (N < 1 || D < 1) return fail(c, TGP_BAD_ARG, "tgp_fit: need N >= 1 and D >= 1");
    if (N > 65536 || D > 4096) return fail(c, TGP_BAD_ARG, "tgp_fit: N <= 65536 and D <= 4096 supported");
    if (n_ls != 1 && n_ls != D) return fail(c, TGP_BAD_ARG, "tgp_fit: n_ls must be 1 or D");
    if (kernel < TGP_RBF || kernel > TGP_MATERN52) return fail(c, TGP_BAD_ARG, "tgp_fit: unknown kernel");
    if (!(constant > 0.0) || !(noise >= 0.0) || !(jitter >= 0.0)) return fail(c, TGP_BAD_ARG, "tgp_fit: constant > 0, noise >= 0, jitter >= 0 required");
    for (int64_t d = 0; d < n_ls; ++d)
        if (!(ls[d] > 0.0)) return fail(c, TGP_BAD_ARG, "tgp_fit: length scales must be > 0");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    c.pre.front = false;   // whatever an earlier fit started early belongs to a factor this one replaces

    const int64_t Np = ((N + NPAD - 1) / NPAD) * NPAD;
    const int64_t Dp = ((D + 3) / 4) * 4;
    {
        const int arc = ensure_fit_buffers(c, Np, D, true);
        if (arc != TGP_OK) return arc;
    }
    if (D != c.D) { c.d_cand = nullptr; c.M = 0; c.d_winner = nullptr; }   // resident candidates / winner record belong to the old D
    c.N = N; c.D = D; c.Np = Np; c.Dp = Dp;
    c.imported = false; c.import_rows = 0;
    c.kernel = kernel; c.constant = constant; c.noise = noise; c.jitter = jitter;
    c.ls.assign((size_t)D, 0.0);
    for (int64_t d = 0; d < D; ++d) c.ls[d] = ls[n_ls == 1 ? 0 : d];

    double mean = 0.0, sd = 1.0;
    std::vector<double> yn((size_t)Np, 0.0);
    normalise_targets(y, N, normalize_y, yn, mean, sd);
    c.y_mean = mean; c.y_std = sd;

    // ---- small problems (N <= 128): one workgroup, one launch, no memcpy (small_kernels.hip) ----
    c.small = allow_small && N <= 2 * NB && small_path_enabled();
    if (c.small) {
        const int64_t Nin = ((N + NB - 1) / NB) * NB;
        int rc = ensure_pinned(c, (size_t)(Nin * Dp + Nin + D) * sizeof(double), (size_t)(8 + 3 * SMALL_GRAD_OUT_STRIDE) * sizeof(double));
        if (rc != TGP_OK) return rc;
        double *in = c.h_pin_in;
        memset(in, 0, (size_t)(Nin * Dp + Nin) * sizeof(double));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t d = 0; d < D; ++d) in[(size_t)i * Dp + d] = X[(size_t)i * D + d] / c.ls[d];   // X / length_scale
        memcpy(in + Nin * Dp, yn.data(), (size_t)N * sizeof(double));
        memcpy(in + Nin * Dp + Nin, c.ls.data(), (size_t)D * sizeof(double));
        // Round 6: ONE launch (the gradient's workgroups run the fit themselves) and a polled completion -- no event
        // record, no stream synchronisation; TGP_SMALL_FUSED=0 / TGP_POLL_US=0 keep round 5's calls (the A/B switches).
        const bool fused = grad_mode && tuning().small_fused != 0;
        if (fused && !c.d_sfg) {
            rc = grow(c, c.d_sfg, c.cap_sfg, small_fit_grad_ws_bytes(), "hipMalloc small fit + gradient workspace");
            if (rc != TGP_OK) return rc;
        }
        const Bell bell = (!grad_mode || fused) ? bell_next(c) : Bell{nullptr, 0, nullptr};
        if (!bell.word) API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
        hipError_t le = fused ? launch_small_fit_grad(c, grad_mode == 2, c.d_pin_out + 8, bell) : launch_small_fit(c, bell);
        c.linv_extent = std::max<int64_t>(c.linv_ld == Np ? c.linv_extent : Np, Nin);   // until the kernel is known to have finished
        c.linv_ld = Np;
        if (le != hipSuccess) return hip_fail(c, le, "launch_small_fit");
        if (grad_mode && !fused) {   // (timed together with the fit: two more event records would cost a third of the call)
            le = launch_small_grad(c, grad_mode == 2, c.d_pin_out + 8);
            if (le != hipSuccess) return hip_fail(c, le, "launch_small_grad");
        }
        if (bell.word) {
            rc = bell_wait(c, bell, "fit sync");
            if (rc != TGP_OK) return rc;
            c.last_fit_ms = bell_ms(c);
        } else {
            API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
            API_HIP(hipStreamSynchronize(c.stream), "fit sync");
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
            c.last_fit_ms = ms;
        }
        const double *res = c.h_pin_out;
        if (res[2] != 0.0) {
            char buf[160];
            snprintf(buf, sizeof buf, "kernel matrix is not positive definite (pivot %d of %lld <= 0)", (int)res[2] - 1, (long long)N);
            return fail(c, TGP_NOT_PD, buf);
        }
        c.linv_extent = Nin; c.linv_ld = Np;
        c.sumlog = res[0];
        c.lml = -0.5 * res[1] - res[0] - (double)N / 2.0 * log(2.0 * M_PI);
        c.normalize_y = normalize_y ? 1 : 0;
        c.h_X.assign(X, X + (size_t)N * D);
        c.h_y.assign(y, y + (size_t)N);
        if (lml) *lml = c.lml;
        if (y_mean) *y_mean = c.y_mean;
        if (y_std) *y_std = c.y_std;
        c.fitted = true; ++c.fit_gen; ++c.fit_gen;
        return TGP_OK;
    }

    // X / length_scale (kernels.py:1556 / 1711), padded rows zero.  Up to 64 MiB of inputs are
    // staged in device-mapped host memory and fetched by the fit's first kernel; the scalars come
    // back the same way: no memcpy, no memset in the call.
    const size_t n_in = (size_t)Np * Dp + (size_t)Np + (size_t)D;
    const bool staged = n_in * sizeof(double) <= ((size_t)64 << 20);
    std::vector<double> xs_heap;
    double *xs;
    if (staged) {
        int rc = ensure_pinned(c, n_in * sizeof(double), (size_t)(8 + 3 + Dp) * sizeof(double));
        if (rc != TGP_OK) return rc;
        xs = c.h_pin_in;
        if (D == Dp) memset(xs + (size_t)N * Dp, 0, (size_t)(Np - N) * Dp * sizeof(double));   // (the rows are about to be overwritten whole)
        else memset(xs, 0, (size_t)Np * Dp * sizeof(double));
        memcpy(xs + (size_t)Np * Dp, yn.data(), (size_t)Np * sizeof(double));
        memcpy(xs + (size_t)Np * Dp + Np, c.ls.data(), (size_t)D * sizeof(double));
    } else {
        xs_heap.assign((size_t)Np * Dp, 0.0);
        xs = xs_heap.data();
    }
    if (staged) {
        // raw rows: the fit's first kernel divides by the length scales (same IEEE division, off the host's critical path)
        if (D == Dp) memcpy(xs, X, (size_t)N * D * sizeof(double));
        else
            for (int64_t i = 0; i < N; ++i) memcpy(xs + (size_t)i * Dp, X + (size_t)i * D, (size_t)D * sizeof(double));
    } else {
        for (int64_t i = 0; i < N; ++i)
            for (int64_t d = 0; d < D; ++d) xs[(size_t)i * Dp + d] = X[(size_t)i * D + d] / c.ls[d];
    }

    const hipEvent_t e0 = c.ev0, e1 = c.ev1;
    // (round 6) a fit whose inputs and scalars travel through the mapped staging buffers is a POLLED call up to Np = 4096:
    // no event record, no stream synchronisation -- a one-wave kernel behind the chain rings the doorbell (launch_ring) and
    // the chain's first kernel leaves the start tick.  Not while the per-launch profiling of tgp_profile_enable is on
    // (bench.py's headline runs: they keep round 5's events), nor with TGP_POLL_US=0.
    const Bell bell = (staged && !c.profiling && Np <= 4096) ? bell_next(c) : Bell{nullptr, 0, nullptr};
    if (!bell.word) API_HIP(hipEventRecord(e0, c.stream), "hipEventRecord");
    if (!staged) {
        API_HIP(hipMemcpyAsync(c.d_Xs, xs, (size_t)Np * Dp * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D Xs");
        API_HIP(hipMemcpyAsync(c.d_ls, c.ls.data(), (size_t)D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D ls");
        API_HIP(hipMemcpyAsync(c.d_yn, yn.data(), (size_t)Np * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D yn");
    }
    if (grad_mode) {
        int rc = ensure_grad_workspace(c);
        if (rc != TGP_OK) return rc;
    }
    // Linv is zero above the diagonal whenever its leading dimension is known (nothing ever writes there) and
    // zero from row linv_extent on; this fit writes everything on and below the diagonal of its Nr rows and
    // skips the panels of pure padding: the zero fill is only needed when an older fit reached further down
    const int64_t Nr = ((N + NB - 1) / NB) * NB;
    const bool always_zero = tuning().linv_zero != 0;   // A/B
    const bool linv_clean = !always_zero && c.linv_ld == Np && c.linv_extent <= Nr;
    // tgp_set_overlap: the front of the resident batch's sweep goes out with this fit (the general sweep's f64 / f32
    // kernels on the shared streams only; the geometry it is issued for is recorded and checked again by tgp_sweep)
    c.pre.issue = 0; c.pre.front = false;
    {
        // (never for tgp_fit_grad: an evaluation of the hyper-parameter objective is followed by another evaluation,
        // not by a sweep)
        const int mode = grad_mode ? 0 : std::min(c.pre.mode, tuning().overlap);
        if (mode > 0 && c.d_cand && c.M > 0 && !c.stream_own && (c.dtype == TGP_F64 || c.dtype == TGP_F32) &&
            mid_sweep_cpw(c, c.M) == 0) {
            int rc = ensure_workspace(c);
            if (rc != TGP_OK) return rc;
            c.pre.issue = mode;
            c.pre.gen = c.fit_gen + 1;
            c.pre.cand = c.d_cand; c.pre.M = c.M; c.pre.Mpad = c.ws_Mpad; c.pre.launch_rows = c.launch_rows; c.pre.chunk = c.chunk;
        }
    }
    struct PrivateFit {                // (ended on every way out of this function, i.e. after the fit's synchronisation)
        Context &c; bool counted;
        ~PrivateFit() { if (counted) { private_fit_end(c.device, c.bg_lease != nullptr); c.bg_lease = nullptr; } }
    } private_fit{c, c.stream_own != nullptr};
    if (private_fit.counted) c.bg_lease = private_fit_begin(c.device, tuning().bg_lease != 0);
    hipError_t le = launch_fit(c, staged ? c.d_pin_in : nullptr, staged ? c.d_pin_out : nullptr, !linv_clean,
                               bell.word ? bell.word + 1 : nullptr);
    const bool pre_issued = c.pre.issue != 0;
    c.pre.issue = 0;
    c.linv_extent = Nr; c.linv_ld = Np;
    if (le != hipSuccess) {
        // a launch that failed half way may have left work on the third stream without having recorded its event
        // (c.pre.pending is set at launch_fit's end): wait for it here, so that no later call frees or rewrites what
        // it still reads
        if (pre_issued && !c.pre.pending && c.stream_pre) (void)hipStreamSynchronize(c.stream_pre);
        return hip_fail(c, le, "launch_fit");
    }

    int flag = 0;
    double scal[2] = {0.0, 0.0};
    if (!staged) {
        API_HIP(hipMemcpyAsync(&flag, c.d_flag, sizeof(int), hipMemcpyDeviceToHost, c.stream), "D2H flag");
        API_HIP(hipMemcpyAsync(scal, c.d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H scal");
    }
    if (!bell.word) API_HIP(hipEventRecord(e1, c.stream), "hipEventRecord");
    if (grad_mode) {   // behind the fit, in front of the call's one synchronisation
        c.grad_staged = staged;
        c.grad_timed = !bell.word;
        le = launch_lml_grad(c, grad_mode == 2, staged ? c.d_pin_out + 8 : c.d_gout, c.grad_timed);
        if (le != hipSuccess) return hip_fail(c, le, "launch_lml_grad");
    }
    if (bell.word) {
        le = launch_ring(c, bell);
        if (le != hipSuccess) return hip_fail(c, le, "launch_ring");
    }
    // the host's copies of the inputs (tgp_fit_append's prefix test, tgp_export_state) while the GPU works: c.fitted is
    // false until the fit has succeeded, so nobody reads them if it does not
    c.h_X.assign(X, X + (size_t)N * D);
    c.h_y.assign(y, y + (size_t)N);
    if (bell.word) {
        const int wrc = bell_wait(c, bell, "fit sync");
        if (wrc != TGP_OK) return wrc;
        c.last_fit_ms = bell_ms(c);      // (the whole call: the LML gradient of tgp_fit_grad included)
    } else {
        API_HIP(hipStreamSynchronize(c.stream), "fit sync");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        c.last_fit_ms = ms;
    }
    if (staged) {
        scal[0] = c.h_pin_out[0];
        scal[1] = c.h_pin_out[1];
        flag = (int)c.h_pin_out[2];
    }
    if (flag != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "kernel matrix is not positive definite (pivot %d of %lld <= 0)", flag - 1, (long long)N);
        return fail(c, TGP_NOT_PD, buf);
    }
    // _gpr.py:609-611: -0.5 y.alpha - sum(log(diag L)) - n/2 log(2 pi)
    c.lml = -0.5 * scal[1] - scal[0] - (double)N / 2.0 * log(2.0 * M_PI);
    c.sumlog = scal[0];
    c.normalize_y = normalize_y ? 1 : 0;
    if (lml) *lml = c.lml;
    if (y_mean) *y_mean = c.y_mean;
    if (y_std) *y_std = c.y_std;
    c.fitted = true; ++c.fit_gen;
    return TGP_OK;
}

int tgp_fit_append(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                   double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                   int normalize_y, double *lml, double *y_mean, double *y_std, int *appended) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) { if (appended) *appended = 0; return h->host->fit(X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std); }
    Context &c = h->c;
    if (appended) *appended = 0;
    bool ok = c.fitted && X && y && ls && D == c.D && N == c.N + 1 && N <= c.Np &&
              kernel == c.kernel && constant == c.constant && noise == c.noise && jitter == c.jitter &&
              (normalize_y ? 1 : 0) == c.normalize_y && (n_ls == 1 || n_ls == D) &&
              (int64_t)c.h_X.size() == c.N * c.D;
    if (ok)
        for (int64_t d = 0; d < D && ok; ++d) ok = (c.ls[d] == ls[n_ls == 1 ? 0 : d]);
    if (ok) ok = memcmp(c.h_X.data(), X, (size_t)c.N * D * sizeof(double)) == 0;
    if (!ok) return tgp_fit(h, X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std);

    c.fitted = false;
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int64_t n_old = c.N, Np = c.Np, Dp = c.Dp;
    double mean = 0.0, sd = 1.0;
    std::vector<double> yn((size_t)Np, 0.0);
    normalise_targets(y, N, normalize_y, yn, mean, sd);
    std::vector<double> xrow((size_t)Dp, 0.0);
    for (int64_t d = 0; d < D; ++d) xrow[d] = X[(size_t)n_old * D + d] / c.ls[d];

    const hipEvent_t e0 = c.ev0, e1 = c.ev1;
    API_HIP(hipEventRecord(e0, c.stream), "hipEventRecord");
    API_HIP(hipMemcpyAsync(c.d_Xs + n_old * Dp, xrow.data(), (size_t)Dp * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D x row");
    API_HIP(hipMemcpyAsync(c.d_yn, yn.data(), (size_t)Np * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D yn");
    c.linv_extent = std::max<int64_t>(c.linv_extent, ((N + NB - 1) / NB) * NB);   // the appended row of Linv
    hipError_t le = launch_fit_append(c, (int)n_old);
    if (le != hipSuccess) return hip_fail(c, le, "launch_fit_append");
    int flag = 0;
    double scal[4] = {0.0, 0.0, 0.0, 0.0};
    API_HIP(hipMemcpyAsync(&flag, c.d_flag, sizeof(int), hipMemcpyDeviceToHost, c.stream), "D2H flag");
    API_HIP(hipMemcpyAsync(scal, c.d_scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H scal");
    API_HIP(hipEventRecord(e1, c.stream), "hipEventRecord");
    API_HIP(hipStreamSynchronize(c.stream), "append sync");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    c.last_fit_ms = ms;
    if (flag != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "kernel matrix is not positive definite (pivot %d of %lld <= 0)", flag - 1, (long long)N);
        return fail(c, TGP_NOT_PD, buf);
    }
    c.N = N;
    if (N > 2 * NB) c.small = false;      // grown out of the small-problem kernels' range
    c.y_mean = mean; c.y_std = sd;
    c.sumlog += scal[3];
    c.lml = -0.5 * scal[1] - c.sumlog - (double)N / 2.0 * log(2.0 * M_PI);
    c.h_X.insert(c.h_X.end(), X + (size_t)n_old * D, X + (size_t)N * D);
    c.h_y.assign(y, y + (size_t)N);
    if (lml) *lml = c.lml;
    if (y_mean) *y_mean = c.y_mean;
    if (y_std) *y_std = c.y_std;
    if (appended) *appended = 1;
    c.fitted = true; ++c.fit_gen;
    return TGP_OK;
} TGP_CATCH

int tgp_fit_grad(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                 double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                 int normalize_y, double *lml, double *y_mean, double *y_std, double *grad) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_fit_grad");
    Context &c = h->c;
    if (!grad) return fail(c, TGP_BAD_ARG, "tgp_fit_grad: grad is NULL");
    const bool ard = n_ls > 1;
    // small problems: fit and gradient as two one-workgroup launches, one synchronisation, no memcpy
    const bool small = N <= 2 * NB && ((D + 3) / 4) * 4 <= 64 && small_path_enabled();
    // (otherwise the gradient needs U = Linv^T and the N^2 workspaces of the blocked path; there too it is
    // launched behind the fit, in front of the call's one synchronisation)
    int rc = fit_impl(h, X, N, D, y, kernel, constant, ls, n_ls, noise, jitter, normalize_y, lml, y_mean, y_std, small,
                      ard ? 2 : 1);
    if (rc != TGP_OK) return rc;
    if (small && c.small) {
        double out[3 + 64];
        {
            const double *sh = c.h_pin_out + 8;   // the shares of the block pairs, added in a fixed order
            const int nsh = N > NB ? 3 : 1, nout = ard ? 3 + (int)c.Dp : 3;
            for (int i = 0; i < nout; ++i) {
                double s = sh[i];
                for (int g = 1; g < nsh; ++g) s += sh[g * SMALL_GRAD_OUT_STRIDE + i];
                out[i] = s;
            }
        }
        c.last_grad_ms[0] = c.last_grad_ms[1] = c.last_grad_ms[2] = 0.0;   // (inside last_fit_ms)
        grad[0] = 0.5 * constant * out[0];
        if (ard) {
            for (int64_t d = 0; d < D; ++d) grad[1 + d] = constant * out[3 + (size_t)d];
        } else {
            grad[1] = 0.5 * constant * out[1];
        }
        grad[1 + n_ls] = 0.5 * noise * out[2];
        return TGP_OK;
    }
    std::vector<double> out((size_t)(3 + c.Dp), 0.0);
    if (c.grad_staged) {
        memcpy(out.data(), c.h_pin_out + 8, (size_t)(ard ? 3 + c.Dp : 3) * sizeof(double));
    } else {
        API_HIP(hipSetDevice(c.device), "hipSetDevice");
        API_HIP(hipMemcpy(out.data(), c.d_gout, (size_t)(ard ? 3 + c.Dp : 3) * sizeof(double), hipMemcpyDeviceToHost), "D2H grad");
    }
    for (int i = 0; i < 3; ++i) {
        float gms = 0.f;
        if (c.grad_timed) (void)hipEventElapsedTime(&gms, c.evg[i], c.evg[i + 1]);   // (a polled call records no stage events: its time is all in last_fit_ms)
        c.last_grad_ms[i] = gms;
    }
    // 0.5 * trace((alpha alpha^T - K^-1) dK/dtheta): _gpr.py:643-647
    grad[0] = 0.5 * constant * out[0];
    if (ard) {
        for (int64_t d = 0; d < D; ++d) grad[1 + d] = constant * out[3 + (size_t)d];
    } else {
        grad[1] = 0.5 * constant * out[1];
    }
    grad[1 + n_ls] = 0.5 * noise * out[2];
    return TGP_OK;
} TGP_CATCH

// State blob: what defines the fitted model, not the factor (the factor is N^2 and is rebuilt in
// milliseconds; X, y and theta are N*(D+1) + D + 3 doubles).  Layout, all little-endian 8-byte
// words: magic "TGPSTAT1", N, D, kernel, normalize_y (int64) | constant, noise, jitter (f64) |
// ls[D] | X[N*D] | y[N].
static const char STATE_MAGIC[8] = {'T', 'G', 'P', 'S', 'T', 'A', 'T', '1'};

int tgp_export_state(tgp_handle h, void *buf, int64_t cap, int64_t *size) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->export_state(buf, cap, size);
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_export_state: no fitted model");
    if (c.imported) return fail(c, TGP_BAD_ARG, "tgp_export_state: this handle holds a factor it received (tgp_import_factor_dev), not the training set it was computed from");
    const int64_t words = 8 + c.D + c.N * c.D + c.N;
    const int64_t need = words * 8;
    if (size) *size = need;
    if (!buf) return size ? TGP_OK : fail(c, TGP_BAD_ARG, "tgp_export_state: buf and size both NULL");
    if (cap < need) return fail(c, TGP_BAD_ARG, "tgp_export_state: buffer too small");
    char *p = static_cast<char *>(buf);
    memcpy(p, STATE_MAGIC, 8); p += 8;
    const int64_t ints[4] = {c.N, c.D, (int64_t)c.kernel, (int64_t)c.normalize_y};
    memcpy(p, ints, sizeof ints); p += sizeof ints;
    const double reals[3] = {c.constant, c.noise, c.jitter};
    memcpy(p, reals, sizeof reals); p += sizeof reals;
    memcpy(p, c.ls.data(), (size_t)c.D * 8); p += c.D * 8;
    memcpy(p, c.h_X.data(), (size_t)c.N * c.D * 8); p += c.N * c.D * 8;
    memcpy(p, c.h_y.data(), (size_t)c.N * 8);
    return TGP_OK;
} TGP_CATCH

int tgp_import_state(tgp_handle h, const void *buf, int64_t size, double *lml) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->import_state(buf, size, lml);
    Context &c = h->c;
    if (!buf || size < 64) return fail(c, TGP_BAD_ARG, "tgp_import_state: blob too short");   // (tgp_fit below joins the third stream)
    const char *p = static_cast<const char *>(buf);
    if (memcmp(p, STATE_MAGIC, 8) != 0) return fail(c, TGP_BAD_ARG, "tgp_import_state: bad magic");
    int64_t ints[4];
    double reals[3];
    memcpy(ints, p + 8, sizeof ints);
    memcpy(reals, p + 40, sizeof reals);
    const int64_t N = ints[0], D = ints[1];
    if (N < 1 || D < 1 || N > 65536 || D > 4096) return fail(c, TGP_BAD_ARG, "tgp_import_state: bad shape");
    if (size != (8 + D + N * D + N) * 8) return fail(c, TGP_BAD_ARG, "tgp_import_state: size does not match the header");
    // the blob carries no alignment promise: copy out before use
    std::vector<double> ls((size_t)D), X((size_t)N * D), y((size_t)N);
    p += 64;
    memcpy(ls.data(), p, (size_t)D * 8); p += D * 8;
    memcpy(X.data(), p, (size_t)N * D * 8); p += N * D * 8;
    memcpy(y.data(), p, (size_t)N * 8);
    return tgp_fit(h, X.data(), N, D, y.data(), (int)ints[2], reals[0], ls.data(), D, reals[1], reals[2],
                   (int)ints[3], lml, nullptr, nullptr);
} TGP_CATCH

// ---- handing a FACTOR over instead of recomputing it (round 6; SURVEY 8e's alternative: "fit on GPU0 + broadcast of L, alpha") ----
int tgp_export_factor_dev(tgp_handle h, tgp_factor *out) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_export_factor_dev");
    Context &c = h->c;
    if (!out) return fail(c, TGP_BAD_ARG, "tgp_export_factor_dev: out is NULL");
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_export_factor_dev: no fitted model");
    memset(out, 0, sizeof *out);
    out->N = c.N; out->D = c.D; out->Np = c.Np; out->Dp = c.Dp; out->fit_gen = c.fit_gen;
    out->kernel = c.kernel; out->normalize_y = c.normalize_y; out->small_path = c.small ? 1 : 0;
    out->constant = c.constant; out->noise = c.noise; out->jitter = c.jitter;
    out->y_mean = c.y_mean; out->y_std = c.y_std; out->lml = c.lml; out->sumlog = c.sumlog;
    out->Xs = c.d_Xs; out->ls = c.d_ls; out->alpha = c.d_alpha; out->Linv = c.d_Linv;
    return TGP_OK;
} TGP_CATCH

int tgp_import_factor_dev(tgp_handle h, const tgp_factor *f, int64_t row0, int64_t rows) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_import_factor_dev");
    Context &c = h->c;
    if (!f || !f->Xs || !f->ls || !f->alpha || !f->Linv) return fail(c, TGP_BAD_ARG, "tgp_import_factor_dev: need a factor with Xs, ls, alpha and Linv");
    const int64_t N = f->N, D = f->D, Np = f->Np, Dp = f->Dp;
    if (N < 1 || N > 65536 || D < 1 || D > 4096 || Np != ((N + NPAD - 1) / NPAD) * NPAD || Dp != ((D + 3) / 4) * 4)
        return fail(c, TGP_BAD_ARG, "tgp_import_factor_dev: inconsistent shape (Np = N rounded up to 256, Dp = D rounded up to 4)");
    if (f->kernel < TGP_RBF || f->kernel > TGP_MATERN52) return fail(c, TGP_BAD_ARG, "tgp_import_factor_dev: unknown kernel");
    if (row0 < 0 || rows < 1 || row0 + rows > Np) return fail(c, TGP_BAD_ARG, "tgp_import_factor_dev: need 0 <= row0, 1 <= rows, row0 + rows <= Np");
    // the rows arrive top down, as a Cholesky finishes them: block 0 starts a new factor, every later block continues it
    if (row0 != 0 && !(c.import_rows == row0 && !c.fitted && c.imported && c.N == N && c.D == D && c.fit_gen_src == f->fit_gen))
        return fail(c, TGP_BAD_ARG, "tgp_import_factor_dev: rows must arrive in order, starting with row0 = 0, all from one fit");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    c.pre.front = false;
    if (row0 == 0) {
        c.fitted = false;
        const int arc = ensure_fit_buffers(c, Np, D, false);
        if (arc != TGP_OK) return arc;
        if (D != c.D) { c.d_cand = nullptr; c.M = 0; c.d_winner = nullptr; }
        c.N = N; c.D = D; c.Np = Np; c.Dp = Dp;
        c.kernel = f->kernel; c.constant = f->constant; c.noise = f->noise; c.jitter = f->jitter;
        c.normalize_y = f->normalize_y; c.y_mean = f->y_mean; c.y_std = f->y_std; c.lml = f->lml; c.sumlog = f->sumlog;
        c.small = f->small_path != 0 && N <= 2 * NB && small_path_enabled();
        c.imported = true; c.import_rows = 0; c.fit_gen_src = f->fit_gen;
        c.h_X.clear(); c.h_y.clear();
        c.ls.assign((size_t)D, 0.0);      // (the host's copy of the length scales: fetched below)
        API_HIP(hipMemcpyAsync(c.d_Xs, f->Xs, (size_t)Np * Dp * sizeof(double), hipMemcpyDefault, c.stream), "D2D Xs");
        API_HIP(hipMemcpyAsync(c.d_ls, f->ls, (size_t)D * sizeof(double), hipMemcpyDefault, c.stream), "D2D ls");
        API_HIP(hipMemcpyAsync(c.d_alpha, f->alpha, (size_t)Np * sizeof(double), hipMemcpyDefault, c.stream), "D2D alpha");
        API_HIP(hipMemcpyAsync(c.ls.data(), f->ls, (size_t)D * sizeof(double), hipMemcpyDefault, c.stream), "D2H ls");
        if (c.dtype != TGP_F64) {
            hipError_t le = launch_f64_to_f32(c, c.d_Xs, c.d_Xs32, (long)(Np * Dp));
            if (le != hipSuccess) return hip_fail(c, le, "f64 -> f32 Xs");
        }
        c.linv_ld = Np; c.linv_extent = Np;     // (whatever the giver's padding rows hold is copied as it is)
    }
    API_HIP(hipMemcpyAsync(c.d_Linv + row0 * Np, static_cast<const double *>(f->Linv) + row0 * Np, (size_t)(rows * Np) * sizeof(double),
                           hipMemcpyDefault, c.stream), "D2D Linv rows");
    if (c.dtype != TGP_F64) {
        hipError_t le = launch_f64_to_f32(c, c.d_Linv + row0 * Np, c.d_Linv32 + row0 * Np, (long)(rows * Np));
        if (le != hipSuccess) return hip_fail(c, le, "f64 -> f32 Linv rows");
    }
    c.import_rows = row0 + rows;
    if (c.import_rows == Np) {
        API_HIP(hipStreamSynchronize(c.stream), "import sync");
        c.fitted = true; ++c.fit_gen;
    }
    return TGP_OK;
} TGP_CATCH

int tgp_debug_read(tgp_handle h, int which, double *out) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->debug_read(which, out);
    Context &c = h->c;
    if (!out) return fail(c, TGP_BAD_ARG, "tgp_debug_read: out is NULL");
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_debug_read: no fitted model");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int64_t N = c.N, Np = c.Np;
    if (which == TGP_BUF_ALPHA) {
        API_HIP(hipMemcpy(out, c.d_alpha, (size_t)N * sizeof(double), hipMemcpyDeviceToHost), "D2H alpha");
        return TGP_OK;
    }
    const double *src = which == TGP_BUF_LINV ? c.d_Linv : c.d_K;
    if (!src) return fail(c, TGP_BAD_ARG, "tgp_debug_read: this handle received its factor (tgp_import_factor_dev): it holds Linv and alpha, not L");
    if (which != TGP_BUF_K && which != TGP_BUF_L && which != TGP_BUF_LINV) return fail(c, TGP_BAD_ARG, "tgp_debug_read: unknown buffer");
    if (which == TGP_BUF_K) return fail(c, TGP_BAD_ARG, "tgp_debug_read: K is overwritten by L after the fit; read L");
    API_HIP(hipMemcpy2D(out, (size_t)N * sizeof(double), src, (size_t)Np * sizeof(double), (size_t)N * sizeof(double), (size_t)N, hipMemcpyDeviceToHost), "D2H matrix");
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = i + 1; j < N; ++j) out[i * N + j] = 0.0;
    return TGP_OK;
} TGP_CATCH

static int ensure_outputs(Context &c, bool mu, bool sg, bool aq) {
    if (c.M > c.out_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        dfree(c.d_mu); dfree(c.d_sigma); dfree(c.d_acq);
        c.out_cap = c.M;
    }
    // every output buffer is sized for out_cap, whichever call creates it
    const size_t bytes = (size_t)std::max<int64_t>(c.out_cap, 1) * sizeof(double);
    if (mu && !c.d_mu) API_HIP(hipMalloc((void **)&c.d_mu, bytes), "hipMalloc mu");
    if (sg && !c.d_sigma) API_HIP(hipMalloc((void **)&c.d_sigma, bytes), "hipMalloc sigma");
    if (aq && !c.d_acq) API_HIP(hipMalloc((void **)&c.d_acq, bytes), "hipMalloc acq");
    return TGP_OK;
}

// The owned candidate buffer, grown to `need` doubles.  A failing hipMalloc must not leave a stale
// capacity or a dangling resident batch behind (the next call would copy / launch through a freed
// pointer): the old batch is forgotten BEFORE the buffer is released and the capacity is only
// restored once the new allocation exists.
static int grow_candidates(Context &c, int64_t need) {
    if (need <= c.cand_cap) return TGP_OK;
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    if (c.d_cand == c.d_cand_owned) { c.d_cand = nullptr; c.M = 0; }
    dfree(c.d_cand_owned);
    c.cand_cap = 0;
    API_HIP(hipMalloc((void **)&c.d_cand_owned, (size_t)need * sizeof(double)), "hipMalloc candidates");
    c.cand_cap = need;
    return TGP_OK;
}

int tgp_set_candidates(tgp_handle h, const double *Xc, int64_t M) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->set_candidates(Xc, M);
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_set_candidates: fit first (D is taken from the model)");
    if (!Xc || M < 1) return fail(c, TGP_BAD_ARG, "tgp_set_candidates: need Xc and M >= 1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int64_t need = M * c.D;
    { const int grc = grow_candidates(c, need); if (grc != TGP_OK) return grc; }
    API_HIP(hipMemcpyAsync(c.d_cand_owned, Xc, (size_t)need * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D candidates");
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    c.d_cand = c.d_cand_owned;
    c.M = M;
    return TGP_OK;
} TGP_CATCH

// The reference-faithful HOST draw with only its sequential part on the host: NumPy's MT19937 stream is continued here
// (host_backend.cpp: the recurrence cannot be run in parallel), a column's 2 M words at a time into one of two pinned
// buffers, each copied to the device while the next is generated; the GPU forms the doubles and the (M, D) layout.
// C3's 262 144 x 32: NumPy 45-66 ms on the host, the library's all-host version 20-25 (page faults of a fresh 67 MB
// array included), this ~6.
int tgp_set_candidates_mt19937(tgp_handle h, uint32_t *key624, int32_t *pos, int64_t M_total, int64_t first_row, int64_t M,
                               const double *lo, const double *hi) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_set_candidates_mt19937");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_set_candidates_mt19937: fit first (D is taken from the model)");
    if (!key624 || !pos || !lo || !hi || M < 1 || first_row < 0 || first_row + M > M_total || *pos < 0 || *pos > 624)
        return fail(c, TGP_BAD_ARG, "tgp_set_candidates_mt19937: need key624, 0 <= pos <= 624, lo, hi and 1 <= rows, first_row + rows <= M_total");
    const int64_t D = c.D;
    std::vector<double> rng((size_t)(2 * D));
    for (int64_t d = 0; d < D; ++d) {
        rng[(size_t)d] = lo[d];
        rng[(size_t)(D + d)] = hi[d] - lo[d];          // NumPy's fscale = high - low
        if (!isfinite(rng[(size_t)(D + d)])) return fail(c, TGP_BAD_ARG, "tgp_set_candidates_mt19937: a range is not finite (NumPy raises OverflowError there)");
    }
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    // owned buffer: candidates (M D doubles) | lo, range (2 D) | the stream's words (2 M D of 4 bytes = M D doubles)
    const int64_t need = 2 * M * D + 2 * D;
    { const int grc = grow_candidates(c, need); if (grc != TGP_OK) return grc; }
    const size_t col_bytes = (size_t)(2 * M) * sizeof(uint32_t);
    if (2 * col_bytes > c.mt_words_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        if (c.h_mt_words) (void)hipHostFree(c.h_mt_words);
        c.h_mt_words = nullptr; c.mt_words_cap = 0;
        API_HIP(hipHostMalloc((void **)&c.h_mt_words, 2 * col_bytes, hipHostMallocDefault), "hipHostMalloc");
        c.mt_words_cap = 2 * col_bytes;
    }
    for (int b = 0; b < 2; ++b)
        if (!c.ev_mt[b]) API_HIP(hipEventCreateWithFlags(&c.ev_mt[b], hipEventDisableTiming), "hipEventCreate");
    double *d_lo = c.d_cand_owned + M * D, *d_range = d_lo + D;
    uint32_t *d_words = reinterpret_cast<uint32_t *>(d_range + D);
    if (c.d_cand == c.d_cand_owned) { c.d_cand = nullptr; c.M = 0; }     // (the resident batch is being overwritten)
    // the caller's RNG state moves only if the whole call succeeds
    uint32_t key[624];
    memcpy(key, key624, sizeof key);
    int32_t p = *pos;
    API_HIP(hipMemcpyAsync(d_lo, rng.data(), (size_t)(2 * D) * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D lo, range");
    bool used[2] = {false, false};
    for (int64_t col = 0; col < D; ++col) {
        const int b = (int)(col & 1);
        uint32_t *buf = c.h_mt_words + (size_t)b * (size_t)(2 * M);
        if (used[b]) API_HIP(hipEventSynchronize(c.ev_mt[b]), "hipEventSynchronize");   // its last copy has left
        // a column consumes 2 M_total words of the stream; this handle's rows are [first_row, first_row + M) of it
        tgp_host::mt19937_skip(key, &p, 2 * first_row);
        tgp_host::mt19937_fill(key, &p, buf, 2 * M);
        tgp_host::mt19937_skip(key, &p, 2 * (M_total - first_row - M));
        API_HIP(hipMemcpyAsync(d_words + (size_t)col * (size_t)(2 * M), buf, col_bytes, hipMemcpyHostToDevice, c.stream), "H2D words");
        API_HIP(hipEventRecord(c.ev_mt[b], c.stream), "hipEventRecord");
        used[b] = true;
    }
    hipError_t le = launch_mt19937_columns(c, d_words, c.d_cand_owned, M, d_lo, d_range);
    if (le != hipSuccess) { (void)hipStreamSynchronize(c.stream); return hip_fail(c, le, "launch_mt19937_columns"); }
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    memcpy(key624, key, sizeof key);
    *pos = p;
    c.d_cand = c.d_cand_owned;
    c.M = M;
    return TGP_OK;
} TGP_CATCH

int tgp_gen_candidates(tgp_handle h, uint64_t seed, uint64_t first_candidate, int64_t M,
                       const double *lo, const double *hi) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_gen_candidates");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_gen_candidates: fit first (D is taken from the model)");
    if (!lo || !hi || M < 1) return fail(c, TGP_BAD_ARG, "tgp_gen_candidates: need lo, hi and M >= 1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int64_t need = M * c.D + 2 * c.D;      // candidates + the bounds behind them
    { const int grc = grow_candidates(c, need); if (grc != TGP_OK) return grc; }
    double *d_lo = c.d_cand_owned + M * c.D, *d_hi = d_lo + c.D;
    API_HIP(hipMemcpyAsync(d_lo, lo, (size_t)c.D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D lo");
    API_HIP(hipMemcpyAsync(d_hi, hi, (size_t)c.D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D hi");
    hipError_t le = launch_gen_candidates(c, c.d_cand_owned, M, seed, first_candidate, d_lo, d_hi);
    if (le != hipSuccess) return hip_fail(c, le, "launch_gen_candidates");
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    c.d_cand = c.d_cand_owned;
    c.M = M;
    return TGP_OK;
} TGP_CATCH

// A kernel reading or writing through a bad pointer faults the GPU: check a borrowed device
// pointer against what the HIP runtime knows (device memory, this GPU, `bytes` left in its
// allocation, 8-byte aligned) before any kernel sees it.
static int check_borrowed(Context &c, const void *p, size_t bytes, const char *who) {
    const std::string w(who);
    hipPointerAttribute_t at;
    hipError_t pe = hipPointerGetAttributes(&at, p);
    if (pe != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, TGP_BAD_ARG, w + ": not a pointer known to the HIP runtime");
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)
        return fail(c, TGP_BAD_ARG, w + ": pointer is not device memory");
    if (at.type == hipMemoryTypeDevice && at.device != c.device)
        return fail(c, TGP_BAD_ARG, w + ": pointer lives on another GPU than this handle");
    hipDeviceptr_t base = nullptr;
    size_t span = 0;
    if (hipMemGetAddressRange(&base, &span, const_cast<void *>(p)) == hipSuccess) {
        const size_t off = (size_t)((const char *)p - (const char *)base);
        if (bytes > span - off) return fail(c, TGP_BAD_ARG, w + ": allocation is smaller than the bytes this call needs");
    } else {
        (void)hipGetLastError();
    }
    if (((uintptr_t)p & 7u) != 0) return fail(c, TGP_BAD_ARG, w + ": pointer must be 8-byte aligned");
    return TGP_OK;
}

// shared by the two LHS entries: D columns, bounds behind the batch in the candidate buffer
static int gen_lhs_into_owned(Context &c, uint64_t seed, uint64_t first, int64_t M, uint64_t n_total,
                              int64_t D, const double *lo, const double *hi, const char *who) {
    const std::string w(who);
    if (!lo || !hi || M < 1) return fail(c, TGP_BAD_ARG, w + ": need lo, hi and M >= 1");
    if (n_total < 1 || n_total > (1ull << 40) || first + (uint64_t)M > n_total)
        return fail(c, TGP_BAD_ARG, w + ": need first_sample + M <= n_total <= 2^40 (LHS sequence exhausted)");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");   // (after the argument checks, as in every other entry)
    const int64_t need = M * D + 2 * D;
    { const int grc = grow_candidates(c, need); if (grc != TGP_OK) return grc; }
    double *d_lo = c.d_cand_owned + M * D, *d_hi = d_lo + D;
    API_HIP(hipMemcpyAsync(d_lo, lo, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D lo");
    API_HIP(hipMemcpyAsync(d_hi, hi, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D hi");
    hipError_t le = launch_gen_lhs(c, c.d_cand_owned, M, D, seed, first, n_total, d_lo, d_hi);
    if (le != hipSuccess) return hip_fail(c, le, "launch_gen_lhs");
    return TGP_OK;
}

int tgp_gen_candidates_lhs(tgp_handle h, uint64_t seed, uint64_t first_sample, int64_t M,
                           uint64_t n_total, const double *lo, const double *hi) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_gen_candidates_lhs");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_gen_candidates_lhs: fit first (D is taken from the model)");
    int rc = gen_lhs_into_owned(c, seed, first_sample, M, n_total, c.D, lo, hi, "tgp_gen_candidates_lhs");
    if (rc != TGP_OK) return rc;
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    c.d_cand = c.d_cand_owned;
    c.M = M;
    return TGP_OK;
} TGP_CATCH

int tgp_lhs_design(tgp_handle h, uint64_t seed, uint64_t first_sample, int64_t M, uint64_t n_total,
                   int64_t D, const double *lo, const double *hi, double *out) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_lhs_design");
    Context &c = h->c;
    if (!out || D < 1 || D > 4096) return fail(c, TGP_BAD_ARG, "tgp_lhs_design: need out and 1 <= D <= 4096");
    int rc = gen_lhs_into_owned(c, seed, first_sample, M, n_total, D, lo, hi, "tgp_lhs_design");
    if (rc != TGP_OK) return rc;
    API_HIP(hipMemcpyAsync(out, c.d_cand_owned, (size_t)(M * D) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H design");
    API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
    if (c.d_cand == c.d_cand_owned) { c.d_cand = nullptr; c.M = 0; }   // the owned batch was overwritten
    return TGP_OK;
} TGP_CATCH

int tgp_read_candidates(tgp_handle h, int64_t first, int64_t count, double *out) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->read_candidates(first, count, out);
    Context &c = h->c;
    if (!c.d_cand || !out || first < 0 || count < 1 || first + count > c.M)
        return fail(c, TGP_BAD_ARG, "tgp_read_candidates: bad range or no candidates");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    API_HIP(hipMemcpy(out, c.d_cand + first * c.D, (size_t)(count * c.D) * sizeof(double), hipMemcpyDeviceToHost), "D2H candidates");
    return TGP_OK;
} TGP_CATCH

int tgp_set_candidates_dev(tgp_handle h, const void *Xc_dev, int64_t M) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_set_candidates_dev");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_set_candidates_dev: fit first");
    if (!Xc_dev || M < 1) return fail(c, TGP_BAD_ARG, "tgp_set_candidates_dev: need a device pointer and M >= 1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int rc = check_borrowed(c, Xc_dev, (size_t)M * (size_t)c.D * sizeof(double), "tgp_set_candidates_dev");
    if (rc != TGP_OK) return rc;
    c.d_cand = reinterpret_cast<const double *>(Xc_dev);
    c.M = M;
    return TGP_OK;
} TGP_CATCH

int tgp_set_winner_out(tgp_handle h, void *rec_dev, int64_t global_offset) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_set_winner_out");
    Context &c = h->c;
    if (!rec_dev) { c.d_winner = nullptr; c.winner_offset = 0; return TGP_OK; }
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_set_winner_out: fit first (D is taken from the model)");
    if (global_offset < 0 || global_offset > ((int64_t)1 << 52)) return fail(c, TGP_BAD_ARG, "tgp_set_winner_out: global_offset must be in [0, 2^52]");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    const int rc = check_borrowed(c, rec_dev, (size_t)(c.D + 2) * sizeof(double), "tgp_set_winner_out");
    if (rc != TGP_OK) return rc;
    c.d_winner = reinterpret_cast<double *>(rec_dev);
    c.winner_offset = global_offset;
    return TGP_OK;
} TGP_CATCH

int tgp_winner_wait(tgp_handle h, void *stream) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_winner_wait");
    Context &c = h->c;
    if (!c.d_winner) return fail(c, TGP_BAD_ARG, "tgp_winner_wait: no winner record attached (tgp_set_winner_out)");
    if (!c.winner_recorded) return TGP_OK;      // no sweep has packed a record yet: nothing to wait for
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c.ev_winner, 0), "hipStreamWaitEvent");
    return TGP_OK;
} TGP_CATCH

int tgp_get_candidate(tgp_handle h, int64_t idx, double *out_row) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->read_candidates(idx, 1, out_row);
    Context &c = h->c;
    if (!c.d_cand || !out_row || idx < 0 || idx >= c.M) return fail(c, TGP_BAD_ARG, "tgp_get_candidate: bad index or no candidates");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(hipMemcpy(out_row, c.d_cand + idx * c.D, (size_t)c.D * sizeof(double), hipMemcpyDeviceToHost), "D2H candidate");
    return TGP_OK;
} TGP_CATCH

// Sweep workspace: grow-only, so a loop that alternates batch sizes (plots, 1-point calls, the
// big sweep) does not re-allocate.  Leading dimensions are per call.
static int ensure_workspace(Context &c) {
    c.pre.front = false;   // whoever asks for the workspace is about to overwrite what a fit's early front left in it (tgp_sweep looks first)
    const size_t elt = c.dtype != TGP_F64 ? 4 : 8;
    const size_t kelt = c.dtype == TGP_F32X3 ? 6 : elt;   // bytes per element of the cross-kernel slab (three bf16 planes; two fp16 planes = 4)
    // chunk: a cross-kernel slab of about 256 MiB per launch, multiple of 1024.  Measured on the
    // four BASELINE configs (TGP_CHUNK sweeps, profiles/README.md): 128 MiB costs 1-5 % (twice
    // the launches, half the tiles per launch to balance), 512 MiB and more lose L2 locality.
    int64_t chunk = (int64_t)((256ull << 20) / ((size_t)c.Np * elt));   // (same candidates per launch for the three-plane slab: 384 MiB)
    chunk = std::max<int64_t>(1024, (chunk / 1024) * 1024);
    chunk = std::min<int64_t>(chunk, 65536);
    if (const long v = tuning_chunk_now(); v >= 1024) chunk = (v / 1024) * 1024;   // tuning knob (multiple of 1024), read at every call
    const int64_t mpad = ((c.M + 255) / 256) * 256;   // a multiple of every candidate-tile width in use
    if (mpad <= chunk) chunk = mpad;   // single group
    int rc;
    if ((rc = grow(c, c.d_Cs, c.cap_Cs, (size_t)mpad * c.Dp * elt, "hipMalloc Cs")) != TGP_OK) return rc;
    // The cross-kernel slab.  f64 / f32: one launch pair (cross-kernel, contraction) covers several GROUPS of
    // `chunk` candidates; inside the contraction the tiles walk the slab group by group (sweep_tile()), so the
    // light tail of one group is filled by the heavy head of the next and the cross-kernel runs in fewer,
    // larger launches.  Measured on one box (C3, ms per step): 1 group per launch 36.4, 2: 35.1, 4: 35.0,
    // 8: 35.1, all 16: 35.4 (the slab then always comes back from HBM, never from the Infinity Cache);
    // C4 and C2 within 0.5 % whatever the grouping.  Default: TGP_SLAB_GB = 1 (four groups of 256 MiB).
    // The split-operand dtypes keep a launch per group (two slots).
    const bool per_group = c.dtype == TGP_F32X3 || c.dtype == TGP_F32H2;
    int64_t launch_rows = chunk;
    if (!per_group) {
        const double slab_gb = tuning().slab_gb;
        const int64_t cap = (int64_t)(slab_gb * 1073741824.0 / (double)((size_t)c.Np * kelt));
        const int64_t groups = std::max<int64_t>(1, std::min<int64_t>((mpad + chunk - 1) / chunk, cap / chunk));
        launch_rows = groups * chunk;
    }
    if ((rc = grow(c, c.d_Ks[0], c.cap_Ks[0], (size_t)std::min<int64_t>(launch_rows, std::max<int64_t>(mpad, chunk)) * c.Np * kelt, "hipMalloc Ks")) != TGP_OK) return rc;
    if (per_group && (rc = grow(c, c.d_Ks[1], c.cap_Ks[1], (size_t)chunk * c.Np * kelt, "hipMalloc Ks")) != TGP_OK) return rc;
    c.launch_rows = launch_rows;
    if ((rc = grow(c, c.d_part, c.cap_part, (size_t)(c.Np / SW_BM) * mpad * sizeof(double), "hipMalloc part")) != TGP_OK) return rc;
    if ((rc = grow(c, c.d_mupart, c.cap_mupart, (size_t)KS_JS * mpad * sizeof(double), "hipMalloc mupart")) != TGP_OK) return rc;
    c.chunk = chunk;
    c.ws_Mpad = mpad;
    const size_t nblk = (size_t)((c.M + NB - 1) / NB + 1);   // the small-problem sweep reduces 64 candidates per block
    if ((rc = grow(c, c.d_bval, c.cap_bval, nblk * sizeof(double), "hipMalloc bval")) != TGP_OK) return rc;
    if ((rc = grow(c, c.d_bidx, c.cap_bidx, nblk * sizeof(long long), "hipMalloc bidx")) != TGP_OK) return rc;
    return TGP_OK;
}

// the small-problem sweep only needs the per-block arg-max partials
static int ensure_small_workspace(Context &c) {
    const size_t nblk = (size_t)((c.M + 31) / 32 + 1);   // (the one-launch sweep of N <= 512 reduces 32 candidates per workgroup)
    int rc;
    if ((rc = grow(c, c.d_bval, c.cap_bval, nblk * sizeof(double), "hipMalloc bval")) != TGP_OK) return rc;
    if ((rc = grow(c, c.d_bidx, c.cap_bidx, nblk * sizeof(long long), "hipMalloc bidx")) != TGP_OK) return rc;
    return TGP_OK;
}

int tgp_sweep(tgp_handle h, int acq, double sf, double incumbent, double param, double *mu,
              double *sigma, double *acq_out, double *best_val, int64_t *best_idx,
              int64_t *n_clamped) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return h->host->sweep(acq, sf, incumbent, param, mu, sigma, acq_out, best_val, best_idx, n_clamped);
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_sweep: no fitted model");
    if (!c.d_cand || c.M < 1) return fail(c, TGP_BAD_ARG, "tgp_sweep: no candidates set");
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_sweep: unknown acquisition");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_sweep: sf must be +1 or -1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    const bool had_front = c.pre.front;
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const bool small = c.small && c.N <= 2 * NB;
    const bool mid = mid_sweep_cpw(c, c.M) != 0;
    c.last_sweep_f64 = small || mid || c.dtype == TGP_F64;   // (the one-workgroup / one-launch kernels only exist in f64)
    int rc = (small || mid) ? ensure_small_workspace(c) : ensure_workspace(c);
    if (rc != TGP_OK) return rc;
    // the front a fit started (tgp_set_overlap) is used when it belongs to the resident fit, batch and workspace geometry
    c.pre.usable = !small && !mid && had_front && c.pre.gen == c.fit_gen && c.pre.cand == c.d_cand && c.pre.M == c.M &&
                   c.pre.Mpad == c.ws_Mpad && c.pre.launch_rows == c.launch_rows && c.pre.chunk == c.chunk;
    c.pre.front = false;   // one sweep per front: the slab it filled is overwritten from here on
    rc = ensure_outputs(c, mu != nullptr, sigma != nullptr, acq_out != nullptr);
    if (rc != TGP_OK) return rc;

    // every sweep's last kernel leaves [best value, best index, clamp count] in device-mapped host memory and
    // hands the counters back at zero: no D2H copy, no memset behind it (TGP_SWEEP_ZC=0: the copies, A/B)
    const bool zc_off = tuning().sweep_zc == 0;
    const bool zc = small || mid || !zc_off;
    if (zc && (rc = ensure_pinned(c, 0, 8 * sizeof(double))) != TGP_OK) return rc;
    const hipEvent_t e0 = c.ev0, e1 = c.ev1;
    // (round 6) the small-problem sweep that only returns its record -- the arg-max of a trial -- is a polled call: the
    // last kernel rings the doorbell, no event, no stream synchronisation (doorbell.hpp)
    const Bell bell = ((small || mid) && !mu && !sigma && !acq_out) ? bell_next(c) : Bell{nullptr, 0, nullptr};
    const auto t_host0 = std::chrono::steady_clock::now();
    if (!bell.word) API_HIP(hipEventRecord(e0, c.stream), "hipEventRecord");
    hipError_t le;
    if (mid) {
        le = launch_mid_sweep(c, c.d_cand, acq, sf, incumbent, param, mu ? c.d_mu : nullptr,
                              sigma ? c.d_sigma : nullptr, acq_out ? c.d_acq : nullptr, c.d_pin_out, bell);
    } else if (small) {
        le = launch_small_sweep(c, c.d_cand, acq, sf, incumbent, param, mu ? c.d_mu : nullptr,
                                sigma ? c.d_sigma : nullptr, acq_out ? c.d_acq : nullptr);
        if (le == hipSuccess) le = launch_argmax_final(c, acq != TGP_ACQ_NONE ? (long)((c.M + NB - 1) / NB) : 0L, c.d_pin_out, bell);
    } else {
        c.sweep_res_host = zc ? c.d_pin_out : nullptr;
        le = launch_sweep(c, acq, sf, incumbent, param, mu != nullptr, sigma != nullptr, acq_out != nullptr);
        c.sweep_res_host = nullptr;
        c.pre.usable = false;
    }
    if (le != hipSuccess) return hip_fail(c, le, "launch_sweep");
    double bv = 0.0;
    long long bi[2] = {0, 0};
    if (!zc) {
        if (acq != TGP_ACQ_NONE) API_HIP(hipMemcpyAsync(&bv, c.d_best, sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H best");
        API_HIP(hipMemcpyAsync(bi, c.d_besti, 2 * sizeof(long long), hipMemcpyDeviceToHost, c.stream), "D2H besti");
        API_HIP(hipMemsetAsync(c.d_besti, 0, 4 * sizeof(long long), c.stream), "memset counters");   // zero between calls
    }
    if (!bell.word) API_HIP(hipEventRecord(e1, c.stream), "hipEventRecord");
    if (c.d_winner && acq != TGP_ACQ_NONE) {
        // the winner record is packed: what tgp_winner_wait makes another stream (RCCL's) wait for -- the ordering no
        // longer rests on this call ending in a stream synchronisation
        if (!c.ev_winner) API_HIP(hipEventCreateWithFlags(&c.ev_winner, hipEventDisableTiming), "hipEventCreate");
        API_HIP(hipEventRecord(c.ev_winner, c.stream), "hipEventRecord");
        c.winner_recorded = true;
    }
    const size_t bytes = (size_t)c.M * sizeof(double);
    if (mu) API_HIP(hipMemcpyAsync(mu, c.d_mu, bytes, hipMemcpyDeviceToHost, c.stream), "D2H mu");
    if (sigma) API_HIP(hipMemcpyAsync(sigma, c.d_sigma, bytes, hipMemcpyDeviceToHost, c.stream), "D2H sigma");
    if (acq_out) API_HIP(hipMemcpyAsync(acq_out, c.d_acq, bytes, hipMemcpyDeviceToHost, c.stream), "D2H acq");
    if (bell.word) {
        const int wrc = bell_wait(c, bell, "sweep sync");
        if (wrc != TGP_OK) return wrc;
        c.last_sweep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();   // (launch to doorbell, host clock)
    } else {
        API_HIP(hipStreamSynchronize(c.stream), "sweep sync");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        c.last_sweep_ms = ms;
    }
    if (c.profiling) prof_collect(c);
    if (zc) { bv = c.h_pin_out[0]; bi[0] = (long long)c.h_pin_out[1]; bi[1] = (long long)c.h_pin_out[2]; }
    if (acq != TGP_ACQ_NONE) {
        if (best_val) *best_val = bv;
        if (best_idx) *best_idx = (bi[0] >= c.M) ? 0 : (int64_t)bi[0];
    }
    if (n_clamped) *n_clamped = (int64_t)bi[1];
    return TGP_OK;
} TGP_CATCH

int tgp_acq_grad(tgp_handle h, const double *Xq, int64_t m, int acq, double sf, double incumbent,
                 double param, double *val, double *grad) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_acq_grad");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_acq_grad: no fitted model");
    if (!Xq || !val || !grad || m < 1 || m > 4096) return fail(c, TGP_BAD_ARG, "tgp_acq_grad: need Xq, val, grad and 1 <= m <= 4096");
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_acq_grad: unknown acquisition");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_acq_grad: sf must be +1 or -1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    // Round 6: a POLLED call when the batch fits the pinned staging -- the points are read from, value + gradient
    // written to, device-mapped host memory, and the last workgroup rings the call's doorbell (doorbell.hpp): no
    // memcpy, no event, no stream synchronisation.  N <= 128 in ONE launch (a workgroup per point), above that the
    // four general kernels with value + gradient formed by the reduction's last workgroup.
    const int64_t D = c.D;
    const size_t zc_in = (size_t)(m * D) * sizeof(double), zc_out = (size_t)(8 + m + m * D) * sizeof(double);
    const bool small_q = small_refine_fits(c) && small_path_enabled() && tuning().small_query != 0;
    Bell bell{nullptr, 0, nullptr};
    if (zc_in <= ((size_t)1 << 20) && zc_out <= ((size_t)1 << 20)) bell = bell_next(c);
    if (bell.word) {
        int prc = ensure_pinned(c, zc_in, zc_out);
        if (prc != TGP_OK) return prc;
        memcpy(c.h_pin_in, Xq, zc_in);
        double *o_val = c.d_pin_out + 8, *o_grad = o_val + m;
        hipError_t le;
        if (small_q) {
            le = launch_small_query(c, c.d_pin_in, (int)m, acq, sf, incumbent, param, o_val, o_grad, bell);
        } else {
            const int64_t per = query_ws_doubles(c);   // launch_query's workspace
            const int64_t need = m * (2 * D + 1) + m * per;     // (the layout of the copying path below, so both share d_qws)
            if (need > c.qws_cap) {
                API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
                dfree(c.d_qws);
                c.qws_cap = 0;
                API_HIP(hipMalloc((void **)&c.d_qws, (size_t)need * sizeof(double)), "hipMalloc query workspace");
                c.qws_cap = need;
            }
            le = launch_query(c, c.d_pin_in, (int)m, acq, sf, incumbent, param, c.d_qws + m * (2 * D + 1), o_val, o_grad, bell);
        }
        if (le != hipSuccess) return hip_fail(c, le, "launch_query");
        int wrc = bell_wait(c, bell, "query sync");
        if (wrc != TGP_OK) return wrc;
        memcpy(val, c.h_pin_out + 8, (size_t)m * sizeof(double));
        memcpy(grad, c.h_pin_out + 8 + m, (size_t)(m * D) * sizeof(double));
        return TGP_OK;
    }
    // [Xq (m D) | val (m) | grad (m D) | workspace]
    const int64_t per = query_ws_doubles(c);   // launch_query's workspace
    const int64_t need = m * (2 * c.D + 1) + m * per;
    if (need > c.qws_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        dfree(c.d_qws);
        API_HIP(hipMalloc((void **)&c.d_qws, (size_t)need * sizeof(double)), "hipMalloc query workspace");
        c.qws_cap = need;
    }
    double *d_Xq = c.d_qws, *d_val = d_Xq + m * c.D, *d_grad = d_val + m, *d_ws = d_grad + m * c.D;
    API_HIP(hipMemcpyAsync(d_Xq, Xq, (size_t)(m * c.D) * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D Xq");
    hipError_t le = launch_query(c, d_Xq, (int)m, acq, sf, incumbent, param, d_ws, d_val, d_grad);
    if (le != hipSuccess) return hip_fail(c, le, "launch_query");
    API_HIP(hipMemcpyAsync(val, d_val, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H val");
    API_HIP(hipMemcpyAsync(grad, d_grad, (size_t)(m * c.D) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H grad");
    API_HIP(hipStreamSynchronize(c.stream), "query sync");
    return TGP_OK;
} TGP_CATCH

int tgp_sweep_topk(tgp_handle h, int acq, double sf, double incumbent, double param, int64_t k,
                   double *vals, int64_t *idxs, int64_t *n_clamped) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_sweep_topk");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_sweep_topk: no fitted model");
    if (!c.d_cand || c.M < 1) return fail(c, TGP_BAD_ARG, "tgp_sweep_topk: no candidates set");
    if (acq <= TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_sweep_topk: needs an acquisition");
    if (k < 1 || k > 64 || !vals || !idxs) return fail(c, TGP_BAD_ARG, "tgp_sweep_topk: need 1 <= k <= 64, vals and idxs");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_sweep_topk: sf must be +1 or -1");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const bool small = c.small && c.N <= 2 * NB;
    const bool mid = mid_sweep_cpw(c, c.M) != 0;
    c.last_sweep_f64 = small || mid || c.dtype == TGP_F64;   // (the one-workgroup / one-launch kernels only exist in f64)
    int rc = (small || mid) ? ensure_small_workspace(c) : ensure_workspace(c);
    if (rc != TGP_OK) return rc;
    rc = ensure_outputs(c, false, false, true);           // the (M,) acquisition vector stays on the device
    if (rc != TGP_OK) return rc;
    const int64_t nb = (c.M + 4095) / 4096;
    const size_t ents = (size_t)(2 * nb * k);
    if ((rc = grow(c, c.d_topv, c.cap_topv, ents * sizeof(double), "hipMalloc topk values")) != TGP_OK) return rc;
    if ((rc = grow(c, c.d_topi, c.cap_topi, ents * sizeof(long long), "hipMalloc topk indices")) != TGP_OK) return rc;
    // (round 6) small and one-launch models: a polled call -- the top-k's final pass leaves values, indices and the clamp
    // count in mapped host memory, hands the counters back at zero and rings; no D2H copy, memset, event or synchronisation
    Bell bell{nullptr, 0, nullptr};
    if (small || mid) {
        bell = bell_next(c);
        if (bell.word && (rc = ensure_pinned(c, 0, (size_t)(8 + 2 * k + 1) * sizeof(double))) != TGP_OK) return rc;
    }
    const auto t_host0 = std::chrono::steady_clock::now();
    if (!bell.word) API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
    hipError_t le;
    if (mid) {
        le = launch_mid_sweep(c, c.d_cand, acq, sf, incumbent, param, nullptr, nullptr, c.d_acq, nullptr);
    } else if (small) {
        le = launch_small_sweep(c, c.d_cand, acq, sf, incumbent, param, nullptr, nullptr, c.d_acq);
        if (le == hipSuccess) le = launch_argmax_final(c, (long)((c.M + NB - 1) / NB), nullptr);
    } else {
        le = launch_sweep(c, acq, sf, incumbent, param, false, false, true);
    }
    if (le != hipSuccess) return hip_fail(c, le, "launch_sweep");
    long off = 0;
    le = launch_topk(c, c.d_acq, (long)c.M, (int)k, c.d_topv, c.d_topi, &off, bell.word ? c.d_pin_out + 8 : nullptr, bell);
    if (le != hipSuccess) return hip_fail(c, le, "launch_topk");
    if (bell.word) {
        const int wrc = bell_wait(c, bell, "topk sync");
        if (wrc != TGP_OK) return wrc;
        c.last_sweep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
        const double *rec = c.h_pin_out + 8;
        for (int64_t i = 0; i < k; ++i) { vals[i] = rec[i]; idxs[i] = (int64_t)rec[k + i]; }     // (-1: fewer than k candidates)
        if (n_clamped) *n_clamped = (int64_t)rec[2 * k];
        if (c.profiling) prof_collect(c);
        return TGP_OK;
    }
    std::vector<long long> hi((size_t)k);
    long long bi[2] = {0, 0};
    API_HIP(hipMemcpyAsync(vals, c.d_topv + off, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H topk values");
    API_HIP(hipMemcpyAsync(hi.data(), c.d_topi + off, (size_t)k * sizeof(long long), hipMemcpyDeviceToHost, c.stream), "D2H topk indices");
    API_HIP(hipMemcpyAsync(bi, c.d_besti, 2 * sizeof(long long), hipMemcpyDeviceToHost, c.stream), "D2H besti");
    API_HIP(hipMemsetAsync(c.d_besti, 0, 4 * sizeof(long long), c.stream), "memset counters");
    API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
    API_HIP(hipStreamSynchronize(c.stream), "topk sync");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
    c.last_sweep_ms = ms;
    if (c.profiling) prof_collect(c);
    for (int64_t i = 0; i < k; ++i) idxs[i] = (hi[(size_t)i] == 0x7fffffffffffffffLL) ? -1 : (int64_t)hi[(size_t)i];   // -1: fewer than k candidates
    if (n_clamped) *n_clamped = (int64_t)bi[1];
    return TGP_OK;
} TGP_CATCH

int tgp_acq_refine(tgp_handle h, const double *X0, int64_t R, const double *lo, const double *hi,
                   int acq, double sf, double incumbent, double param, int64_t max_iter,
                   double *x_out, double *val_out, int64_t *status_out, int64_t *iterations) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_acq_refine");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_acq_refine: no fitted model");
    if (!X0 || !lo || !hi || !x_out || !val_out || R < 1 || R > 4096)
        return fail(c, TGP_BAD_ARG, "tgp_acq_refine: need X0, lo, hi, x_out, val_out and 1 <= R <= 4096");
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_acq_refine: unknown acquisition");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_acq_refine: sf must be +1 or -1");
    if (max_iter < 1) return fail(c, TGP_BAD_ARG, "tgp_acq_refine: max_iter >= 1");
    for (int64_t d = 0; d < c.D; ++d)
        if (!(lo[d] <= hi[d])) return fail(c, TGP_BAD_ARG, "tgp_acq_refine: need lo <= hi in every dimension");
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int64_t D = c.D, m = R;
    if (small_refine_fits(c) && small_path_enabled()) {
        // small problems: one launch, one workgroup per restart running its whole optimisation; starts and
        // bounds are read from, results written to, device-mapped host memory (no memcpy)
        const size_t n_in = (size_t)(R * D + 2 * D), n_out = (size_t)(8 + R * D + 4 * R);
        int prc = ensure_pinned(c, n_in * sizeof(double), n_out * sizeof(double));
        if (prc != TGP_OK) return prc;
        memcpy(c.h_pin_in, X0, (size_t)(R * D) * sizeof(double));
        memcpy(c.h_pin_in + R * D, lo, (size_t)D * sizeof(double));
        memcpy(c.h_pin_in + R * D + D, hi, (size_t)D * sizeof(double));
        double *o_x = c.d_pin_out + 8, *o_v = o_x + R * D, *o_info = o_v + R;
        API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
        hipError_t le = launch_small_refine(c, c.d_pin_in, c.d_pin_in + R * D, c.d_pin_in + R * D + D, (int)R, acq, sf,
                                            incumbent, param, (int)std::min<int64_t>(max_iter, 1 << 30), 1e-5,
                                            2.220446049250313e-09, o_x, o_v, o_info);
        if (le != hipSuccess) return hip_fail(c, le, "launch_small_refine");
        API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
        API_HIP(hipStreamSynchronize(c.stream), "refine sync");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
        c.last_sweep_ms = ms;
        const double *h_x = c.h_pin_out + 8, *h_v = h_x + R * D, *h_info = h_v + R;
        memcpy(x_out, h_x, (size_t)(R * D) * sizeof(double));
        memcpy(val_out, h_v, (size_t)R * sizeof(double));
        int64_t ev = 0;
        for (int64_t r = 0; r < R; ++r) {
            if (status_out) status_out[r] = (int64_t)h_info[3 * r];
            ev = std::max<int64_t>(ev, (int64_t)h_info[3 * r + 2]);
        }
        if (iterations) *iterations = ev;      // evaluations of the slowest restart (what the lock-step path counts)
        return TGP_OK;
    }
    // query workspace as tgp_acq_grad: [Xq (m D) | val (m) | grad (m D) | workspace]
    const int64_t per = query_ws_doubles(c);
    const int64_t need = m * (2 * D + 1) + m * per;
    if (need > c.qws_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        dfree(c.d_qws);
        API_HIP(hipMalloc((void **)&c.d_qws, (size_t)need * sizeof(double)), "hipMalloc query workspace");
        c.qws_cap = need;
    }
    // optimiser state: [state (R stride) | history of the eight-wave teams (D > 1024) | lo (D) | hi (D) | x_out (R D) |
    //                   v_out (R) | info (2 R) | active (int)]
    const int64_t stride = refine_state_stride((int)D);
    const int64_t hist = refine_hist_doubles((int)D, (int)R);
    const size_t rf_need = (size_t)(R * stride + hist + 2 * D + R * D + R + 2 * R + 2) * sizeof(double);
    int rc = grow(c, c.d_rf, c.cap_rf, rf_need, "hipMalloc refine state");
    if (rc != TGP_OK) return rc;
    double *d_Xq = c.d_qws, *d_val = d_Xq + m * D, *d_grad = d_val + m, *d_ws = d_grad + m * D;
    double *d_state = c.d_rf, *d_lo = d_state + R * stride + hist, *d_hi = d_lo + D, *d_xo = d_hi + D;
    double *d_vo = d_xo + R * D, *d_info = d_vo + R;
    int *d_active = reinterpret_cast<int *>(d_info + 2 * R);
    API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
    API_HIP(hipMemcpyAsync(d_Xq, X0, (size_t)(R * D) * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D X0");
    API_HIP(hipMemcpyAsync(d_lo, lo, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D lo");
    API_HIP(hipMemcpyAsync(d_hi, hi, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D hi");
    hipError_t le = launch_refine_clip(c, d_Xq, d_lo, d_hi, (int)R);
    if (le != hipSuccess) return hip_fail(c, le, "launch_refine_clip");
    // every iteration: one batched value + gradient evaluation of all R trial points (the closed-form
    // kernels of tgp_acq_grad), then one optimiser step for all restarts.  The host only looks at
    // the count of unfinished restarts, every 4th iteration.
    int64_t it = 0;
    int active = (int)R;
    for (; it <= max_iter; ++it) {
        // (the wave step turns the evaluation's sums into value + gradient itself, one launch fewer)
        const bool fused = true;
        le = launch_query(c, d_Xq, (int)m, acq, sf, incumbent, param, d_ws, fused ? nullptr : d_val, d_grad);
        if (le != hipSuccess) return hip_fail(c, le, "launch_query");
        le = launch_refine_step(c, d_state, d_Xq, d_val, d_grad, d_lo, d_hi, (int)R, (int)it, 1e-5, 2.220446049250313e-09, d_active,
                                fused ? query_red(c, d_ws, (int)m) : nullptr, acq, sf, incumbent, param);
        if (le != hipSuccess) return hip_fail(c, le, "launch_refine_step");
        if ((it & 3) == 3 || it == max_iter) {
            API_HIP(hipMemcpyAsync(&active, d_active + (it & 1), sizeof(int), hipMemcpyDeviceToHost, c.stream), "D2H active");
            API_HIP(hipStreamSynchronize(c.stream), "refine sync");
            if (active == 0) { ++it; break; }
        }
    }
    le = launch_refine_collect(c, d_state, (int)R, d_xo, d_vo, d_info);
    if (le != hipSuccess) return hip_fail(c, le, "launch_refine_collect");
    std::vector<double> info((size_t)(2 * R));
    API_HIP(hipMemcpyAsync(x_out, d_xo, (size_t)(R * D) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H x");
    API_HIP(hipMemcpyAsync(val_out, d_vo, (size_t)R * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H values");
    API_HIP(hipMemcpyAsync(info.data(), d_info, (size_t)(2 * R) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H info");
    API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
    API_HIP(hipStreamSynchronize(c.stream), "refine sync");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
    c.last_sweep_ms = ms;
    if (status_out)
        for (int64_t r = 0; r < R; ++r) status_out[r] = (int64_t)info[(size_t)(2 * r)];
    if (iterations) *iterations = it;
    return TGP_OK;
} TGP_CATCH

// The gradient stage as the reference runs it -- scipy.optimize.minimize(method='L-BFGS-B', maxiter = 15000) on the
// negated acquisition from every start (turbo/modules/auxiliary_optimisers.py:80-99) -- inside the library: one
// host_lbfgsb.hpp optimiser per restart, all of them in LOCK-STEP on the calling thread (reverse communication needs no
// threads): every round gathers the trial points of the restarts still running, ONE batched closed-form value +
// gradient evaluation serves them all (tgp_acq_grad: a point's value does not depend on the batch it travels in), and
// every optimiser takes its step.  What turbo_amd/auxiliary_optimisers.py did with a Python thread and a SciPy
// optimiser per restart meeting at a rendezvous; same walk per restart, no interpreter between two rounds.
int tgp_acq_lbfgsb(tgp_handle h, const double *X0, int64_t R, const double *lo, const double *hi,
                   int acq, double sf, double incumbent, double param, int64_t max_iter,
                   double *x_out, double *val_out, int64_t *status_out, int64_t *evaluations) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_acq_lbfgsb");
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_acq_lbfgsb: no fitted model");
    if (!X0 || !lo || !hi || !x_out || !val_out || R < 1 || R > 4096)
        return fail(c, TGP_BAD_ARG, "tgp_acq_lbfgsb: need X0, lo, hi, x_out, val_out and 1 <= R <= 4096");
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_acq_lbfgsb: unknown acquisition");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_acq_lbfgsb: sf must be +1 or -1");
    if (max_iter < 1) return fail(c, TGP_BAD_ARG, "tgp_acq_lbfgsb: max_iter >= 1");
    const int D = (int)c.D;
    for (int d = 0; d < D; ++d)
        if (!(lo[d] <= hi[d])) return fail(c, TGP_BAD_ARG, "tgp_acq_lbfgsb: need lo <= hi in every dimension");
    std::vector<HostLbfgsb> opts;
    opts.reserve((size_t)R);
    std::vector<std::vector<double>> xt((size_t)R, std::vector<double>((size_t)D));
    std::vector<char> started((size_t)R, 0), live((size_t)R, 1);
    std::vector<int64_t> rounds((size_t)R, 0);
    for (int64_t r = 0; r < R; ++r) {
        opts.emplace_back(lo, hi, D);
        for (int d = 0; d < D; ++d) xt[(size_t)r][(size_t)d] = HostLbfgsb::clip(X0[r * D + d], lo[d], hi[d]);
    }
    std::vector<double> Xq((size_t)R * D), val((size_t)R), grad((size_t)R * D), gt((size_t)D);
    std::vector<int64_t> who((size_t)R);
    int64_t ev = 0, n_live = R;
    while (n_live > 0) {
        // restarts whose next point is the one they evaluated last (a search giving up on "no further progress") are
        // answered from that evaluation, as SciPy answers them from its cache; the others go into this round's batch
        int64_t m = 0;
        for (int64_t r = 0; r < R; ++r) {
            if (!live[(size_t)r]) continue;
            HostLbfgsb &o = opts[(size_t)r];
            while (live[(size_t)r] && started[(size_t)r] && o.evaluated(xt[(size_t)r])) {
                gt = o.g_eval;
                o.step(xt[(size_t)r], gt, o.f_eval, false, 1e-5, 2.220446049250313e-09);
                if (o.status != 0 || o.iters >= max_iter || ++rounds[(size_t)r] >= 15000) { live[(size_t)r] = 0; --n_live; }
            }
            if (!live[(size_t)r]) continue;
            memcpy(&Xq[(size_t)(m * D)], xt[(size_t)r].data(), (size_t)D * sizeof(double));
            who[(size_t)m++] = r;
        }
        if (m == 0) break;
        const int rc = tgp_acq_grad(h, Xq.data(), m, acq, sf, incumbent, param, val.data(), grad.data());
        if (rc != TGP_OK) return rc;
        ev += m;
        for (int64_t i = 0; i < m; ++i) {
            const int64_t r = who[(size_t)i];
            HostLbfgsb &o = opts[(size_t)r];
            for (int d = 0; d < D; ++d) gt[(size_t)d] = -grad[(size_t)(i * D + d)];     // maximise = minimise the negation
            o.step(xt[(size_t)r], gt, -val[(size_t)i], !started[(size_t)r], 1e-5, 2.220446049250313e-09);
            started[(size_t)r] = 1;
            if (o.status != 0 || o.iters >= max_iter || ++rounds[(size_t)r] >= 15000) { live[(size_t)r] = 0; --n_live; }
        }
    }
    for (int64_t r = 0; r < R; ++r) {
        const HostLbfgsb &o = opts[(size_t)r];
        for (int d = 0; d < D; ++d) x_out[r * D + d] = o.x[(size_t)d];
        val_out[r] = -o.phi;
        if (status_out) status_out[r] = o.status;
    }
    if (evaluations) *evaluations = ev;
    return TGP_OK;
} TGP_CATCH

// tgp_fit_optimise above the one-launch sizes: every start is a host_lbfgsb.hpp optimiser (SciPy's L-BFGS-B restated:
// the iterates scikit-learn's fit walks) driving tgp_fit_grad
// (fit + LML gradient on the GPU); the starts run side by side, a C++ thread and a handle on a stream of its own
// each (start j of a thread's share after the other: thread t takes starts t, t + T, ...), with no interpreter
// between two evaluations.  Every start walks the iterates it walks alone -- its own optimiser state, its own
// handle -- so the result does not depend on the number of threads.
static int fit_optimise_streams(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                                const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                                double jitter, int normalize_y, int64_t max_iter, double *theta_out, double *f_out,
                                int64_t *status_out, int64_t *evaluations) {
    Context &c = h->c;
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    const int P = (int)(2 + n_ls);
    // measured through the plugin path (tools/bench_latency.py): one evaluation is a serial chain of ~35 launches that
    // leaves the chip idle up to N ~ 1000; beyond, two chains already share the CUs
    const int threads_env = tuning().hyper_threads;
    // (round 4, late, with the workers' inverses in line and every start on a worker: three starts at N = 700 / 1000 / 1500
    // take 18.6 / 27.1 / 36.8 ms on three threads against 32 / 42 / 53 on one; at 2048 the chains fill the chip: 120 vs 117)
    // (round 5, late, re-measured with L-BFGS-B in the library and the one-outer-block fits: three starts side by side
    // now pay well beyond 1536 -- N = 2048 55.8 -> 42.2 ms, 3000 110 -> 82, 4096 240 -> 206, 6144 482 -> 440, 8192 984 ->
    // 911 -- two threads never do (2 + 1); the limit is the workers' workspaces, 4 N^2 doubles each: 2 GiB at 8192)
    int T = threads_env > 0 ? threads_env : (N <= 1536 ? 4 : (N <= 8192 ? 3 : 1));
    T = (int)std::min<int64_t>(T, S);
    // With more than one thread EVERY start runs on a worker handle (a private stream each) and the caller's handle sits
    // out: on its shared main stream it ended up serialised with one of the workers in the first factory of a fresh
    // process (an evaluation 0.35 -> 0.83 ms on both; N = 500, three starts: 23-27 ms instead of 11.6; tools/diag_streams.py
    // shows the pairing).  No probe saw it -- one launch on each stream, chains of memory-dependent launches from one
    // thread, launch + wait cycles from two threads all ran side by side -- so re-creating streams until a probe passes
    // is no cure; the workers do not show it among themselves.
    std::vector<tgp_handle> workers((size_t)T, nullptr);
    struct Borrowed {   // the pool is this call's from here to the return
        tgp_handle h; bool held = false;
        ~Borrowed() { if (held) (void)tgp_workers_release(h); }
    } borrowed{h};
    if (T > 1) {
        const int rc = tgp_workers_acquire(h, T, workers.data());
        if (rc != TGP_OK) return rc;
        borrowed.held = true;
    }
    std::vector<int> status((size_t)S, 0), rcs((size_t)T, TGP_OK);
    std::vector<int64_t> evals((size_t)S, 0);
    std::vector<std::string> errs((size_t)T);
    // an entry with log_lo == log_hi is FIXED at that value: the optimiser works on the other entries only, as
    // scikit-learn leaves a fixed hyper-parameter out of theta (kernels.py: Hyperparameter(..., "fixed")) -- not as a
    // coordinate of the box that cannot move, whose gradient would still enter the quasi-Newton pairs
    std::vector<int> fr;
    std::vector<double> lo_f, hi_f;
    for (int k = 0; k < P; ++k)
        if (log_lo[k] < log_hi[k]) { fr.push_back(k); lo_f.push_back(log_lo[k]); hi_f.push_back(log_hi[k]); }
    const int Pf = (int)fr.size();
    auto run_share = [&](int t) {
      try {
        tgp_handle hw = T == 1 ? h : workers[(size_t)t];
        std::vector<double> ls((size_t)n_ls), grad((size_t)P), th((size_t)P), xt((size_t)Pf), gt((size_t)Pf);
        for (int64_t s = t; s < S; s += T) {
            // th = the full vector log(constant, length scale(s), noise); the optimiser sees its free entries only
            for (int k = 0; k < P; ++k) th[(size_t)k] = HostLbfgsb::clip(theta0[s * P + k], log_lo[k], log_hi[k]);
            HostLbfgsb opt(lo_f.data(), hi_f.data(), Pf);
            for (int k = 0; k < Pf; ++k) xt[(size_t)k] = th[(size_t)fr[(size_t)k]];
            // max_iter bounds ACCEPTED iterations (opt.iters), as in the one-launch path and as SciPy's maxiter does;
            // the line searches' trial evaluations have a cap of their own, SciPy's maxfun = 15000 (round-4 advisor:
            // every trial used to count against max_iter, so an ARD fit could stop early with status 0)
            int64_t it = 0;
            for (; opt.iters < max_iter && it < 15000; ++it) {
                if (it > 0 && opt.evaluated(xt)) {      // the search's best point once more: answered from the last evaluation
                    gt = opt.g_eval;
                    opt.step(xt, gt, opt.f_eval, false, 1e-5, 2.220446049250313e-09);
                    if (opt.status != 0) break;
                    continue;
                }
                for (int k = 0; k < Pf; ++k) th[(size_t)fr[(size_t)k]] = xt[(size_t)k];
                const double constant = exp(th[0]), noise = exp(th[(size_t)(P - 1)]);
                for (int64_t d = 0; d < n_ls; ++d) ls[(size_t)d] = exp(th[(size_t)(1 + d)]);
                double lml = 0.0, phit;
                const int rc = tgp_fit_grad(hw, X, N, D, y, kernel, constant, ls.data(), n_ls, noise, jitter, normalize_y,
                                            &lml, nullptr, nullptr, grad.data());
                if (rc == TGP_NOT_PD) {            // -inf likelihood, zero gradient (_gpr.py:586-589)
                    phit = INFINITY;
                    for (int k = 0; k < Pf; ++k) gt[(size_t)k] = 0.0;
                } else if (rc != TGP_OK) {
                    rcs[(size_t)t] = rc;
                    errs[(size_t)t] = tgp_last_error(hw);
                    return;
                } else {
                    phit = -lml;
                    for (int k = 0; k < Pf; ++k) gt[(size_t)k] = -grad[(size_t)fr[(size_t)k]];
                }
                ++evals[(size_t)s];
                if (Pf == 0) {                     // nothing is free: the one evaluation is the answer
                    opt.phi = phit;
                    opt.status = 1;
                    break;
                }
                opt.step(xt, gt, phit, it == 0, 1e-5, 2.220446049250313e-09);   // SciPy's L-BFGS-B defaults: pgtol, factr 1e7 x eps
                if (opt.status != 0) break;
            }
            status[(size_t)s] = opt.status;
            for (int k = 0; k < Pf; ++k) th[(size_t)fr[(size_t)k]] = opt.x[(size_t)k];
            for (int k = 0; k < P; ++k) theta_out[s * P + k] = th[(size_t)k];
            f_out[s] = opt.phi;
        }
      } catch (const std::bad_alloc &) {    // (no exception leaves a worker thread: that would be std::terminate)
        rcs[(size_t)t] = TGP_NO_MEMORY;
        errs[(size_t)t] = "out of host memory";
      } catch (...) {
        rcs[(size_t)t] = TGP_HIP_ERROR;
        errs[(size_t)t] = "unexpected exception in a start's thread";
      }
    };
    (void)run_on_start_threads(T, run_share);   // (the library's parked start threads; shares no thread could be had for run on this one)
    for (int t = 0; t < T; ++t)
        if (rcs[(size_t)t] != TGP_OK) return fail(c, rcs[(size_t)t], "tgp_fit_optimise: " + errs[(size_t)t]);
    int64_t ev = 0;
    for (int64_t s = 0; s < S; ++s) {
        if (status_out) status_out[s] = status[(size_t)s];
        ev += evals[(size_t)s];
    }
    if (evaluations) *evaluations = ev;
    return TGP_OK;
}

static int check_optimise_args(Context &c, const char *fn, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                               const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                               int64_t max_iter, const double *theta_out, const double *f_out) {
    const std::string name(fn);
    if (!X || !y || !theta0 || !log_lo || !log_hi || !theta_out || !f_out)
        return fail(c, TGP_BAD_ARG, name + ": need X, y, theta0, log_lo, log_hi, theta_out, f_out");
    if (kernel < 0 || kernel > 3) return fail(c, TGP_BAD_ARG, name + ": unknown kernel");
    if (N < 1 || D < 1 || D > 4096 || (n_ls != 1 && n_ls != D) || S < 1 || S > 64 || max_iter < 1)
        return fail(c, TGP_BAD_ARG, name + ": needs N >= 1, 1 <= D <= 4096, n_ls 1 or D, 1 <= S <= 64, max_iter >= 1");
    for (int64_t i = 0; i < 2 + n_ls; ++i) {
        // (the noise entry fixed at log 0: a kernel without a noise term)
        if (i == 1 + n_ls && log_lo[i] == -INFINITY && log_hi[i] == -INFINITY) continue;
        if (!(log_lo[i] <= log_hi[i]) || !isfinite(log_lo[i]) || !isfinite(log_hi[i]))
            return fail(c, TGP_BAD_ARG, name + ": bounds must be finite with lo <= hi (the noise entry may be fixed at -inf: no noise term)");
    }
    return TGP_OK;
}

int tgp_fit_lbfgsb(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                   const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                   double jitter, int normalize_y, int64_t max_iter, double *theta_out, double *f_out,
                   int64_t *status_out, int64_t *evaluations) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_fit_lbfgsb");
    Context &c = h->c;
    const int rc = check_optimise_args(c, "tgp_fit_lbfgsb", X, N, D, y, kernel, theta0, S, n_ls, log_lo, log_hi, max_iter, theta_out, f_out);
    if (rc != TGP_OK) return rc;
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    return fit_optimise_streams(h, X, N, D, y, kernel, theta0, S, n_ls, log_lo, log_hi, jitter, normalize_y, max_iter,
                                theta_out, f_out, status_out, evaluations);
} TGP_CATCH

int tgp_fit_optimise(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                     const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                     double jitter, int normalize_y, int64_t max_iter, double *theta_out, double *f_out,
                     int64_t *status_out, int64_t *evaluations) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_fit_optimise");
    Context &c = h->c;
    const int rc0 = check_optimise_args(c, "tgp_fit_optimise", X, N, D, y, kernel, theta0, S, n_ls, log_lo, log_hi, max_iter, theta_out, f_out);
    if (rc0 != TGP_OK) return rc0;
    const int64_t Dp = ((D + 3) / 4) * 4, P = 2 + n_ls;
    if (N > 2 * NB || Dp > 64 || P > 64 || !small_path_enabled() || log_lo[P - 1] == -INFINITY)
        return fit_optimise_streams(h, X, N, D, y, kernel, theta0, S, n_ls, log_lo, log_hi, jitter, normalize_y, max_iter,
                                    theta_out, f_out, status_out, evaluations);
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    double mean = 0.0, sd = 1.0;
    std::vector<double> yn((size_t)N, 0.0);
    normalise_targets(y, N, normalize_y, yn, mean, sd);
    // inputs through device-mapped host memory (each start copies X and the targets once), results the same way
    const size_t n_in = (size_t)(N * D + N + S * P + 2 * P), n_out = (size_t)(8 + S * P + S + 3 * S);
    int rc = ensure_pinned(c, n_in * sizeof(double), n_out * sizeof(double));
    if (rc != TGP_OK) return rc;
    double *in = c.h_pin_in;
    memcpy(in, X, (size_t)(N * D) * sizeof(double));
    memcpy(in + N * D, yn.data(), (size_t)N * sizeof(double));
    memcpy(in + N * D + N, theta0, (size_t)(S * P) * sizeof(double));
    memcpy(in + N * D + N + S * P, log_lo, (size_t)P * sizeof(double));
    memcpy(in + N * D + N + S * P + P, log_hi, (size_t)P * sizeof(double));
    const size_t ws_need = (size_t)S * (size_t)small_hyper_workspace_doubles((int)N, (int)D, (int)Dp) * sizeof(double);
    rc = grow(c, c.d_rf, c.cap_rf, ws_need, "hipMalloc hyper workspace");
    if (rc != TGP_OK) return rc;
    const double *d_in = c.d_pin_in;
    double *o_theta = c.d_pin_out + 8, *o_f = o_theta + S * P, *o_info = o_f + S;
    const double *h_theta = c.h_pin_out + 8, *h_f = h_theta + S * P, *h_info = h_f + S;
    bool timed_out = false;
    // A start whose three workgroups never met (status 3: the barrier's time budget ran out, e.g. CUs taken
    // away by HSA_CU_MASK or by another process on the card) invalidates the launch: nothing of it reaches
    // the caller, and the fit runs once more with one workgroup per start -- no barrier, no way to wait.
    for (int attempt = 0; attempt < 2; ++attempt) {
        API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
        hipError_t le = launch_small_hyper(c, kernel, d_in, d_in + N * D, d_in + N * D + N, d_in + N * D + N + S * P,
                                           d_in + N * D + N + S * P + P, (int)S, (int)N, (int)D, (int)Dp, (int)n_ls,
                                           (int)std::min<int64_t>(max_iter, 1 << 30), jitter, c.d_rf, o_theta, o_f, o_info,
                                           attempt > 0);
        if (le != hipSuccess) return hip_fail(c, le, "launch_small_hyper");
        API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
        API_HIP(hipStreamSynchronize(c.stream), "hyper sync");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
        c.last_fit_ms = ms;
        timed_out = false;
        for (int64_t s = 0; s < S; ++s) timed_out = timed_out || (int64_t)h_info[3 * s] == 3;
        if (!timed_out) break;
    }
    if (timed_out) return fail(c, TGP_HIP_ERROR, "tgp_fit_optimise: the barrier between a start's workgroups timed out, and so did the relaunch with one workgroup per start");
    memcpy(theta_out, h_theta, (size_t)(S * P) * sizeof(double));
    memcpy(f_out, h_f, (size_t)S * sizeof(double));
    int64_t ev = 0;
    for (int64_t s = 0; s < S; ++s) {
        if (status_out) status_out[s] = (int64_t)h_info[3 * s];
        ev += (int64_t)h_info[3 * s + 2];
    }
    if (evaluations) *evaluations = ev;
    return TGP_OK;
} TGP_CATCH

int tgp_evaluate(tgp_handle h, const double *Xc, int64_t M, int acq, double sf, double incumbent,
                 double param, double *mu, double *sigma, double *acq_out, double *best_val,
                 int64_t *best_idx, int64_t *n_clamped) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) {
        const int rc = h->host->set_candidates(Xc, M);
        return rc != TGP_OK ? rc : h->host->sweep(acq, sf, incumbent, param, mu, sigma, acq_out, best_val, best_idx, n_clamped);
    }
    Context &c = h->c;
    if (!c.fitted) return fail(c, TGP_NOT_FITTED, "tgp_evaluate: no fitted model");
    if (!Xc || M < 1) return fail(c, TGP_BAD_ARG, "tgp_evaluate: need Xc and M >= 1");
    if (acq < TGP_ACQ_NONE || acq > TGP_ACQ_SIGMA) return fail(c, TGP_BAD_ARG, "tgp_evaluate: unknown acquisition");
    if (sf != 1.0 && sf != -1.0) return fail(c, TGP_BAD_ARG, "tgp_evaluate: sf must be +1 or -1");
    const size_t in_bytes = (size_t)M * (size_t)c.D * sizeof(double);
    const bool mid = mid_sweep_cpw(c, M) != 0;
    const bool zero_copy = ((c.small && c.N <= 2 * NB) || mid) && in_bytes <= ((size_t)8 << 20) && M <= 262144;
    if (zero_copy) c.last_sweep_f64 = true;   // (otherwise tgp_sweep below says)
    if (!zero_copy) {
        int rc = tgp_set_candidates(h, Xc, M);
        if (rc != TGP_OK) return rc;
        return tgp_sweep(h, acq, sf, incumbent, param, mu, sigma, acq_out, best_val, best_idx, n_clamped);
    }
    // Small (N <= 128) or mid-size (N <= 256) model, small batch (the plot path: turbo/plotting/trials.py:371,448,574-577; 1-point
    // calls of a foreign optimiser): candidates and results travel through pinned, device-mapped
    // host memory -- two launches, one synchronisation, no memcpy call.
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    if (in_bytes > c.pin_cand_cap) {
        API_HIP(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
        if (c.h_pin_cand) (void)hipHostFree(c.h_pin_cand);
        c.h_pin_cand = nullptr; c.d_pin_cand = nullptr; c.pin_cand_cap = 0;
        const size_t cap = std::max<size_t>(in_bytes, (size_t)1 << 20);
        API_HIP(hipHostMalloc((void **)&c.h_pin_cand, cap, hipHostMallocMapped), "hipHostMalloc");
        API_HIP(hipHostGetDevicePointer((void **)&c.d_pin_cand, c.h_pin_cand, 0), "hipHostGetDevicePointer");
        c.pin_cand_cap = cap;
    }
    int rc = ensure_pinned(c, 0, (size_t)(8 + 3 * M) * sizeof(double));
    if (rc != TGP_OK) return rc;
    memcpy(c.h_pin_cand, Xc, in_bytes);
    c.d_cand = c.d_pin_cand;              // resident (in host memory the GPU can read) until replaced
    c.M = M;
    rc = ensure_small_workspace(c);
    if (rc != TGP_OK) return rc;
    double *o_res = c.d_pin_out, *o_mu = c.d_pin_out + 8, *o_sg = o_mu + M, *o_aq = o_sg + M;
    // (round 6) a polled call.  N <= 128: the record's kernel is a launch of its own behind the sweep, so when it rings every
    // mean / deviation / acquisition value the sweep wrote into mapped host memory is out.  The one-launch sweep of
    // 128 < N <= 256: every workgroup's output stores come from the wave that takes its ticket behind a system-scope
    // fence, and the last workgroup rings
    const Bell bell = bell_next(c);
    const auto t_host0 = std::chrono::steady_clock::now();
    if (!bell.word) API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
    hipError_t le;
    if (mid) {   // one launch: the last workgroup writes the result record
        le = launch_mid_sweep(c, c.d_cand, acq, sf, incumbent, param, mu ? o_mu : nullptr, sigma ? o_sg : nullptr,
                              acq_out ? o_aq : nullptr, o_res, bell);
    } else {
        le = launch_small_sweep(c, c.d_cand, acq, sf, incumbent, param, mu ? o_mu : nullptr,
                                sigma ? o_sg : nullptr, acq_out ? o_aq : nullptr);
        if (le == hipSuccess) le = launch_argmax_final(c, acq != TGP_ACQ_NONE ? (long)((M + NB - 1) / NB) : 0L, o_res, bell);
    }
    if (le != hipSuccess) return hip_fail(c, le, "launch_small_sweep");
    if (bell.word) {
        const int wrc = bell_wait(c, bell, "evaluate sync");
        if (wrc != TGP_OK) return wrc;
        c.last_sweep_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    } else {
        API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
        API_HIP(hipStreamSynchronize(c.stream), "evaluate sync");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
        c.last_sweep_ms = ms;
    }
    const double *res = c.h_pin_out;
    if (mu) memcpy(mu, res + 8, (size_t)M * sizeof(double));
    if (sigma) memcpy(sigma, res + 8 + M, (size_t)M * sizeof(double));
    if (acq_out) memcpy(acq_out, res + 8 + 2 * M, (size_t)M * sizeof(double));
    if (acq != TGP_ACQ_NONE) {
        if (best_val) *best_val = res[0];
        if (best_idx) *best_idx = (int64_t)res[1];
    }
    if (n_clamped) *n_clamped = (int64_t)res[2];
    return TGP_OK;
} TGP_CATCH

int tgp_predict_batch(tgp_handle h, int64_t T, const int64_t *Ns, int64_t D, const double *const *Xs,
                      const double *const *ys, int kernel, const double *constants, const double *ls,
                      const double *noises, const double *jitters, int normalize_y, const double *Xc,
                      int64_t M, double *mu, double *sigma, double *lml, int64_t *n_clamped) try {
    if (!h) return TGP_BAD_ARG;
    HOST_NA("tgp_predict_batch");
    Context &c = h->c;
    if (!Ns || !Xs || !ys || !constants || !ls || !noises || !jitters || !Xc || !mu)
        return fail(c, TGP_BAD_ARG, "tgp_predict_batch: NULL argument");
    if (T < 1 || T > 4096 || D < 1 || D > 4096 || M < 1)
        return fail(c, TGP_BAD_ARG, "tgp_predict_batch: need 1 <= T <= 4096, 1 <= D <= 4096, M >= 1");
    if (kernel < TGP_RBF || kernel > TGP_MATERN52) return fail(c, TGP_BAD_ARG, "tgp_predict_batch: unknown kernel");
    int64_t nmax = 0;
    for (int64_t t = 0; t < T; ++t) {
        if (Ns[t] < 1 || Ns[t] > 4 * NB) return fail(c, TGP_BAD_ARG, "tgp_predict_batch: every model needs 1 <= N <= 256 (use tgp_fit + tgp_evaluate per model beyond that)");
        nmax = std::max(nmax, Ns[t]);
        if (!Xs[t] || !ys[t]) return fail(c, TGP_BAD_ARG, "tgp_predict_batch: NULL model data");
        if (!(constants[t] > 0.0) || !(noises[t] >= 0.0) || !(jitters[t] >= 0.0)) return fail(c, TGP_BAD_ARG, "tgp_predict_batch: constant > 0, noise >= 0, jitter >= 0 required");
        for (int64_t d = 0; d < D; ++d)
            if (!(ls[t * D + d] > 0.0)) return fail(c, TGP_BAD_ARG, "tgp_predict_batch: length scales must be > 0");
    }
    API_HIP(hipSetDevice(c.device), "hipSetDevice");
    API_HIP(pre_join(c), "hipStreamWaitEvent");
    // models of up to 128 points: the small-problem kernels (leading dimension 128); a batch with a larger model:
    // the one-workgroup fit and the one-launch sweep of 128 < N <= 256 for every model of it (leading dimension 256)
    const bool mid = nmax > 2 * NB;
    const int64_t Dp = ((D + 3) / 4) * 4, NPB = mid ? 4 * NB : 2 * NB;
    const int64_t in_stride = NPB * Dp + NPB + D + (D & 1);                 // doubles per model in the pinned input
    const size_t fa = small_fit_args_bytes(), sa = small_sweep_args_bytes();
    const size_t args_off = (size_t)(T * in_stride) * sizeof(double);
    const size_t in_bytes = args_off + (size_t)T * (fa + sa) + 64;
    int rc = ensure_pinned(c, in_bytes, (size_t)(3 * T + 8) * sizeof(double));
    if (rc != TGP_OK) return rc;
    // device: per-model workspaces | counters (4 T long long) | mu (T M) | sigma (T M) | candidates (M D)
    const int64_t wsd = mid ? mid_batch_ws_doubles(D, Dp) : small_batch_ws_doubles(D, Dp);
    const size_t dev_need = (size_t)(T * wsd + 4 * T + 2 * T * M + M * D) * sizeof(double);
    if ((rc = grow(c, c.d_batch, c.cap_batch, dev_need, "hipMalloc batch workspace")) != TGP_OK) return rc;
    double *d_ws = c.d_batch;
    long long *d_cnt = reinterpret_cast<long long *>(d_ws + T * wsd);
    double *d_mu = d_ws + T * wsd + 4 * T, *d_sg = d_mu + T * M, *d_xc = d_sg + T * M;

    char *pin = reinterpret_cast<char *>(c.h_pin_in);
    char *pin_dev = reinterpret_cast<char *>(c.d_pin_in);
    void *fit_args = pin + args_off, *sweep_args = pin + args_off + (size_t)T * fa;
    std::vector<double> ymean((size_t)T), ystd((size_t)T), yn((size_t)NPB);
    for (int64_t t = 0; t < T; ++t) {
        const int64_t N = Ns[t], Nin = ((N + NB - 1) / NB) * NB;
        double *in = c.h_pin_in + t * in_stride;
        memset(in, 0, (size_t)(Nin * Dp + Nin) * sizeof(double));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t d = 0; d < D; ++d) in[(size_t)i * Dp + d] = Xs[t][(size_t)i * D + d] / ls[t * D + d];
        std::fill(yn.begin(), yn.end(), 0.0);
        normalise_targets(ys[t], N, normalize_y, yn, ymean[(size_t)t], ystd[(size_t)t]);
        memcpy(in + Nin * Dp, yn.data(), (size_t)N * sizeof(double));
        memcpy(in + Nin * Dp + Nin, ls + t * D, (size_t)D * sizeof(double));
        (mid ? fill_mid_batch_args : fill_small_batch_args)(
            fit_args, sweep_args, t, c.d_pin_in + t * in_stride, d_ws + t * wsd, c.d_pin_out + 8 + 3 * t, d_cnt + 4 * t,
            d_xc, d_mu + t * M, sigma ? d_sg + t * M : nullptr, N, D, Dp, M, constants[t], noises[t], jitters[t],
            ymean[(size_t)t], ystd[(size_t)t]);
    }
    API_HIP(hipEventRecord(c.ev0, c.stream), "hipEventRecord");
    API_HIP(hipMemsetAsync(d_cnt, 0, (size_t)(4 * T) * sizeof(long long), c.stream), "memset counters");
    API_HIP(hipMemcpyAsync(d_xc, Xc, (size_t)(M * D) * sizeof(double), hipMemcpyHostToDevice, c.stream), "H2D points");
    hipError_t le = mid ? launch_mid_batch(c, kernel, pin_dev + args_off, pin_dev + args_off + (size_t)T * fa, T, M)
                        : launch_small_batch(c, kernel, pin_dev + args_off, pin_dev + args_off + (size_t)T * fa, T, M, true, true);
    if (le != hipSuccess) return hip_fail(c, le, "launch_small_batch");
    std::vector<long long> cnt((size_t)(4 * T));
    API_HIP(hipMemcpyAsync(mu, d_mu, (size_t)(T * M) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H mu");
    if (sigma) API_HIP(hipMemcpyAsync(sigma, d_sg, (size_t)(T * M) * sizeof(double), hipMemcpyDeviceToHost, c.stream), "D2H sigma");
    API_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)(4 * T) * sizeof(long long), hipMemcpyDeviceToHost, c.stream), "D2H counters");
    API_HIP(hipEventRecord(c.ev1, c.stream), "hipEventRecord");
    API_HIP(hipStreamSynchronize(c.stream), "batch sync");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c.ev0, c.ev1);
    c.last_sweep_ms = ms;
    int64_t clamped = 0;
    for (int64_t t = 0; t < T; ++t) {
        const double *res = c.h_pin_out + 8 + 3 * t;
        if (res[2] != 0.0) {
            char buf[200];
            snprintf(buf, sizeof buf, "model %lld of the batch: kernel matrix is not positive definite (pivot %d of %lld <= 0)",
                     (long long)t, (int)res[2] - 1, (long long)Ns[t]);
            return fail(c, TGP_NOT_PD, buf);
        }
        if (lml) lml[t] = -0.5 * res[1] - res[0] - (double)Ns[t] / 2.0 * log(2.0 * M_PI);
        clamped += (int64_t)cnt[(size_t)(4 * t + 1)];
    }
    if (n_clamped) *n_clamped = clamped;
    return TGP_OK;
} TGP_CATCH

int tgp_predict(tgp_handle h, const double *Xc, int64_t M, double *mu, double *sigma) try {
    return tgp_evaluate(h, Xc, M, TGP_ACQ_NONE, 1.0, 0.0, 0.0, mu, sigma, nullptr, nullptr, nullptr, nullptr);
} TGP_CATCH

int tgp_profile_enable(tgp_handle h, int on) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return TGP_OK;
    h->c.profiling = on != 0;
    return TGP_OK;
} TGP_CATCH

int tgp_profile_reset(tgp_handle h) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) return TGP_OK;
    Context &c = h->c;
    prof_collect(c);
    c.trmm_launches = c.kstar_launches = 0;
    c.trmm_ms = c.kstar_ms = 0.0;
    c.trmm_flops = 0.0;
    return TGP_OK;
} TGP_CATCH

int tgp_profile_read(tgp_handle h, int64_t *trmm_launches, double *trmm_ms, int64_t *kstar_launches,
                     double *kstar_ms, double *last_fit_ms, double *last_sweep_ms) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) {
        if (trmm_launches) *trmm_launches = 0;
        if (trmm_ms) *trmm_ms = 0.0;
        if (kstar_launches) *kstar_launches = 0;
        if (kstar_ms) *kstar_ms = 0.0;
        if (last_fit_ms) *last_fit_ms = h->host->last_fit_ms;
        if (last_sweep_ms) *last_sweep_ms = h->host->last_sweep_ms;
        return TGP_OK;
    }
    Context &c = h->c;
    prof_collect(c);
    if (trmm_launches) *trmm_launches = c.trmm_launches;
    if (trmm_ms) *trmm_ms = c.trmm_ms;
    if (kstar_launches) *kstar_launches = c.kstar_launches;
    if (kstar_ms) *kstar_ms = c.kstar_ms;
    if (last_fit_ms) *last_fit_ms = c.last_fit_ms;
    if (last_sweep_ms) *last_sweep_ms = c.last_sweep_ms;
    return TGP_OK;
} TGP_CATCH

int tgp_last_timings(tgp_handle h, double *out, int64_t n) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) {
        if (!out || n < 1) { h->host->err = "tgp_last_timings: need out and n >= 1"; return TGP_BAD_ARG; }
        for (int64_t i = 0; i < n; ++i) out[i] = i == 0 ? h->host->last_fit_ms : (i == 1 ? h->host->last_sweep_ms : (i == 6 ? 1.0 : 0.0));   // [6]: the host backend is float64 throughout
        return TGP_OK;
    }
    Context &c = h->c;
    if (!out || n < 1) return fail(c, TGP_BAD_ARG, "tgp_last_timings: need out and n >= 1");
    // [6] (round 6): 1 when the last sweep ran in float64 whatever the handle's dtype (the small-problem / one-launch
    // kernels), 0 when it ran in the handle's arithmetic, -1 before the first sweep
    // [7..11] (round 6, a debugging aid): phase stamps of the last POLLED small fit in microseconds from the kernel's start --
    // inputs staged, first kernel-matrix tile in LDS, first block factored, fit done, call done (0 when the call was not polled)
    double v[12] = {c.last_fit_ms, c.last_sweep_ms, c.last_grad_ms[0], c.last_grad_ms[1], c.last_grad_ms[2], c.trmm_flops,
                    (double)c.last_sweep_f64, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c.h_bell && c.h_bell[1]) {
        const unsigned long long t0 = c.h_bell[1];
        const int src[5] = {3, 4, 5, 6, 2};
        for (int k = 0; k < 5; ++k) v[7 + k] = c.h_bell[src[k]] >= t0 ? (double)(c.h_bell[src[k]] - t0) * 1e-2 : 0.0;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = i < 12 ? v[i] : 0.0;
    return TGP_OK;
} TGP_CATCH

int tgp_sweep_geometry(tgp_handle h, int64_t *chunk, int64_t *n_padded) try {
    if (!h) return TGP_BAD_ARG;
    if (h->host) { if (chunk) *chunk = 16; if (n_padded) *n_padded = h->host->N; return TGP_OK; }
    if (chunk) *chunk = h->c.chunk;
    if (n_padded) *n_padded = h->c.Np;
    return TGP_OK;
} TGP_CATCH

}  // extern "C"
