// tgp_internal.hpp -- state shared by the C-ABI layer and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/turbogp.h"
#include "doorbell.hpp"
#include "lds_opt_in.hpp"
#include "tuning.hpp"

namespace tgp {


constexpr int NB = 64;          // Cholesky / inverse block
constexpr int NPAD = 256;       // N is padded to a multiple of this (identity padding)
constexpr int SW_BM = 128;      // sweep tile: rows of Linv
constexpr int SW_BN = 128;      // sweep tile: candidates
constexpr int KS_JS = 8;        // most training-point splits of the cross-kernel grid (rows of mupart)
constexpr int FIN_BLOCK = 256;  // finalize block = candidates per arg-max partial

struct ProfSeg { int a, b, kind; double flops; };   // pooled events a -> b bracket one launch; kind: 0 trmm, 1 kstar; flops: the contraction's algorithmic flops of that launch (a launch a fit took row tiles of has fewer)

// The front of the resident batch's sweep, started INSIDE a fit (tgp_set_overlap; sweep_kernels.hip presweep_*)
struct PreSweep {
    int mode = 0;             // 0 off, 1 = candidate scaling + launch pair 0's cross-kernel, 2 = + the contraction's early row tiles
    int issue = 0;            // set by tgp_fit around launch_fit: the mode to issue for THIS fit (0: nothing)
    bool front = false;       // a fit issued the front for the batch recorded below
    int rows128 = 0;          // 128-row tiles [0, rows128) of launch pair 0 contracted inside that fit
    long gen = -1;            // fit_gen that fit leaves behind when it succeeds
    const double *cand = nullptr;
    int64_t M = 0, Mpad = 0, launch_rows = 0, chunk = 0;   // the batch and workspace geometry it was issued for
    bool pending = false;     // work on the device's third stream the main stream has not been told to wait for
    bool usable = false;      // set by tgp_sweep around launch_sweep: the front belongs to the resident fit, batch and geometry
    hipEvent_t ev = nullptr;     // third stream: end of the front
    hipEvent_t ev_in = nullptr;  // main stream: Xs / length scales staged
};

struct Context {
    int device = 0;
    int dtype = TGP_F64;
    hipStream_t stream = nullptr;    // everything runs in order on this stream (the device's shared main stream: not owned) ...
    unsigned long long *d_stamp = nullptr;   // TGP_STAMP_FILE (debug): in-kernel time stamps of the panel chain
    hipStream_t stream_own = nullptr; // tgp_set_private_stream: this handle's own main stream (owned), else null
    hipStream_t bg_lease = nullptr;   // the background stream this fit of a private-stream handle holds on loan (private_fit_begin), else null
    hipStream_t stream_bg = nullptr; // ... except the inverse factor's GEMMs behind the panel chain (the device's shared background stream: not owned)
    hipStream_t stream_pre = nullptr; // ... and the front of the next sweep inside a fit (the device's shared third stream: not owned)
    PreSweep pre;
    std::vector<hipEvent_t> ev_la;   // the events that order the two (no timing)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // brackets of the last fit / sweep (last_*_ms)
    hipEvent_t evg[4] = {nullptr, nullptr, nullptr, nullptr};   // stages of the last LML gradient
    double last_grad_ms[3] = {0.0, 0.0, 0.0};  // K^-1 = U U^T | pairwise weights + traces | ARD products
    std::string err;

    // ---- fitted state (device, f64) ----
    bool fitted = false;
    int64_t N = 0, D = 0, Np = 0;
    int64_t Dp = 0;                // D rounded up to a multiple of 4: row stride of Xs / Cs
    int kernel = TGP_RBF;
    double constant = 1.0, noise = 0.0, jitter = 0.0;
    double y_mean = 0.0, y_std = 1.0, lml = 0.0;
    std::vector<double> ls;        // D entries (broadcast when isotropic)
    std::vector<double> h_X;       // host copy of the training inputs (N, D): prefix test of tgp_fit_append
    std::vector<double> h_y;       // host copy of the raw targets (N,): tgp_export_state
    int normalize_y = 1;
    double sumlog = 0.0;           // sum(log(diag L)) of the resident factor
    double *d_t1 = nullptr, *d_t2 = nullptr;   // (Np,) scratch vectors of the row append
    double *d_Xs = nullptr;        // (Np, Dp) X / ls, rows >= N and columns >= D zero
    double *d_ls = nullptr;        // (D,)
    double *d_K = nullptr;         // (Np, Np) K, then L in the lower triangle
    double *d_Linv = nullptr;      // (Np, Np) L^-1, zeros above the diagonal
    double *d_W = nullptr;         // (Np, Np) workspace of the triangular inverse (T^T above the block diagonal)
    double *d_U = nullptr;         // (Np, Np) Linv^T (upper triangular), so every merge product is NT
    double *d_Dinv = nullptr;      // (2, Np/NB, NB, NB): inverses of the diagonal blocks | the diagonal blocks of L while they wait to be written into K
    double *d_Apan = nullptr;      // (2, Np/NB, NB, NB): the unsolved blocks of the current panel, two slots used in turn (fused_panel_kernel)
    double *d_yn = nullptr;        // (Np,) normalised y
    double *d_z = nullptr;         // (Np,) Linv * yn
    double *d_alpha = nullptr;     // (Np,)
    double *d_apart = nullptr;     // (Np/128, Np) shares of alpha = Linv^T z, one row per 128-row slice of Linv, then (Np/128,) sums of z^2
    double *d_scal = nullptr;      // [0] sum log diag, [1] yn . alpha
    int *d_flag = nullptr;         // first failing pivot + 1, or 0
    // LML-gradient workspace (allocated on first tgp_fit_grad)
    double *d_gpart = nullptr;     // (tiles, 3 + Dp) partial sums: [S_c, S_iso, S_diag, gd[0..Dp)] per 64 x 64 tile
    bool grad_staged = false;      // the last tgp_fit_grad left its sums in the pinned result buffer (+8), not in d_gout
    bool grad_timed = true;        // ... and recorded the events its stage times are read from (false: a polled call)
    double *d_gout = nullptr;      // [S_c, S_iso, S_diag, gd[Dp]]
    int64_t g_cap_Np = 0, g_cap_Dp = 0;
    double *d_qws = nullptr;       // small-batch query workspace (tgp_acq_grad)
    int64_t qws_cap = 0;
    double *d_rf = nullptr;        // on-device optimiser state (tgp_acq_refine)
    size_t cap_rf = 0;
    double *d_batch = nullptr;     // tgp_predict_batch: per-model workspaces, outputs, counters
    size_t cap_batch = 0;
    double *d_topv = nullptr;      // top-k workspace (tgp_sweep_topk)
    long long *d_topi = nullptr;
    size_t cap_topv = 0, cap_topi = 0;
    float *d_Xs32 = nullptr;       // f32 copies for the f32 sweep
    unsigned short *d_Linv16 = nullptr;   // TGP_F32X3: Linv32 as three bf16 planes (3, Np, Np), cut before the first sweep after a fit
    long fit_gen = 0, linv16_gen = -1;    // which fit the planes belong to
    float linv16_sb = 0.f;                // TGP_F32H2: the cross-kernel scale the stored 1 / (s_a s_b) was formed with
    unsigned *d_x2scal = nullptr;         // TGP_F32H2: [bits of max|Linv32|, bits of 1 / (s_a s_b)]
    float *d_Linv32 = nullptr;
    int64_t cap_Np = 0, cap_D = 0;
    bool cap_full = false;         // the buffers include what a FIT needs (K, the inverse's workspaces), not only what a sweep needs
    bool imported = false;         // the resident factor was received (tgp_import_factor_dev), not computed: no training set on the host
    int64_t import_rows = 0;       // rows of Linv received so far of a factor that is arriving block by block
    int64_t fit_gen_src = -1;      // ... and the giver's fit generation they belong to

    // ---- small-problem path (N <= 128): pinned, device-mapped staging ----
    bool small = false;            // the resident fit came from small_fit_kernel
    int64_t linv_extent = 0;       // rows / columns of d_Linv from this on are zero ...
    int64_t linv_ld = 0;           // ... for this leading dimension (0: unknown -> clear everything)
    double *h_pin_in = nullptr, *d_pin_in = nullptr;     // host / device view of the input staging
    double *h_pin_out = nullptr, *d_pin_out = nullptr;   // ... of the result staging
    size_t pin_in_cap = 0, pin_out_cap = 0;              // bytes
    double *h_pin_cand = nullptr, *d_pin_cand = nullptr; // candidates handed over by tgp_evaluate on that path
    size_t pin_cand_cap = 0;
    uint32_t *h_mt_words = nullptr;                      // tgp_set_candidates_mt19937: two pinned column buffers of the stream's words
    size_t mt_words_cap = 0;                             // bytes
    hipEvent_t ev_mt[2] = {nullptr, nullptr};            // ... and "this buffer's copy has left" (no timing)
    // ---- polled completion (doorbell.hpp): coherent device-mapped [sequence number, start tick, end tick, -] ----
    unsigned long long *h_bell = nullptr, *d_bell = nullptr;
    unsigned long long bell_seq = 0;   // number of the last polled call issued on this handle
    unsigned *d_ticket = nullptr;      // ticket counters of the polled multi-workgroup kernels (zero between launches)
    double *d_sfg = nullptr;           // small fit + gradient in one launch: the workspaces of the two workgroups that only contribute a block pair
    size_t cap_sfg = 0;

    // ---- candidates ----
    const double *d_cand = nullptr;   // (M, D) f64 row-major
    double *d_cand_owned = nullptr;
    int64_t cand_cap = 0;             // elements owned
    int64_t M = 0;

    // ---- sweep workspace ----
    int64_t chunk = 0;            // candidates per GROUP of a trmm launch (the slab the caches re-serve: about 256 MiB)
    int64_t launch_rows = 0;      // candidates per trmm launch = rows of the slab (a multiple of chunk; the whole batch when it fits)
    void *d_Cs = nullptr;                       // (Mpad, Dp) scaled candidates, compute dtype
    void *d_Ks[2] = {nullptr, nullptr};         // (chunk, Np) cross-kernel slab, two slots
    double *d_part = nullptr;                   // (Np/SW_BM, Mpad) partial ||v||^2
    double *d_mupart = nullptr;                 // (KS_JS, Mpad) partial K*.alpha
    int64_t ws_Mpad = 0;                        // leading dimension of Cs / part / mupart for this sweep
    size_t cap_Cs = 0, cap_Ks[2] = {0, 0}, cap_part = 0, cap_mupart = 0, cap_bval = 0, cap_bidx = 0;   // bytes
    double *d_mu = nullptr, *d_sigma = nullptr, *d_acq = nullptr;   // (M,) optional outputs
    int64_t out_cap = 0;
    double *d_bval = nullptr;     // per finalize block arg-max value
    long long *d_bidx = nullptr;  // per finalize block arg-max index
    double *sweep_res_host = nullptr;   // set by tgp_sweep around launch_sweep: device-mapped [best value, best index, clamp count] the sweep's last kernel fills (no D2H copy, no memset behind it), or null
    double *d_winner = nullptr;   // borrowed (D + 2) record [value, global index, row] or null (tgp_set_winner_out)
    int64_t winner_offset = 0;    // global index of candidate 0 of the resident batch
    hipEvent_t ev_winner = nullptr;   // recorded on the stream behind the kernel that packs the record (tgp_winner_wait)
    bool winner_recorded = false;
    double *d_best = nullptr;     // [0] value
    long long *d_besti = nullptr; // [0] index, [1] clamp count, [2] ticket counter of mid_sweep_kernel (zero between launches), [3] spare

    // ---- profiling ----
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;   // timing events, created once and reused (no create / destroy in the sweep loop)
    size_t ev_used = 0;
    std::vector<ProfSeg> segs;
    int64_t trmm_launches = 0, kstar_launches = 0;
    double trmm_ms = 0.0, kstar_ms = 0.0;
    double trmm_flops = 0.0;      // algorithmic flops (rows^2 per candidate over the rows a launch covered) of the timed contraction launches
    double last_fit_ms = 0.0, last_sweep_ms = 0.0;
    int last_sweep_f64 = -1;      // 1: the last sweep ran in f64 whatever the dtype (one-workgroup / one-launch kernels), 0: in the handle's arithmetic, -1: none yet
};

// launchers (fit_kernels.hip / sweep_kernels.hip); return hipSuccess or the first error
hipStream_t private_fit_begin(int device, bool may_borrow);   // fit_kernels.hip: a fit of a private-stream handle starts; the device's background stream if it gets it on loan, else null
void private_fit_end(int device, bool held);
hipError_t launch_fit(Context &c, const double *staged_in, double *res_host, bool zero_linv = true, unsigned long long *start_stamp = nullptr);   // start_stamp: device view of a mapped word the first kernel leaves its wall_clock64() in (a polled call)
hipError_t launch_ring(Context &c, const Bell &bell);   // a one-wave kernel behind everything queued on c.stream: rings the doorbell   // staged_in: device-mapped [Xs | yn | ls] or null (already in HBM); res_host: device-mapped [sum log, yn.alpha, flag] or null; zero_linv: false when Linv is known to be zero above the diagonal and from row Nr on
// the device's shared main / background stream (fit_kernels.hip); either pointer may be null
hipError_t device_streams(int device, hipStream_t *main, hipStream_t *bg, hipStream_t *pre = nullptr);   // main != null takes a reference
void device_streams_release(int device);
void device_stream_status(int device, int *bg_ok, int *pre_ok);   // 1 runs beside the main stream, 0 serialised (one hardware queue), -1 not probed
hipError_t launch_lml_grad(Context &c, bool ard, double *gout, bool timed = true);   // timed = false: no event records between its stages
hipError_t launch_f64_to_f32(Context &c, const double *in, float *out, long n);   // out[i] = (float)in[i] on c.stream (fit_kernels.hip: the cast every fit path uses)
// N <= 128, Dp <= 64, behind launch_small_fit: one workgroup per block pair (1 or 3), workgroup g leaving
// [S_c, S_iso, S_diag, gd[0..Dp)] for its pair at out + g * SMALL_GRAD_OUT_STRIDE; the caller adds them
constexpr int SMALL_GRAD_OUT_STRIDE = 72;
hipError_t launch_small_grad(Context &c, bool ard, double *out);
hipError_t launch_query(Context &c, const double *d_Xq, int m, int acq, double sf, double incumbent,
                        double param, double *d_ws, double *d_val, double *d_grad,   // d_val == nullptr: the sums only
                        const Bell &bell = Bell{nullptr, 0, nullptr});   // bell.word != null: value + gradient formed by the reduction's last workgroup, which rings it (d_Xq / d_val / d_grad may then be device-mapped host memory)
int64_t query_ws_doubles(const Context &c);   // doubles of d_ws per query point
double *query_red(const Context &c, double *d_ws, int m);   // per query point [k.alpha, v.v, gm (D), gv (D)]
hipError_t launch_gen_candidates(Context &c, double *dst, int64_t M, unsigned long long seed,
                                 unsigned long long first_candidate, const double *d_lo,
                                 const double *d_hi);
hipError_t launch_gen_lhs(Context &c, double *dst, int64_t M, int64_t D, unsigned long long seed,
                          unsigned long long first_sample, unsigned long long n_total,
                          const double *d_lo, const double *d_hi);
// d_words: per column the 2 M tempered MT19937 outputs of its M draws; dst (M, D) = lo + range * u, NumPy's arithmetic (sweep_kernels.hip)
hipError_t launch_mt19937_columns(Context &c, const void *d_words, double *dst, int64_t M, const double *d_lo,
                                  const double *d_range);
hipError_t launch_fit_append(Context &c, int n_old);
// done_host != null (a polled call): the final pass also leaves [k values | k indices as doubles, -1 = none | clamp count]
// in that device-mapped host record, hands the sweep's counters back at zero and rings the bell
hipError_t launch_topk(Context &c, const double *d_vals, long M, int k, double *ws_v, long long *ws_i,
                       long *final_off, double *done_host = nullptr, const Bell &bell = Bell{nullptr, 0, nullptr});
hipError_t launch_refine_clip(Context &c, double *d_xt, const double *d_lo, const double *d_hi, int R);
hipError_t launch_refine_step(Context &c, double *d_state, double *d_xt, const double *d_val,
                              const double *d_grad, const double *d_lo, const double *d_hi, int R,
                              int it, double pgtol, double ftol, int *d_active,   // d_active: 2 ints, used in turn
                              const double *d_red = nullptr, int acq = 0, double sf = 1.0, double incumbent = 0.0,
                              double param = 0.0);   // d_red (D <= 64 only): value + gradient taken from launch_query's sums here
hipError_t launch_refine_collect(Context &c, const double *d_state, int R, double *d_x, double *d_v, double *d_info);
long refine_state_stride(int D);
long refine_hist_doubles(int D, int R);   // behind the R states in the same buffer
// N <= 128, D <= 64: the whole stage in one launch, one workgroup per restart (refine_kernels.hip);
// d_info: (3 R) status, accepted steps, evaluations.  Pointers may be device-mapped host memory.
bool small_refine_fits(const Context &c);
hipError_t launch_small_refine(Context &c, const double *d_x0, const double *d_lo, const double *d_hi, int R,
                               int acq, double sf, double incumbent, double param, int max_iter,
                               double pgtol, double ftol, double *d_x, double *d_v, double *d_info);
// N <= 128, D <= 64 (small_refine_fits): tgp_acq_grad in one launch, one workgroup per point; the pointers may be
// device-mapped host memory; bell.word != null: the last workgroup rings it (refine_kernels.hip)
hipError_t launch_small_query(Context &c, const double *d_xq, int m, int acq, double sf, double incumbent, double param,
                              double *d_val, double *d_grad, const Bell &bell);
hipError_t launch_small_fit(Context &c, const Bell &bell);   // bell.word != null: the kernel rings it when the results are out (doorbell.hpp)
// N <= 128, Dp <= 64: fit + LML gradient in ONE launch (one workgroup per block pair, each running the fit itself);
// gout_host: device-mapped host memory for the pairs' sums (as launch_small_grad leaves them); needs c.d_sfg
hipError_t launch_small_fit_grad(Context &c, bool ard, double *gout_host, const Bell &bell);
size_t small_fit_grad_ws_bytes();
// N <= 128: the hyper-parameter fit in one launch, one workgroup per start (small_kernels.hip).  theta = log(constant,
// length scale(s), noise); d_ws: S * small_hyper_workspace_doubles() doubles; d_info (3 S): status, steps, evaluations
long small_hyper_workspace_doubles(int N, int D, int Dp);
hipError_t launch_small_hyper(Context &c, int kernel, const double *d_X, const double *d_yn, const double *d_theta0,
                              const double *d_blo, const double *d_bhi, int S, int N, int D, int Dp, int n_ls,
                              int max_iter, double jitter, double *d_ws, double *d_theta, double *d_f, double *d_info,
                              bool one_wg_per_start = false);
size_t small_fit_args_bytes();
size_t small_sweep_args_bytes();
int64_t small_batch_ws_doubles(int64_t D, int64_t Dp);
// ... and for batches with models of 128 < N <= 256 (one-workgroup fit with the tiles in the model's workspace, one-launch sweep)
int64_t mid_batch_ws_doubles(int64_t D, int64_t Dp);
void fill_mid_batch_args(void *fit_args, void *sweep_args, int64_t t, const double *in_dev, double *ws_dev,
                         double *res_dev, long long *counters_dev, const double *cand_dev, double *mu_dev,
                         double *sigma_dev, int64_t N, int64_t D, int64_t Dp, int64_t M, double constant,
                         double noise, double jitter, double y_mean, double y_std);
hipError_t launch_mid_batch(Context &c, int kernel, const void *fit_args_dev, const void *sweep_args_dev, int64_t T, int64_t M);
void fill_small_batch_args(void *fit_args, void *sweep_args, int64_t t, const double *in_dev, double *ws_dev,
                           double *res_dev, long long *counters_dev, const double *cand_dev, double *mu_dev,
                           double *sigma_dev, int64_t N, int64_t D, int64_t Dp, int64_t M, double constant,
                           double noise, double jitter, double y_mean, double y_std);
hipError_t launch_small_batch(Context &c, int kernel, const void *fit_args_dev, const void *sweep_args_dev,
                              int64_t T, int64_t M, bool fit, bool sweep);
hipError_t launch_argmax_final(Context &c, long nblk, double *res_host, const Bell &bell = Bell{nullptr, 0, nullptr});   // bell.word != null: rings when the record is out (the call's last kernel)
// 128 < N <= 512: the whole sweep -- cross-kernel tile, contraction, acquisition, arg-max, winner record -- in ONE launch
// (small_kernels.hip, mid_sweep_kernel); res_host: optional zero-copy record [best value, best index, clamp count].
// mid_sweep_cpw: candidates per workgroup (64 up to N = 256, 32 up to N = 512 and moderate batches), 0 = the general sweep
int mid_sweep_cpw(const Context &c, int64_t M);   // M: the batch about to be swept
hipError_t launch_mid_sweep(Context &c, const double *cand, int acq, double sf, double incumbent,
                            double param, double *mu, double *sigma, double *acqv, double *res_host,
                            const Bell &bell = Bell{nullptr, 0, nullptr});   // bell.word != null: the last workgroup rings when the record is out
hipError_t launch_small_sweep(Context &c, const double *cand, int acq, double sf, double incumbent,
                              double param, double *mu, double *sigma, double *acqv);
hipError_t launch_sweep(Context &c, int acq, double sf, double incumbent, double param,
                        bool want_mu, bool want_sigma, bool want_acq);
// inside launch_fit, on the third stream (c.pre.mode > 0, candidates resident, workspace ensured): the candidate scaling and
// launch pair 0's cross-kernel (needs Xs, the length scales) / the contraction's 128-row tiles whose rows of Linv are
// final (rows < rows_final), at most budget128 of them in all
hipError_t presweep_front(Context &c, hipStream_t st);
hipError_t presweep_rows(Context &c, hipStream_t st, int rows_final, int budget128);

// Profiling marks on the sweep's stream: ONE event between consecutive launches (the end of one
// launch is the start of the next), so a chunk costs two records instead of four.  prof_mark
// returns the event's index or -1 (profiling off); prof_seg names the launch between two marks.
int prof_mark(Context &c, hipStream_t s);
void prof_seg(Context &c, int a, int b, int kind, double flops = 0.0);

}  // namespace tgp
