/*
 * turbogp.h -- C-ABI of libturbogp.so: the MI355X-native GP-surrogate hot path of mbway/turbo.
 *
 * The reference has no FFI on this path: it is pure Python that calls scikit-learn / SciPy
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference call site
 * (file:line under /root/reference, or sklearn/... for the third-party code it delegates to)
 * whose work it replaces.  The Python plugin classes in turbo_amd/ bind these symbols with
 * ctypes (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, int status return, no torch / C++ types.
 *   - All host arrays are caller-owned, row-major float64.  The library copies on entry and
 *     never keeps a caller pointer past return, except the *_dev candidate entry which borrows a
 *     DEVICE pointer until the next tgp_set_candidates* / tgp_destroy.
 *   - One handle = one GPU.  Calls on one handle must not overlap; a handle may be used from a
 *     different host thread than the one that created it (every entry calls hipSetDevice).
 *     Every call is synchronous: it returns with its results on the host.  All handles on one
 *     device share one HIP stream pair (a main stream and a CU-masked background stream the fit
 *     overlaps its inverse factor on; created with the first handle, released with the last), so
 *     calls on DIFFERENT handles of one device may be issued from different threads but do not
 *     overlap on the GPU.  The library never touches the null stream.
 *   - dtype selects the arithmetic of the candidate sweep (cross-kernel + triangular
 *     contraction).  The fit (kernel matrix, Cholesky, inverse factor, alpha) is always f64.
 */
#ifndef TURBOGP_H
#define TURBOGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgp_handle_s *tgp_handle;

enum tgp_status {
    TGP_OK = 0,
    TGP_NOT_PD = 1,      /* kernel matrix not positive definite -> numpy.linalg.LinAlgError
                            (sklearn/gaussian_process/_gpr.py:348-358) */
    TGP_BAD_ARG = 2,     /* AssertionError / ValueError on the Python side */
    TGP_HIP_ERROR = 3,   /* RuntimeError; tgp_last_error() holds the HIP message */
    TGP_NOT_FITTED = 4,
    TGP_NO_MEMORY = 5,   /* host allocation failed -> MemoryError */
    TGP_NO_DEVICE = 6    /* tgp_create: no HIP device visible to this process */
};

/* TGP_F32X3 (opt-in): the sweep at f32 accuracy on the bf16 matrix pipe -- every f32 operand of the
 * triangular contraction split into three bf16 planes, six products accumulated in f32;
 * everything else as TGP_F32. */
/* TGP_F32H2 (opt-in): the same from two scaled fp16 planes and three products -- half the matrix
 * work and 4 bytes per element; same accuracy class. */
enum tgp_dtype { TGP_F64 = 0, TGP_F32 = 1, TGP_F32X3 = 2, TGP_F32H2 = 3 };

/* unit-amplitude stationary kernels: sklearn/gaussian_process/kernels.py RBF :1553-1565,
 * Matern nu=0.5/1.5/2.5 :1717-1724 */
enum tgp_kernel { TGP_RBF = 0, TGP_MATERN12 = 1, TGP_MATERN32 = 2, TGP_MATERN52 = 3 };

/* turbo/modules/acquisition_functions.py: UCB :147-158 (TGP_ACQ_SIGMA = its beta=inf branch),
 * PI :225-247, EI :336-358.  TGP_ACQ_NONE = predict only. */
enum tgp_acq { TGP_ACQ_NONE = 0, TGP_ACQ_UCB = 1, TGP_ACQ_PI = 2, TGP_ACQ_EI = 3, TGP_ACQ_SIGMA = 4 };

/* buffers readable through tgp_debug_read (parity tests only) */
enum tgp_buffer { TGP_BUF_K = 0, TGP_BUF_L = 1, TGP_BUF_LINV = 2, TGP_BUF_ALPHA = 3 };

/* ---- lifetime ------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (index among visible devices); TGP_NO_DEVICE when the
 * process sees no GPU.
 * device == TGP_DEVICE_HOST: a context that never calls HIP -- the RELOAD path.  The reference pickles
 * every trial's model (turbo/recorder.py:117-155) and the plot path queries the reloaded models in
 * whatever process loaded the recorder (turbo/recorder.py:157-163, turbo/plotting/trials.py:192-195,
 * :371, :448, :574-577), which need not own an MI355X.  A host handle serves tgp_fit, tgp_fit_append
 * (as a full fit), tgp_export_state / tgp_import_state (the same blob), tgp_debug_read (L, alpha),
 * tgp_set_candidates, tgp_read_candidates, tgp_get_candidate, tgp_sweep, tgp_evaluate, tgp_predict and
 * the timing queries, always in float64 (csrc/host_backend.cpp: plain C++, its own arithmetic -- not
 * the HIP kernels, not the test oracle); every other entry returns TGP_BAD_ARG on it. */
#define TGP_DEVICE_HOST (-1)
int tgp_create(int device, int dtype, tgp_handle *out);
int tgp_destroy(tgp_handle h);
/* Message of the last failing call on this handle (or of tgp_create when h == NULL). */
const char *tgp_last_error(tgp_handle h);
/* "turbogp <version> gfx950" */
const char *tgp_version(void);

/* ---- fit ------------------------------------------------------------------------------- */

/* Fixed-hyper-parameter GP fit.  Replaces SciKitGPSurrogate.construct_model -> model.fit
 * (turbo/modules/surrogates.py:294-326, :318) i.e. sklearn _gpr.py:272-282 (y normalisation),
 * :346-347 (K = c*k(X,X) + noise*I, K.diag += jitter), :349 (lower Cholesky), :360-364 (alpha)
 * and :584-613 (log marginal likelihood, returned through `lml`).
 *   X   (N, D) row-major, y (N,)
 *   ls  length scale(s): n_ls == 1 (isotropic) or n_ls == D (ARD)
 * Outputs (nullable): lml, y_mean, y_std.
 * Returns TGP_NOT_PD when a pivot is <= 0 or not finite. */
int tgp_fit(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y,
            int kernel, double constant, const double *ls, int64_t n_ls,
            double noise, double jitter, int normalize_y,
            double *lml, double *y_mean, double *y_std);

/* Fit (exactly as tgp_fit) plus the gradient of the log marginal likelihood with respect to the
 * LOG hyper-parameters, `grad` = [d/dlog(constant), d/dlog(ls[0..n_ls)), d/dlog(noise)]
 * (2 + n_ls doubles; the noise entry is 0 when noise == 0).  Replaces
 * log_marginal_likelihood(theta, eval_gradient=True) (sklearn _gpr.py:537-652: K^-1 by
 * cho_solve(L, I) and 0.5*einsum("ijl,jik->kl", alpha alpha^T - K^-1, K_gradient)) that the
 * optimiser loop of GaussianProcessRegressor.fit (:296-337) evaluates per L-BFGS-B step, reached
 * from turbo/modules/surrogates.py:313-318 whenever training_iterations > 0.  The model is left
 * fitted at these hyper-parameters. */
int tgp_fit_grad(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y,
                 int kernel, double constant, const double *ls, int64_t n_ls,
                 double noise, double jitter, int normalize_y,
                 double *lml, double *y_mean, double *y_std, double *grad);

/* Fit like tgp_fit, but when the handle already holds a fit with the same kernel, hyper-parameters,
 * jitter and normalisation whose training inputs are bit-for-bit the first N-1 rows of X, extend
 * the factor by ONE row in O(N^2) (four triangular GEMVs) instead of refactorising in O(N^3):
 *   l = Linv k,  lambda = sqrt(kappa - l.l),  L' = [[L,0],[l^T,lambda]],
 *   Linv' = [[Linv,0],[-(Linv^T l)^T/lambda, 1/lambda]],  alpha' = Linv'^T Linv' yn'.
 * That is the reference's loop: the Optimiser appends exactly one trial per iteration and refits
 * from scratch (turbo/optimiser.py:93-96, :335-336 -> surrogates.py:318).  Any mismatch (or a new
 * 256-row padding block) falls back to tgp_fit.  *appended (nullable) tells which path ran. */
int tgp_fit_append(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y,
                   int kernel, double constant, const double *ls, int64_t n_ls,
                   double noise, double jitter, int normalize_y,
                   double *lml, double *y_mean, double *y_std, int *appended);

/* Persistence (turbo/recorder.py:141-147 pickles every trial's model with dill; turbo/utils.py:72-90
 * loads them back, possibly in another process).  The blob holds what DEFINES the model -- kernel,
 * hyper-parameters, X, y: (8 + D + N*D + N) * 8 bytes -- not the O(N^2) factor, which import rebuilds
 * by running tgp_fit (milliseconds), so a Recorder that keeps one model per trial stays O(N*D) per
 * trial as with the reference's sklearn objects (which hold X_train_, y_train_, L_, alpha_).
 *   export: *size receives the bytes needed; buf == NULL only queries the size.
 *   import: status of the fit it runs (TGP_NOT_PD possible); lml nullable. */
int tgp_export_state(tgp_handle h, void *buf, int64_t cap, int64_t *size);
int tgp_import_state(tgp_handle h, const void *buf, int64_t size, double *lml);

/* ---- Handing a FACTOR over instead of recomputing it (round 6) -------------------------------------------------
 * SURVEY.md 8e's alternative to the replicated fit ("fit on GPU0 + broadcast of L, alpha"); what scikit-learn keeps as
 * L_ / alpha_ (sklearn/gaussian_process/_gpr.py:349,360), reached from turbo/modules/surrogates.py:318.
 * With the candidate batch sharded over G GPUs the fit is the part that does not shrink (C3, 8 shards: 2.2 of 6.4 ms).
 * A handle can instead RECEIVE what a sweep needs -- the scaled training points, the length scales, alpha, the scalars
 * and the inverse factor Linv -- from device memory: another handle's buffers in the same process (tgp_export_factor_dev),
 * or a buffer a collective filled (an RCCL broadcast into a torch tensor).  Only float64 state travels; an f32 / f32h2 /
 * f32x3 handle cuts its own copies with the casts its own fit uses, so a receiver's sweep returns the giver's bytes.
 *
 * tgp_export_factor_dev fills `out` with the giver's shapes, hyper-parameters, scalars and DEVICE pointers INTO its own
 * buffers: valid until the giver's next fit / import / destroy, never owned by the caller.
 * tgp_import_factor_dev copies rows [row0, row0 + rows) of Linv (leading dimension Np) into the handle, on its stream.
 * The rows must arrive in order -- top down, as a Cholesky finishes them (the inverse factor's row block i is final once
 * panel i is): row0 = 0 starts a new factor (shapes, scalars, Xs, ls, alpha are taken from `f` then), every later call
 * continues it, and the call that delivers row Np - 1 completes it: the handle is fitted from then on (TGP_NOT_FITTED
 * before).  One call with rows = Np imports the whole factor.  A handle that received its factor holds no training
 * set: tgp_export_state and tgp_debug_read(L) refuse, tgp_fit_append refits; everything that evaluates the model
 * (sweeps, tgp_acq_grad, the gradient stage) works.  Host handles: TGP_BAD_ARG. */
typedef struct tgp_factor {
    int64_t N, D, Np, Dp;        /* Np = N rounded up to 256, Dp = D rounded up to 4 */
    int64_t fit_gen;             /* the giver's fit generation: the blocks of one import must all carry the same */
    int32_t kernel, normalize_y; /* tgp_kernel; as given to tgp_fit */
    int32_t small_path, reserved;/* 1: the giver's fit came from the one-workgroup kernels (N <= 128): the receiver sweeps with them too */
    double constant, noise, jitter, y_mean, y_std, lml, sumlog;
    const void *Xs;              /* (Np, Dp) f64: X / length_scale, padded rows and columns zero */
    const void *ls;              /* (D) f64 */
    const void *alpha;           /* (Np) f64 */
    const void *Linv;            /* (Np, Np) f64 row-major: L^-1, zeros above the diagonal */
} tgp_factor;
int tgp_export_factor_dev(tgp_handle h, tgp_factor *out);
int tgp_import_factor_dev(tgp_handle h, const tgp_factor *f, int64_t row0, int64_t rows);

/* Copy a fitted buffer to the host (tests): K / L / LINV are (N, N) row-major (L and LINV
 * lower-triangular with zeros above the diagonal), ALPHA is (N,). */
int tgp_debug_read(tgp_handle h, int which, double *out);

/* ---- candidates ------------------------------------------------------------------------- */

/* Upload an (M, D) float64 candidate batch; it stays resident in HBM for later sweeps.
 * Replaces the hand-over of `random_x` at turbo/modules/auxiliary_optimisers.py:60-61. */
int tgp_set_candidates(tgp_handle h, const double *Xc, int64_t M);
/* Borrow an (M, D) float64 row-major batch that already lives in this GPU's memory
 * (e.g. a torch tensor's data_ptr()).  The pointer is checked against the HIP runtime's records
 * (device memory, this GPU, at least M*D doubles left in its allocation) -> TGP_BAD_ARG. */
int tgp_set_candidates_dev(tgp_handle h, const void *Xc_dev, int64_t M);
/* Fill the resident batch with M uniform candidates drawn ON the GPU: x[d] = lo[d] + (hi[d] -
 * lo[d]) * u, u from Philox-4x32-10 keyed by `seed`; candidate i of this call is number
 * first_candidate + i of the stream, so shards of one batch on several GPUs are disjoint pieces
 * of the same stream.  Device-side counterpart of random_selector
 * (turbo/modules/naive_selectors.py:39-46; that one draws from the global NumPy RNG, so the
 * values differ by design -- opt-in). */
int tgp_gen_candidates(tgp_handle h, uint64_t seed, uint64_t first_candidate, int64_t M,
                       const double *lo, const double *hi);
/* Latin hypercube design on the GPU: samples first_sample .. first_sample + M - 1 of an
 * n_total-point design.  Sample i, dimension d:  lo_d + (hi_d - lo_d) * ((pi_d(i) + u_id) / n_total)
 * with pi_d a keyed pseudo-random permutation of the n_total strata (4-round Feistel network over
 * Philox-4x32-10, cycle-walked) and u_id uniform in [0, 1) from the Philox stream of `seed`: every
 * sample is computed on its own, so shards are rows of one design.  Device-side counterpart of
 * LHS_selector (turbo/modules/naive_selectors.py:58-83: arange + rand per stratum, then
 * np.random.permutation per column from the global NumPy RNG -- the values differ by design).
 *   tgp_gen_candidates_lhs  fills the resident candidate batch (needs a fitted model for D)
 *   tgp_lhs_design          hands an (M, D) design back to the host; needs no model (the reference
 *                           uses the selector for the pre-phase trials, before any fit) */
int tgp_gen_candidates_lhs(tgp_handle h, uint64_t seed, uint64_t first_sample, int64_t M,
                           uint64_t n_total, const double *lo, const double *hi);
int tgp_lhs_design(tgp_handle h, uint64_t seed, uint64_t first_sample, int64_t M, uint64_t n_total,
                   int64_t D, const double *lo, const double *hi, double *out);
/* Read rows first .. first + count - 1 of the resident batch back: (count, D). */
int tgp_read_candidates(tgp_handle h, int64_t first, int64_t count, double *out);
/* Read candidate row `idx` back (the argmax row: auxiliary_optimisers.py:64-65). */
int tgp_get_candidate(tgp_handle h, int64_t idx, double *out_row);

/* ---- sweep ------------------------------------------------------------------------------ */

/* Posterior mean / std and acquisition over the resident candidate batch, plus its argmax.
 * Replaces  acq(random_x)  ->  model.predict(X, return_std_dev=True)
 * (turbo/modules/acquisition_functions.py:152,230,341 -> surrogates.py:332-338 -> sklearn
 * _gpr.py:443-494) and the argsort/[0] of auxiliary_optimisers.py:63-66.
 *   acq        enum tgp_acq;  sf = +1 ('max') or -1 ('min');  incumbent = best raw y so far
 *   param      beta (UCB) or xi (PI / EI)
 *   mu, sigma, acq_out   nullable (M,) host outputs
 *   best_val, best_idx   nullable; argmax of the acquisition (of -inf if all NaN), lowest index
 *                        wins ties.  With TGP_ACQ_NONE they are left untouched.
 *   n_clamped  nullable; number of candidates whose variance was < 0 and clamped to 0
 *              (sklearn _gpr.py:479-485 warns once in that case) */
int tgp_sweep(tgp_handle h, int acq, double sf, double incumbent, double param,
              double *mu, double *sigma, double *acq_out,
              double *best_val, int64_t *best_idx, int64_t *n_clamped);

/* Sharded arg-max (the candidate batch is cut into contiguous shards, one per GPU,
 * and the per-shard winners are combined; the reference's own arg-max is the argsort/[0] of
 * turbo/modules/auxiliary_optimisers.py:63-66 over the whole batch).  Attach a DEVICE buffer of
 * D + 2 doubles on this GPU: every later tgp_sweep with an acquisition also packs
 *     [best value, (double)(global_offset + best index), candidate row (D)]
 * into it, on the device and before tgp_sweep returns, so the all-gather between GPUs (RCCL) can
 * read it in place -- no D2H of the row, no host-built tensor.  global_offset = global index of
 * candidate 0 of this handle's shard.  The record is written on the library's stream and complete
 * when tgp_sweep returns (the call synchronises that stream), which is what lets another stream --
 * RCCL on the caller's -- read it without an event.  The buffer is borrowed (checked like
 * tgp_set_candidates_dev) until replaced, detached with rec_dev == NULL, a fit with another D,
 * or tgp_destroy. */
int tgp_set_winner_out(tgp_handle h, void *rec_dev, int64_t global_offset);
/* (round 6) Make `stream` -- a hipStream_t of the CALLER on this handle's device, passed as a plain pointer; NULL is
 * the legacy default stream -- wait for the winner record of the last tgp_sweep: the library records an event on its
 * own stream right behind the kernel that packs the record, and this call is one hipStreamWaitEvent on it.  It makes
 * the cross-stream ordering of the multi-GPU exchange explicit (the record is written on the library's stream and
 * read by RCCL on the caller's, turbo_amd/distributed.py): correct today because tgp_sweep returns only after a
 * synchronisation of its stream, and correct tomorrow if a sweep ever returns earlier.  No sweep with a record
 * yet: nothing to wait for, TGP_OK.  No record attached: TGP_BAD_ARG. */
int tgp_winner_wait(tgp_handle h, void *stream);

/* Acquisition value AND gradient with respect to the query point for a small batch (m <= 4096)
 * of host points Xq (m, D): val (m,), grad (m, D).  TGP_ACQ_NONE returns the posterior mean and its
 * gradient.  Serves the gradient stage of the auxiliary optimiser
 * (turbo/modules/auxiliary_optimisers.py:69-112, which differentiates 1-point acq calls by finite
 * differences); always float64, independent of the handle's sweep dtype. */
int tgp_acq_grad(tgp_handle h, const double *Xq, int64_t m, int acq, double sf, double incumbent,
                 double param, double *val, double *grad);

/* The sweep of tgp_sweep, returning the k <= 64 BEST candidates instead of the single best:
 * vals[0..k) descending, idxs[0..k) their indices (lowest index first among equal values, NaN
 * last, -1 when the batch holds fewer than k candidates).  This is
 * `best_ids = np.argsort(random_y)[:start_from_best]` of the reference's gradient stage
 * (turbo/modules/auxiliary_optimisers.py:63-66, :77-79) without the (M,) acquisition vector leaving
 * the GPU. */
int tgp_sweep_topk(tgp_handle h, int acq, double sf, double incumbent, double param, int64_t k,
                   double *vals, int64_t *idxs, int64_t *n_clamped);

/* The hyper-parameter fit inside the library: S <= 64 starts theta0 (S, P), theta = log(constant, length
 * scale(s), noise) with P = 2 + n_ls (n_ls = 1 or D), are each optimised to a local maximum of the log
 * marginal likelihood inside [log_lo, log_hi].  What GaussianProcessRegressor.fit does with
 * optimizer='fmin_l_bfgs_b' and n_restarts_optimizer = S - 1 (sklearn _gpr.py:296-337, :654-670, reached
 * from turbo/modules/surrogates.py:313-318), the restarts side by side.
 *
 * tgp_fit_lbfgsb: every start is walked by L-BFGS-B itself (csrc/host_lbfgsb.hpp: the algorithm behind
 *   scipy.optimize.minimize(method='L-BFGS-B') restated -- generalised Cauchy point, subspace minimisation,
 *   More-Thuente line search, 10 pairs, SciPy's defaults pgtol 1e-5 and factr 1e7): the iterates, the number
 *   of evaluations and the optimum are the ones scikit-learn's fit arrives at on the same objective (to the
 *   rounding of the objective).  A host thread and a worker handle (tgp_workers_acquire) per start drive
 *   tgp_fit_grad from inside the library -- no interpreter between two evaluations.  With more than one start
 *   the caller's handle is not touched at all; with a single start (or one thread) it runs the evaluations
 *   itself and is left holding the LAST evaluation of the LAST start -- not a model to use.
 * tgp_fit_optimise: the library's choice of optimiser.
 *   N <= 128, D <= 64:  ONE launch, a workgroup per start (kernel matrix, Cholesky, inverse factor, alpha,
 *       LML, its gradient and one step of a projected L-BFGS per iteration, with L-BFGS-B's stopping rules):
 *       no host round trip per evaluation -- other iterates than SciPy's, the same or a better optimum.  The
 *       handle's fitted model is left untouched.
 *   larger problems:    tgp_fit_lbfgsb.
 * An entry with log_lo == log_hi is a FIXED hyper-parameter: tgp_fit_lbfgsb leaves it out of the optimiser's vector
 * (as scikit-learn leaves a "fixed" hyper-parameter out of theta; theta0's value there is replaced by the bound), the
 * one-launch path treats it as a coordinate that cannot move.  The noise entry may be fixed at -INFINITY: a kernel
 * without a noise term (noise = 0; such a call always takes the tgp_fit_lbfgsb route).
 * Either way the caller picks the best start and fits it with tgp_fit.
 *   max_iter: accepted iterations per start (SciPy's maxiter); line-search trials do not count against it
 *       (they have SciPy's maxfun = 15000 of their own).
 *   theta_out (S, P), f_out (S) = -LML at theta_out
 *   status_out (S, nullable): 1 converged, 0 stopped by max_iter (SciPy's status 1), 2 no acceptable step
 *       from the last iterate (SciPy's status 2, ABNORMAL_TERMINATION_IN_LNSRCH; theta_out is that iterate)
 *   evaluations (nullable): LML + gradient evaluations over all starts */
int tgp_fit_lbfgsb(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                   const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                   double jitter, int normalize_y, int64_t max_iter, double *theta_out, double *f_out,
                   int64_t *status_out, int64_t *evaluations);
int tgp_fit_optimise(tgp_handle h, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                     const double *theta0, int64_t S, int64_t n_ls, const double *log_lo, const double *log_hi,
                     double jitter, int normalize_y, int64_t max_iter, double *theta_out, double *f_out,
                     int64_t *status_out, int64_t *evaluations);

/* The gradient stage itself on the device: R <= 4096 restarts X0 (R, D) are refined together by a
 * projected L-BFGS (memory 8; a line search that asks for sufficient decrease 1e-4 and the curvature
 * condition 0.9 as L-BFGS-B's does, lengthening a step whose slope is still steep; bounds lo / hi
 * per dimension; stopping rules of SciPy's L-BFGS-B defaults: projected gradient <= 1e-5 or
 * relative reduction <= 2.2e-9) that MAXIMISES the acquisition.  N <= 128 and D <= 64: ONE launch,
 * a workgroup per restart runs its whole optimisation with the model in LDS.  Larger models:
 * every iteration is one batched closed-form value + gradient evaluation (the kernels of
 * tgp_acq_grad) and one optimiser step for all restarts in lock-step, all resident on the GPU.
 * Replaces the loop of scipy.optimize.minimize(method='L-BFGS-B') runs over finite-difference
 * gradients at turbo/modules/auxiliary_optimisers.py:80-99.
 *   x_out (R, D), val_out (R): refined points and their acquisition values
 *   status_out (R, nullable): 1 converged, 2 no progress from the start point, 0 stopped by max_iter
 *   iterations (nullable): value + gradient evaluations made (by the slowest restart) */
int tgp_acq_refine(tgp_handle h, const double *X0, int64_t R, const double *lo, const double *hi,
                   int acq, double sf, double incumbent, double param, int64_t max_iter,
                   double *x_out, double *val_out, int64_t *status_out, int64_t *iterations);

/* The gradient stage as the reference runs it, inside the library: every restart X0 (R, D), R <= 4096, walked by
 * L-BFGS-B itself (csrc/host_lbfgsb.hpp, SciPy's algorithm and defaults: the optimiser behind
 * scipy.optimize.minimize(method='L-BFGS-B', options=dict(maxiter=15000)) at
 * turbo/modules/auxiliary_optimisers.py:80-99) on the negated acquisition with its closed-form gradient, the
 * restarts in lock-step: one batched value + gradient evaluation (the kernels of tgp_acq_grad) per round serves every
 * restart still running.  Each restart walks what it would walk alone -- a point's value does not depend on the batch
 * it is evaluated in.  Where tgp_acq_refine runs an optimiser of its own on the device (other iterates, one launch for
 * small models), this one reproduces SciPy's walk with no interpreter between two rounds.
 *   max_iter: accepted iterations per restart (SciPy's maxiter)
 *   x_out (R, D), val_out (R): the end points and their acquisition values
 *   status_out (R, nullable): 1 converged (SciPy's success), 0 stopped by max_iter, 2 no acceptable step
 *   evaluations (nullable): value + gradient evaluations over all restarts */
int tgp_acq_lbfgsb(tgp_handle h, const double *X0, int64_t R, const double *lo, const double *hi,
                   int acq, double sf, double incumbent, double param, int64_t max_iter,
                   double *x_out, double *val_out, int64_t *status_out, int64_t *evaluations);

/* One call = tgp_set_candidates + tgp_sweep: what ONE call of the reference's acquisition
 * instance does, acq(X) -> model.predict(X, return_std_dev=True) -> formula
 * (turbo/modules/acquisition_functions.py:152,230,341; surrogates.py:332-338), and what the plot
 * path repeats per stored model on 200 .. 10^4 points (turbo/plotting/trials.py:371,448,574-577).
 * Arguments as tgp_sweep; the batch stays resident afterwards.  For a model of N <= 256 and a batch
 * of up to 8 MB the candidates and results travel through pinned host memory the GPU reads and
 * writes directly: one or two kernel launches (128 < N <= 256: one), one synchronisation, no memcpy. */
int tgp_evaluate(tgp_handle h, const double *Xc, int64_t M, int acq, double sf, double incumbent,
                 double param, double *mu, double *sigma, double *acq_out, double *best_val,
                 int64_t *best_idx, int64_t *n_clamped);

/* Many stored models, one batch of points: what the plot path does when it walks the recorder's
 * trials and predicts the same grid with every trial's model (turbo/plotting/trials.py:371,448,
 * 574-577; turbo/plotting/surrogates.py:23-24,61-65).  Each of the T models is given by what
 * defines it -- X_t (N_t, D), y_t, hyper-parameters -- with N_t <= 256; all share the kernel kind,
 * D and normalize_y.  The T fits run as ONE launch of T workgroups, the T sweeps as one launch
 * (a batch whose largest model has more than 128 points runs every model of it on the kernels of
 * 128 < N <= 256: group the models by size for the faster small-problem kernels); the handle's
 * resident model, if any, is not touched.
 *   ls (T, D) row-major (broadcast an isotropic length scale); mu, sigma (T, M) row-major
 *   (sigma, lml nullable); n_clamped: total over the batch.
 * TGP_NOT_PD names the failing model in tgp_last_error. */
int tgp_predict_batch(tgp_handle h, int64_t T, const int64_t *Ns, int64_t D, const double *const *Xs,
                      const double *const *ys, int kernel, const double *constants, const double *ls,
                      const double *noises, const double *jitters, int normalize_y, const double *Xc,
                      int64_t M, double *mu, double *sigma, double *lml, int64_t *n_clamped);

/* Convenience = tgp_evaluate(TGP_ACQ_NONE): ModelInstance.predict
 * (turbo/modules/surrogates.py:332-338). */
int tgp_predict(tgp_handle h, const double *Xc, int64_t M, double *mu, double *sigma);

/* ---- one process, several GPUs ------------------------------------------------------------ */

/* The sharded candidate sweep of one node behind the C-ABI (across processes the
 * Python side does the same with torch.distributed / RCCL).  A tgp_multi owns one tgp_handle per
 * listed device and drives each from its own host thread:
 *   tgp_multi_fit             tgp_fit replicated on every device (identical inputs -> identical
 *                             factor, checked; no exchange)
 *   tgp_multi_set_candidates  contiguous shards of ceil(M / n) rows of the (M, D) host batch
 *   tgp_multi_gen_candidates  every device draws its rows of the one Philox stream (tgp_gen_candidates)
 *   tgp_multi_sweep           tgp_sweep on every shard at once; the winners are reduced with the
 *                             library-wide rule (largest value, lowest GLOBAL index, NaN never wins):
 *                             best_val, best_idx (global), best_row (D, nullable), acq_out (M, nullable).
 *                             = the argsort()[0] of turbo/modules/auxiliary_optimisers.py:63-66 over
 *                             the whole batch.
 * Errors as for the single-device calls; tgp_multi_last_error names the device.  The same device
 * may be listed more than once (each entry gets its own context and stream). */
typedef struct tgp_multi_s *tgp_multi;
int tgp_multi_create(int n, const int *device_ids, int dtype, tgp_multi *out);
int tgp_multi_destroy(tgp_multi m);
const char *tgp_multi_last_error(tgp_multi m);
int tgp_multi_size(tgp_multi m);
int tgp_multi_handle(tgp_multi m, int i, tgp_handle *out);
int tgp_multi_fit(tgp_multi m, const double *X, int64_t N, int64_t D, const double *y, int kernel,
                  double constant, const double *ls, int64_t n_ls, double noise, double jitter,
                  int normalize_y, double *lml, double *y_mean, double *y_std);
int tgp_multi_set_candidates(tgp_multi m, const double *Xc, int64_t M);
int tgp_multi_gen_candidates(tgp_multi m, uint64_t seed, int64_t M, const double *lo, const double *hi);
int tgp_multi_sweep(tgp_multi m, int acq, double sf, double incumbent, double param, double *best_val,
                    int64_t *best_idx, double *best_row, double *acq_out);

/* Handles of one device share one stream (see the conventions above).  A handle switched to a
 * private stream (on != 0) submits to a stream of its own instead, so calls made on several such
 * handles from several host threads overlap on the GPU -- what HipGPSurrogate uses to run the
 * restarts of the hyper-parameter fit (sklearn _gpr.py:326-337) side by side above N = 128, where a
 * single evaluation leaves most of the chip idle.  Synchronises the handle's current stream. */
int tgp_set_private_stream(tgp_handle h, int on);

/* The device's pool of WORKER handles (round 5): handles on private streams that run the concurrent starts of a
 * hyper-parameter fit above the one-launch sizes (sklearn _gpr.py:326-337, n_restarts_optimizer; reached from
 * turbo/modules/surrogates.py:313-318) -- in the library's own threads (tgp_fit_optimise) or in the host's
 * (HipGPSurrogate drives tgp_fit_grad on them with SciPy).  ONE pool per device whatever the number of handles and
 * factories in the process, at most four workers: the HIP runtime deals a process's streams onto a fixed number of
 * hardware queues and two streams on one queue run one after the other (a second factory with workers of its own
 * doubled the time of a hyper-parameter fit in round 4).
 *   tgp_workers_acquire   out[0..n) = n worker handles (1 <= n <= 4) of h's device, created on demand; the pool is
 *                         LOCKED for the calling thread until tgp_workers_release (one hyper-parameter fit at a time per
 *                         device; a second caller blocks).  The workers accept every call a handle accepts except
 *                         tgp_destroy and tgp_workers_*; they belong to the library and live until the last handle
 *                         made by tgp_create on that device is destroyed.
 *   tgp_workers_release   unlocks the pool; must be called by the thread that acquired it (TGP_BAD_ARG otherwise; a
 *                         second acquire by the thread that holds the pool is refused the same way instead of deadlocking). */
int tgp_workers_acquire(tgp_handle h, int n, tgp_handle *out);
int tgp_workers_release(tgp_handle h);

/* Fit and sweep overlapped (round 5).  The reference runs them back to back -- turbo/optimiser.py:336-340:
 * construct_model, then the acquisition's maximisation over ONE vectorised batch
 * (turbo/modules/auxiliary_optimisers.py:59-66) -- and so does this library, call by call.  But the batch does not
 * depend on the model: when the candidates of the NEXT sweep are already resident (tgp_set_candidates[_dev],
 * tgp_gen_candidates[_lhs]) at the time tgp_fit / tgp_fit_grad is called, the front of that sweep can run INSIDE
 * the fit, beside its latency-bound panel chain, on the device's third stream:
 *   mode 1   the candidates' scaling by the new length scales and the first launch pair's cross-kernel
 *            (needs nothing of the fit but X / length_scale);
 *   mode 2   ... and the contraction of that pair over the row tiles whose rows of L^-1 are already final.
 *   mode 0   (default) nothing: fit and sweep strictly one after the other.
 * tgp_sweep then skips what is done; any call that changes the candidates, the fit or the sweep workspace in between
 * simply discards the front.  Same kernels, same arithmetic and the same order of every sum either way: results are
 * bit-identical to mode 0.  Applies to the general sweep of f64 / f32 handles (N > 256) on the shared streams; a no-op
 * elsewhere.  The environment's TGP_OVERLAP caps the mode for A/B runs.
 * A batch borrowed with tgp_set_candidates_dev is READ during the next tgp_fit once a mode is armed: it has to stay
 * valid (and unchanged) from the arming until the sweep that consumes it, not only during tgp_sweep. */
int tgp_set_overlap(tgp_handle h, int mode);

/* (round 6) What the device's shared streams probed as when they were created (the first tgp_create on the device):
 * out3[0] = 1 when the BACKGROUND stream really runs beside the main one (the inverse factor behind the Cholesky's
 * panel chain), 0 when the runtime put both on one hardware queue -- they then run one after the other and every
 * large fit takes about twice as long --, -1 not probed (TGP_BG_PROBE=0); out3[1] likewise for the THIRD stream
 * (tgp_set_overlap; 0: there is none and the overlap is a no-op); out3[2] = GPU_MAX_HW_QUEUES of the process
 * environment (0: unset).  The HIP runtime deals streams onto 4 hardware queues unless that variable said otherwise
 * BEFORE its first call in the process; turbo_amd/_lib.py sets 8 at import and warns, from this entry, when another
 * library (torch) initialised the runtime first and the probes came back serialised. */
int tgp_stream_status(tgp_handle h, int *out3);
/* Every environment switch (TGP_*) of the library with the value in force in this process, one
 * "NAME=value<TAB># what it selects" line each (csrc/tuning.hpp: the ONE table they are all read from).  Writes at
 * most cap bytes including the terminating 0; returns the size needed (or -1). */
int64_t tgp_tuning(char *buf, int64_t cap);

/* ---- the reference's HOST candidate draw, outside the interpreter ------------------------------ */

/* The batch of the reference's random_selector (turbo/modules/naive_selectors.py:39-46: one
 * np.random.uniform(pmin, pmax, size=(M, 1)) per parameter from NumPy's GLOBAL legacy RNG, hstacked), bit for bit:
 * key624 / pos are MT19937's state words and position as np.random.get_state() returns them; on return they are
 * where NumPy's own D calls would have left them (np.random.set_state) and out (M, D) row-major holds the batch,
 * column c = lo[c] + (hi[c] - lo[c]) * u with u the stream's next M doubles.  Plain host code (no handle, no GPU,
 * in the host-only library too): NumPy spends 66 ms on C3's 262 144 x 32 draw -- twice the GPU step it feeds --
 * this 10-20.  turbo_amd/naive_selectors.py random_selector calls it for large batches. */
int tgp_mt19937_uniform_columns(uint32_t *key624, int32_t *pos, int64_t M, int64_t D, const double *lo,
                                const double *hi, double *out);

/* The same draw made RESIDENT (GPU handles): the stream's words are generated on the host -- the one part that cannot
 * run in parallel -- and copied a column at a time while the next is generated; the GPU forms the doubles and the
 * (rows, D) layout in NumPy's arithmetic.  The resident batch is rows [first_row, first_row + rows) of the M_total-row
 * batch NumPy would have drawn -- a rank's shard of ONE batch: every rank with the same RNG state gets its
 * own rows of the same batch, and key624 / pos end behind the WHOLE batch on all of them (the other rows' words are
 * passed over).  tgp_read_candidates returns those doubles bit for bit; on any failure key624 / pos are untouched.
 * D is the fitted model's.  C3's batch: 2.2 ms instead of 46-66 ms of NumPy + the upload. */
int tgp_set_candidates_mt19937(tgp_handle h, uint32_t *key624, int32_t *pos, int64_t M_total, int64_t first_row,
                               int64_t rows, const double *lo, const double *hi);

/* ---- measurement ------------------------------------------------------------------------ */

/* Turn per-kernel HIP-event timing on/off (on the library's own stream). */
int tgp_profile_enable(tgp_handle h, int on);
/* Totals since the last tgp_profile_reset: launches and milliseconds of the dominant sweep
 * kernel (trmm_sumsq), of the cross-kernel build, and of the whole last fit / sweep. */
int tgp_profile_read(tgp_handle h, int64_t *trmm_launches, double *trmm_ms,
                     int64_t *kstar_launches, double *kstar_ms,
                     double *last_fit_ms, double *last_sweep_ms);
int tgp_profile_reset(tgp_handle h);
/* Device times (ms, hipEvents on the library's stream) of the last calls, first n of:
 * [fit, sweep, LML gradient: K^-1 = U U^T, LML gradient: pairwise weights and traces,
 *  LML gradient: ARD products, (not a time) the algorithmic flops of the contraction launches timed since
 *  tgp_profile_reset: rows^2 per candidate over the rows each launch covered -- the numerator that goes with
 *  tgp_profile_read's trmm_ms when a fit took row tiles of a launch (tgp_set_overlap),
 *  (not a time, round 6) 1 when the last sweep ran in float64 whatever the handle's dtype -- the one-workgroup
 *  kernels for N <= 128 and the one-launch sweep above them only exist in f64 --, 0 when it ran in the handle's
 *  arithmetic, -1 before the first sweep: what CandidateSweep asks before it re-forms the winner's value in f64].
 * The short polled calls (small fit, fit + gradient; doorbell.hpp) report the kernels' own wall_clock64() span;
 * slots [7..11] (a debugging aid) hold that kernel's phase stamps in microseconds from its start: inputs staged,
 * first kernel-matrix tile in LDS, first block factored, fit done, call done (0 when the last fit was not polled). */
int tgp_last_timings(tgp_handle h, double *out, int64_t n);
/* Candidates per trmm launch (chunk) and padded N used by the sweep, for the roofline maths. */
int tgp_sweep_geometry(tgp_handle h, int64_t *chunk, int64_t *n_padded);

#ifdef __cplusplus
}
#endif
#endif /* TURBOGP_H */
