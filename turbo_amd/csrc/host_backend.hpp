// host_backend.hpp -- the GP posterior WITHOUT a GPU, inside libturbogp.so (tgp_create(TGP_DEVICE_HOST)).
//
// Why it exists: the reference pickles every trial's model (turbo/recorder.py:117-155) and the plot
// path reloads and queries them "often in another process" (turbo/recorder.py:157-163,
// turbo/plotting/trials.py:192-195, :371, :448, :574-577) -- a process that need not own an MI355X.
// This is a RELOAD path, not a second product path: the Python factory only selects it for a model
// that was unpickled on a machine without a HIP device, and says so with a warning.
//
// Independent of the HIP path and of the test oracle: plain C++17, no HIP header, no NumPy/SciPy.
// Arithmetic (the same sklearn calls the HIP kernels replace):
//   y normalisation      sklearn/gaussian_process/_gpr.py:272-282
//   kernel matrix        kernels.py RBF :1553-1560, Matern :1708-1738 (direct sum of squared
//                        differences of X / length_scale), diagonal c + noise + jitter (_gpr.py:347)
//   Cholesky             _gpr.py:349: left-looking by 64-column panels, rows of a panel in parallel
//   alpha, LML           _gpr.py:360-364, :584-613 (two triangular substitutions)
//   predict              _gpr.py:443-494: K*, mu = s_y K* alpha + y_mean, v = L^-1 K*^T by forward
//                        substitution on tiles of 16 candidates, var = (c + noise) - |v|^2, clamp
//   acquisition          turbo/modules/acquisition_functions.py UCB :147-158, PI :225-247, EI :336-358
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tgp_host {

struct HostGP {
    std::string err;
    bool fitted = false;
    int64_t N = 0, D = 0;
    int kernel = 0, normalize_y = 1;
    double constant = 1.0, noise = 0.0, jitter = 0.0;
    double y_mean = 0.0, y_std = 1.0, lml = 0.0, sumlog = 0.0;
    std::vector<double> ls;      // (D,) broadcast when isotropic
    std::vector<double> X, y;    // raw inputs (the state blob)
    std::vector<double> Xs;      // (N, D) X / ls
    std::vector<double> L;       // (N, N) row-major, lower triangle (zeros above)
    std::vector<double> alpha;   // (N,)
    std::vector<double> cand;    // (M, D) resident candidates
    int64_t M = 0;
    double last_fit_ms = 0.0, last_sweep_ms = 0.0;

    // all return a tgp_status; err holds the message of a failure
    int fit(const double *X, int64_t N, int64_t D, const double *y, int kernel, double constant,
            const double *ls, int64_t n_ls, double noise, double jitter, int normalize_y,
            double *lml, double *y_mean, double *y_std);
    int export_state(void *buf, int64_t cap, int64_t *size);
    int import_state(const void *buf, int64_t size, double *lml);
    int debug_read(int which, double *out);
    int set_candidates(const double *Xc, int64_t M);
    int read_candidates(int64_t first, int64_t count, double *out);
    int sweep(int acq, double sf, double incumbent, double param, double *mu, double *sigma,
              double *acq_out, double *best_val, int64_t *best_idx, int64_t *n_clamped);
};

// The candidate batch of the reference's random_selector (turbo/modules/naive_selectors.py:39-46: one
// np.random.uniform(pmin, pmax, size=(M, 1)) per parameter from NumPy's GLOBAL legacy RNG, hstacked) reproduced bit
// for bit outside the interpreter: the MT19937 stream continued from (key, pos) exactly as numpy/random/src/mt19937
// does (regenerate 624 words at pos == 624, temper on read), a double from two outputs as legacy random_sample
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53, a draw as low + (high - low) * u in two roundings -- column c takes draws
// [c M, (c + 1) M) of the stream and lands in out[i * D + c].  key / pos are left where NumPy's own calls would have
// left them.  At BASELINE's C3 (262 144 x 32) NumPy takes 66 ms for that draw on the GPU box's host, 2x the GPU step
// it feeds.  Returns a tgp_status.
int mt19937_uniform_columns(uint32_t *key, int32_t *pos, int64_t M, int64_t D, const double *lo, const double *hi,
                            double *out);
// the next n outputs of that stream (tempered 32-bit words) into dst, (key, pos) advanced: the raw material of the
// draw above, for the GPU to finish (tgp_set_candidates_mt19937)
void mt19937_fill(uint32_t *key, int32_t *pos, uint32_t *dst, int64_t n);
// ... and n outputs passed over unread (the rows of a column that belong to other ranks' shards)
void mt19937_skip(uint32_t *key, int32_t *pos, int64_t n);

}  // namespace tgp_host
