"""Test infrastructure: a stand-in for ``turbo_amd._lib.NativeGP`` that answers from the CPU oracle
(oracle/gp_oracle.py), so the HOST side of the package -- plugin contract, sharding, the winner
exchange, bench.py's step function -- can be exercised where there is no GPU.  Never imported by
the product."""
import time

import numpy as np

from oracle import gp_oracle as o


class OracleBackedContext:
    """stands in for turbo_amd._lib.NativeGP: same methods, answers from oracle/gp_oracle.py"""
    ACQ = {1: "ucb", 2: "pi", 3: "ei"}

    def __init__(self, device=0, dtype="f64"):
        self.model = None
        self.appended = False
        self.last_fit_ms = 0.0
        self.last_sweep_ms = 0.0

    def fit(self, X, y, kind, constant, length_scale, noise, jitter, normalize_y, append=False):
        t0 = time.perf_counter()
        self.model = o.fit(X, y, kind, constant, length_scale, noise, jitter, normalize_y)
        self.last_fit_ms = (time.perf_counter() - t0) * 1e3
        self.y = np.asarray(y)
        self.N, self.D = np.asarray(X).shape
        return self.model.lml, self.model.y_mean, self.model.y_std

    def profile_read(self):
        return {"last_fit_ms": self.last_fit_ms, "last_sweep_ms": self.last_sweep_ms, "trmm_launches": 0, "trmm_ms": 0.0,
                "kstar_launches": 0, "kstar_ms": 0.0}

    def profile_enable(self, on):
        pass

    def profile_reset(self):
        pass

    def sweep_geometry(self):
        return 0, 0

    def set_candidates(self, Xc):
        self.Xc = np.array(Xc, dtype=np.float64)
        self.M = len(self.Xc)

    def get_candidate(self, idx):
        return self.Xc[idx].copy()

    def sweep(self, acq=0, sf=1.0, incumbent=0.0, param=0.0, want_mu=False, want_sigma=False, want_acq=False):
        t0 = time.perf_counter()
        mu, sg = o.predict(self.model, self.Xc)
        self.last_sweep_ms = (time.perf_counter() - t0) * 1e3
        out = dict(mu=mu if want_mu else None, sigma=sg if want_sigma else None, acq=None,
                   best_val=float("nan"), best_idx=-1, n_clamped=0)
        if acq:
            ext = "max" if sf > 0 else "min"
            a = sg.copy() if acq == 4 else o.acquisition(self.ACQ[acq], mu, sg, ext, param, incumbent)
            out["acq"] = a if want_acq else None
            out["best_idx"] = int(np.argmax(a))
            out["best_val"] = float(a[out["best_idx"]])
        return out
