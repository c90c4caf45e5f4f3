"""Child process of test_alternate_kernel_paths: the TGP_* tuning switches are read once per
process, so each selection of kernels gets its own process.  Compares fit + sweep with the
oracle at two sizes and prints 'alt-paths ok'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbo_amd as ta                      # noqa: E402
from oracle import gp_oracle as o            # noqa: E402

# (the third size: a SMALLER fit on the same handle after a larger one -- Linv must be clean where the new fit skips --
# and the one-launch sweep of 128 < N <= 256)
gp_prev = {}
for (N, D, M, kind, dtype, tol) in [(1100, 6, 3000, "matern52", "f64", 1e-7), (2304, 9, 5000, "rbf", "f32", 5e-3), (200, 6, 3000, "matern52", "f64", 1e-7)]:
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, (M, D))
    ls = float(np.sqrt(D / 6.0))
    om = o.fit(X, y, kind, 1.0, ls, 1e-3, 1e-10, True)
    mu, sg = o.predict(om, Xc)
    gp = gp_prev.setdefault(dtype, ta.NativeGP(0, dtype))
    lml, _, _ = gp.fit(X, y, kind, 1.0, ls, 1e-3, 1e-10, True)
    assert abs(lml - om.lml) <= 1e-9 * abs(om.lml), (lml, om.lml)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_L), om.L, rtol=1e-8, atol=1e-11)
    gp.set_candidates(Xc)
    r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    np.testing.assert_allclose(r["mu"], mu, rtol=tol, atol=tol * om.y_std)
    np.testing.assert_allclose(r["sigma"] ** 2, sg ** 2, rtol=tol, atol=tol * 1.001 * om.y_std ** 2)
    assert r["best_idx"] == int(np.argmax(r["acq"]))
    g_lml, g = gp.fit_grad(X, y, kind, 1.0, ls, 1e-3, 1e-10, True)
    o_lml, og = o.lml_and_grad(X, y, kind, 1.0, ls, 1e-3, 1e-10, True)
    np.testing.assert_allclose(g, og, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(og).max()))
print("alt-paths ok")
