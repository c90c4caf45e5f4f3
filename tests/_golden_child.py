"""Child process of test_golden_cases_on_the_blocked_path: every static golden case (all of them
N <= 64, so the default process runs them on the small-problem kernels) once more with whatever
TGP_* switches the parent set -- TGP_SMALL=0 sends them down the blocked multi-launch path.
Prints 'golden-child ok'."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import turbo_amd as ta                      # noqa: E402
from conftest import STATIC_CASES, golden_path   # noqa: E402

ACQ = {"ei": (ta._lib.ACQ_EI, 0.01), "pi": (ta._lib.ACQ_PI, 0.01), "ucb2": (ta._lib.ACQ_UCB, 2.0)}
for name in STATIC_CASES:
    with np.load(golden_path(name), allow_pickle=False) as z:
        c = {k: z[k] for k in z.files}
    ls = c["length_scale"]
    gp = ta.NativeGP(0, "f64")
    lml, ym, ys = gp.fit(c["X"], c["y"], str(c["kind"]), float(c["constant"]), ls, float(c["noise"]),
                         float(c["jitter"]), bool(c["normalize_y"]))
    assert abs(lml - float(c["lml"])) <= 1e-9 * abs(float(c["lml"])) + 1e-8, (name, lml)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_L), c["L"], rtol=1e-9, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_ALPHA), c["alpha"], rtol=1e-6,
                               atol=1e-8 * np.abs(c["alpha"]).max(), err_msg=name)
    vat = 1e-9 * (float(c["constant"]) + float(c["noise"])) * float(c["y_std"]) ** 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = gp.evaluate(c["Xc"], want_mu=True, want_sigma=True)
    np.testing.assert_allclose(r["mu"], c["mus"], rtol=1e-5, atol=1e-9, err_msg=name)
    np.testing.assert_allclose(r["sigma"] ** 2, c["sigmas"] ** 2, rtol=1e-5, atol=vat, err_msg=name)
    s_floor = np.sqrt(vat)
    well = c["sigmas"] > 100 * s_floor
    for key, (enum, param) in ACQ.items():
        for ext, sf in (("min", -1.0), ("max", 1.0)):
            inc = float(c["incumbent_" + ext])
            r = gp.evaluate(c["Xc"], enum, sf, inc, param, want_acq=True)
            want = c["acq_%s_%s" % (key, ext)]
            scale = max(1.0, float(np.abs(want).max()))
            np.testing.assert_allclose(r["acq"][well], want[well], rtol=1e-5, atol=1e-9 * scale, err_msg=name + key + ext)
            assert r["best_idx"] == int(np.argmax(r["acq"])) and r["best_val"] == r["acq"][r["best_idx"]]
print("golden-child ok")
