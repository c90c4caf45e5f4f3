"""Pin the oracle's LML gradient to golden vectors produced by the reference
wrapper's scikit-learn model.  CPU only."""
import numpy as np
import pytest

from conftest import golden_path
from oracle import gp_oracle as o

GRAD_CASES = ["grad_rbf_iso_3d", "grad_rbf_ard_3d", "grad_matern52_ard_5d", "grad_matern32_iso_5d",
              "grad_matern12_iso_5d_nowhite", "grad_default_matern52_white"]


def load(name):
    with np.load(golden_path(name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", GRAD_CASES)
def test_lml_gradient(name):
    c = load(name)
    noise = float(c["noise"])
    ls = c["length_scale"]
    ls = float(ls[0]) if len(ls) == 1 else ls
    lml, grad = o.lml_and_grad(c["X"], c["y"], str(c["kind"]), float(c["constant"]), ls,
                               None if noise < 0 else noise, float(c["jitter"]), True)
    assert lml == pytest.approx(float(c["lml"]), rel=1e-10)
    np.testing.assert_allclose(grad, c["grad"], rtol=1e-7, atol=1e-9)
    # finite differences of the oracle's own LML agree with its gradient
    theta = np.log(np.concatenate([[float(c["constant"])], np.atleast_1d(ls), [] if noise < 0 else [noise]]))
    eps = 1e-6
    for p in range(len(theta)):
        vals = []
        for sgn in (1, -1):
            t = theta.copy()
            t[p] += sgn * eps
            e = np.exp(t)
            nl = len(np.atleast_1d(ls))
            l2 = e[1:1 + nl]
            vals.append(o.lml_and_grad(c["X"], c["y"], str(c["kind"]), e[0], float(l2[0]) if nl == 1 else l2,
                                       None if noise < 0 else e[-1], float(c["jitter"]), True)[0])
        assert (vals[0] - vals[1]) / (2 * eps) == pytest.approx(grad[p], rel=2e-4, abs=1e-5)
