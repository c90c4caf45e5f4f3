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


# name -> (kind, full hyper-parameter vector [constant, length scale(s), noise or None] as a function of the recorded
# free ones): the reference's optimisation traces (tests/golden/make_golden_hyper.py, make_golden_hyper2.py)
OPT_TRACES = {
    "opt_default_2d": ("matern52", lambda hp: (hp[0], hp[1], hp[2])),
    "opt_rbf_ard_4d": ("rbf", lambda hp: (hp[0], hp[1:5], hp[5])),
    "opt_matern32_iso_5d_mid": ("matern32", lambda hp: (hp[0], hp[1], hp[2])),
    "opt_fixed_noise_3d": ("matern52", lambda hp: (hp[0], hp[1], 1e-2)),
    "opt_fixed_constant_ard_3d": ("rbf", lambda hp: (1.0, hp[0:3], hp[3])),
    "opt_nowhite_matern52_3d": ("matern52", lambda hp: (hp[0], hp[1], None)),
}


@pytest.mark.parametrize("name", sorted(OPT_TRACES))
def test_the_oracle_at_the_references_optimised_hyper_parameters(name):
    """every trial of every optimisation trace: at the hyper-parameters the reference's fit arrived at, the oracle's log
    marginal likelihood is the reference's (GaussianProcessRegressor.log_marginal_likelihood_value_, _gpr.py:332-337)
    and so is its posterior at the recorded query points -- and its LML gradient vanishes there in every free
    coordinate that is not on a bound"""
    t = load(name)
    kind, full = OPT_TRACES[name]
    jitter = float(t["alpha"]) if "alpha" in t else 1e-10
    for k, n in enumerate(t["sizes"]):
        X, y = t["X"][:n], t["y"][:n]
        c, ls, noise = full(t["hp_%d" % k])
        m = o.fit(X, y, kind, c, ls, 0.0 if noise is None else noise, jitter, True)
        lml, grad = o.lml_and_grad(X, y, kind, c, ls, noise, jitter, True)
        assert lml == pytest.approx(float(t["lml_%d" % k]), rel=1e-9, abs=1e-9)
        mu, sg = o.predict(m, t["X"][-16:])
        np.testing.assert_allclose(mu, t["mu_%d" % k], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(sg, t["sigma_%d" % k], rtol=1e-6, atol=1e-8)
        hp = t["hp_%d" % k]
        inner = (hp > 1.1e-5) & (hp < 0.9e5)
        # the recorded vector holds the free hyper-parameters only; the oracle's gradient is over [c, ls..., noise]
        free = {"opt_fixed_noise_3d": slice(0, -1), "opt_fixed_constant_ard_3d": slice(1, None)}.get(name, slice(None))
        g = np.asarray(grad)[free]
        assert len(g) == len(hp)
        assert np.max(np.abs(g[inner]), initial=0.0) < 2e-2 * max(1.0, abs(lml)), (k, g)
