"""Drop-in check of the HOST side against the real reference loop (CPU only, build container only).

The reference's ``turbo.Optimiser`` (turbo/optimiser.py:227-357) is run UNMODIFIED with this
package's plugin classes in its ``surrogate`` / ``acquisition`` / ``aux_optimiser`` slots.  There is
no GPU here, so the ctypes context (``turbo_amd._lib.NativeGP``) is replaced by a stand-in that
answers from the CPU oracle -- test infrastructure only; what is under test is the plugin contract:
call order and arguments of the Optimiser, listener events, the RNG draw sequence of the candidate
sweep, fitting_info / maximisation_info keys, and the Recorder's dill round trip of the models.
With the seeds of the golden Branin trace the run must reproduce the reference's own choices.
"""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import golden_path
from oracle import gp_oracle as o

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "turbo")),
                                reason="needs the reference checkout (build container only)")


from oracle_context import OracleBackedContext


@pytest.fixture
def reference(monkeypatch):
    sys.path.insert(0, REF)
    if not hasattr(np, "asscalar"):                      # harness-side shim
        monkeypatch.setattr(np, "asscalar", lambda a: np.asarray(a).item(), raising=False)
    import turbo as tb
    import turbo.modules as tm
    import turbo_amd as ta
    monkeypatch.setattr(ta._lib, "NativeGP", OracleBackedContext)
    monkeypatch.setattr(ta._lib, "load", lambda: None)
    yield tb, tm, ta
    sys.path.remove(REF)


def branin(x, y):
    from math import pi
    return float((y - (5.1 / (4 * pi ** 2)) * x ** 2 + 5 * x / pi - 6) ** 2 + 10 * (1 - 1 / (8 * pi)) * np.cos(x) + 10)


def test_unmodified_optimiser_runs_with_the_plugins_and_reproduces_the_golden_trace(reference, tmp_path):
    tb, tm, ta = reference
    with np.load(golden_path("branin_trace"), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    np.random.seed(42)                                    # demos/Branin-Hoo.ipynb cell 4
    op = tb.Optimiser(branin, 'min', [('x', -5., 10.), ('y', 0., 15.)], pre_phase_trials=4,
                      settings_preset=None)
    op.latent_space = tm.NoLatentSpace()
    op.pre_phase_select = tm.LHS_selector(num_total=4)
    op.fallback = tm.Fallback(selector=tm.random_selector())
    # the three slots of the hot path, filled by this package
    op.surrogate = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1.0),
                                                       normalize_y=True, optimizer=None),
                                     training_iterations=1)
    op.acquisition = ta.EI(xi=0.01)
    op.aux_optimiser = ta.RandomAndQuasiNewton(num_random=1024, grad_restarts=0, start_from_best=0)
    with pytest.raises(AttributeError):                   # the Optimiser's typo guard still applies
        op.surogate = None

    events = []

    class Spy(tm.Listener):
        def surrogate_fitted(self, trial_num):
            events.append(("fitted", trial_num))

        def acquisition_maximised(self, trial_num):
            events.append(("maximised", trial_num))

        def selection_finished(self, trial_num, x, selection_info):
            events.append(("selected", trial_num, np.array(x, copy=True), dict(selection_info)))
    op.register_listener(Spy())
    rec = tb.Recorder(op)
    n_trials = 16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        op.run(max_trials=n_trials)

    np.testing.assert_allclose(np.vstack(op.rt.trial_xs), t["trial_xs"][:n_trials], rtol=0, atol=1e-9)
    np.testing.assert_allclose(op.rt.trial_ys, t["trial_ys"][:n_trials], rtol=1e-9)
    sel = [e for e in events if e[0] == "selected" and e[3]["type"] != "pre_phase"]
    assert [e[1] for e in sel] == list(range(4, n_trials))
    for k, e in enumerate(sel):
        info = e[3]
        assert set(info) >= {"type", "model", "fitting_info", "acq_info", "maximisation_info"}
        assert info["fitting_info"]["iterations"] == 1 and info["acq_info"] == {"xi": 0.01}
        assert info["maximisation_info"]["max_acq"] == pytest.approx(float(t["max_acq"][k]), rel=1e-8)
        assert info["model"].get_log_likelihood() == pytest.approx(float(t["lml"][k]), rel=1e-10)
        assert info["model"].get_hyper_param_names() == ["k1__k1__constant_value", "k1__k2__length_scale",
                                                         "k2__noise_level"]
    assert [e[0] for e in events if e[0] != "selected"][:2] == ["fitted", "maximised"]

    # Recorder: dill + gzip round trip of the recorded models (turbo/recorder.py:117-163)
    path = str(tmp_path / "run")
    rec.save_compressed(path)
    rec2 = tb.Recorder.load_compressed(path)
    m_old = rec.trials[n_trials - 1].selection_info["model"]
    m_new = rec2.trials[n_trials - 1].selection_info["model"]
    grid = np.random.RandomState(0).uniform([-5, 0], [10, 15], size=(50, 2))
    np.testing.assert_array_equal(m_new.predict(grid), m_old.predict(grid))
    acq = rec2.get_acquisition_function(n_trials - 1)     # rebuilt through our factory
    assert acq(grid).shape == (50,) and acq.get_name() == "EI"


def test_hyper_parameter_fit_through_the_unmodified_optimiser(reference):
    """training_iterations > 0 with warm starts, as in demos/Branin-Hoo.ipynb cell 7"""
    tb, tm, ta = reference

    class OracleGrad(OracleBackedContext):
        def fit_grad(self, X, y, kind, constant, length_scale, noise, jitter, normalize_y):
            self.fit(X, y, kind, constant, length_scale, noise, jitter, normalize_y)
            lml, g = o.lml_and_grad(X, y, kind, constant, length_scale, noise, jitter, normalize_y)
            return lml, g
    ta._lib.NativeGP = OracleGrad
    np.random.seed(1)
    op = tb.Optimiser(branin, 'min', [('x', -5., 10.), ('y', 0., 15.)], pre_phase_trials=4, settings_preset=None)
    op.latent_space = tm.NoLatentSpace()
    op.pre_phase_select = tm.LHS_selector(num_total=4)
    op.fallback = tm.Fallback(selector=tm.random_selector())
    # (optimizer='scipy': SciPy drives the context's objective from Python -- the stand-in context has no L-BFGS-B of its
    # own, the GPU library's is tests/test_host_lbfgs.py's and the GPU suite's business)
    op.surrogate = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1.0), normalize_y=True,
                                                       random_state=0, optimizer='scipy'), training_iterations=2)
    op.acquisition = ta.UCB(beta=1)
    op.aux_optimiser = ta.CandidateSweep(num_random=256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        op.run(max_trials=9)
    assert op.surrogate._last_model_params is not None and len(op.surrogate._last_model_params) == 3
    assert np.isfinite(op.rt.trial_ys).all() and len(op.rt.trial_ys) == 9
