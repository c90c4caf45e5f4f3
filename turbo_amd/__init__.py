"""turbo_amd: the GP-surrogate inner loop of mbway/turbo on MI355X (gfx950).

Plugin classes with the reference's Surrogate / AcquisitionFunction / auxiliary-optimiser API
over hand-written HIP kernels behind a ctypes C-ABI (include/turbogp.h).  Usage with an
unmodified reference Optimiser (settings_preset=None):

    import turbo as tb, turbo.modules as tm, turbo_amd as ta
    op = tb.Optimiser(f, 'min', bounds, pre_phase_trials=4, settings_preset=None)
    op.latent_space = tm.NoLatentSpace(); op.pre_phase_select = tm.LHS_selector(4)
    op.fallback = tm.Fallback(selector=tm.random_selector())
    op.surrogate = ta.HipGPSurrogate(model_params=dict(kernel=..., optimizer=None, normalize_y=True))
    op.acquisition = ta.EI(xi=0.01)
    op.aux_optimiser = ta.CandidateSweep(num_random=262144)
"""
from .kernels import GPKernel
from .bounds import Bounds
from .surrogates import Surrogate, HipGPSurrogate
from .acquisition_functions import AcquisitionFunction, UCB, PI, EI
from .auxiliary_optimisers import CandidateSweep, RandomAndQuasiNewton
from .naive_selectors import random_selector, LHS_selector
from ._lib import TurboGPLibraryError, NativeGP, LIB_PATH

__all__ = ['GPKernel', 'Bounds', 'Surrogate', 'HipGPSurrogate', 'AcquisitionFunction', 'UCB', 'PI',
           'EI', 'CandidateSweep', 'RandomAndQuasiNewton', 'random_selector', 'LHS_selector',
           'TurboGPLibraryError', 'NativeGP', 'LIB_PATH']
