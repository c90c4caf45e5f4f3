import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(GOLDEN, name + ".npz")


STATIC_CASES = [
    "rbf_iso_2d", "rbf_iso_8d", "rbf_iso_32d", "matern52_ard_16d", "matern32_iso_64d",
    "matern12_iso_4d_nowhite", "default_matern52_white_branin", "rbf_3d_no_normalise",
    "rbf_3d_constant_y", "rbf_5d_ragged", "rbf_2d_single_point",
]


@pytest.fixture(params=STATIC_CASES)
def golden_case(request):
    import numpy as np
    with np.load(golden_path(request.param), allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    d["_name"] = request.param
    return d
