"""Round-3 kernels against their own A/B switches, bit for bit (GPU):

  * the fused panel launch (fused_panel_kernel: solve + next pivot + rank-64 update in one launch, the
    pivot workgroup's products cut to their non-zero halves) leaves the SAME L, Linv, alpha as the
    two-launch chain -- SHA-256 over the buffers at sizes either side of every block boundary;
  * the one-launch hyper-parameter fit with three workgroups per start (64 < N <= 128, a bounded barrier
    per start) returns the SAME theta, -LML and status as one workgroup per start, up to 64 starts;
  * the host backend (tgp_create(TGP_DEVICE_HOST)) agrees with the GPU on random models.

The switches are read once per process, so every setting runs tools/stress_round3.py in a child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, **env):
    e = dict(os.environ, TGP_STRESS_QUICK="1")
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_round3.py"), mode], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def test_fused_panel_chain_is_bit_identical_to_the_two_launch_chain():
    two = _run("fit", TGP_PANEL_FUSE="0")
    assert two.count("\n") == 12
    assert _run("fit", TGP_PANEL_FUSE="1", TGP_PANEL_FUSE_TILES="100000") == two     # every outer block fused
    assert _run("fit") == two                                                         # the default choice per outer block


def test_three_workgroups_per_start_equal_one():
    one = _run("hyper", TGP_HYPER_WGS="1")
    assert one.count("\n") == 5 and '"status": [1]' in one
    assert _run("hyper") == one


def test_host_backend_agrees_with_the_gpu_on_random_models():
    assert '"idx": 0' in _run("host")
