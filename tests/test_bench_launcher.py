"""`python bench.py --gpus N` with N > 1 and no RANK in the environment must start its own ranks
(torch.distributed.run as a child process), let rank 0's one JSON line through and hand back the child's
exit code.  Here (no GPU) the ranks run over gloo against the oracle-backed stand-in context of the test
suite; on the GPU box tests/test_gpu_round4.py runs the same command with two gloo ranks on the card."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_gpus2_starts_its_own_ranks_and_prints_one_line():
    res = _run({"BENCH_BACKEND": "gloo", "BENCH_TEST_CONTEXT": "oracle_context:OracleBackedContext",
                "OMP_NUM_THREADS": "2"},
               "--gpus", "2", "--config", "c1", "--steps", "2", "--warmup", "1", "--no-opt-in")
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["config"]["M_per_gpu"] == 32768 and out["config"]["M_total"] == 65536
    assert set(out["amdahl_bound"]["speedup_max_by_gpus"]) == {"2", "4", "8"}
    assert "standin" in out and out["value"] is None      # a rehearsal says so and carries no number


def test_bench_launcher_hands_back_the_ranks_failure():
    # no GPU here and no stand-in: every rank must fail loudly, and the launcher must not hide it
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    res = _run({"BENCH_BACKEND": "gloo"}, "--gpus", "2", "--config", "c1", "--steps", "1", "--warmup", "0",
               "--no-opt-in", "--no-cpu-baseline")
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_launcher_sets_the_ipc_mode_rccl_needs_on_this_pool(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert bench.launch_ranks(["--gpus", "8", "--steps", "3"], 8) == 7
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in c and c[c.index("--nproc-per-node") + 1] == "8"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "8", "--steps", "3"]
