"""Run by tests/test_gpu_configs.py in a process of its own: bench.py's multi-GPU step with the RCCL
backend on a process group of ONE rank -- the device-packed winner record, the all-gather on the
GPU (torch.distributed 'nccl' = RCCL) and the local reduce, on the one card a test box has.  The
N > 1 case itself runs only on the driver's 8-GPU node; the gloo tests cover its rank arithmetic."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import socket

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:      # a free port, not a fixed one (two runs on one host would collide)
    _s = socket.socket()
    _s.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(_s.getsockname()[1])
    _s.close()
os.environ["RANK"] = "0"
os.environ["WORLD_SIZE"] = "1"
os.environ["LOCAL_RANK"] = "0"

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
import bench
import turbo_amd as ta

cfg = dict(bench.CONFIGS["c1"])
X, y, ls = bench.synth_train(cfg)
Xc, m_local, offset, m_job = bench.shard_candidates(cfg, 0, 1, False)
gp = ta.NativeGP(0, cfg["dtype"])
gp.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
cand = torch.from_numpy(Xc).to("cuda:0")
gp.set_candidates_dev(cand.data_ptr(), m_local, keepalive=cand)
rec = torch.zeros(cfg["D"] + 2, dtype=torch.float64, device="cuda:0")
gp.set_winner_out(rec.data_ptr(), offset, keepalive=rec)
step = bench.build_step(gp, cfg, X, y, ls, float(y.min()), 2, offset, rec, "nccl")   # world = 2: take the exchange path
for _ in range(3):
    r = step()
dist.barrier()
torch.cuda.synchronize()
assert r["job_best_val"] == r["best_val"] and r["job_best_idx"] == offset + r["best_idx"], r
np.testing.assert_array_equal(np.asarray(r["job_best_row"]).reshape(-1), np.asarray(gp.get_candidate(r["best_idx"])).reshape(-1))
dist.destroy_process_group()
print("rccl world-1 exchange ok")
