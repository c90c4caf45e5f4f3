#!/usr/bin/env python3
"""bench.py -- GP-surrogate inner loop on MI355X: one step = one pass of the hot path
(fit: kernel matrix + Cholesky + inverse factor + alpha;  sweep: cross-kernel + triangular
contraction + EI/UCB/PI + arg-max) over a synthetic batch that is resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config c3]

N > 1: one rank per GPU under torch.distributed.run.  The driver starts the ranks itself; invoked
plainly (`python bench.py --gpus 8`, no RANK in the environment) this script starts them as a CHILD
process before anything touches the GPU (`launch_ranks`).  Every rank holds the same training set (fit replicated, no comms) and a contiguous shard of the config's ONE batch
of M candidates, ceil(M / N) rows each (strong scaling, as BASELINE.json's configs 3/4 and
SURVEY.md 8e shard it; --weak gives every GPU M candidates of its own instead); the only exchange
is one all-gather of the per-rank winner records (RCCL), read from the device buffer the sweep
packed them into.
Rank 0 prints ONE JSON line (contract in the task statement; BASELINE.json names the metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs (SURVEY.md section 8d fixes the synthetic inputs and hyper-parameters)
CONFIGS = {
    "c1": dict(D=8, N=512, M=65536, kind="rbf", ard=False, acq="ucb", param=2.0, dtype="f64", noise=1e-4, cfg=1),
    "c2": dict(D=16, N=2048, M=131072, kind="matern52", ard=True, acq="ei", param=0.01, dtype="f64", noise=1e-4, cfg=2),
    "c3": dict(D=32, N=4096, M=262144, kind="rbf", ard=False, acq="ei", param=0.01, dtype="f32", noise=1e-2, cfg=3),
    "c4": dict(D=64, N=8192, M=1048576, kind="matern32", ard=False, acq="pi", param=0.01, dtype="f32", noise=1e-2, cfg=4),
}
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # dense MFMA peaks, MI355X_MICROARCH.md / BASELINE.md
ACQ_ENUM = {"ucb": 1, "pi": 2, "ei": 3}


def synth_train(cfg):
    c = cfg["cfg"]
    D, N = cfg["D"], cfg["N"]
    rng = np.random.RandomState(1000 + c)
    X = rng.uniform(0, 1, size=(N, D))
    rng = np.random.RandomState(2000 + c)
    w = rng.normal(size=D) / np.sqrt(D)
    y = np.sin(3 * X @ w) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
    iso = float(np.sqrt(D / 6.0))
    ls = iso * (0.5 + np.arange(D) / (D - 1.0)) if cfg["ard"] else iso
    return X, y, ls


def synth(cfg, rank, m_local):
    """training set + `m_local` candidates of stream `rank` (rank 0 = THE batch of the config;
    other ranks' streams are only used by weak scaling)"""
    X, y, ls = synth_train(cfg)
    rng = np.random.RandomState(3000 + cfg["cfg"] + 7919 * rank)    # per-shard candidate stream
    Xc = rng.uniform(0, 1, size=(m_local, cfg["D"]))
    return X, y, Xc, ls


def shard_candidates(cfg, rank, world, weak):
    """(Xc_local, m_local, global_offset, m_job).  Strong scaling (default, BASELINE configs 3/4,
    SURVEY.md 8e): the config's ONE batch of M candidates -- the same rows the 1-GPU run sweeps --
    cut into contiguous shards of ceil(M / G); weak (--weak): M candidates of its own per GPU."""
    from turbo_amd.distributed import shard_plan
    m_local, offset, m_job = shard_plan(cfg["M"], world, rank, weak)
    if weak:
        Xc = synth(cfg, rank, m_local)[2]
    else:
        Xc = synth(cfg, 0, cfg["M"])[2][offset:offset + m_local]
    return np.ascontiguousarray(Xc), m_local, offset, m_job


def exchange_winner(gp, r, rec, offset, backend):
    """the ONLY exchange of the data path: one all-gather of per-rank winner records
    [value, global index, row] and the same local reduce on every rank"""
    from turbo_amd.distributed import allgather_argmax, allgather_records
    if rec is not None and backend == "nccl":
        return allgather_records(rec, ctx=gp)         # packed on the GPU by the sweep, read in place (tgp_winner_wait orders the two streams)
    if rec is not None:
        return allgather_records(rec.cpu())           # gloo rehearsal on a GPU box
    # no device record: a CPU stand-in context (tests/test_distributed_gloo.py)
    return allgather_argmax(r["best_val"], gp.get_candidate(r["best_idx"]), offset + r["best_idx"])


def build_step(gp, cfg, X, y, ls, inc, world, offset, rec, backend, m_local=None):
    """one step = one pass of the hot path: fit + sweep on the resident shard (+ winner exchange).
    A rank whose shard is empty (more GPUs than candidates) still fits -- the fit is replicated -- and
    enters the exchange with a record that cannot win (-inf, an index past the batch)."""
    empty = m_local is not None and m_local == 0

    def step():
        gp.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
        if empty:
            r = dict(best_val=float("-inf"), best_idx=0, n_clamped=0, sweep_ms=0.0)
            if rec is not None:
                rec.zero_()
                rec[0] = float("-inf")
                rec[1] = float(offset)
        else:
            r = gp.sweep(ACQ_ENUM[cfg["acq"]], -1.0, inc, cfg["param"])
        if world > 1:
            if empty and rec is None:
                from turbo_amd.distributed import allgather_argmax
                v, row, gi = allgather_argmax(float("-inf"), np.zeros(cfg["D"]), offset)
            else:
                v, row, gi = exchange_winner(gp, r, rec, offset, backend)
            r = dict(r, job_best_val=v, job_best_row=row, job_best_idx=gi)
        return r
    return step


def plugin_leg(cfg, X, y, ls, inc, device, c_abi_ms, reps=3):
    """The SAME step through the plugin classes only -- what turbo/optimiser.py:334-346 (_select_trial) calls:
    HipGPSurrogate.construct_model (fixed theta) -> EI|PI|UCB.construct_function -> CandidateSweep.__call__ -- for the
    three candidate sources of CandidateSweep:
      host_draw            the reference-faithful default: the M x D uniform numbers of NumPy's global RNG, one column at a
                           time (turbo/modules/naive_selectors.py:39-46) -- since round 6 the generator's stream is continued
                           inside the library and the batch finished on the GPU (tgp_set_candidates_mt19937): the same
                           numbers, the same np.random state afterwards, the same chosen point
      host_draw_numpy      the same batch formed by NumPy's own loop on the host and uploaded (rounds 1-5's default; what a
                           candidate generator of the caller's own costs)
      device_rng           device_rng_seed=...: the batch is drawn on the GPU (Philox), never crosses PCIe
      device_rng_prefetch  + prefetch_next=True: the next trial's batch is drawn behind this sweep and the next fit starts
                           its sweep inside itself (tgp_set_overlap 2)
    One process, one GPU, wall clock per trial (median of `reps` after one warm trial).  Outside the timed region of
    `value`; reported under "plugin"."""
    import turbo_amd as ta
    D, M = cfg["D"], cfg["M"]
    kern = ta.GPKernel(cfg["kind"], 1.0, ls, cfg["noise"])
    bounds = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(D)])
    out = {"workload": "C%d through HipGPSurrogate / %s / CandidateSweep(num_random=%d), one trial = construct_model (fixed theta) "
                       "+ construct_function + sweep" % (cfg["cfg"], cfg["acq"].upper(), M),
           "c_abi_ms_per_step": c_abi_ms, "reps": reps}
    def numpy_selector(num_points, latent_bounds):      # the reference's loop, by NumPy itself
        return np.hstack([np.random.uniform(lo, hi, size=(num_points, 1)) for _, lo, hi in latent_bounds.ordered])
    t0 = time.perf_counter()
    numpy_selector(M, bounds)
    out["numpy_draw_alone_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    ta.random_selector()(M, bounds)
    out["host_draw_alone_ms"] = (time.perf_counter() - t0) * 1e3      # (the library's all-host continuation of the same stream)
    for name, kw in (("host_draw", {}), ("host_draw_numpy", dict(gen_random=numpy_selector)), ("device_rng", dict(device_rng_seed=7)),
                     ("device_rng_prefetch", dict(device_rng_seed=7, prefetch_next=True))):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=kern, optimizer=None, normalize_y=True, alpha=1e-10),
                                training_iterations=1, dtype=cfg["dtype"], device=device, incremental=False)
        aux = ta.CandidateSweep(num_random=M, **kw)
        fac = {"ei": lambda: ta.EI(xi=cfg["param"]), "pi": lambda: ta.PI(xi=cfg["param"]), "ucb": lambda: ta.UCB(beta=cfg["param"])}[cfg["acq"]]()
        res = {}

        def trial(t):
            model, _ = sur.construct_model(t, X, y)
            args_ = () if cfg["acq"] == "ucb" else (inc,)
            f, _ = fac.construct_function(t, model, "min", *args_)
            res["x"], res["info"] = aux(bounds, f)
        np.random.seed(3000 + cfg["cfg"])
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            trial(0)
            ts = []
            for t in range(1, 1 + (1 if (name.startswith("host_draw") and M * D > 3e7) else reps)):   # (the two host variants: the same trials, so the same batches and the same max_acq)
                t1 = time.perf_counter()
                trial(t)
                ts.append(time.perf_counter() - t1)
        ms = float(np.median(ts)) * 1e3
        out[name] = {"ms_per_trial": ms, "evals_per_s": M / (ms * 1e-3), "vs_c_abi": ms / c_abi_ms if c_abi_ms else None,
                     "max_acq": float(res["info"]["max_acq"])}
        sur.close()
    return out


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, X, y, Xc, ls, budget_s=20.0):
    """The oracle (NumPy/SciPy port of the sklearn arithmetic the reference calls) on this box's
    host cores: fit timed in full, the sweep on a bounded candidate sample, extrapolated
    linearly in M (rows are independent)."""
    from oracle import gp_oracle as o
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    M = cfg["M"]
    t0 = time.perf_counter()
    om = o.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
    fit_s = time.perf_counter() - t0
    chunk = int(min(M, max(1024, (2 << 30) // (3 * 8 * cfg["N"]))))    # 3 M x N f64 temporaries <= 2 GiB
    done, sweep_s = 0, 0.0
    inc = float(y.min())
    while done < min(M, len(Xc)) and (sweep_s < budget_s or done == 0):
        part = Xc[done:done + chunk]
        t0 = time.perf_counter()
        o.sweep(om, part, cfg["acq"], "min", cfg["param"], inc)
        sweep_s += time.perf_counter() - t0
        done += len(part)
    step_s = fit_s + sweep_s * (M / done)
    return {"value": M / step_s, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": "oracle.fit in full (%.2f s incl. LML terms) + oracle.sweep on %d of %d candidates "
                      "(%.2f s), sweep extrapolated linearly to M" % (fit_s, done, M, sweep_s),
            "fit_ms": fit_s * 1e3, "sweep_evals_per_s": done / sweep_s, "host_cpus": os.cpu_count(),
            "cpu_model": cpu_model_name()}


def sklearn_leg(cfg, X, y, Xc, ls, budget_s=10.0):
    """The arithmetic exactly as the reference reaches it: scikit-learn's own
    GaussianProcessRegressor with fixed theta -- what SciKitGPSurrogate.construct_model builds
    (turbo/modules/surrogates.py:315-318) and ModelInstance.predict calls (:332-338) -- plus the
    restated EI/PI/UCB on its (mu, sigma).  Reported beside the port; None when scikit-learn is
    not importable on this box."""
    try:
        import sklearn
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process import kernels as K
    except ImportError:
        return None
    from oracle import gp_oracle as o
    if cfg["kind"] == "rbf":
        base = K.RBF(length_scale=ls, length_scale_bounds="fixed")
    else:
        nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[cfg["kind"]]
        base = K.Matern(length_scale=ls, nu=nu, length_scale_bounds="fixed")
    kern = K.ConstantKernel(1.0, "fixed") * base + K.WhiteKernel(cfg["noise"], "fixed")
    gpr = GaussianProcessRegressor(kernel=kern, alpha=1e-10, optimizer=None, normalize_y=True)
    t0 = time.perf_counter()
    gpr.fit(X, y)
    fit_s = time.perf_counter() - t0
    M = cfg["M"]
    chunk = int(min(M, max(1024, (2 << 30) // (4 * 8 * cfg["N"]))))    # sklearn holds ~4 M x N f64 temporaries
    inc = float(y.min())
    done, sweep_s = 0, 0.0
    while done < min(M, len(Xc)) and (sweep_s < budget_s or done == 0):
        part = Xc[done:done + chunk]
        t0 = time.perf_counter()
        mu, sd = gpr.predict(part, return_std=True)
        acq = o.acquisition(cfg["acq"], mu.flatten(), sd, "min", cfg["param"], inc)
        int(np.argmax(acq))
        sweep_s += time.perf_counter() - t0
        done += len(part)
    step_s = fit_s + sweep_s * (M / done)
    return {"value": M / step_s, "unit": "evals/s", "version": sklearn.__version__,
            "fit_ms": fit_s * 1e3, "sweep_evals_per_s": done / sweep_s,
            "sample": "GaussianProcessRegressor(optimizer=None).fit in full (%.2f s) + predict(return_std=True) "
                      "+ acquisition on %d of %d candidates (%.2f s), extrapolated linearly to M"
                      % (fit_s, done, M, sweep_s)}


def hyper_bench(args):
    """--config hyper: one evaluation of the hyper-parameter objective = tgp_fit_grad (kernel matrix,
    Cholesky, inverse factor, alpha, K^-1 = U U^T, the gradient's pairwise trace pass), what
    GaussianProcessRegressor.fit evaluates per L-BFGS-B step (sklearn _gpr.py:579-650, reached from
    turbo/modules/surrogates.py:313-318 whenever training_iterations > 0).  Inputs are C2's family
    (16D Matern-5/2 ARD) at N = --hyper-n.  Algorithmic flops per evaluation:
    N^3/3 (factor) + N^3/3 (inverse factor) + N^3/3 (lower triangle of U U^T) + N^2 (3D/2 + 20)."""
    import turbo_amd as ta
    N, D = args.hyper_n, 16
    cfg = dict(CONFIGS["c2"], N=N, M=1)
    X, y, ls = synth_train(cfg)
    gp = ta.NativeGP(0, "f64")
    call = lambda: gp.fit_grad(X, y, "matern52", 1.0, ls, 1e-4, 1e-10, True)
    import torch
    torch.cuda.synchronize()
    # torch's device initialisation is followed, some tens of ms later, by ONE 35-60 ms stall of whatever
    # call is then in flight (measured: tools/scratch runs with and without torch on the device; without
    # torch there is none) - at 0.1 ms per evaluation that lands inside a short timed region, so the
    # warm-up also lasts at least 0.5 s of wall clock
    t_w = time.perf_counter()
    n_w = 0
    while n_w < args.warmup or time.perf_counter() - t_w < 0.5:
        call()
        n_w += 1
    tim = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lml, grad = call()
        tim.append(gp.last_timings())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    call_dev_ms = float(np.median([t["fit_ms"] for t in tim]))      # a polled call (up to N = 4096): the WHOLE evaluation, from the kernels' own clock
    # The stage times come from event records between the launches, which a polled call does not make: a second, shorter
    # loop with the per-launch profiling on (tgp_profile_enable: round 5's events + stream synchronisation) supplies them.
    # `value` / `ms_per_step` belong to the default calls above.
    gp.profile_enable(True)
    tim = []
    for _ in range(3):
        call()
    t1 = time.perf_counter()
    n_ev = max(5, min(args.steps, 40))
    for _ in range(n_ev):
        call()
        tim.append(gp.last_timings())
    dt_events = (time.perf_counter() - t1) / n_ev
    gp.profile_enable(False)
    med = {k: float(np.median([t[k] for t in tim])) for k in tim[0] if k.endswith("_ms")}     # (the time slots of tgp_last_timings only)
    dev_ms = med["fit_ms"] + med["grad_kinv_ms"] + med["grad_pairwise_ms"] + med["grad_ard_ms"]
    flops = float(N) ** 3 + float(N) ** 2 * (1.5 * D + 20.0)
    ach = flops / (dev_ms * 1e-3) / 1e12
    kinv_flops = float(N) ** 3 / 3.0
    small = med["grad_kinv_ms"] <= 0.0   # N <= 128: fit and gradient are two one-workgroup launches timed together as fit_ms
    kinv = ({"name": "small_grad_kernel (one workgroup; timed inside fit_ms)", "ms": None, "algorithmic_flops": kinv_flops,
             "achieved": None, "frac": None} if small else
            {"name": "gemm_nt_glds_kernel<double, KN_UPPER_A, TM_LOWER> (K^-1 = U U^T)",
             "ms": med["grad_kinv_ms"], "algorithmic_flops": kinv_flops,
             "achieved": kinv_flops / (med["grad_kinv_ms"] * 1e-3) / 1e12,
             "frac": kinv_flops / (med["grad_kinv_ms"] * 1e-3) / 1e12 / PEAK_TFLOPS["f64"]})
    out = {
        "metric": "hyper-parameter objective evaluations/sec (log marginal likelihood + gradient, N training)",
        "value": 1.0 / dt, "unit": "evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "hyper: one tgp_fit_grad evaluation, 16D matern52 ARD, N=%d" % N, "N": N, "D": D},
        "device_ms": dev_ms, "stages_ms": med, "call_device_ms": call_dev_ms,
        "ms_per_step_with_events": dt_events * 1e3,
        "stages_note": "stages_ms / device_ms: from event records between the launches (tgp_profile_enable on, a separate loop of %d evaluations); "
                       "ms_per_step / value: the default calls (polled up to N = 4096, no events), call_device_ms their kernels' own start-to-doorbell time" % n_ev,
        "small_fit_phases_us": tim[-1].get("small_fit_phases_us") if small else None,   # N <= 128: inputs staged | K tile | factored | fit done | gradient done, from the kernel's start
        "roofline": {"bound": "mfma", "kernel": "whole evaluation (fit + K^-1 + trace pass)",
                     "achieved": ach, "peak": PEAK_TFLOPS["f64"], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS["f64"],
                     "traffic": None, "algorithmic_flops": flops,
                     "dominant_grad_kernel": kinv,
                     "fit": {"ms": med["fit_ms"], "algorithmic_flops": 2.0 * float(N) ** 3 / 3.0,
                             "frac": 2.0 * float(N) ** 3 / 3.0 / (med["fit_ms"] * 1e-3) / 1e12 / PEAK_TFLOPS["f64"]}},
    }
    if not args.no_cpu_baseline:
        from oracle import gp_oracle as o
        try:
            from threadpoolctl import threadpool_info
            threads = max([p_.get("num_threads", 1) for p_ in threadpool_info()] + [1])
        except Exception:
            threads = os.cpu_count() or 1
        t1 = time.perf_counter()
        olml, ograd = o.lml_and_grad(X, y, "matern52", 1.0, ls, 1e-4, 1e-10, True)
        port_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": 1.0 / port_s, "unit": "evals/s", "cores": int(threads), "kind": "port",
                               "sample": "one oracle.lml_and_grad evaluation at the same N (%.2f s)" % port_s,
                               "host_cpus": os.cpu_count(), "cpu_model": cpu_model_name(),
                               "lml_rel_err_vs_gpu": abs(lml - olml) / abs(olml),
                               "grad_max_rel_err_vs_gpu": float(np.max(np.abs(grad - ograd)) / max(1.0, float(np.abs(ograd).max())))}
        try:
            import sklearn
            from sklearn.gaussian_process import GaussianProcessRegressor
            from sklearn.gaussian_process import kernels as K
            kern = K.ConstantKernel(1.0) * K.Matern(length_scale=ls, nu=2.5) + K.WhiteKernel(1e-4)
            gpr = GaussianProcessRegressor(kernel=kern, alpha=1e-10, optimizer=None, normalize_y=True).fit(X, y)
            t1 = time.perf_counter()
            gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
            sk_s = time.perf_counter() - t1
            out["cpu_baseline"]["sklearn"] = {"value": 1.0 / sk_s, "unit": "evals/s", "version": sklearn.__version__,
                                              "sample": "one GaussianProcessRegressor.log_marginal_likelihood(theta, eval_gradient=True) (%.2f s)" % sk_s}
        except ImportError:
            pass
    print(json.dumps(out), flush=True)


# The host driver on this pool shares device memory between processes through dmabuf only; without this
# RCCL's intra-node transport (and any CUDA-tensor IPC) fails with `hipIpcGetMemHandle: invalid argument`.
# Must be in the environment before the HIP runtime loads, so it is set before torch is imported.
# GPU_MAX_HW_QUEUES: the runtime deals a process's streams onto this many hardware queues (4 by default) and streams on
# one queue run one after the other; the library keeps three streams per device beside torch's own.  turbo_amd sets the
# same default when it is imported, but torch initialises the runtime first here.
RCCL_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "GPU_MAX_HW_QUEUES": "8"}


def launch_ranks(argv, n):
    """`python bench.py --gpus N` (N > 1) with no RANK in the environment: start the N ranks as a child
    `python -m torch.distributed.run` on this very script (never os.exec*: nothing here has touched the GPU,
    but a child keeps that true whatever is imported later), let the ranks' stdout through -- rank 0 prints
    the one JSON line -- and return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k, v in RCCL_ENV.items():
        env.setdefault(k, v)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["hyper"])
    ap.add_argument("--hyper-n", type=int, default=4096, help="N of --config hyper")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default=None, choices=["f32", "f32x3", "f32h2"],
                    help="override the sweep arithmetic of an f32 configuration: f32x3 = f32 accuracy from three "
                         "bf16 planes on the bf16 matrix pipe (opt-in; the default and the headline stay f32)")
    ap.add_argument("--no-opt-in", action="store_true",
                    help="skip the extra, untimed-for-the-headline measurement of the opt-in f32h2 sweep")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: M candidates per GPU instead of one batch of M cut into shards")
    ap.add_argument("--overlap", type=int, default=0, choices=[0, 1, 2],
                    help="tgp_set_overlap for the TIMED steps: 0 (default) = the strictly serial schedule, which is what the "
                         "library and the plugin classes run unless asked otherwise -- the headline `value` belongs to it; "
                         "1 = candidate scaling + first cross-kernel inside the fit, 2 = + early contraction row tiles "
                         "(bit-identical results).  With 0 the opt-in schedule 2 is measured on a few steps OUTSIDE the timed "
                         "region and reported under `overlapped_schedule`")
    ap.add_argument("--no-plugin", action="store_true",
                    help="skip the untimed-for-the-headline `plugin` leg (the same step through HipGPSurrogate / EI|PI|UCB / "
                         "CandidateSweep, three candidate sources)")
    ap.add_argument("--shard-of", type=int, default=1, metavar="G",
                    help="ONE GPU running rank 0's share of a G-way strong split (ceil(M / G) candidates, no exchange): "
                         "the compute side of the scaling curve measured where no multi-GPU node is at hand; the line "
                         "says so (config.shard_of) and is NOT a multi-GPU measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))
    for k, v in RCCL_ENV.items():
        os.environ.setdefault(k, v)
    if args.config == "hyper":
        assert args.gpus == 1, "--config hyper is a single-GPU measurement"
        return hyper_bench(args)
    cfg = dict(CONFIGS[args.config])
    if args.dtype is not None:
        assert cfg["dtype"] == "f32", "--dtype only applies to the f32 configurations (c3, c4)"
        cfg["dtype"] = args.dtype

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    # BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    # TEST HOOK (tests/test_bench_launcher.py only): "module:Class" of a stand-in context under tests/, so
    # the launcher, the rank plumbing and the output line can be exercised where there is no GPU.  The line
    # then says so ("standin") and carries no value: it is never a measurement.
    standin = os.environ.get("BENCH_TEST_CONTEXT")
    if standin:
        assert backend == "gloo", "BENCH_TEST_CONTEXT is a CPU rehearsal: BENCH_BACKEND=gloo"
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if not standin:
        torch.cuda.set_device(local_rank)
    devname = "cpu" if standin else "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    import turbo_amd as ta

    X, y, ls = synth_train(cfg)
    if args.shard_of > 1:
        assert world == 1 and not args.weak, "--shard-of is a single-process measurement of a strong split"
        Xc, m_local, offset, _ = shard_candidates(cfg, 0, args.shard_of, False)
        m_job = m_local          # what THIS process sweeps; value = its own throughput
    else:
        Xc, m_local, offset, m_job = shard_candidates(cfg, rank, world, args.weak)
    inc = float(y.min())
    if standin:
        import importlib
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        mod, cls = standin.split(":")
        make_context = getattr(importlib.import_module(mod), cls)
    else:
        make_context = ta.NativeGP
    gp = make_context(local_rank, cfg["dtype"])
    if hasattr(gp, "set_overlap"):
        gp.set_overlap(args.overlap)
    gp.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
    # candidates resident in HBM before the timed region (a torch tensor owns the memory)
    cand = torch.from_numpy(Xc).to(devname)
    if standin:
        gp.set_candidates(Xc)
    elif m_local > 0:
        gp.set_candidates_dev(cand.data_ptr(), m_local, keepalive=cand)
    rec = None
    if world > 1 and not standin:
        rec = torch.zeros(cfg["D"] + 2, dtype=torch.float64, device="cuda:%d" % local_rank)
        gp.set_winner_out(rec.data_ptr(), offset, keepalive=rec)
    step = build_step(gp, cfg, X, y, ls, inc, world, offset, rec, backend, m_local)

    def fence():
        if dist is not None:
            dist.barrier()
        if not standin:
            torch.cuda.synchronize()

    # The per-launch HIP events stay on for the WHOLE run (warm-up, timed steps, the serial-schedule steps and the
    # full-vector sweeps behind them): the timed region's totals are differences of two reads, and the whole run's
    # average is what a rocprofv3 --kernel-trace --stats summary of this command shows for the same kernel
    # (roofline.whole_run: under --overlap the launches are not all alike, so the two averages differ by design).
    gp.profile_enable(True)
    gp.profile_reset()
    for _ in range(args.warmup):
        step()

    def prof_now():
        p = gp.profile_read()
        p["trmm_flops"] = gp.last_timings()["trmm_flops"] if hasattr(gp, "last_timings") else 0.0
        return p
    prof0 = prof_now()
    fit_ms, sweep_ms = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        p = gp.profile_read()
        fit_ms.append(p["last_fit_ms"])
        sweep_ms.append(p["last_sweep_ms"])
    fence()
    dt = time.perf_counter() - t0
    prof1 = prof_now()
    prof = {k: (prof1[k] - prof0[k]) if k in ("trmm_launches", "trmm_ms", "kstar_launches", "kstar_ms", "trmm_flops") else prof1[k] for k in prof1}
    trmm_flops = prof["trmm_flops"]
    # Outside the timed region: a few steps under the strictly SERIAL schedule, so that the line also carries the fit's and
    # the sweep's own durations (under --overlap the fit's event bracket contains the sweep's front): what roofline.fit and
    # amdahl_bound are formed from.  Never part of `value`.
    serial = None
    overlapped = None
    if args.overlap == 0 and hasattr(gp, "set_overlap") and not standin and m_local > 0:
        # the opt-in schedule (tgp_set_overlap(2): CandidateSweep(prefetch_next=True) arms it) on the same handle, outside
        # the timed region: never part of `value`
        gp.set_overlap(2)
        ts, fs, ss = [], [], []
        for _ in range(4):
            t1 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t1)
            p = gp.profile_read()
            fs.append(p["last_fit_ms"])
            ss.append(p["last_sweep_ms"])
        gp.set_overlap(0)
        overlapped = {"mode": 2, "ms_per_step": float(np.median(ts[1:])) * 1e3, "fit_ms_incl_front": float(np.median(fs[1:])),
                      "sweep_ms_behind_front": float(np.median(ss[1:])),
                      "evals_per_s": m_job / float(np.median(ts[1:])),
                      "note": "tgp_set_overlap(2) on the same handle, 3 steps outside the timed region (rank 0's own clock): the front of the "
                              "sweep runs inside the fit; opt-in (CandidateSweep(prefetch_next=True)), NOT the headline"}
    if args.overlap > 0 and hasattr(gp, "set_overlap") and not standin:
        gp.set_overlap(0)
        ts, fs, ss = [], [], []
        for _ in range(4):
            t1 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t1)
            p = gp.profile_read()
            fs.append(p["last_fit_ms"])
            ss.append(p["last_sweep_ms"])
        gp.set_overlap(args.overlap)
        serial = {"ms_per_step": float(np.median(ts[1:])) * 1e3, "fit_ms": float(np.median(fs[1:])), "sweep_ms": float(np.median(ss[1:])),
                  "note": "--overlap 0 on the same handle, 3 steps outside the timed region (rank 0's own clock)"}
    chunk, n_pad = gp.sweep_geometry()
    # also the variant that hands the whole (M,) acquisition vector to the host
    # (outside the timed region; wall clock around the call, D2H included)
    full_s = []
    for _ in range(3 if m_local > 0 else 0):
        t1 = time.perf_counter()
        gp.sweep(ACQ_ENUM[cfg["acq"]], -1.0, inc, cfg["param"], want_acq=True)
        full_s.append(time.perf_counter() - t1)
    full_vec_s = float(np.median(full_s)) if full_s else float("inf")
    prof_all = prof_now()
    gp.profile_enable(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64,
                         device=("cuda:%d" % local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Beside the headline (never part of it): the same step with the OPT-IN sweep arithmetic 'f32h2'
    # (f32 accuracy from two scaled fp16 planes on the fp16 matrix pipe), same
    # shard, same protocol, and whether it picks the same winner.  Reported under "opt_in".
    opt_in = None
    if cfg["dtype"] == "f32" and not args.no_opt_in and not standin:
        r_main = step()
        gp2 = ta.NativeGP(local_rank, "f32h2")
        gp2.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
        if m_local > 0:
            gp2.set_candidates_dev(cand.data_ptr(), m_local, keepalive=cand)
        rec2 = None
        if world > 1:
            rec2 = torch.zeros(cfg["D"] + 2, dtype=torch.float64, device="cuda:%d" % local_rank)
            gp2.set_winner_out(rec2.data_ptr(), offset, keepalive=rec2)
        step2 = build_step(gp2, cfg, X, y, ls, inc, world, offset, rec2, backend, m_local)
        for _ in range(max(args.warmup, 1)):
            r2 = step2()
        n2 = max(1, min(args.steps, 5))
        fit2, sweep2 = [], []
        fence()
        t1 = time.perf_counter()
        for _ in range(n2):
            r2 = step2()
            p2 = gp2.profile_read()
            fit2.append(p2["last_fit_ms"])
            sweep2.append(p2["last_sweep_ms"])
        fence()
        dt2 = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dt2], dtype=torch.float64,
                             device=("cuda:%d" % local_rank) if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        key = "job_best_idx" if world > 1 else "best_idx"
        opt_in = {"dtype": "f32h2", "steps": n2, "ms_per_step": dt2 / n2 * 1e3, "value": m_job / (dt2 / n2),
                  "unit": "evals/s", "fit_ms": float(np.median(fit2)), "sweep_ms": float(np.median(sweep2)),
                  "same_winner_as_f32": bool(int(r2[key]) == int(r_main[key])),
                  "note": "opt-in sweep arithmetic: f32 accuracy from two scaled fp16 planes, three fp16 MFMAs per "
                          "product (DESIGN.md section 4, tests/test_gpu_configs.py); NOT the headline"}
        del gp2

    if rank == 0:
        N = cfg["N"]
        total = m_job
        ms_per_step = dt / args.steps * 1e3
        # dominant kernel: trmm_sumsq.  Algorithmic flops per launch = N^2 per candidate
        # (the triangular solve's N(N+1)/2 FMA) x the candidates of one launch.
        # With --overlap 2 the fit has already contracted the first row tiles of the step's first launch: the library
        # counts every timed launch's OWN algorithmic flops (rows^2 per candidate over the rows it covered).
        launches = max(prof["trmm_launches"], 1)
        cands_per_launch = m_local * args.steps / launches
        avg_ms = prof["trmm_ms"] / launches
        flops_per_launch = (trmm_flops / launches) if trmm_flops > 0 else cands_per_launch * float(N) * N
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # f32x3: six bf16 MFMAs per f32 product, priced against the dense bf16 peak (6 x the algorithmic flops)
        x3 = cfg["dtype"] in ("f32x3", "f32h2")
        peak = 2516.6 if x3 else PEAK_TFLOPS[cfg["dtype"]]
        if x3:
            achieved *= 6.0 if cfg["dtype"] == "f32x3" else 3.0   # MFMA products per algorithmic multiply
        # HBM traffic of the dominant kernel: NOT a quantity of this run -- PMC counters need passes of
        # their own (MI355X_MICROARCH.md), so the figure is the one profiles/collect.sh measured for
        # this config with rocprofv3 --pmc and profiles/summarize_pmc.py wrote down
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath) and not x3:
            with open(tpath) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
            traffic_source = "profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this command)" % args.config
        fit_med, sweep_med = float(np.median(fit_ms)), float(np.median(sweep_ms))
        fit_own = serial["fit_ms"] if serial else fit_med          # the fit / the sweep alone (serial schedule)
        sweep_own = serial["sweep_ms"] if serial else sweep_med
        # whole-step algorithmic flops: per candidate N^2 + 3 N D + 4 N, the fit N^3 / 3 + N^2 (3 D / 2 + 8)
        Dd = cfg["D"]
        step_flops = m_local * (float(N) * N + 3.0 * N * Dd + 4.0 * N) + float(N) ** 3 / 3.0 + float(N) ** 2 * (1.5 * Dd + 8.0)
        fit_flops = float(N) ** 3 / 3.0
        # the replicated fit bounds fixed-M scaling (Amdahl): step(1 GPU) / (fit + sweep / G)
        sweep_1gpu = sweep_own * (1 if (args.weak or world == 1) else world)
        amdahl = {str(g): (fit_own + sweep_1gpu) / (fit_own + sweep_1gpu / g) for g in (2, 4, 8)}
        out = {
            "metric": "acquisition evals/sec (M candidates, N training) + GP-fit ms",
            "value": total / (dt / args.steps),
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # (the N = 1 line is the baseline of the default N > 1 runs, which cut the SAME batch into shards: "strong"
            # unless --weak gives every GPU a batch of its own)
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": "C%d: %dD %s%s, N=%d observed, M=%d candidates %s, %s, fit + sweep per step"
                                   % (cfg["cfg"], cfg["D"], cfg["kind"], " ARD" if cfg["ard"] else "", N,
                                      cfg["M"], "per GPU" if (args.weak or world == 1) else "in all (ceil(M/G) per GPU)",
                                      cfg["acq"].upper()),
                       "N": N, "D": cfg["D"], "M_per_gpu": m_local, "M_total": total,
                       **({"shard_of": args.shard_of,
                           "note": "ONE GPU running rank 0's share of a %d-way strong split of M=%d; no exchange; not a multi-GPU measurement"
                                   % (args.shard_of, cfg["M"])} if args.shard_of > 1 else {}),
                       "parallelism": "candidate-shard x%d (contiguous), fit replicated, one all-gather of winners" % world},
            "schedule": "serial (tgp_set_overlap 0: the library's and the plugin classes' default)" if args.overlap == 0 else
                        "overlapped (tgp_set_overlap %d: opt-in; the default run of this script measures the serial schedule)" % args.overlap,
            "overlap": args.overlap,
            "overlapped_schedule": overlapped,
            "overlap_note": None if args.overlap == 0 else
            "tgp_set_overlap(%d): the front of the resident batch's sweep (candidate scaling, first cross-kernel%s) runs INSIDE "
            "tgp_fit on a third stream, so fit_ms (events around the fit) is longer and sweep_ms shorter than under --overlap 0; "
            "ms_per_step is the honest total.  serial_schedule holds the two intervals under --overlap 0 (outside the timed region); roofline.fit and amdahl_bound use those"
            % (args.overlap, ", early row tiles of the first contraction" if args.overlap > 1 else ""),
            "serial_schedule": serial,
            "fit_ms": fit_med,
            "sweep_ms": sweep_med,
            "sweep_evals_per_s": (total / (sweep_med * 1e-3)) if sweep_med > 0 else None,
            "amdahl_bound": {"speedup_max_by_gpus": amdahl, "fit_ms_replicated": fit_own, "sweep_ms_one_gpu": sweep_1gpu,
                             "note": "fixed M over G GPUs with the fit replicated: (fit + sweep) / (fit + sweep / G) of the SERIAL schedule, before any exchange"},
            "sweep_full_vector_evals_per_s_per_gpu": m_local / full_vec_s,
            "roofline": {"bound": "mfma", "kernel": ("trmm_sumsq_bf16x3_kernel (6 bf16 MFMA flops per algorithmic flop, bf16 dense peak)" if cfg["dtype"] == "f32x3" else "trmm_sumsq_f16x2_kernel (3 fp16 MFMA flops per algorithmic flop, fp16 dense peak)") if x3 else "trmm_sumsq_glds[_big]_kernel",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "step_frac": None if x3 else step_flops / (ms_per_step * 1e-3) / 1e12 / peak,
                         "step_algorithmic_flops": step_flops,
                         "fit": {"ms": fit_own, "algorithmic_flops": fit_flops, "dtype": "f64",
                                 "frac": fit_flops / (fit_own * 1e-3) / 1e12 / PEAK_TFLOPS["f64"] if fit_own > 0 else None,
                                 "note": "the fit alone (serial schedule)"},
                         "launches": int(prof["trmm_launches"]), "avg_launch_ms": avg_ms,
                         "whole_run": {"launches": int(prof_all["trmm_launches"]),
                                       "avg_launch_ms": prof_all["trmm_ms"] / max(prof_all["trmm_launches"], 1),
                                       "note": "every contraction launch of this process (warm-up, timed, serial-schedule and full-vector "
                                               "sweeps): the average a rocprofv3 --kernel-trace --stats summary of this command shows"},
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "candidates_per_launch": cands_per_launch, "chunk": chunk,
                         "kstar_avg_ms": prof["kstar_ms"] / max(prof["kstar_launches"], 1)},
        }
        if opt_in is not None:
            out["opt_in"] = opt_in
        if standin:
            out["standin"] = "%s: a CPU stand-in context of the test suite, NOT a measurement" % standin
            out["value"] = None
        if world == 1 and not args.no_plugin and not standin and args.shard_of == 1 and args.dtype is None:
            out["plugin"] = plugin_leg(cfg, X, y, ls, inc, local_rank, ms_per_step)
        if world == 1 and not args.no_cpu_baseline and not standin:
            out["cpu_baseline"] = cpu_baseline(cfg, X, y, Xc, ls)
            sk = sklearn_leg(cfg, X, y, Xc, ls)
            if sk is not None:
                out["cpu_baseline"]["sklearn"] = sk
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
