#!/usr/bin/env python3
"""Why an evaluation of the hyper-parameter objective costs more beside two others than alone: the same
tgp_fit_grad on pooled workers, first ONE worker alone, then THREE side by side (a host thread each), for
rocprofv3 --kernel-trace; `analyse` splits every hardware queue's kernels into evaluations (at each
fit_prologue_kernel) and prints, per phase, the kernels' own time and the gaps between them.

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/trace_side_by_side.py run [N] [evals]
    python3 tools/trace_side_by_side.py analyse out/*/*_kernel_trace.csv"""
import collections
import csv
import os
import sys
import threading
import time

import numpy as np


def run(N, evals):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import turbo_amd as ta
    rng = np.random.RandomState(N + 8)
    X = rng.uniform(0, 1, (N, 8))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    gp = ta.NativeGP(0, "f64")
    k = ta.GPKernel("matern52", 1.0, float(np.sqrt(8 / 6.0)), 1e-2)

    def loop(w, n, out, i):
        t0 = time.perf_counter()
        for _ in range(n):
            w.fit_grad(X, y, k.kind, k.constant, k.length_scale, k.noise_level, 1e-10, True)
        out[i] = (time.perf_counter() - t0) / n * 1e3

    with gp.workers(3) as ws:
        for w in ws:
            loop(w, 20, [0], 0)
        out = [0.0]
        loop(ws[0], evals, out, 0)
        print("alone: %.3f ms per evaluation" % out[0], flush=True)
        time.sleep(0.05)     # the phases are told apart by this hole in the trace
        out = [0.0] * 3
        th = [threading.Thread(target=loop, args=(ws[i], evals, out, i)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        print("three side by side: %s ms per evaluation" % " ".join("%.3f" % v for v in out), flush=True)


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # phases: split at the largest hole between two consecutive kernels of the whole trace after the warm-up
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    run_end = np.maximum.accumulate(np.array(ends))
    skip = len(rows) // 8                                       # (the warm-up's own pauses are not the phase boundary)
    i = skip + int(np.argmax(np.array(starts[skip + 1:]) - run_end[skip:-1]))
    cut = starts[i + 1]
    for phase, sel in (("alone", [r for r in rows if int(r["Start_Timestamp"]) < cut]), ("three side by side", [r for r in rows if int(r["Start_Timestamp"]) >= cut])):
        byq = collections.defaultdict(list)
        for r in sel:
            byq[r["Queue_Id"]].append(r)
        print("== %s" % phase)
        for q, l in sorted(byq.items()):
            idx = [i for i, r in enumerate(l) if "fit_prologue" in r["Kernel_Name"]]
            if len(idx) < 30:
                continue
            busy, gaps, span, nk = [], [], [], []
            per_kernel = collections.defaultdict(list)
            for a, b in zip(idx[10:-1], idx[11:]):
                ev = l[a:b]
                s = [int(r["Start_Timestamp"]) for r in ev]
                e = [int(r["End_Timestamp"]) for r in ev]
                busy.append(sum(y - x for x, y in zip(s, e)) / 1e3)
                gaps.append(sum(max(0, s[i + 1] - e[i]) for i in range(len(ev) - 1)) / 1e3)
                span.append((e[-1] - s[0]) / 1e3)
                nk.append(len(ev))
                for r, x, y2 in zip(ev, s, e):
                    per_kernel[r["Kernel_Name"].replace("tgp::", "").replace("void ", "").split("(")[0][:48]].append((y2 - x) / 1e3)
            print("queue %s: %d evaluations of %.0f launches: span %.0f us = kernels %.0f + gaps %.0f (%.1f us per gap)" %
                  (q, len(busy), np.mean(nk), np.mean(span), np.mean(busy), np.mean(gaps), np.mean(gaps) / max(np.mean(nk) - 1, 1)))
            top = sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]))[:8]
            for name, d in top:
                print("      %-48s %5.1f launches per evaluation, %6.1f us each" % (name, len(d) / len(busy), np.mean(d)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1000, int(sys.argv[3]) if len(sys.argv) > 3 else 150)
    else:
        analyse(sys.argv[2])
