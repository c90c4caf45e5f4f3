#!/usr/bin/env python3
"""Reads a TGP_STAMP_FILE dump (fit_kernels.hip: in-kernel wall_clock64 stamps of every fused panel
launch, 10 ns ticks) and prints per launch: workgroup 0's phases and the spread of the other
workgroups' start / end times, all relative to the launch's first stamp."""
import sys

import numpy as np

S = 2048
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, S)
prev_end = None
for i, row in enumerate(a):
    wg = row[8:].reshape(-1, 2)
    live = wg[:, 0] > 0
    if not live.any():
        continue
    t0 = int(wg[live, 0].min())
    rel = lambda v: (int(v) - t0) / 100.0   # us
    w0 = wg[0]
    others = wg[1:][live[1:]]
    gap = "" if prev_end is None else " gap %.2f" % ((t0 - prev_end) / 100.0)
    line = "launch %3d mode %d wgs %4d%s | wg0 start %.2f loads %.2f inLDS %.2f factored %.2f end %.2f" % (
        i, i & 1, int(live.sum()), gap, rel(w0[0]), rel(row[0]) if row[0] else -1, rel(row[1]), rel(row[2]), rel(w0[1]))
    if len(others):
        done = others[others[:, 1] > 0]
        line += " | others start %.2f..%.2f end %.2f..%.2f dur mean %.2f max %.2f" % (
            rel(others[:, 0].min()), rel(others[:, 0].max()), rel(done[:, 1].min()), rel(done[:, 1].max()),
            float(np.mean(done[:, 1].astype(np.int64) - done[:, 0].astype(np.int64))) / 100.0,
            float(np.max(done[:, 1].astype(np.int64) - done[:, 0].astype(np.int64))) / 100.0)
    print(line)
    prev_end = int(max(wg[live, 1].max(), w0[1]))
