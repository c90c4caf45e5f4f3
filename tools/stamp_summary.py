#!/usr/bin/env python3
"""Reads a TGP_STAMP_FILE dump (fit_kernels.hip: in-kernel wall_clock64 stamps of every fused panel
launch, 10 ns ticks) and prints per launch: workgroup 0's phases and the spread of the other
workgroups' start / end times, all relative to the launch's first stamp.

Round 6: workgroup 0 also leaves WHERE it ran (HW_ID + XCC_ID), and <file>.cus holds the CUs each of the fit's streams
reaches (main | background | third: 16384 probe workgroups each).  With it every launch line says which CU the pivot
workgroup sat on and whether the background stream's GEMMs can run there, and the summary at the end splits the factor
phase (block in LDS -> factored and inverted) by that -- the question the 20-26 us panels of round 5 left open."""
import os
import sys

import numpy as np

S = 2048
PROBE = 16384


def where(v):
    """(xcc, se, sh, cu) of a stamp word: HW_ID in the low word (cu [11:8], sh [12], se [15:13]), XCC_ID in the high one"""
    hw, xcc = int(v) & 0xffffffff, (int(v) >> 32) & 0xf
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)


a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, S)
sets = None
if os.path.exists(sys.argv[1] + ".cus"):
    p = np.fromfile(sys.argv[1] + ".cus", dtype=np.uint64).reshape(3, PROBE)
    sets = [set(where(v) for v in row if int(v) >> 63) for row in p]
    print("CUs reached: main stream %d, background stream %d, third stream %d (of %d seen in all)" % (
        len(sets[0]), len(sets[1]), len(sets[2]), len(sets[0] | sets[1] | sets[2])))
prev_end = None
factor_us = {True: [], False: []}
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
    if row[3]:
        cu = where(row[3])
        in_bg = sets is not None and cu in sets[1]
        line += " | wg0 on xcc %d se %d sh %d cu %2d%s" % (cu + (" (a background-stream CU)" if in_bg else " (NOT a background-stream CU)" if sets else "",))
        if row[1] and row[2]:
            factor_us[in_bg].append((i, (int(row[2]) - int(row[1])) / 100.0))
    if len(others):
        done = others[others[:, 1] > 0]
        line += " | others start %.2f..%.2f end %.2f..%.2f dur mean %.2f max %.2f" % (
            rel(others[:, 0].min()), rel(others[:, 0].max()), rel(done[:, 1].min()), rel(done[:, 1].max()),
            float(np.mean(done[:, 1].astype(np.int64) - done[:, 0].astype(np.int64))) / 100.0,
            float(np.max(done[:, 1].astype(np.int64) - done[:, 0].astype(np.int64))) / 100.0)
    print(line)
    prev_end = int(max(wg[live, 1].max(), w0[1]))
if sets is not None:
    for key, name in ((True, "pivot workgroup on a CU the background stream reaches"), (False, "pivot workgroup on a CU it does not reach")):
        v = factor_us[key]
        if v:
            d = np.array([x[1] for x in v])
            slow = [x[0] for x in v if x[1] > 15.0]
            print("factor phase, %s: %d launches, median %.2f us, max %.2f us, slower than 15 us: %d %s" % (
                name, len(v), float(np.median(d)), float(d.max()), len(slow), slow))
        else:
            print("factor phase, %s: no launches" % name)
