#!/usr/bin/env python3
"""Per-queue timeline of the LAST step (from its fit_prologue_kernel on) of a rocprofv3 --kernel-trace csv:
which kernels ran on which hardware queue, when (us from the step's start) and for how long.

    python tools/trace_timeline.py run_kernel_trace.csv [--all]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "fit_prologue" in r["Kernel_Name"]]
    seg = rows[idx[-1]:] if idx else rows
    t0 = int(seg[0]["Start_Timestamp"])
    byq = collections.defaultdict(list)
    for r in seg:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3
        e = (int(r["End_Timestamp"]) - t0) / 1e3
        name = r["Kernel_Name"].replace("tgp::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        byq[r["Queue_Id"]].append((s, e, name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
    for q, l in sorted(byq.items()):
        busy = sum(e - s for s, e, _, _ in l)
        print("queue %s: %d kernels, busy %.0f us, span %.0f .. %.0f us" % (q, len(l), busy, l[0][0], l[-1][1]))
    print()
    short = "--all" not in sys.argv
    for q, l in sorted(byq.items()):
        print("---- queue %s" % q)
        for s, e, n, wg in l:
            if short and e - s < 30 and not any(k in n for k in ("kstar", "trmm", "prep", "finalize", "argmax")):
                continue
            print("%9.0f  +%7.0f us  %-60s %6d wgs" % (s, e - s, n, wg))


if __name__ == "__main__":
    main()
