#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel and apply
the gfx950 corrections of MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE
reports half of the bytes of a wide coalesced read -> x2; WRITE_SIZE is exact.

usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <out.json>
"""
import csv
import json
import sys


def per_kernel(path, counter):
    agg = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


def main():
    fetch, write, needle, out = sys.argv[1:5]
    f = per_kernel(fetch, "FETCH_SIZE")
    w = per_kernel(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fetch_b = 2.0 * sf * 1024 / max(nf, 1)      # x2: gfx950 FETCH_SIZE under-count
        write_b = sw * 1024 / max(nw, 1)
        rows.append(dict(kernel=k, launches=nf, fetch_bytes_per_launch=fetch_b,
                         write_bytes_per_launch=write_b, hbm_bytes_per_launch=fetch_b + write_b))
    sel = [r for r in rows if needle in r["kernel"]]
    res = dict(kernels=rows, selected=sel[0] if sel else None,
               hbm_bytes_per_launch=sel[0]["hbm_bytes_per_launch"] if sel else None,
               note="FETCH_SIZE x2 (gfx950 correction), KiB -> bytes, separate --pmc passes")
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for r in rows:
        print("%-100s n=%4d fetch %10.3f MB  write %10.3f MB" % (r["kernel"][:100], r["launches"],
              r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
