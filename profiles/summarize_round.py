#!/usr/bin/env python3
"""Copy the judged summaries of one `profiles/collect.sh` run into profiles/.

usage: python3 profiles/summarize_round.py <OUT dir of collect.sh> <tag, e.g. r01>

Writes  profiles/<tag>_c3_kernel_stats.csv, <tag>_c3_bench_under_rocprof.json,
        <tag>_c{1,2,3,4}_bench.json, <tag>_c3_pmc_sq.json and traffic_c3.json (what bench.py
        reads for roofline.traffic), and prints the numbers the README quotes.
The dominant kernel is picked as the trmm_sumsq kernel with the largest total duration.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("nothing matches " + pattern)
    return hits[0]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    stats = one(os.path.join(out, "stats", "**", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    trmm = max((r for r in rows if "trmm_sumsq" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
    name = trmm["Name"]
    print("dominant kernel:", name)
    print("  rocprof average %.5f ms over %s launches (%s %% of kernel time)"
          % (float(trmm["AverageNs"]) / 1e6, trmm["Calls"], trmm["Percentage"]))
    shutil.copy(stats, os.path.join(HERE, "%s_c3_kernel_stats.csv" % tag))
    under = json.load(open(os.path.join(out, "bench_stats.json")))
    print("  HIP-event average of the same run %.5f ms" % under["roofline"]["avg_launch_ms"])
    shutil.copy(os.path.join(out, "bench_stats.json"), os.path.join(HERE, "%s_c3_bench_under_rocprof.json" % tag))

    fetch = one(os.path.join(out, "pmc_fetch", "**", "*_counter_collection.csv"))
    write = one(os.path.join(out, "pmc_write", "**", "*_counter_collection.csv"))
    needle = name.split("(")[0].replace("void ", "")
    subprocess.check_call([sys.executable, os.path.join(HERE, "summarize_pmc.py"), fetch, write, needle,
                           os.path.join(HERE, "traffic_c3.json")], stdout=subprocess.DEVNULL)
    t = json.load(open(os.path.join(HERE, "traffic_c3.json")))
    print("  traffic per launch: %.1f MB" % (t["hbm_bytes_per_launch"] / 1e6))

    sq = one(os.path.join(out, "pmc_sq", "**", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sq)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {}
    for k, v in agg.items():
        if "trmm_sumsq" in k or "kstar" in k:
            d = {c: sum(x) / len(x) for c, x in v.items()}
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / d["dur_ns"]
            d["mfma_busy_frac_of_simd_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
            res[k] = d
            print("  %-60s clock %.2f GHz  MFMA busy %.3f  LDS conflicts %.0f"
                  % (k[:60], d["clock_GHz"], d["mfma_busy_frac_of_simd_cycles"], d.get("SQ_LDS_BANK_CONFLICT", 0)))
    json.dump(res, open(os.path.join(HERE, "%s_c3_pmc_sq.json" % tag), "w"), indent=1)

    for c in ("c1", "c2", "c3", "c4"):
        src = os.path.join(out, "bench_%s.json" % c)
        if not os.path.exists(src) or os.path.getsize(src) == 0:
            print(c, "missing")
            continue
        shutil.copy(src, os.path.join(HERE, "%s_%s_bench.json" % (tag, c)))
        d = json.load(open(src))
        cb = d.get("cpu_baseline")
        print("%s  %.2f ms/step  fit %.2f  sweep %.2f  %.3f M evals/s  frac %.4f (%.1f TFLOP/s)  kstar %.1f us%s"
              % (c, d["ms_per_step"], d["fit_ms"], d["sweep_ms"], d["value"] / 1e6, d["roofline"]["frac"],
                 d["roofline"]["achieved"], d["roofline"]["kstar_avg_ms"] * 1e3,
                 "  cpu port %.0f ev/s (x%.0f), sklearn %.0f ev/s" % (cb["value"], d["value"] / cb["value"],
                                                                    cb.get("sklearn", {}).get("value", float("nan")))
                 if cb else ""))

    # round-2 extras, when the collection has them
    extras = [("bench_hyper_%s.json" % n, "%s_hyper_n%s_bench.json" % (tag, n)) for n in (32, 128, 512, 2048, 4096)]
    extras += [("latency_small.jsonl", "%s_latency_small.jsonl" % tag), ("fit_sizes.jsonl", "%s_fit_sizes.jsonl" % tag),
               ("mfma_f64_peak.txt", "%s_mfma_f64_peak_run.txt" % tag), ("gradient_stage.jsonl", "%s_gradient_stage.jsonl" % tag), ("trial_loop.jsonl", "%s_trial_loop.jsonl" % tag),
               ("bench_c3_f32x3.json", "%s_c3_f32x3_bench.json" % tag), ("bench_c4_f32x3.json", "%s_c4_f32x3_bench.json" % tag),
               ("bench_c3_f32h2.json", "%s_c3_f32h2_bench.json" % tag), ("bench_c4_f32h2.json", "%s_c4_f32h2_bench.json" % tag),
               ("split_accuracy.jsonl", "%s_split_accuracy.jsonl" % tag)]
    for src, dst in extras:
        sp = os.path.join(out, src)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            shutil.copy(sp, os.path.join(HERE, dst))
    for sub, dst in (("stats_hyper", "%s_hyper_n4096_kernel_stats.csv" % tag), ("stats_c1", "%s_c1_kernel_stats.csv" % tag)):
        hits = sorted(glob.glob(os.path.join(out, sub, "**", "*_kernel_stats.csv"), recursive=True))
        if hits:
            shutil.copy(hits[0], os.path.join(HERE, dst))
    for n in (32, 128, 512, 2048, 4096):
        sp = os.path.join(out, "bench_hyper_%s.json" % n)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            d = json.load(open(sp))
            cb = d.get("cpu_baseline", {})
            print("hyper N=%d  %.3f ms/evaluation  stages %s  sklearn %.2f ms" % (n, d["ms_per_step"], {k: round(v, 3) for k, v in d["stages_ms"].items() if v},
                  1e3 / cb["sklearn"]["value"] if "sklearn" in cb else float("nan")))


if __name__ == "__main__":
    main()
