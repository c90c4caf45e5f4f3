#!/usr/bin/env python3
"""Round-3 summaries of one `profiles/collect_r3.sh` run: for EVERY BASELINE config (c1..c4) the
rocprofv3 per-kernel stats, the bench line printed under the profiler, the PMC traffic of the
dominant kernel (traffic_<cfg>.json, what bench.py quotes as roofline.traffic) and the SQ / GRBM
counters (clock, MFMA pipe busy, LDS bank conflicts); the hyper-parameter objective's and the small
path's kernel stats; the un-profiled bench lines and tables.

usage: python3 profiles/summarize_round3.py <OUT dir of collect_r3.sh> <tag, e.g. r03> [--on-box]
  --on-box  (run by collect_r3.sh on the GPU box) writes the json summaries INTO <OUT>/summ so the raw
            counter csv files need not travel; without it the summaries are copied into profiles/.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_pmc  # noqa: E402


PMC_MIN_NS = 200000      # shortest kernel whose PMC-run duration is trusted for clock / busy ratios


def first(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def traffic(out, cfg, needle, dst):
    fetch = first(os.path.join(out, "pmc_fetch_" + cfg, "**", "*_counter_collection.csv"))
    write = first(os.path.join(out, "pmc_write_" + cfg, "**", "*_counter_collection.csv"))
    if not fetch or not write:
        return None
    f = summarize_pmc.per_kernel(fetch, "FETCH_SIZE")
    w = summarize_pmc.per_kernel(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fb = 2.0 * sf * 1024 / max(nf, 1)
        wb = sw * 1024 / max(nw, 1)
        rows.append(dict(kernel=k, launches=nf, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                         hbm_bytes_per_launch=fb + wb))
    sel = [r for r in rows if needle in r["kernel"]]
    res = dict(kernels=rows, selected=sel[0] if sel else None,
               hbm_bytes_per_launch=sel[0]["hbm_bytes_per_launch"] if sel else None,
               note="FETCH_SIZE x2 (gfx950 correction), KiB -> bytes, separate --pmc passes (profiles/collect_r3.sh)")
    json.dump(res, open(dst, "w"), indent=1)
    return res


def sq(out, cfg, dst):
    path = first(os.path.join(out, "pmc_sq_" + cfg, "**", "*_counter_collection.csv"))
    if not path:
        return None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {}
    for k, v in agg.items():
        if not any(s in k for s in ("trmm_sumsq", "kstar", "fused_panel", "pivot_update", "mfma_gemm", "gemm_nt", "gemm64_glds", "mid_sweep")):
            continue
        d = {c: sum(x) / len(x) for c, x in v.items()}
        # Under --pmc the timestamps of a SHORT kernel do not cover the window the counters were open for (round 3
        # derived 2.6-3.6 GHz for the fit's 20-80 us kernels on a 2.4 GHz part): the ratios are only formed for
        # kernels that run at least PMC_MIN_NS; for the others the raw counter averages are all that is kept.
        if d.get("GRBM_GUI_ACTIVE", 0) > 0 and d["dur_ns"] >= PMC_MIN_NS:
            d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / d["dur_ns"]        # summed over the 8 XCDs
            d["mfma_busy_frac_of_simd_cycles"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
        else:
            d["ratios_omitted"] = "kernel shorter than %d us under --pmc: duration does not cover the counted window" % (PMC_MIN_NS // 1000)
        d["launches"] = len(v["dur_ns"])
        res[k] = d
    json.dump(res, open(dst, "w"), indent=1)
    return res


def main():
    out, tag = sys.argv[1], sys.argv[2]
    on_box = "--on-box" in sys.argv
    dst_dir = os.path.join(out, "summ") if on_box else HERE
    os.makedirs(dst_dir, exist_ok=True)
    for cfg in ("c1", "c2", "c3", "c4"):
        stats = first(os.path.join(out, "stats_" + cfg, "**", "*_kernel_stats.csv"))
        pre = os.path.join(out, "summ")
        if stats:
            rows = list(csv.DictReader(open(stats)))
            trmm = max((r for r in rows if "trmm_sumsq" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
            name = trmm["Name"]
            shutil.copy(stats, os.path.join(dst_dir, "%s_%s_kernel_stats.csv" % (tag, cfg)))
            under = os.path.join(out, "bench_stats_%s.json" % cfg)
            ev = None
            if os.path.exists(under) and os.path.getsize(under):
                shutil.copy(under, os.path.join(dst_dir, "%s_%s_bench_under_rocprof.json" % (tag, cfg)))
                rf = json.load(open(under))["roofline"]
                ev = rf.get("whole_run", {}).get("avg_launch_ms", rf["avg_launch_ms"])   # (round 5: the whole run's average -- what the csv averages over)
            print("%s dominant %s: rocprof average %.5f ms over %s launches; HIP events of the same run %s ms"
                  % (cfg, name.split("(")[0][-60:], float(trmm["AverageNs"]) / 1e6, trmm["Calls"], "%.5f" % ev if ev else "-"))
            needle = name.split("(")[0].replace("void ", "")
            t = traffic(out, cfg, needle, os.path.join(dst_dir, "traffic_%s.json" % cfg))
            if t and t["hbm_bytes_per_launch"]:
                print("   traffic per launch %.1f MB" % (t["hbm_bytes_per_launch"] / 1e6))
            s = sq(out, cfg, os.path.join(dst_dir, "%s_%s_pmc_sq.json" % (tag, cfg)))
            for k, d in (s or {}).items():
                if "trmm_sumsq" in k or "kstar" in k:
                    print("   %-70s clock %.2f GHz  MFMA busy %.3f  LDS conflicts %.0f" % (
                        k[:70], d.get("clock_GHz", 0), d.get("mfma_busy_frac_of_simd_cycles", 0), d.get("SQ_LDS_BANK_CONFLICT", 0)))
        elif not on_box:
            # summaries made on the box
            for fn in ("%s_%s_kernel_stats.csv" % (tag, cfg), "%s_%s_bench_under_rocprof.json" % (tag, cfg),
                       "traffic_%s.json" % cfg, "%s_%s_pmc_sq.json" % (tag, cfg)):
                sp = os.path.join(pre, fn)
                if os.path.exists(sp):
                    shutil.copy(sp, os.path.join(HERE, fn))
        src = os.path.join(out, "bench_%s.json" % cfg)
        if os.path.exists(src) and os.path.getsize(src):
            if not on_box:
                shutil.copy(src, os.path.join(HERE, "%s_%s_bench.json" % (tag, cfg)))
            d = json.load(open(src))
            cb = d.get("cpu_baseline")
            print("%s  %.3f ms/step  fit %.3f  sweep %.3f  %.3f M evals/s  frac %.4f  step_frac %s  kstar %.1f us%s" % (
                cfg, d["ms_per_step"], d["fit_ms"], d["sweep_ms"], d["value"] / 1e6, d["roofline"]["frac"],
                d["roofline"].get("step_frac"), d["roofline"]["kstar_avg_ms"] * 1e3,
                "  cpu port %.0f ev/s (x%.0f)" % (cb["value"], d["value"] / cb["value"]) if cb else ""))
    if on_box:
        for sub, dst in [("stats_hyper_%d" % n, "%s_hyper_n%d_kernel_stats.csv" % (tag, n)) for n in (512, 2048, 4096)] + \
                        [("stats_small", "%s_small_path_kernel_stats.csv" % tag)]:
            st = first(os.path.join(out, sub, "**", "*_kernel_stats.csv"))
            if st:
                shutil.copy(st, os.path.join(dst_dir, dst))
        return
    pre = os.path.join(out, "summ")
    for fn in sorted(os.listdir(pre)) if os.path.isdir(pre) else []:
        if "hyper" in fn or "small_path" in fn:
            shutil.copy(os.path.join(pre, fn), os.path.join(HERE, fn))
    extras = [("bench_hyper_%s.json" % n, "%s_hyper_n%s_bench.json" % (tag, n)) for n in (32, 128, 512, 2048, 4096)]
    extras += [("latency_small.jsonl", "%s_latency_small.jsonl" % tag), ("fit_sizes.jsonl", "%s_fit_sizes.jsonl" % tag),
               ("gradient_stage.jsonl", "%s_gradient_stage.jsonl" % tag), ("trial_loop.jsonl", "%s_trial_loop.jsonl" % tag),
               ("fit_chain_stamps_n4096.txt", "%s_fit_chain_stamps_n4096.txt" % tag),
               ("hyper_fit.jsonl", "%s_hyper_fit.jsonl" % tag), ("two_factories.jsonl", "%s_two_factories.jsonl" % tag),
               ("overlap_ab.txt", "%s_overlap_ab.txt" % tag), ("tuning_table.txt", "%s_tuning_table.txt" % tag),
               ("hyper_side_by_side.txt", "%s_hyper_side_by_side.txt" % tag),
               # round 6: the short calls, polled one-launch paths beside round 5's calls
               ("host_draw.jsonl", "%s_host_draw.jsonl" % tag),
               ("short_calls.jsonl", "%s_short_calls.jsonl" % tag),
               ("short_calls_round5_calls.jsonl", "%s_short_calls_round5_calls.jsonl" % tag),
               ("gradient_stage_round5_calls.jsonl", "%s_gradient_stage_round5_calls.jsonl" % tag),
               ("trial_loop_round5_calls.jsonl", "%s_trial_loop_round5_calls.jsonl" % tag)]
    for c in ("c3", "c4"):
        for g in (1, 2, 4, 8):
            for sfx in ("", "_overlap2", "_serial"):
                extras.append(("%s_shard_of_%d%s.json" % (c, g, sfx), "%s_%s_shard_of_%d%s.json" % (tag, c, g, sfx)))
    for src, dst in extras:
        sp = os.path.join(out, src)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            shutil.copy(sp, os.path.join(HERE, dst))
    for n in (32, 128, 512, 2048, 4096):
        sp = os.path.join(out, "bench_hyper_%s.json" % n)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            d = json.load(open(sp))
            print("hyper N=%d  %.3f ms/evaluation  stages %s" % (n, d["ms_per_step"], {k: round(v, 3) for k, v in d["stages_ms"].items() if v}))


if __name__ == "__main__":
    main()
