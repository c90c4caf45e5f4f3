#!/bin/bash
# tools/gpu/r6_skip_xcc.sh OUT -- round 6: the pivot workgroup's XCD kept out of the background / third streams' masks
# (TGP_BG_SKIP_XCC=-1, default) against round 5's masks (-2): fit latency over the sizes, the chain's stamps at N = 4096,
# the C3 8-way shard and the C3 step.
set -u
OUT=${1:-gpurun_out/r6_skip_xcc}
mkdir -p "$OUT"
for rep in 1 2; do for v in -1 -2; do
    TGP_BG_SKIP_XCC=$v timeout -k 10 200 python3 tools/bench_fit.py 512 1024 2048 3072 4096 6144 8192 > "$OUT/fit_skip${v}_$rep.jsonl" 2> "$OUT/fit_skip${v}_$rep.err"; echo "fit $v rep $rep rc=$?"
done; done
for v in -1 -2; do
    TGP_BG_SKIP_XCC=$v TGP_STAMP_FILE=$OUT/st$v.bin timeout -k 10 200 python3 tools/bench_fit.py 4096 --reps 3 > /dev/null 2>&1
    python3 tools/stamp_summary.py $OUT/st$v.bin > "$OUT/stamps_skip$v.txt" 2>&1; rm -f $OUT/st$v.bin $OUT/st$v.bin.cus; tail -2 "$OUT/stamps_skip$v.txt"
    TGP_BG_SKIP_XCC=$v timeout -k 10 300 python3 bench.py --config c3 --shard-of 8 --steps 20 --warmup 3 --no-cpu-baseline --no-opt-in --no-plugin > "$OUT/c3_shard8_skip$v.json" 2> "$OUT/c3_shard8_skip$v.err"; echo "shard8 $v rc=$?"
    TGP_BG_SKIP_XCC=$v timeout -k 10 300 python3 bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in --no-plugin > "$OUT/c3_skip$v.json" 2> "$OUT/c3_skip$v.err"; echo "c3 $v rc=$?"
done
python3 - <<PY
import json, glob
for v in ("-1", "-2"):
    rows = {}
    for f in sorted(glob.glob("$OUT/fit_skip%s_*.jsonl" % v)):
        for line in open(f):
            d = json.loads(line)
            rows.setdefault(d["N"], []).append(d["fit_ms_device"])
    print("skip", v, {n: [round(x, 3) for x in r] for n, r in rows.items()})
    for name in ("c3_shard8", "c3"):
        d = json.load(open("$OUT/%s_skip%s.json" % (name, v)))
        print("  ", name, "ms_per_step %.3f fit %.3f sweep %.3f" % (d["ms_per_step"], d["fit_ms"], d["sweep_ms"]))
PY
