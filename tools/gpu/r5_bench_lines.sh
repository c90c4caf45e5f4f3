# the four un-profiled bench lines (with cpu_baseline) on the final build -> profiles/r05_cN_bench.json
set -u
O=${1:-gpurun_out/r5lines}; mkdir -p $O
for c in c3 c1 c2; do
    python3 bench.py --config $c --steps 10 --warmup 3 > "$O/bench_$c.json" 2> "$O/bench_$c.err"; echo "bench $c rc=$?"
done
python3 bench.py --config c4 --steps 2 --warmup 1 > "$O/bench_c4.json" 2> "$O/bench_c4.err"; echo "bench c4 rc=$?"
python3 - $O <<'PY'
import json,sys
for c in ("c1","c2","c3","c4"):
    d=json.load(open("%s/bench_%s.json"%(sys.argv[1],c)))
    print(c,"ms/step %.3f value %.4g frac %.4f fit(own) %.3f fitfrac %.3f serial %s amdahl8 %.2f"%(d["ms_per_step"],d["value"],d["roofline"]["frac"],d["roofline"]["fit"]["ms"],d["roofline"]["fit"]["frac"],{k:(round(v,3) if isinstance(v,float) else v) for k,v in (d["serial_schedule"] or {}).items() if k!="note"},d["amdahl_bound"]["speedup_max_by_gpus"]["8"]))
PY
