# the -m gpu suite, smoke(), the default bench line, the 8-way shards
set -u
O=${1:-gpurun_out/r5suite}
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest -m gpu rc=$rc"; tail -15 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
python3 bench.py --no-cpu-baseline --no-opt-in > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for spec in "c1 1" "c3 8" "c4 8"; do set -- $spec
  python3 bench.py --config $1 --shard-of $2 --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in > $O/bench_$1_s$2.json 2>> $O/bench_default.err
done
python3 - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.load(open(f)); print("%-28s ms/step %8.3f fit %.3f sweep %.3f host+gaps %.3f frac %.3f"%(f.split("/")[-1],d["ms_per_step"],d["fit_ms"],d["sweep_ms"],d["ms_per_step"]-d["fit_ms"]-d["sweep_ms"],d["roofline"]["frac"]))
PY
