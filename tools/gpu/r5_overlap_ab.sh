# round 5: bench.py --overlap 0|1|2 on every BASELINE config and on the 8-way shards of C3 / C4, twice in alternation
set -u
O=${1:-gpurun_out/r5ab}
mkdir -p $O
for rep in 1 2; do
for spec in "c1 1" "c2 1" "c3 1" "c3 8" "c4 8"; do
  set -- $spec; c=$1; g=$2
  case $c in c4) ST="--steps 6 --warmup 2";; c1) ST="--steps 100 --warmup 10";; *) ST="--steps 20 --warmup 3";; esac
  for ov in 0 1 2; do
    timeout -k 10 300 python3 bench.py --config $c --shard-of $g --overlap $ov $ST --no-cpu-baseline --no-opt-in > $O/${c}_s${g}_ov${ov}_r$rep.json 2> $O/${c}_s${g}_ov${ov}_r$rep.err
    python3 - $O/${c}_s${g}_ov${ov}_r$rep.json $c $g $ov <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%s G=%s overlap %s  ms/step %8.3f  fit %.3f sweep %.3f  frac %.3f"%(sys.argv[2],sys.argv[3],sys.argv[4],d["ms_per_step"],d["fit_ms"],d["sweep_ms"],d["roofline"]["frac"]),flush=True)
except Exception as e: print(sys.argv[2:],"failed",e)
PY
  done
done
done
