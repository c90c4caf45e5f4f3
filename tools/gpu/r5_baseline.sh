# round 5, before any change: the compute side of the strong-scaling curve on the round-4 build (ONE GPU),
# and a kernel trace of the C3 shard-of-8 step (what runs when inside the fit)
set -u
O=${1:-gpurun_out/r5base}
mkdir -p $O
for c in c3 c4; do for g in 1 2 4 8; do
  case $c in c4) ST="--steps 3 --warmup 1";; *) ST="--steps 10 --warmup 3";; esac
  timeout -k 10 300 python3 bench.py --config $c --shard-of $g $ST --no-cpu-baseline --no-opt-in > $O/${c}_shard_of_$g.json 2> $O/${c}_$g.err; echo "$c shard-of $g rc=$?"
done; done
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for c in ("c3","c4"):
    t1=None
    for g in (1,2,4,8):
        d=json.load(open("%s/%s_shard_of_%d.json"%(O,c,g)))
        t=d["ms_per_step"]; t1=t1 or t
        print("%s G=%d  M_local %7d  ms/step %8.3f  fit %.3f sweep %.3f  -> %.2fx"%(c,g,d["config"]["M_per_gpu"],t,d["fit_ms"],d["sweep_ms"],t1/t))
PY
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace_c3_s8" -o run -- python3 $R/bench.py --config c3 --shard-of 8 --steps 3 --warmup 2 --no-cpu-baseline --no-opt-in > "$R/$O/trace_c3_s8.json" 2> "$R/$O/trace_c3_s8.err"; echo "trace rc=$?"
cd $R
ls -la $O/trace_c3_s8/* | head
