# the compute side of the strong-scaling curve on ONE GPU: rank 0's share of a G-way split of the config's batch
set -u
O=gpurun_out/r3shard
mkdir -p $O
for c in c3 c4; do for g in 1 2 4 8; do
  case $c in c4) ST="--steps 3 --warmup 1";; *) ST="--steps 10 --warmup 3";; esac
  timeout -k 10 300 python3 bench.py --config $c --shard-of $g $ST --no-cpu-baseline --no-opt-in > $O/${c}_shard_of_$g.json 2> $O/${c}_$g.err; echo "$c shard-of $g rc=$?"
done; done
python3 - <<'PY'
import json
for c in ("c3","c4"):
    t1=None
    for g in (1,2,4,8):
        d=json.load(open("gpurun_out/r3shard/%s_shard_of_%d.json"%(c,g)))
        t=d["ms_per_step"]; t1=t1 or t
        print("%s G=%d  M_local %7d  ms/step %8.3f  fit %.3f sweep %.3f  -> speed-up of the compute side %.2fx (Amdahl bound %.2fx)"%(c,g,d["config"]["M_per_gpu"],t,d["fit_ms"],d["sweep_ms"],t1/t, json.load(open("gpurun_out/r3shard/%s_shard_of_1.json"%c))["amdahl_bound"]["speedup_max_by_gpus"].get(str(g),1.0) if g>1 else 1.0))
PY
