# round 5: the sweep's front inside the fit -- tests, then the A/B (bench.py --overlap 0|1|2) for C3 / C4 and their 8-way shards
set -u
O=${1:-gpurun_out/r5ov}
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_round5.py -x -q > $O/pytest_round5.txt 2>&1; rc=$?; echo "round5 tests rc=$rc"; tail -5 $O/pytest_round5.txt
[ $rc -eq 0 ] || exit $rc
for c in c3; do for g in 1 8; do for ov in 0 1 2; do
  timeout -k 10 300 python3 bench.py --config $c --shard-of $g --overlap $ov --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in > $O/${c}_s${g}_ov$ov.json 2> $O/${c}_s${g}_ov$ov.err; echo "$c shard-of $g overlap $ov rc=$?"
done; done; done
python3 - $O <<'PY'
import json,sys
O=sys.argv[1]
for c in ("c3",):
    for g in (1,8):
        for ov in (0,1,2):
            try: d=json.load(open("%s/%s_s%d_ov%d.json"%(O,c,g,ov)))
            except Exception as e: print(c,g,ov,"?",e); continue
            print("%s G=%d overlap %d  ms/step %8.3f  fit %.3f sweep %.3f  trmm frac %.3f kstar %.3f"%(c,g,ov,d["ms_per_step"],d["fit_ms"],d["sweep_ms"],d["roofline"]["frac"],d["roofline"]["kstar_avg_ms"]))
PY
