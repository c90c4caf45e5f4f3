# the -m gpu suite, the default bench line, the small-problem latency table
set -u
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
cut -c1-400 $O/bench_c3.json
timeout -k 10 300 python3 tools/bench_latency.py > $O/latency_small.jsonl 2> $O/latency.err; echo "latency rc=$?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r3e/latency_small.jsonl"):
    d=json.loads(l)
    if 'gpu_fit_optimised_ms' in d: print(' ',d['N'],d['D'],d['M'],'default optimised fit %.3f ms (lml %.4f) | device %.3f | sklearn %.1f'%(d['gpu_fit_optimised_ms'],d['gpu_fit_optimised_lml'],d['gpu_fit_optimised_device_ms'],d.get('sklearn_fit_optimised_ms',float('nan'))))
PY
