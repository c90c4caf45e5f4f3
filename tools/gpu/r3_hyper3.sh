set -u
O=gpurun_out/r3j
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hyper or device_opt or one_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python3 tools/bench_latency.py > $O/latency_small.jsonl 2> $O/latency.err; echo "latency rc=$?"
TGP_HYPER_WGS=1 timeout -k 10 300 python3 tools/bench_latency.py > $O/latency_small_wgs1.jsonl 2> $O/latency1.err; echo "latency1 rc=$?"
python3 - <<'PY'
import json
for fn in ("gpurun_out/r3j/latency_small.jsonl","gpurun_out/r3j/latency_small_wgs1.jsonl"):
    print(fn)
    for l in open(fn):
        d=json.loads(l)
        if 'gpu_fit_optimised_device_ms' in d: print(' ',d['N'],d['D'],d['M'],'device opt %.3f ms lml %.4f | scipy-driven %.3f lml %.4f'%(d['gpu_fit_optimised_device_ms'],d['gpu_fit_optimised_device_lml'],d['gpu_fit_optimised_ms'],d['gpu_fit_optimised_lml']))
PY
timeout -k 10 300 python3 tools/bench_trial_loop.py > $O/trial_loop.jsonl 2>$O/trial.err; echo "trial rc=$?"; cut -c1-300 $O/trial_loop.jsonl | tail -8
