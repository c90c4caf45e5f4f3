#!/bin/bash
set -e
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -3 $O/suite.log
python tools/bench_fit.py 512 1024 2048 3072 4096 6144 8192 > $O/fit_sizes.jsonl 2> $O/fit.err
python -c "
import json
print(' '.join('%d:%.3f' % (json.loads(l)['N'], json.loads(l)['fit_ms_device']) for l in open('$O/fit_sizes.jsonl')))"
python tools/bench_latency.py > $O/latency.jsonl 2> $O/lat.err
python -c "
import json
for l in open('$O/latency.jsonl'):
    d = json.loads(l); print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in ('N','M','gpu_fit_ms','gpu_fit_device_ms','gpu_predict_ms','gpu_ei_ms','gpu_fit_optimised_ms')})"
