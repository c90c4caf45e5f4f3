#!/bin/bash
set -e
O=gpurun_out/r4k; mkdir -p $O
python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -8
python tools/bench_latency.py > $O/latency.jsonl 2> $O/lat.err
python -c "
import json
for l in open('$O/latency.jsonl'):
    d = json.loads(l); print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in ('N','M','gpu_fit_ms','gpu_predict_ms','gpu_ei_ms')})"
for m in 0 100000000; do TGP_MID_MAXM=$m python bench.py --config c1 --steps 30 --warmup 5 --no-cpu-baseline | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('C1 TGP_MID_MAXM=$m ms/step %.3f fit %.3f sweep %.3f' % (d['ms_per_step'], d['fit_ms'], d['sweep_ms']))"; done
