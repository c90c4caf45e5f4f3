#!/bin/bash
set -e
O=gpurun_out/r4l; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -2 $O/suite.log
for zc in 0 1; do for c in c1 c2 c3; do
TGP_SWEEP_ZC=$zc python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-opt-in | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$c ZC=$zc ms/step %.3f fit %.3f sweep %.3f' % (d['ms_per_step'], d['fit_ms'], d['sweep_ms']))"; done; done
