#!/bin/bash
# the -m gpu suite + the default bench line (+ smoke)
set -e
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -3 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4h/bench_c3.json"))
print("C3 ms/step %.2f fit %.3f sweep %.2f frac %.3f value %.0f" % (d["ms_per_step"], d["fit_ms"], d["sweep_ms"], d["roofline"]["frac"], d["value"]))
PY
python tools/bench_hyper_fit.py > $O/hyper_fit.jsonl 2> $O/hyper_fit.err; cut -c1-400 $O/hyper_fit.jsonl
