#!/bin/bash
# repeatability of the fit beside the background stream (tools/repeat_fit.py), the old k-loop for contrast, fit latency
mkdir -p gpurun_out
timeout -k 10 400 python tools/repeat_fit.py 1100 2304 3000 3700 4096 4500 5000 5500 6000 7000 8192 --trials 300 > gpurun_out/repeat_fit.txt 2>gpurun_out/repeat_fit.err; echo "default rc=$?"
TGP_GEMM64=round4-war timeout -k 10 200 python tools/repeat_fit.py 4096 5000 5500 --trials 300 > gpurun_out/repeat_fit_war.txt 2>/dev/null; echo "old loop rc=$? (1 expected)"
timeout -k 10 300 python tools/bench_fit.py 1024 2048 3072 4096 5120 6144 8192 --reps 30 > gpurun_out/fit_after_fix.txt 2>&1
cat gpurun_out/repeat_fit.txt gpurun_out/repeat_fit_war.txt | cut -c1-150
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/fit_after_fix.txt") if l.startswith("{")]
print("fit ms", " ".join("%d:%.3f" % (r["N"], r["fit_ms_device"]) for r in rows))
PY
