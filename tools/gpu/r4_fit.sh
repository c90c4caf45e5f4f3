#!/bin/bash
# round 4: the f64 GEMM microbenchmark, fit latencies over N (default + A/B switches), the GPU suite, the bench line
set -e
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 200 ./tools/microbench/gemm64_bench 4096 > $O/gemm64_4096.txt 2>&1; cat $O/gemm64_4096.txt
timeout -k 10 200 ./tools/microbench/gemm64_bench 8192 > $O/gemm64_8192.txt 2>&1; head -20 $O/gemm64_8192.txt
SIZES="64 128 256 512 1000 1024 2048 3072 4096 6144 8192"
python tools/bench_fit.py $SIZES > $O/fit_sizes.jsonl 2> $O/fit.err
TGP_GEMM64=reg python tools/bench_fit.py 512 1024 2048 4096 8192 > $O/fit_sizes_gemm64reg.jsonl 2>> $O/fit.err
for t in 2048 4096 6144; do TGP_TRAIL64=$t python tools/bench_fit.py 4096 6144 8192 > $O/fit_trail64_$t.jsonl 2>> $O/fit.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4c/fit_*.jsonl")):
    print(f, " ".join("%d:%.3f" % (json.loads(l)["N"], json.loads(l)["fit_ms_device"]) for l in open(f)))
PY
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || { tail -30 $O/suite.log; exit 1; }
tail -3 $O/suite.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4c/bench_c3.json"))
print("C3 ms/step %.2f fit %.3f sweep %.2f frac %.3f" % (d["ms_per_step"], d["fit_ms"], d["sweep_ms"], d["roofline"]["frac"]))
PY
