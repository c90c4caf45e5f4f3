#!/bin/bash
# round 4, first GPU call: the A/B of the private streams, the new tests, the whole suite, the default bench line
set -e
mkdir -p gpurun_out/r4a
python tools/ab_private_streams.py > gpurun_out/r4a/ab_private_streams.jsonl 2> gpurun_out/r4a/ab.err
cat gpurun_out/r4a/ab_private_streams.jsonl
python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -5
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/suite.log 2>&1 || { tail -30 gpurun_out/r4a/suite.log; exit 1; }
tail -3 gpurun_out/r4a/suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench_c3.json 2> gpurun_out/r4a/bench_c3.err
cut -c1-600 gpurun_out/r4a/bench_c3.json
