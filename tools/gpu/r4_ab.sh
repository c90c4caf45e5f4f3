#!/bin/bash
# the A/B of stream-creation order (tools/ab_private_streams.py) + the round-4 regression tests
set -e
mkdir -p gpurun_out/r4b
python tools/ab_private_streams.py > gpurun_out/r4b/ab_private_streams_fixed.jsonl 2> gpurun_out/r4b/ab.err
cut -c1-200 gpurun_out/r4b/ab_private_streams_fixed.jsonl
python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -5
