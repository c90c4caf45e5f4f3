set -u
O=gpurun_out/r3f
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 300 python3 tools/bench_latency.py > $O/latency_small.jsonl 2> $O/latency.err; echo "latency rc=$?"
grep -i "device\|optimis" $O/latency_small.jsonl | cut -c1-400 | head -20
