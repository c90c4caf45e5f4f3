set -u
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
cut -c1-1500 $O/bench_c3.json
