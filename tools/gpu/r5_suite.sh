# the -m gpu suite, smoke(), the default bench line
set -u
O=${1:-gpurun_out/r5suite}
mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest -m gpu rc=$rc"; tail -15 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
