# the -m gpu suite several times over (flakiness check)
set -u
O=gpurun_out/r3rep
mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "run $i rc=$?"; tail -2 $O/pytest_$i.log
done
