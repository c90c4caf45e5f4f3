set -u
O=gpurun_out/r3c
mkdir -p $O
TGP_STAMP_FILE=$O/stamps.bin timeout -k 10 200 python3 tools/bench_fit.py 4096 --reps 3 > $O/fit.json 2> $O/fit.err; echo "rc=$?"
python3 tools/stamp_summary.py $O/stamps.bin > $O/stamps.txt; head -70 $O/stamps.txt | cut -c1-330
