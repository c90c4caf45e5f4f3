set -u
O=gpurun_out/r3a
mkdir -p $O
S="64 200 300 700 1100 2304 4096 5000"
TGP_PANEL_FUSE=0 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_unfused.jsonl 2> $O/bits_unfused.err; echo "bits0 rc=$?"
TGP_PANEL_FUSE=1 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_fused.jsonl 2> $O/bits_fused.err; echo "bits1 rc=$?"
cmp $O/bits_unfused.jsonl $O/bits_fused.jsonl && echo "BIT-IDENTICAL"
TGP_PANEL_FUSE=0 timeout -k 10 300 python3 tools/bench_fit.py 512 1024 2048 4096 8192 > $O/fit_unfused.jsonl 2> $O/fit_unfused.err; echo "fit0 rc=$?"
TGP_PANEL_FUSE=1 timeout -k 10 300 python3 tools/bench_fit.py 512 1024 2048 4096 8192 --check > $O/fit_fused.jsonl 2> $O/fit_fused.err; echo "fit1 rc=$?"
cat $O/fit_unfused.jsonl $O/fit_fused.jsonl | cut -c1-120
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout -k 10 300 python3 bench.py --config c1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"
cut -c1-900 $O/bench_c3.json; cut -c1-700 $O/bench_c1.json
