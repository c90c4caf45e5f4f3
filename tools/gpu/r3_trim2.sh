set -u
O=gpurun_out/r3l
mkdir -p $O
S="64 200 300 700 1100 2304 4096 5000"
TGP_PANEL_FUSE=0 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_unfused.jsonl 2> $O/bits_unfused.err; echo "bits0 rc=$?"
TGP_PANEL_FUSE=1 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_fused.jsonl 2> $O/bits_fused.err; echo "bits1 rc=$?"
cmp $O/bits_unfused.jsonl $O/bits_fused.jsonl && echo "BIT-IDENTICAL"
for t in 0 128 256 320 448 100000; do TGP_PANEL_FUSE_TILES=$t timeout -k 10 300 python3 tools/bench_fit.py 2048 4096 8192 --reps 30 2>/dev/null | python3 -c "
import sys,json
print('tiles<=$t', ' '.join('N=%d %.4f'%(json.loads(l)['N'],json.loads(l)['fit_ms_device']) for l in sys.stdin))"; done
