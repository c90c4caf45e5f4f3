set -u
O=gpurun_out/r3k
mkdir -p $O
S="64 200 300 700 1100 2304 4096"
TGP_PANEL_FUSE=0 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_unfused.jsonl 2> $O/bits_unfused.err; echo "bits0 rc=$?"
TGP_PANEL_FUSE=1 timeout -k 10 300 python3 tools/fit_bitcheck.py $S > $O/bits_fused.jsonl 2> $O/bits_fused.err; echo "bits1 rc=$?"
cmp $O/bits_unfused.jsonl $O/bits_fused.jsonl && echo "BIT-IDENTICAL"
for f in 0 1; do TGP_PANEL_FUSE=$f timeout -k 10 300 python3 tools/bench_fit.py 512 1024 2048 4096 --reps 40 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fuse=$f N=%d dev %.4f wall %.4f'%(d['N'],d['fit_ms_device'],d['fit_ms_wall']))"; done
TGP_STAMP_FILE=$O/stamps.bin timeout -k 10 200 python3 tools/bench_fit.py 4096 --reps 3 > /dev/null 2>&1
python3 tools/stamp_summary.py $O/stamps.bin > $O/stamps.txt; sed -n 2,6p $O/stamps.txt | cut -c1-200; rm -f $O/stamps.bin
