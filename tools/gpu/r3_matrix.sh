set -u
O=gpurun_out/r3d
mkdir -p $O
for fuse in 0 1; do for cus in 192 224 240 0; do for ob in 512 256; do
  TGP_PANEL_FUSE=$fuse TGP_BG_CUS=$cus TGP_OB=$ob timeout -k 10 120 python3 tools/bench_fit.py 4096 --reps 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fuse=$fuse cus=$cus ob=$ob N=%d dev %.3f wall %.3f'%(d['N'],d['fit_ms_device'],d['fit_ms_wall']))"
done; done; done | tee $O/matrix.txt
