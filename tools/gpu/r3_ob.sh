set -u
for ob in 512 256; do for t in 0 128 256; do TGP_OB=$ob TGP_PANEL_FUSE_TILES=$t timeout -k 10 300 python3 tools/bench_fit.py 3000 4096 6000 8192 --reps 30 2>/dev/null | python3 -c "
import sys,json
print('ob=$ob tiles<=$t', ' '.join('N=%d %.4f'%(json.loads(l)['N'],json.loads(l)['fit_ms_device']) for l in sys.stdin))"; done; done
