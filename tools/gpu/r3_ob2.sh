set -u
for ob in 512 256; do TGP_OB=$ob timeout -k 10 300 python3 tools/bench_fit.py 700 1024 1280 1536 2048 2560 3072 --reps 30 2>/dev/null | python3 -c "
import sys,json
print('ob=$ob', ' '.join('N=%d %.4f'%(json.loads(l)['N'],json.loads(l)['fit_ms_device']) for l in sys.stdin))"; done
