set -u
O=gpurun_out/r3g
mkdir -p $O
for ob in 512 256; do
  TGP_OB=$ob timeout -k 10 120 python3 tools/bench_fit.py 300 512 700 1024 --reps 40 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ob=$ob N=%d dev %.4f wall %.4f'%(d['N'],d['fit_ms_device'],d['fit_ms_wall']))"
done | tee $O/ob.txt
for ob in 512 256; do
TGP_OB=$ob timeout -k 10 120 python3 bench.py --config c1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c1 ob=$ob ms/step %.4f fit %.4f sweep %.4f frac %.3f kstar %.4f trmm %.4f'%(d['ms_per_step'],d['fit_ms'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['kstar_avg_ms'],d['roofline']['avg_launch_ms']))"
done | tee -a $O/ob.txt
