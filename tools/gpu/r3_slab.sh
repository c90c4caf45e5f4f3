set -u
O=gpurun_out/r3i
mkdir -p $O
for gb in 0.2 0.55 1.1 2.2 40; do
TGP_SLAB_GB=$gb timeout -k 10 300 python3 bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c3 slab_gb=$gb: ms/step %.4f sweep %.4f frac %.4f kstar_total %.3f trmm_total %.3f launches %d'%(d['ms_per_step'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['kstar_avg_ms']*d['roofline']['launches']/10,d['roofline']['avg_launch_ms']*d['roofline']['launches']/10,d['roofline']['launches']))"
done
for gb in 0.2 1.1 4.4 40; do
TGP_SLAB_GB=$gb timeout -k 10 300 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c4 slab_gb=$gb: ms/step %.4f sweep %.4f frac %.4f kstar_total %.3f trmm_total %.3f launches %d'%(d['ms_per_step'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['kstar_avg_ms']*d['roofline']['launches']/2,d['roofline']['avg_launch_ms']*d['roofline']['launches']/2,d['roofline']['launches']))"
done
for gb in 0.2 40; do
TGP_SLAB_GB=$gb timeout -k 10 300 python3 bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c2 slab_gb=$gb: ms/step %.4f sweep %.4f frac %.4f launches %d'%(d['ms_per_step'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['launches']))"
done
