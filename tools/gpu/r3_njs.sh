set -u
for c in c3 c2 c1; do for j in 1 2 4 8; do
TGP_KS_JS=$j timeout -k 10 300 python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$c njs=$j: ms/step %.4f sweep %.4f kstar_avg %.4f trmm_avg %.4f'%(d['ms_per_step'],d['sweep_ms'],d['roofline']['kstar_avg_ms'],d['roofline']['avg_launch_ms']))"
done; done
