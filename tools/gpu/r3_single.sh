set -u
O=gpurun_out/r3h
mkdir -p $O
for c in c3 c1 c2; do
timeout -k 10 300 python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>$O/bench_$c.err | tee $O/bench_$c.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$c ms/step %.4f fit %.4f sweep %.4f frac %.4f step_frac %.4f kstar %.4f trmm %.4f launches %d'%(d['ms_per_step'],d['fit_ms'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['step_frac'],d['roofline']['kstar_avg_ms'],d['roofline']['avg_launch_ms'],d['roofline']['launches']))"
done
timeout -k 10 300 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-opt-in 2>$O/bench_c4.err | tee $O/bench_c4.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c4 ms/step %.4f fit %.4f sweep %.4f frac %.4f kstar %.4f trmm %.4f launches %d'%(d['ms_per_step'],d['fit_ms'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['kstar_avg_ms'],d['roofline']['avg_launch_ms'],d['roofline']['launches']))"
TGP_SLAB_GB=0.2 timeout -k 10 300 python3 bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c3 per-group launches: ms/step %.4f sweep %.4f frac %.4f launches %d'%(d['ms_per_step'],d['sweep_ms'],d['roofline']['frac'],d['roofline']['launches']))"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
