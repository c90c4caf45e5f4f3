# kstar loop variants (libturbogp.so.ksN built with -DTGP_KS_VARIANT=N): bench.py C3 / C1 / C2, serial schedule, kstar's own time
set -u
for v in "" .ks1 .ks2 ""; do
  for c in c3 c1 c2; do
    case $c in c1) ST="--steps 50 --warmup 5";; *) ST="--steps 10 --warmup 3";; esac
    TGP_LIBRARY=$PWD/turbo_amd/csrc/libturbogp.so$v python3 bench.py --config $c --overlap 0 $ST --no-cpu-baseline --no-opt-in 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('variant%-5s %s ms/step %8.3f sweep %.3f kstar_avg %.4f'%('$v','$c',d['ms_per_step'],d['sweep_ms'],d['roofline']['kstar_avg_ms']))"
  done
done
