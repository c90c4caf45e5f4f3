#!/bin/bash
# fit latency at N = 2048 / 4096 / 8192 under the tuning switches that the faster f64 kernels may have moved
O=gpurun_out/r4e; mkdir -p $O; : > $O/knobs.txt
run() { echo "== $*" >> $O/knobs.txt; env "$@" python tools/bench_fit.py 2048 4096 8192 --reps 15 2>> $O/err.txt | python -c "
import sys, json
print(' '.join('%d:%.3f' % (json.loads(l)['N'], json.loads(l)['fit_ms_device']) for l in sys.stdin))" >> $O/knobs.txt; }
run A=0
for t in 0 64 256 512 100000; do run TGP_PANEL_FUSE_TILES=$t; done
for b in 128 160 224 0; do run TGP_BG_CUS=$b; done
run TGP_OB=256
run TGP_OB=1024
run TGP_BGINV=0
run TGP_MERGE64=512
run TGP_MERGE64=4096
run A=1
cat $O/knobs.txt
