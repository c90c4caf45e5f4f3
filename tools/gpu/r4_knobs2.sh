#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O; : > $O/knobs2.txt
run() { echo "== $*" >> $O/knobs2.txt; env "$@" python tools/bench_fit.py 1024 2048 3072 4096 6144 8192 --reps 15 2>> $O/err.txt | python -c "
import sys, json
print(' '.join('%d:%.3f' % (json.loads(l)['N'], json.loads(l)['fit_ms_device']) for l in sys.stdin))" >> $O/knobs2.txt; }
run A=0
run TGP_PANEL_FUSE_TILES=512
run TGP_PANEL_FUSE_TILES=384
run TGP_OB=1024 TGP_BG_CUS=224
run TGP_OB=1024 TGP_BG_CUS=224 TGP_PANEL_FUSE_TILES=256
run TGP_OB=1024 TGP_BG_CUS=240
run TGP_OB=1024 TGP_BG_CUS=208
run TGP_BG_CUS=208
run TGP_BG_CUS=240
run TGP_OB=768
cat $O/knobs2.txt
