#!/bin/bash
# tools/gpu/r6_latency.sh OUT -- round 6: the short calls (small fit, fit + gradient, acquisition gradient) with the
# polled one-launch paths (default) beside round 5's calls (TGP_POLL_US=0 TGP_SMALL_FUSED=0 TGP_SMALL_QUERY=0), same box.
# Run from the repo root on the GPU box.  A step that is killed (timeout) ends the script: no GPU step after a hang.
set -u
OUT=${1:-gpurun_out/r6_latency}
mkdir -p "$OUT"
step() {   # step NAME SECONDS cmd...
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "$name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name was killed: stopping"; exit $rc; fi
}
OLD="env TGP_POLL_US=0 TGP_SMALL_FUSED=0 TGP_SMALL_QUERY=0"
step short_calls_new 240 python3 tools/bench_short_calls.py
step short_calls_old 240 $OLD python3 tools/bench_short_calls.py
step short_calls_new_1thread 240 env TGP_HYPER_THREADS=1 python3 tools/bench_short_calls.py
step gradient_stage_new 300 python3 tools/bench_gradient_stage.py
step gradient_stage_old 300 $OLD python3 tools/bench_gradient_stage.py
step trial_loop_new 300 python3 tools/bench_trial_loop.py
step trial_loop_old 300 $OLD python3 tools/bench_trial_loop.py
