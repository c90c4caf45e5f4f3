#!/bin/bash
# profiles/collect_r6.sh [OUT] -- round 6: the evidence behind EVERY config's roofline line, run ON the GPU box from the repo
# root (via gpurun); writes under OUT (default gpurun_out/prof6).  Afterwards, in the container:
#     python3 profiles/summarize_round3.py OUT r06
# Same order as rounds 4-5 (PMC passes first, traffic_cN.json written on the box, then the un-profiled lines).  The bench
# line's default schedule is the SERIAL one now (bench.py --overlap 0: what the library and the plugin classes run unless
# asked otherwise); the line carries the overlapped schedule and the plugin leg as extras.  New this round: the short calls
# (tools/bench_short_calls.py, the gradient stage, the trial loop) with the polled one-launch paths beside round 5's calls.
# NOTE for readers of the pmc_sq files: under --pmc the runtime serialises kernels, so the library's stream-overlap probes
# fail and every schedule falls back to serial -- the PMC files describe the serial schedule, the kernel-trace csv the real one.
set -u
R=$PWD
OUT=${1:-gpurun_out/prof6}
PHASE=${2:-all}      # pmc | lines | latency | all  (one gpurun call holds at most 20 minutes: the three phases fit one call each)
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
NOX="--no-cpu-baseline --no-opt-in --no-plugin"
if [ $PHASE = pmc ] || [ $PHASE = all ]; then
for c in c3 c1 c2 c4; do
    case $c in c4) ST="--steps 2 --warmup 1";; c1) ST="--steps 20 --warmup 3";; *) ST="--steps 5 --warmup 2";; esac
    B="python3 $R/bench.py --config $c $ST $NOX"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats_$c" -o run -- $B > "$R/$OUT/bench_stats_$c.json" 2> "$R/$OUT/stats_$c.err"; echo "$c stats rc=$?"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$OUT/pmc_fetch_$c" -o run -- $B > /dev/null 2> "$R/$OUT/pmc_fetch_$c.err"; echo "$c fetch rc=$?"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$OUT/pmc_write_$c" -o run -- $B > /dev/null 2> "$R/$OUT/pmc_write_$c.err"; echo "$c write rc=$?"
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d "$R/$OUT/pmc_sq_$c" -o run -- $B > /dev/null 2> "$R/$OUT/pmc_sq_$c.err"; echo "$c sq rc=$?"
done
for n in 512 2048 4096; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats_hyper_$n" -o run -- python3 $R/bench.py --config hyper --hyper-n $n --steps 20 --warmup 5 --no-cpu-baseline > "$R/$OUT/bench_stats_hyper_$n.json" 2> "$R/$OUT/stats_hyper_$n.err"; echo "hyper $n stats rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats_small" -o run -- python3 $R/tools/bench_trial_loop.py > "$R/$OUT/trial_loop_under_rocprof.jsonl" 2> "$R/$OUT/stats_small.err"; echo "small stats rc=$?"
cd "$R"
python3 profiles/summarize_round3.py "$OUT" r06 --on-box > "$OUT/summary_pmc.txt" 2>&1; echo "pmc summary rc=$?"
cp "$OUT"/summ/traffic_c*.json profiles/ && echo "traffic files refreshed"
cp profiles/traffic_c*.json "$OUT"/ 2>/dev/null    # (profiles/ does not travel back from the box: the copies under OUT do)
fi
cd "$R"
if [ $PHASE = lines ] || [ $PHASE = all ]; then
# the full default lines (CPU baseline, plugin leg, opt-in arithmetic) of every config
for c in c3 c1 c2; do
    python3 bench.py --config $c --steps 10 --warmup 3 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"
done
python3 bench.py --config c4 --steps 2 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
# the compute side of the strong-scaling curve, both schedules
for c in c3 c4; do for g in 1 2 4 8; do for ov in 0 2; do
    case $c in c4) ST="--steps 3 --warmup 1";; *) ST="--steps 10 --warmup 3";; esac
    sfx=""; [ $ov -eq 2 ] && sfx="_overlap2"
    timeout -k 10 300 python3 bench.py --config $c --shard-of $g --overlap $ov $ST $NOX > $OUT/${c}_shard_of_${g}${sfx}.json 2> $OUT/${c}_shard_of_${g}${sfx}.err; echo "$c shard-of $g overlap $ov rc=$?"
done; done; done
for n in 32 128 512 2048 4096; do
    python3 bench.py --config hyper --hyper-n $n --steps 200 --warmup 20 > "$OUT/bench_hyper_$n.json" 2> "$OUT/bench_hyper_$n.err"; echo "bench hyper $n rc=$?"
done
fi
if [ $PHASE = latency ] || [ $PHASE = all ]; then
python3 tools/bench_latency.py > "$OUT/latency_small.jsonl" 2> "$OUT/latency_small.err"; echo "latency rc=$?"
python3 tools/bench_fit.py 64 128 256 512 1024 2048 3072 4096 6144 8192 > "$OUT/fit_sizes.jsonl" 2> "$OUT/fit_sizes.err"; echo "fit sizes rc=$?"
OLD="env TGP_POLL_US=0 TGP_SMALL_FUSED=0 TGP_SMALL_QUERY=0 TGP_SMALL_LIVE=0"
python3 tools/bench_short_calls.py > "$OUT/short_calls.jsonl" 2> "$OUT/short_calls.err"; echo "short calls rc=$?"
$OLD python3 tools/bench_short_calls.py > "$OUT/short_calls_round5_calls.jsonl" 2> "$OUT/short_calls_round5_calls.err"; echo "short calls (round 5's) rc=$?"
python3 tools/bench_gradient_stage.py > "$OUT/gradient_stage.jsonl" 2> "$OUT/gradient_stage.err"; echo "gradient stage rc=$?"
$OLD python3 tools/bench_gradient_stage.py > "$OUT/gradient_stage_round5_calls.jsonl" 2> "$OUT/gradient_stage_round5_calls.err"; echo "gradient stage (round 5's) rc=$?"
python3 tools/bench_trial_loop.py > "$OUT/trial_loop.jsonl" 2> "$OUT/trial_loop.err"; echo "trial loop rc=$?"
$OLD python3 tools/bench_trial_loop.py > "$OUT/trial_loop_round5_calls.jsonl" 2> "$OUT/trial_loop_round5_calls.err"; echo "trial loop (round 5's) rc=$?"
python3 tools/bench_hyper_fit.py > "$OUT/hyper_fit.jsonl" 2> "$OUT/hyper_fit.err"; echo "hyper fit rc=$?"
python3 tools/ab_private_streams.py two_factories > "$OUT/two_factories.jsonl" 2> "$OUT/two_factories.err"; echo "two factories rc=$?"
python3 tools/bench_host_draw.py --gpu --reps 5 > "$OUT/host_draw.jsonl" 2> "$OUT/host_draw.err"; echo "host draw rc=$?"
python3 -c "
import turbo_amd as ta
for k, (v, doc) in ta._lib.tuning().items(): print('%-22s = %-8s %s' % (k, v, doc))" > "$OUT/tuning_table.txt" 2>&1; echo "tuning rc=$?"
TGP_STAMP_FILE=$OUT/stamps_n4096.bin python3 tools/bench_fit.py 4096 --reps 3 > /dev/null 2>&1; python3 tools/stamp_summary.py $OUT/stamps_n4096.bin > "$OUT/fit_chain_stamps_n4096.txt" 2>&1; echo "stamps rc=$?"
rm -f $OUT/stamps_n4096.bin $OUT/stamps_n4096.bin.cus
fi
python3 profiles/summarize_round3.py "$OUT" r06 --on-box > "$OUT/summary_$PHASE.txt" 2>&1; echo "summary rc=$?"
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "*_counter_collection.csv" -size +4M -delete
du -sh "$OUT"
