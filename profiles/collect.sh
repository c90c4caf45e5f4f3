#!/bin/bash
# profiles/collect.sh [OUT] -- run ON the GPU box from the repo root (via gpurun); writes everything
# under OUT (default gpurun_out/prof).  Afterwards, in the container:
#     python3 profiles/summarize_round.py OUT r02     # copies the judged summaries into profiles/
# Passes (separate runs, as MI355X_MICROARCH.md prescribes for PMC):
#   1. rocprofv3 --kernel-trace --stats        per-kernel durations of the default bench command
#   2. rocprofv3 --pmc FETCH_SIZE              L2 -> fabric read traffic  (x2 on gfx950)
#   3. rocprofv3 --pmc WRITE_SIZE              write traffic
#   4. rocprofv3 --pmc GRBM/SQ counters        clock, MFMA busy, LDS bank conflicts
#   5. un-profiled bench lines for C3 (default), C1, C2, C4
set -u
R=$PWD
OUT=${1:-gpurun_out/prof}
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o runc -- $B > "$R/$OUT/bench_stats.json" 2> "$R/$OUT/stats.err"; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$OUT/pmc_fetch" -o runc -- $B > /dev/null 2> "$R/$OUT/pmc_fetch.err"; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$OUT/pmc_write" -o runc -- $B > /dev/null 2> "$R/$OUT/pmc_write.err"; echo "write rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d "$R/$OUT/pmc_sq" -o runc -- $B > /dev/null 2> "$R/$OUT/pmc_sq.err"; echo "sq rc=$?"
cd "$R"
for c in c3 c1 c2; do
    python3 bench.py --config $c --steps 10 --warmup 2 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"
done
python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
# round 2: the "next" rows and the fit
for n in 32 128 512 2048 4096; do
    python3 bench.py --config hyper --hyper-n $n --steps 200 --warmup 20 > "$OUT/bench_hyper_$n.json" 2> "$OUT/bench_hyper_$n.err"; echo "bench hyper $n rc=$?"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats_hyper" -o runc -- python3 $R/bench.py --config hyper --hyper-n 4096 --steps 5 --warmup 2 --no-cpu-baseline > "$R/$OUT/bench_hyper_stats.json" 2> "$R/$OUT/stats_hyper.err"; echo "hyper stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats_c1" -o runc -- python3 $R/bench.py --config c1 --steps 20 --warmup 2 --no-cpu-baseline > "$R/$OUT/bench_c1_stats.json" 2> "$R/$OUT/stats_c1.err"; echo "c1 stats rc=$?"
cd "$R"
python3 tools/bench_latency.py > "$OUT/latency_small.jsonl" 2> "$OUT/latency_small.err"; echo "latency rc=$?"
python3 tools/bench_fit.py 64 128 256 512 1024 2048 4096 8192 > "$OUT/fit_sizes.jsonl" 2> "$OUT/fit_sizes.err"; echo "fit sizes rc=$?"
[ -x tools/microbench/mfma_f64_peak ] && tools/microbench/mfma_f64_peak > "$OUT/mfma_f64_peak.txt" 2>&1; echo "f64 peak rc=$?"
python3 tools/bench_gradient_stage.py > "$OUT/gradient_stage.jsonl" 2> "$OUT/gradient_stage.err"; echo "gradient stage rc=$?"
python3 bench.py --dtype f32x3 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_c3_f32x3.json" 2> "$OUT/bench_c3_f32x3.err"; echo "bench c3 f32x3 rc=$?"
python3 bench.py --config c4 --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4_f32x3.json" 2> "$OUT/bench_c4_f32x3.err"; echo "bench c4 f32x3 rc=$?"
python3 bench.py --dtype f32h2 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_c3_f32h2.json" 2> "$OUT/bench_c3_f32h2.err"; echo "bench c3 f32h2 rc=$?"
python3 bench.py --config c4 --dtype f32h2 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4_f32h2.json" 2> "$OUT/bench_c4_f32h2.err"; echo "bench c4 f32h2 rc=$?"
python3 tools/bench_split_accuracy.py > "$OUT/split_accuracy.jsonl" 2> "$OUT/split_accuracy.err"; echo "split accuracy rc=$?"
python3 tools/bench_trial_loop.py > "$OUT/trial_loop.jsonl" 2> "$OUT/trial_loop.err"; echo "trial loop rc=$?"
