set -u
mkdir -p gpurun_out/la
S="2048 4096 8192"
python3 tools/bench_fit.py $S --check > gpurun_out/la/base.jsonl 2>&1
TGP_LOOKAHEAD=1 python3 tools/bench_fit.py $S --check > gpurun_out/la/la_nomask.jsonl 2>&1
for c in 128 192 224; do
TGP_LOOKAHEAD=1 TGP_BG_CUS=$c python3 tools/bench_fit.py $S --check > gpurun_out/la/la_$c.jsonl 2>&1
done
grep -h fit_ms gpurun_out/la/*.jsonl | cut -c1-200
tail -2 gpurun_out/la/la_192.jsonl
