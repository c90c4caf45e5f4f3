# soak: the round-3 kernels repeated, every repetition must reproduce the first one's digests (a race would not)
set -u
O=gpurun_out/r3soak
mkdir -p $O
timeout -k 10 200 python3 tools/stress_round3.py fit > $O/fit_ref.jsonl 2>/dev/null; echo "ref rc=$?"
timeout -k 10 200 python3 tools/stress_round3.py hyper > $O/hyper_ref.jsonl 2>/dev/null
bad=0
for i in $(seq 1 ${1:-12}); do
  timeout -k 10 200 python3 tools/stress_round3.py fit > $O/fit_i.jsonl 2>/dev/null || bad=$((bad+1))
  cmp -s $O/fit_ref.jsonl $O/fit_i.jsonl || { bad=$((bad+1)); cp $O/fit_i.jsonl $O/fit_diff_$i.jsonl; }
  timeout -k 10 200 python3 tools/stress_round3.py hyper > $O/hyper_i.jsonl 2>/dev/null || bad=$((bad+1))
  cmp -s $O/hyper_ref.jsonl $O/hyper_i.jsonl || { bad=$((bad+1)); cp $O/hyper_i.jsonl $O/hyper_diff_$i.jsonl; }
  echo "rep $i bad=$bad"
done
echo "SOAK bad=$bad"
