set -u
O=gpurun_out/r3s
mkdir -p $O
TGP_PANEL_FUSE=0 timeout -k 10 400 python3 tools/stress_round3.py fit > $O/fit0.jsonl 2> $O/fit0.err; echo "fit0 rc=$?"
TGP_PANEL_FUSE=1 TGP_PANEL_FUSE_TILES=100000 timeout -k 10 400 python3 tools/stress_round3.py fit > $O/fit1.jsonl 2> $O/fit1.err; echo "fit1 rc=$?"
timeout -k 10 400 python3 tools/stress_round3.py fit > $O/fit2.jsonl 2> $O/fit2.err; echo "fit2 (default) rc=$?"
cmp $O/fit0.jsonl $O/fit1.jsonl && cmp $O/fit0.jsonl $O/fit2.jsonl && echo "FIT BIT-IDENTICAL ($(wc -l < $O/fit0.jsonl) sizes)"
TGP_HYPER_WGS=1 timeout -k 10 400 python3 tools/stress_round3.py hyper > $O/hyper1.jsonl 2> $O/hyper1.err; echo "hyper1 rc=$?"
timeout -k 10 400 python3 tools/stress_round3.py hyper > $O/hyper3.jsonl 2> $O/hyper3.err; echo "hyper3 rc=$?"
cmp $O/hyper1.jsonl $O/hyper3.jsonl && echo "HYPER BIT-IDENTICAL"; cut -c1-160 $O/hyper3.jsonl
timeout -k 10 400 python3 tools/stress_round3.py host 2> $O/host.err | tee $O/host.json; echo "host rc=$?"; tail -3 $O/host.err
