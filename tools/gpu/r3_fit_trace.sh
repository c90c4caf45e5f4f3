set -u
R=$PWD
O=$R/gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for f in 0 1; do
  export TGP_PANEL_FUSE=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$f -o t -- python3 $R/tools/bench_fit.py 4096 --reps 5 > $O/fit$f.json 2> $O/trace$f.err; echo "trace$f rc=$?"
done
cd $R
python3 tools/trace_summary.py $O/trace0 > $O/summary0.txt
python3 tools/trace_summary.py $O/trace1 > $O/summary1.txt
cat $O/summary0.txt $O/summary1.txt
