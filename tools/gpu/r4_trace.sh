#!/bin/bash
# rocprofv3 kernel trace of one fit at N = 4096 and N = 8192 -> tools/trace_summary.py; in-kernel stamps of the panel chain
set -u
R=$PWD
O=$R/gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for n in 4096 8192; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$n -o t -- python3 $R/tools/bench_fit.py $n --reps 5 > $O/fit$n.json 2> $O/trace$n.err; echo "trace$n rc=$?"
done
cd $R
for n in 4096 8192; do python3 tools/trace_summary.py $O/trace$n > $O/summary$n.txt; cat $O/summary$n.txt; done
TGP_STAMP_FILE=$O/stamps.bin timeout -k 10 200 python3 tools/bench_fit.py 4096 --reps 3 > $O/fit_stamp.json 2> $O/fit_stamp.err; echo "rc=$?"
python3 tools/stamp_summary.py $O/stamps.bin > $O/stamps.txt; tail -18 $O/stamps.txt | cut -c1-250
