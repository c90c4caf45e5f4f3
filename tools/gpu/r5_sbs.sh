#!/bin/bash
# evaluations of the hyper-parameter objective alone and three side by side, kernel trace per hardware queue
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5sbs
mkdir -p $O
for N in 1000 500; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$N -o run -- python3 $R/tools/trace_side_by_side.py run $N 150 > $O/run_$N.txt 2>&1
  f=$(ls $O/t$N/*kernel_trace.csv $O/t$N/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 $R/tools/trace_side_by_side.py analyse $f > $O/analysis_$N.txt 2>&1
  rm -rf $O/t$N
  cat $O/run_$N.txt $O/analysis_$N.txt
done
