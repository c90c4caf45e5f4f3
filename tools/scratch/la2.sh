set -u
R=$PWD
mkdir -p gpurun_out/la2
export TMPDIR=/tmp
cd /tmp
export TGP_BG_CUS=192
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/la2/tr -o run -- python3 $R/tools/bench_fit.py 4096 --reps 3 > $R/gpurun_out/la2/out.txt 2>&1
echo rc=$?
