# kernel trace of the last step of bench.py under an overlap mode: r5_trace.sh <out> <overlap> <shard-of> [config]
set -u
O=${1:-gpurun_out/r5trace}; OV=${2:-1}; G=${3:-8}; C=${4:-c3}
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -o run -- python3 $R/bench.py --config $C --shard-of $G --overlap $OV --steps 3 --warmup 2 --no-cpu-baseline --no-opt-in > "$R/$O/bench.json" 2> "$R/$O/bench.err"; echo "trace rc=$?"
cd $R
python3 tools/trace_timeline.py $O/trace/run_kernel_trace.csv > $O/timeline.txt 2>&1
tail -5 $O/timeline.txt
