# does the bench line's whole-run HIP-event average of the dominant kernel agree with rocprofv3's csv of the same command?
set -u
R=$PWD; O=$R/gpurun_out/chk; mkdir -p $O
export TMPDIR=/tmp
for c in c2 c3 c1; do
  case $c in c1) ST="--steps 20 --warmup 3";; *) ST="--steps 5 --warmup 2";; esac
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$c -o run -- python3 $R/bench.py --config $c $ST --no-cpu-baseline --no-opt-in > $O/b_$c.json 2> $O/b_$c.err
  cd $R
  python3 - $O $c <<'PY'
import json,csv,glob,sys
O,c=sys.argv[1],sys.argv[2]
d=json.load(open("%s/b_%s.json"%(O,c)))["roofline"]
rows=list(csv.DictReader(open(glob.glob("%s/st_%s/**/run_kernel_stats.csv"%(O,c),recursive=True)[0])))
t=max((r for r in rows if "trmm_sumsq" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
a=float(t["AverageNs"])/1e6
print("%s timed avg %.5f (%d launches) | whole run avg %.5f (%d launches) | csv avg %.5f (%s calls) -> whole-run / csv = %.4f"%(c,d["avg_launch_ms"],d["launches"],d["whole_run"]["avg_launch_ms"],d["whole_run"]["launches"],a,t["Calls"],d["whole_run"]["avg_launch_ms"]/a))
PY
done
