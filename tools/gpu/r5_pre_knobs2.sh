set -u
O=${1:-gpurun_out/r5knobs2}
mkdir -p $O
run() {
  tag=$1; shift
  for g in 8 1; do
    env "$@" timeout -k 10 200 python3 bench.py --config c3 --shard-of $g --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in $EXTRA > $O/${tag}_s$g.json 2> $O/${tag}_s$g.err
    python3 - $O/${tag}_s$g.json "$tag" $g <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-34s G=%s  ms/step %7.3f  fit %.3f sweep %.3f"%(sys.argv[2],sys.argv[3],d["ms_per_step"],d["fit_ms"],d["sweep_ms"]),flush=True)
except Exception as e: print(sys.argv[2],sys.argv[3],"failed",e)
PY
  done
}
EXTRA="--overlap 0"; run serial TGP_PRE_CU0=0
EXTRA="--overlap 1"; run front_192 TGP_PRE_CU0=0
EXTRA="--overlap 1"; run front_128hi TGP_PRE_CUS=128 TGP_PRE_CU0=128
EXTRA="--overlap 1"; run front_unmasked TGP_PRE_CUS=0
EXTRA="--overlap 2"
run rows8_192_81k TGP_PRE_CU0=0
run rows8_128hi_64k TGP_PRE_CUS=128 TGP_PRE_CU0=128 TGP_PRE_LDS_KB=64
run rows12_192_64k TGP_PRE_LDS_KB=64 TGP_PRE_TILES=12
run rows8_unmasked_64k TGP_PRE_CUS=0 TGP_PRE_LDS_KB=64
EXTRA="--overlap 0"; run serial_again TGP_PRE_CU0=0
python3 tools/bench_fit.py 512 1024 2048 4096 8192
