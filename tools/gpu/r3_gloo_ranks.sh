# bench.py's N > 1 path on the ONE GPU of a test box: 4 ranks over gloo sharing the card (BENCH_BACKEND=gloo) --
# a functional rehearsal of the multi-rank step (shards, device-packed winner records, the exchange, max-over-ranks
# timing); NOT a scaling measurement: the ranks time-share one GPU
set -u
O=gpurun_out/r3g4
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in c1 c3; do
BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 4 --config $c --steps 3 --warmup 1 --no-opt-in > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
tail -1 $O/bench_$c.json | cut -c1-700
done
python3 bench.py --config c1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1-rank c1 value', d['value'])"
