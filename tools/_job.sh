set -e
export TMPDIR=/tmp
O=gpurun_out/r4m
mkdir -p $O
SEPT_BENCH_DEVICE=0 SEPT_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/dp2.json 2> $O/dp2.err || tail -20 $O/dp2.err
python - <<PY
import json
d=json.loads(open("$O/dp2.json").read().strip().splitlines()[-1]); print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"].get("host_enqueue_ms_per_step"), d["config"]["parallelism"])
PY
