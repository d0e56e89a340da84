set -e
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
run() { n=$1; shift
  env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-ref-batch > $O/bench_$n.json 2> $O/bench_$n.err || { tail -5 $O/bench_$n.err; return 0; }
  python - <<PY
import json
d=json.load(open("$O/bench_$n.json")); print("$n", d["value"], d["ms_per_step"])
PY
}
run q4 A=1
run q6 DEBUG_HIP_FORCE_GRAPH_QUEUES=6
run q8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run q12 DEBUG_HIP_FORCE_GRAPH_QUEUES=12
run q16 DEBUG_HIP_FORCE_GRAPH_QUEUES=16
run q5 DEBUG_HIP_FORCE_GRAPH_QUEUES=5
run q3 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
