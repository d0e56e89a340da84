set -e
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
python -m pytest tests -m gpu -q --tb=line > $O/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" $O/gpu_tests.log | cut -c1-300 | head -20
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_1.json 2> $O/bench_1.err
python - <<PY
import json
d=json.load(open("$O/bench_1.json")); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"])
PY
