set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -q --tb=line > gpurun_out/r2g/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" gpurun_out/r2g/gpu_tests.log | cut -c1-300 | head -40
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err || { tail -30 gpurun_out/r2g/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2g/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
PY
