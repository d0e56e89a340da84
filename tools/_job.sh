set -e
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
python3 tools/bench_parts2.py 2>&1 | grep -v amdgpu
python -m pytest tests -m gpu -q --tb=line > $O/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" $O/gpu_tests.log | cut -c1-300 | head -20
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2r/bench.json")); print(d["value"], d["ms_per_step"], d["vs_cpu_baseline"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["one_thread"]["value"])
PY
