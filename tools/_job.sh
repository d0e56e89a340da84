set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q --tb=line > gpurun_out/r2h/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" gpurun_out/r2h/gpu_tests.log | cut -c1-300 | head -40
for v in 1 0 1 0; do
SEPT_BN_DGRAD_SUMS=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2h/bench_$v.json 2> gpurun_out/r2h/bench_$v.err || { tail -30 gpurun_out/r2h/bench_$v.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r2h/bench_$v.json")); print("dgrad_sums=$v", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"])
PY
done
