set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
python -m pytest tests -m gpu -q --tb=line > gpurun_out/r2l/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" gpurun_out/r2l/gpu_tests.log | cut -c1-300 | head -30
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2l/bench_$i.json 2> gpurun_out/r2l/bench_$i.err || { tail -30 gpurun_out/r2l/bench_$i.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r2l/bench_$i.json")); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"])
PY
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2l/replay -- python3 bench.py --replay-only --steps 30 --warmup 3 > gpurun_out/r2l/replay.log 2>&1
python3 tools/replay_stats.py gpurun_out/r2l/replay 30 gpurun_out/r2l/replay_kernel_stats.csv
