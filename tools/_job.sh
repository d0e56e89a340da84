set -e
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_bn_conv1_gpu.py -m gpu -q --tb=short -k "folded_in" 2>&1 | tail -5
python3 tools/bench_parts2.py 2>&1 | grep -v amdgpu
for v in 1 0; do
SEPT_BN_APPLY_DGRAD=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || { tail -30 $O/bench_$v.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r2t/bench_$v.json")); print("apply_in_dgrad=$v", d["value"], d["ms_per_step"], d["reference_batch"]["ms_per_step"])
PY
done
