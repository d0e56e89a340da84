set -e
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
for m in 0 1 2 4 8 16 32 256 30; do timeout -k 10 60 ./tools/mel_abl_$m.bin; done > $O/mel_ablation.txt
cat $O/mel_ablation.txt
python -m pytest tests -m gpu -q --tb=line > $O/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" $O/gpu_tests.log | cut -c1-300 | head -20
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3d/bench.json")); print(d["value"], d["ms_per_step"], d["mel"])
PY
