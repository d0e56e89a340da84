set -e
export TMPDIR=/tmp
O=gpurun_out/r2v
mkdir -p $O
SEPT_BENCH_NO_WGRAD=1 python3 tools/bench_conv.py 224 2>&1 | grep "^conv"
python - <<'PY'
import os, sys, torch
sys.path.insert(0, "speech-emotion-privacy-trust_amd")
from sept_amd import ops
B=224
for (H, W, ci, co, mode) in [(100, 64, 64, 32, 1), (50, 32, 128, 64, 1)]:
    x = torch.randn(B, H, W, ci, device="cuda").bfloat16()
    w = torch.randn((ci, co, 5, 5), device="cuda") * 0.05
    wt = ops.conv5x5_prep_weights(w, mode); y = ops.conv5x5(x, wt)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(20): ops.conv5x5(x, wt, out=y)
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / 20
    print(f"conv {ci}->{co} {H}x{W}: {ms*1e3:.1f} us {2.0*B*H*W*ci*co*25/ms/1e9:.1f} TFLOP/s")
PY
python -m pytest tests -m gpu -q --tb=line > $O/gpu_tests.log 2>&1 || true
grep -E "^/root|^E |Error|passed|failed" $O/gpu_tests.log | cut -c1-300 | head -20
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2v/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["reference_batch"]["ms_per_step"])
PY
