set -e
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -x -k "hand_scheduled" > $O/t.log 2>&1 || true
tail -25 $O/t.log
