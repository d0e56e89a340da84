set -e
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
python -m pytest tests/test_small_ops_gpu.py -m gpu -q --tb=short -x -k "gru" > $O/t.log 2>&1 || true
tail -2 $O/t.log
python tools/bench_gru.py
