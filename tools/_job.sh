set -e
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
python -m pytest tests/test_bn_conv1_gpu.py -m gpu -q --tb=short -x > $O/t.log 2>&1 || true
tail -3 $O/t.log
timeout -k 10 300 python tools/bench_parts2.py 2>&1 | head -6
