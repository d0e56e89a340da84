set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r2b/gpu_tests.log; exit 1; }
tail -5 gpurun_out/r2b/gpu_tests.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err || { tail -30 gpurun_out/r2b/bench.err; exit 1; }
cat gpurun_out/r2b/bench.json
