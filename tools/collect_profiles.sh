#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh gpurun_out/<dir> <round tag, e.g. r02>
# kernel trace of the replayed step, SQ counter passes (MFMA / LDS / wait fractions, clocks), instruction counts of
# the mel kernel, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) and the bench line itself.
set -e
export TMPDIR=/tmp
O=$1; R=${2:-r04}
mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/${R}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/replay -- python3 bench.py --replay-only --steps 30 --warmup 3 > $O/replay.log 2>&1
python3 tools/replay_stats.py $O/replay 30 $O/${R}_replay_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/full.log 2>&1
cp $(ls $O/full/*/*kernel_stats.csv | head -1) $O/${R}_bench_step_kernel_stats.csv
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_conv -- python3 tools/bench_conv.py 224 > $O/sq_conv.log 2>&1
python3 tools/pmc_sq_to_json.py $O/sq_conv $O/${R}_pmc_sq_conv.json "rocprofv3 --pmc $SQ --kernel-trace --output-format csv -- python3 tools/bench_conv.py 224"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/sq_mel -- python3 tools/bench_mel.py --iters 10 > $O/sq_mel.log 2>&1
python3 tools/pmc_sq_to_json.py $O/sq_mel $O/${R}_pmc_sq_mel.json "rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -- python3 tools/bench_mel.py --iters 10"
SQ3="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $O/inst_mel -- python3 tools/bench_mel.py --iters 10 > $O/inst_mel.log 2>&1
python3 tools/pmc_sq_to_json.py $O/inst_mel $O/${R}_pmc_insts_mel.json "rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -- python3 tools/bench_mel.py --iters 10"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-batch > $O/sq_step.log 2>&1
python3 tools/pmc_sq_to_json.py $O/sq_step $O/${R}_pmc_sq_step.json "rocprofv3 --pmc $SQ --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-batch"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-batch > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-batch > $O/write.log 2>&1
python3 tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/${R}_pmc_traffic.json
python3 tools/step_traffic.py $O/${R}_pmc_traffic.json $O/${R}_replay_kernel_stats.csv > $O/${R}_step_traffic.txt
head -3 $O/${R}_step_traffic.txt
# kernel-only timings of every (n_fft, n_mels) the reference extracts (mel1 = 800, mel2 = 1600, default 1024, MFCC 400)
for nf in 800 1600 1024 400; do for m in 80 128; do python3 tools/bench_mel.py --n_fft $nf --mels $m --iters 30 >> $O/${R}_mel_timings.txt; done; done
python3 tools/bench_mel.py --n_fft 400 --hop 200 --mels 128 --iters 30 >> $O/${R}_mel_timings.txt   # the MFCC front end
SEPT_STAMPS=1 python3 tools/step_stamps.py > $O/${R}_step_stamps.txt 2>/dev/null
# the in-kernel launch clock behind the bench line's roofline figure: against HIP events alone / on two streams, and the
# instrumented capture against the plain one
python3 tools/kclock_check.py > $O/${R}_kclock_check.txt 2>/dev/null
rm -rf $O/replay $O/full $O/sq_conv $O/sq_mel $O/inst_mel $O/sq_step $O/pmc_fetch $O/pmc_write
ls -la $O/${R}_*
