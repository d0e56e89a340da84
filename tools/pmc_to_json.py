"""Builds profiles/rNN_pmc_traffic.json from two rocprofv3 counter passes of bench.py:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Counter unit is KiB.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half
the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section); the factor is checked on
sept_bn_stats_partial_kernel<4>, whose input is exactly B*H*W*32*2 bytes."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def tag_of(name):
    m = re.search(r"sept_conv5x5_mfma_kernel<(\d+), (\d+)((?:, -?\d+)*)>", name)
    if m:   # ninth template argument = loader form (1: BatchNorm backward apply, 2: pool-first activation): ops.py's tags
        rest = [v.strip() for v in m.group(3).split(",") if v.strip()]
        ld = {"1": "+bnapply", "2": "+act"}.get(rest[6], "") if len(rest) >= 7 else ""
        return f"conv5x5_mfma<{m.group(1)},{m.group(2)}>{ld}"
    m = re.search(r"sept_conv5x5_wgrad_kernel<(\d+), (\d+)", name)
    if m:
        return f"conv5x5_wgrad<{m.group(1)},{m.group(2)}>"
    if "sept_mel_stft_kernel" in name or "sept_mel_shfl_kernel" in name:
        return "mel"
    return None


def read(dirname, counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, out):
    fe, wr = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        f = sum(fe.get(name, [0])) / max(len(fe.get(name, [])), 1)
        w = sum(wr.get(name, [0])) / max(len(wr.get(name, [])), 1)
        kernels[name[:90]] = {"tag": tag_of(name), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                              "hbm_bytes_per_launch": int((2 * f + w) * 1024), "launches": len(fe.get(name, []))}
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) of `python3 bench.py "
            "--steps 3 --warmup 1 --no-cpu-baseline --no-ref-batch` (32 clips = 224 windows, F=80).  Counter unit KiB.  "
            "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half the bytes of 16-B/lane "
            "streaming reads (MI355X_MICROARCH.md, HBM section); check: sept_bn_stats_partial_kernel<4> reads "
            "exactly 224*200*80*32*2 = 229.4 MB.  Means over all launches of a kernel name (the mel kernel mixes "
            "the B=32 step launches with the B=256 timing launches).")
    json.dump({"_note": note, "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
