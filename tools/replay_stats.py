#!/usr/bin/env python3
"""Per-kernel statistics of the REPLAYED step only, from a rocprofv3 kernel trace of
`python3 bench.py --replay-only --steps K --warmup W --no-cpu-baseline`:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_replay -- python3 bench.py --replay-only --steps 30
    python tools/replay_stats.py gpurun_out/prof_replay 30 profiles/r02_replay_kernel_stats.csv

A step starts with its STFT->mel launch; the last K of those mark the timed replays (everything before -- eager
warm-up steps, the two graph warm-ups -- is dropped), so the averages are those of the kernels as they run inside
the captured two-stream step, nothing else mixed in.  Also prints the per-step sum of kernel durations and the
wall time from the first kernel start to the last kernel end of the window."""
import csv
import glob
import sys
from collections import OrderedDict


def main(run_dir, steps, out_csv):
    steps = int(steps)
    f = glob.glob(f"{run_dir}/**/*kernel_trace.csv", recursive=True)
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "sept_mel_" in r["Kernel_Name"]]
    if len(starts) < steps:
        raise SystemExit(f"only {len(starts)} step starts in the trace, need {steps}")
    win = rows[starts[-steps]:]
    agg = OrderedDict()
    for r in win:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(name, [0, 0, 10 ** 18, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    wall = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, a[0], round(a[0] / steps, 2), a[1], round(a[1] / a[0], 1), a[2], a[3], round(100.0 * a[1] / total, 2)])
        w.writerow([f"# window: last {steps} replayed steps; kernel time per step {total / steps / 1e3:.1f} us on two streams; "
                    f"wall per step (first start to last end) {wall / steps / 1e3:.1f} us", "", "", "", "", "", "", ""])
    print(f"wrote {out_csv}: {len(agg)} kernels, {total / steps / 1e3:.1f} us of kernels per step, wall {wall / steps / 1e3:.1f} us per step")


if __name__ == "__main__":
    main(*sys.argv[1:4])
