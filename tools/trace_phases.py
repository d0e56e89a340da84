"""Per-phase average duration of one kernel in a rocprofv3 kernel trace of bench.py (default flags):
launch order is warm-up + probe steps (eager, two streams), graph replays (warm + timed), the eager
timing steps after the timed region, and the 'alone' steps (branches serialised).

    python tools/trace_phases.py <kernel_trace.csv> "sept_conv5x5_mfma_kernel<64, 32" [launches_per_step=2]
"""
import csv
import statistics as st
import sys


def main(path, needle, per_step=2, warmup=3, probe=2, graph_warm=2, steps=10, post=3, alone=2):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if needle in r["Kernel_Name"]]
    cuts = [("eager warm-up + probe", (warmup + probe) * per_step), ("graph replays", (graph_warm + steps) * per_step),
            ("eager timing steps", post * per_step), ("alone", alone * per_step)]
    i = 0
    print(f"{len(d)} launches of {needle!r}")
    for name, n in cuts:
        part = d[i:i + n]
        i += n
        if part:
            print(f"  {name:24s} n={len(part):3d}  mean {st.mean(part):7.1f} us  min {min(part):7.1f}  max {max(part):7.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2)
