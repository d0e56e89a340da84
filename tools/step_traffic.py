#!/usr/bin/env python3
"""HBM bytes per replayed step = sum over kernels of (bytes per launch from the PMC passes) x (launches per replayed step):

    python tools/step_traffic.py profiles/r03_pmc_traffic.json profiles/r03_replay_kernel_stats.csv > profiles/r03_step_traffic.txt

(profiles/rNN_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2 * FETCH + WRITE) * 1024 per
launch -- tools/pmc_to_json.py; profiles/rNN_replay_kernel_stats.csv: tools/replay_stats.py.)"""
import csv
import json
import sys


def main(traffic_json, stats_csv):
    tj = json.load(open(traffic_json))["kernels"]
    rows = []
    for r in csv.reader(open(stats_csv)):
        if not r or r[0].startswith("#") or r[0] == "Name":
            continue
        rows.append((r[0], float(r[2]), float(r[4]) / 1e3))
    total, missing = 0.0, []
    out = []
    for name, calls, us in rows:
        hit = tj.get(name[:90])
        if hit is None:   # names are cut to 90 characters in the traffic file
            cand = [v for k, v in tj.items() if name.startswith(k) or k.startswith(name[:60])]
            hit = cand[0] if cand else None
        if hit is None:
            missing.append(name)
            continue
        b = hit["hbm_bytes_per_launch"] * calls
        total += b
        out.append((b, calls, hit["hbm_bytes_per_launch"], us, name))
    print(f"HBM bytes per replayed step: {total / 1e9:.3f} GB  ({len(out)} kernels matched, {len(missing)} without counters)")
    print(f"{'MB/step':>9} {'calls':>5} {'MB/launch':>10} {'us/launch':>10}  kernel")
    for b, calls, per, us, name in sorted(out, reverse=True):
        print(f"{b / 1e6:9.1f} {calls:5.1f} {per / 1e6:10.2f} {us:10.1f}  {name[:100]}")
    for m in missing:
        print("no counters:", m[:100])


if __name__ == "__main__":
    main(*sys.argv[1:3])
