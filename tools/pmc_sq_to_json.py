#!/usr/bin/env python3
"""profiles/rNN_pmc_sq.json from a rocprofv3 SQ-counter pass (own run: --pmc with --kernel-trace only):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \\
              SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python3 <cmd>
    python tools/pmc_sq_to_json.py <dir> profiles/r02_pmc_sq_<what>.json "<command line>"

Per kernel (mean over its launches): the raw counters and the derived figures the north star asks for --
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)   [cycles: MI355X_MICROARCH.md, counter units]
  lds_busy_frac    = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles),  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  clock_GHz        = GRBM_GUI_ACTIVE / 8 XCDs / duration;  kernel cycles = GRBM_GUI_ACTIVE / 8
  wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(run_dir, out, cmd=""):
    cnt = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{run_dir}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"].replace("(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    for f in glob.glob(f"{run_dir}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].replace("(anonymous namespace)::", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    kernels = {}
    for name, cs in cnt.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = dur.get(name)
        k = {"launches": len(next(iter(cs.values()))), "counters": {c: round(v) for c, v in sorted(m.items())}}
        if d:
            k["mean_duration_us_under_pmc"] = round(sum(d) / len(d) / 1e3, 1)
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc > 0:
            if d:
                k["clock_GHz"] = round(cyc / (sum(d) / len(d)), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                k["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
            if "SQ_LDS_IDX_ACTIVE" in m:
                k["lds_busy_frac"] = round(m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for key, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY"),
                           ("valu_active_frac", "SQ_ACTIVE_INST_VALU")):
                if c in m:
                    k[key] = round(m[c] / wc, 4)
        kernels[name[:110]] = k
    json.dump({"_command": cmd, "_note": __doc__.split("Per kernel", 1)[1].strip(), "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
