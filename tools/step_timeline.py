"""Per-queue timeline of ONE replayed step from a rocprofv3 kernel trace (bench.py --replay-only):
   python tools/step_timeline.py <rocprof output dir> [step index]
Columns: hardware queue, start (us from the step's mel kernel), duration (us), kernel."""
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\([A-Za-z]+Args\)", "", n)
    n = re.sub(r"\((?!.*<).*$", "", n)
    return n[:72]


def main():
    d, step = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "sept_mel_" in r["Kernel_Name"]]
    s, e = idx[step], idx[step + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:e]:
        st, en = (int(r["Start_Timestamp"]) - t0) / 1000, (int(r["End_Timestamp"]) - t0) / 1000
        print(f"{r['Queue_Id']:>3} {st:8.1f} {en - st:7.1f}  {short(r['Kernel_Name'])}")
    print(f"step span {(int(rows[e]['Start_Timestamp']) - t0) / 1000:.1f} us")


if __name__ == "__main__":
    main()
