"""Per-kernel means of the counters in a rocprofv3 --pmc --output-format csv run directory."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if pat in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
