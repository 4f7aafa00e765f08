"""Sum rocprofv3 --pmc counter_collection csv rows per kernel name:  python tools/pmc_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    launches = max(n[(k, c)] for c in d)
    print(f"{k:62s} launches {launches:4d}  " + "  ".join(f"{c}={v:.3e}" for c, v in sorted(d.items())))
