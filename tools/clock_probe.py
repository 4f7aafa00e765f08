"""Average shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run:
   python tools/clock_probe.py <dir>     (GRBM_GUI_ACTIVE cycles / kernel duration)"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
acc = defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        ns, name = dur[r["Dispatch_Id"]]
        k = name.split("(")[0].replace("void (anonymous namespace)::", "")[:48]
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += ns
        acc[k][2] += 1
for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if ns > 0:
        print(f"{k:50s} launches {n:4d}  total {ns / 1e6:8.3f} ms  GRBM_GUI_ACTIVE/ns = {cyc / ns:6.3f} (GHz if one counter instance)")
