"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel and counter: python tools/pmc_summary.py DIR [DIR ...]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
print("kernel,counter,dispatches,mean,sum")
for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%s,%d,%.6g,%.6g' % (k, c, n, s / n, s))
