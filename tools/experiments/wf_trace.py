"""per-launch durations of kernel 6's two kernels from a rocprofv3 --kernel-trace CSV: python tools/wf_trace.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for key in ("pt_kernel_wf", "wf_walk"):
    xs = [d(r) for r in rows if key in r["Kernel_Name"]]
    if xs:
        print("%-14s n=%d total %.1f ms  mean %.3f  median %.3f  max %.3f | first 24: %s" % (key, len(xs), sum(xs), sum(xs) / len(xs), sorted(xs)[len(xs) // 2], max(xs), [round(x, 2) for x in xs[:24]]))
k = [r for r in rows if "pt_kernel_wf" in r["Kernel_Name"] or "wf_walk" in r["Kernel_Name"]]
if k:
    span = (int(k[-1]["End_Timestamp"]) - int(k[0]["Start_Timestamp"])) / 1e6
    busy = sum(d(r) for r in k)
    print("span %.1f ms, kernels busy %.1f ms (%.0f %%)" % (span, busy, 100 * busy / span))
