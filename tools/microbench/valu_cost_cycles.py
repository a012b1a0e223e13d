"""Joins the output of tools/microbench/valu_cost.bin (VALU_COST lines) with the GRBM_GUI_ACTIVE counts of the same dispatches
(rocprofv3 --pmc GRBM_GUI_ACTIVE ... -- tools/microbench/valu_cost.bin): shader cycles per wave64 instruction per SIMD with the clock
the GPU actually held, at 1 and 4 waves per SIMD.   usage: python valu_cost_cycles.py OUTPUT.txt PMC_DIR"""
import csv, glob, os, re, sys
rows = {}
for line in open(sys.argv[1]):
    if line.startswith("VALU_COST"):
        kv = dict(re.findall(r'(\w+)=("[^"]*"|\S+)', line))
        rows[(int(kv["op"]), int(kv["wps"]))] = kv
grbm = {}
for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k<(\d+), *(\d+)>", r["Kernel_Name"])
        if not m or r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        key = (int(m.group(1)), int(m.group(2)))
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if key not in grbm or dur > grbm[key][1]:      # the measurement is the longer of the two dispatches of a kernel (the other calibrates)
            grbm[key] = (float(r["Counter_Value"]), dur)
print("%-52s %4s %10s %12s %12s %10s %12s" % ("instruction(s)", "wps", "ms", "ns/inst/SIMD", "cycles/inst", "clock GHz", "memtime/ns"))
for key in sorted(rows, key=lambda k: (k[1], k[0])):
    kv = rows[key]
    n = float(kv["inst_per_wave"]) * key[1]
    ns = float(kv["ns_per_inst_per_simd_realtime"])
    if key in grbm:
        cyc_total, dur = grbm[key]
        cyc = cyc_total / 8.0 / n          # the counter sums the 8 XCDs; n = instructions a SIMD issued
        clk = cyc_total / 8.0 / dur
        print("%-52s %4d %10.2f %12.4f %12.3f %10.3f %12.4f" % (kv["name"].strip('"'), key[1], dur / 1e6, ns, cyc, clk, float(kv["memtime_ticks_per_ns"])))
    else:
        print("%-52s %4d %10.2f %12.4f %12s %10s %12.4f" % (kv["name"].strip('"'), key[1], float(kv["event_ms"]), ns, "-", "-", float(kv["memtime_ticks_per_ns"])))
