"""Build profiles/pt_kernel_model.json (what bench.py's `roofline` carries over) from rocprofv3 --pmc passes of the bench
workload.  usage:
    python tools/make_pt_model.py --samples N --source TEXT --out profiles/pt_kernel_model.json DIR [DIR ...]
DIRs hold *_counter_collection.csv of separate passes (SQ set, GRBM, FETCH_SIZE, WRITE_SIZE); N = pixel-samples traced by the
pt_kernel dispatches of ONE pass (every pass runs the same command).

Formulas:
  valu_insts_per_sample      = SQ_INSTS_VALU / N
  valu_issue_cycles_per_inst = sum_class(count_class * cycles_class) / SQ_INSTS_VALU, the SIMD cycles the kernel's own instruction
                               mix needs per wave64 instruction.  Round 5: the class costs are MEASURED IN CYCLES on the MI355X
                               (tools/microbench/valu_cost.hip + valu_cost_cycles.py, profiles/r05/valu_cost_microbench.txt: dispatches
                               of >= 20 ms of independent instructions at this kernel's occupancy, 4 waves per SIMD; cycles =
                               GRBM_GUI_ACTIVE / 8 of the dispatch, which held 2.38 GHz -- s_memtime against s_memrealtime gives the
                               same clock): f32 add/mul/fma 2.56 (the one stream under which the clock drops, to 2.0-2.1 GHz),
                               f64 add/mul/fma/ldexp/div_fixup 4.25, conversions 4.25, v_mul_lo_u32 4.26, integer add/xor/and/or/
                               shift/alignbit/lshl_add/bfe 3.65, compare + select pairs 3.67, lone v_cmp / v_cndmask_e64 / f32 min/max /
                               min3/max3 / v_readlane / v_writelane 4.25, v_mov 2.55, 64-bit add 2 x 4.29, f32 rcp/rsq/sqrt 8.28,
                               f64 rcp/rsq/sqrt 16.3.  (One wave per SIMD: 5.1 cycles for everything but the transcendentals, 9.2 / 17.2.)
                               Rounds 3-4 measured the same ratios on 0.2-0.7 ms dispatches and ANCHORED them at "f64 fma = 4 cycles";
                               the measured 4.25 moves every cost up by 6 %.  Classes from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F{32,64},
                               _INT32, _INT64, _CVT; "other" = SQ_INSTS_VALU minus their sum (compares, selects, min/max, logic, moves,
                               lane moves), priced at 3.65 = its cheapest members besides v_mov: a LOWER bound of the class's cost for
                               these kernels, so the peak is an upper bound and `frac` a lower bound.
  valu_busy_measured         = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)   (can exceed 1: active intervals of the
                               waves of one SIMD overlap)
  lane_utilisation           = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  clock_ghz                  = GRBM_GUI_ACTIVE / 8 / kernel time            (the counter sums the 8 XCDs)
  valu_busy                  = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
  hbm_bytes_per_sample       = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / N      (FETCH_SIZE x2: gfx950 correction, MI355X_MICROARCH.md)
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=float, required=True)
ap.add_argument("--source", default="")
ap.add_argument("--kernel", default="pt_kernel")
ap.add_argument("--out", required=True)
ap.add_argument("dirs", nargs="+")
a = ap.parse_args()


def kernel_source_sha16():
    """hash of the device sources the model was measured on (bench.py compares it with the tree it runs in)"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-raytracer_amd", "csrc")
    h = hashlib.sha256()
    for rel in ("device/kernels.hip", "device/wavefront.inc", "device/sppm.inc", "device/device.h", "common/flat.h", "common/rng.h", "common/detlog.h", "common/schedule.h", "host/schedule.cpp"):
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

tot = defaultdict(float)      # counter -> sum over pt_kernel dispatches
dur = defaultdict(float)      # counter -> kernel ns of the pass that counted it
names = set()
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if a.kernel not in r["Kernel_Name"]:
                continue
            names.add(r["Kernel_Name"].split("(")[0])
            c = r["Counter_Name"]
            tot[c] += float(r["Counter_Value"])
            key = (c, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

N = a.samples
m = {"kernel": sorted(names), "samples_in_pass": N, "source": a.source, "kernel_source_sha16": kernel_source_sha16(),
     "counters": {k: tot[k] for k in sorted(tot)}}
OTHER_CYCLES = 3.65
CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.56, "SQ_INSTS_VALU_MUL_F32": 2.56, "SQ_INSTS_VALU_FMA_F32": 2.56, "SQ_INSTS_VALU_TRANS_F32": 8.28,
                "SQ_INSTS_VALU_ADD_F64": 4.25, "SQ_INSTS_VALU_MUL_F64": 4.25, "SQ_INSTS_VALU_FMA_F64": 4.25, "SQ_INSTS_VALU_TRANS_F64": 16.3,
                "SQ_INSTS_VALU_INT64": 8.58, "SQ_INSTS_VALU_CVT": 4.25, "SQ_INSTS_VALU_INT32": 3.65}
COST_ANCHOR = ("cycles = GRBM_GUI_ACTIVE / 8 of >= 20 ms dispatches of independent instructions at 4 waves per SIMD, clock held 2.38 GHz "
               "(profiles/r05/valu_cost_microbench.txt); 'other' at its cheapest members besides v_mov (3.65): frac is a lower bound")
if "SQ_INSTS_VALU" in tot:
    m["valu_insts_per_sample"] = tot["SQ_INSTS_VALU"] / N
    if all(c in tot for c in CLASS_CYCLES):
        classed = sum(tot[c] for c in CLASS_CYCLES)
        cyc = sum(tot[c] * k for c, k in CLASS_CYCLES.items()) + OTHER_CYCLES * (tot["SQ_INSTS_VALU"] - classed)
        m["valu_issue_cycles_per_inst"] = cyc / tot["SQ_INSTS_VALU"]
        m["valu_mix_per_sample"] = {c.replace("SQ_INSTS_VALU_", "").lower(): tot[c] / N for c in CLASS_CYCLES}
        m["valu_mix_per_sample"]["other"] = (tot["SQ_INSTS_VALU"] - classed) / N
        m["valu_class_cycles"] = {c.replace("SQ_INSTS_VALU_", "").lower(): k for c, k in CLASS_CYCLES.items()}
        m["valu_class_cycles"]["other"] = OTHER_CYCLES
        m["valu_cost_anchor"] = COST_ANCHOR
    if "SQ_ACTIVE_INST_VALU" in tot:
        m["valu_active_cycles_per_inst_measured"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"]
if "SQ_THREAD_CYCLES_VALU" in tot and "SQ_ACTIVE_INST_VALU" in tot:
    m["lane_utilisation"] = tot["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot["SQ_ACTIVE_INST_VALU"])
if "GRBM_GUI_ACTIVE" in tot and dur["GRBM_GUI_ACTIVE"] > 0:
    m["clock_ghz"] = tot["GRBM_GUI_ACTIVE"] / 8.0 / dur["GRBM_GUI_ACTIVE"]
    m["kernel_ms_in_pass"] = dur["GRBM_GUI_ACTIVE"] / 1e6
    if "SQ_ACTIVE_INST_VALU" in tot:
        m["valu_busy_measured"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0)
    if "valu_issue_cycles_per_inst" in m:  # fraction of the SIMDs' cycles that the kernel's instruction mix needs at the very least, in the PMC pass itself
        m["valu_issue_frac_in_pass"] = tot["SQ_INSTS_VALU"] * m["valu_issue_cycles_per_inst"] / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0)
if "SQ_WAVE_CYCLES" in tot:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in tot:
            m["share_" + k.lower()] = tot[k] / tot["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
    m["hbm_fetch_bytes"] = 2.0 * 1024.0 * tot.get("FETCH_SIZE", 0.0)
    m["hbm_write_bytes"] = 1024.0 * tot.get("WRITE_SIZE", 0.0)
    m["hbm_bytes_per_sample"] = (m["hbm_fetch_bytes"] + m["hbm_write_bytes"]) / N
json.dump(m, open(a.out, "w"), indent=1)
print(json.dumps({k: v for k, v in m.items() if k != "counters"}, indent=1))
