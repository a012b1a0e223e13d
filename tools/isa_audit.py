#!/usr/bin/env python3
"""ISA audit of one kernel in a hipcc -save-temps .s file: per basic block, the number of VALU / SALU / LDS / VMEM / scratch /
branch instructions, with the block's loop depth and the source-level label the compiler left on it.

  python tools/isa_audit.py <file.s> <mangled-kernel-name-substring> [--dump]

VALU classes follow tools/make_pt_model.py (f64 arithmetic, f32 arithmetic, transcendental, 64-bit integer, conversions, other)."""
import re
import sys
from collections import OrderedDict


def classify(mn):
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"
    if mn.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_memtime"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def valu_class(mn):
    m = mn
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f64", m):
        return "trans64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f16)", m):
        return "trans32"
    if re.match(r"v_(add|mul|fma|fmac|mad|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|trunc|floor|ceil|rndne|fract|min|max)_f64", m):
        return "f64"
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mad|fmaak|fmamk)_f32", m) or m.startswith("v_pk_"):
        return "f32"
    if re.match(r"v_cvt_", m):
        return "cvt"
    if re.match(r"v_(lshlrev|lshrrev|ashrrev|add_co|addc_co|mad_u64|mul_lo|mul_hi|mad_i64)_(b64|u64|i64|u32|i32)", m) and ("64" in m or "mul_" in m or "mad_u64" in m):
        return "int64/mul"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if key in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S*:", l):
            start = i
            break
        if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {"label": "", "depth": 0, "n": {}, "vc": {}, "lines": []}
    i = start + 1
    while i < len(lines) and not lines[i].startswith(".Lfunc_end") and ".amdhsa_kernel" not in lines[i]:
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        mb = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m or mb:
            cur = m.group(1) if m else "bb.%s" % mb.group(1)
            lab = (m.group(2) if m else mb.group(2)) or ""
            blocks[cur] = {"label": lab.strip("; ").strip(), "depth": 0, "n": {}, "vc": {}, "lines": []}
            j = i + 1
            while j < len(lines) and lines[j].strip().startswith(";"):
                d = re.search(r"Depth[= ](\d+)", lines[j])
                if d:
                    blocks[cur]["depth"] = max(blocks[cur]["depth"], int(d.group(1)))
                j += 1
            d = re.search(r"Depth[= ](\d+)", l)
            if d:
                blocks[cur]["depth"] = max(blocks[cur]["depth"], int(d.group(1)))
        else:
            s = l.strip()
            if s and not s.startswith((";", ".", "//")):
                mn = s.split()[0]
                c = classify(mn)
                b = blocks[cur]
                b["n"][c] = b["n"].get(c, 0) + 1
                if c == "valu":
                    vc = valu_class(mn)
                    b["vc"][vc] = b["vc"].get(vc, 0) + 1
                b["lines"].append(s)
        i += 1
    tot = {}
    print("%-12s %5s %5s %5s %4s %5s %7s %6s  %s" % ("block", "depth", "VALU", "SALU", "LDS", "VMEM", "scratch", "branch", "label / VALU classes"))
    for k, b in blocks.items():
        n = b["n"]
        if not n:
            continue
        for c, v in n.items():
            tot[c] = tot.get(c, 0) + v
        vcs = " ".join("%s:%d" % kv for kv in sorted(b["vc"].items()))
        print("%-12s %5d %5d %5d %4d %5d %7d %6d  %s | %s" % (k, b["depth"], n.get("valu", 0) + n.get("valu_lane", 0), n.get("salu", 0), n.get("lds", 0), n.get("vmem", 0),
                                                          n.get("scratch", 0), n.get("branch", 0), b["label"][:70], vcs))
        if dump:
            for s in b["lines"]:
                print("        " + s)
    print("TOTAL", tot)


if __name__ == "__main__":
    main()
