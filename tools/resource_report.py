"""Register / scratch / LDS report of every kernel in csrc/device/kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_report.py [-DFLAG ...] > profiles/rNN/kernel_resource_usage.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(ROOT, "rust-raytracer_amd")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(pkg, "csrc"), "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(pkg, "csrc/device/kernels.hip"), "-o", "/tmp/kernels_resource.o"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in err.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["/usr/bin/c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
print("# " + " ".join(cmd[:1] + cmd[1:8] + sys.argv[1:]) + " ... -Rpass-analysis=kernel-resource-usage")
print("%-62s %5s %5s %5s %8s %9s %10s %10s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "waves/SIMD", "SGPR spill", "VGPR spill"))
for r in rows:
    print("%-62s %5s %5s %5s %8s %9s %10s %10s" % (r["name"][-62:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                                   r.get("Occupancy"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
