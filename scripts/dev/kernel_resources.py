"""Compact per-kernel register / scratch / LDS / occupancy table for a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/dev/kernel_resources.py [file.hip] [name filter]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd", "csrc", "das_kernels.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-jump-tables", "--cuda-device-only",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
for r in rows:
    name = re.sub(r"\(.*", "", r["name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt in name:
        print("%-70s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
