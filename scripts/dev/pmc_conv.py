#!/usr/bin/env python3
"""MFMA-pipe utilisation and shader clock of the detector's convolution kernels (dev tool; GPU box).

  python3 scripts/dev/pmc_conv.py [--half] [--tag r03]   -> gpurun_out/<tag>_pmc_conv_{f32,f16}.txt  (+ the raw CSVs next to it)

One rocprofv3 pass (kernel trace + SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE) around scripts/conv_layer_table.py;
per kernel template:  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)  (four matrix pipes per CU),
                      clock     = GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import argparse, csv, glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--half", action="store_true")
ap.add_argument("--tag", default="r03")
ap.add_argument("--f32-mode", choices=["split", "native"], default="split")
a = ap.parse_args()
kind = "f16" if a.half else ("f32" if a.f32_mode == "split" else "f32_native")
out = os.path.join(ROOT, "gpurun_out", "%s_pmc_conv_%s" % (a.tag, kind))
cmd = ["rocprofv3", "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "-d", out, "-o", "r", "--output-format", "csv", "--",
       "python3", os.path.join(ROOT, "scripts", "conv_layer_table.py"), "--no-miopen", "--reps", "2", "--f32-mode", a.f32_mode] + (["--half"] if a.half else [])
rc = subprocess.call(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=500, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if rc != 0:
    sys.exit("rocprofv3 failed (%d)" % rc)
cc = sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True))[-1]
kt = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = {}
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"]
    m = re.search(r"conv_(dma|igemm|patch)_kernelI(DF16_|f)((?:Li\d+E)*)(?:Lb(\d)E)?(?:Lb(\d)E)?", name)          # (float16 instantiations come out mangled)
    if m:
        short = "conv_%s_kernel<%s>" % (m.group(1), ", ".join(["f16" if m.group(2) != "f" else "f32"] + re.findall(r"Li(\d+)E", m.group(3)) +
                                                              (["cat" if m.group(4) == "1" else "plain"] if m.group(4) else []) + (["split"] if m.group(5) == "1" else [])))
    elif "bf::" in name and "conv_" in name:
        short = name[name.index("conv_"):].split("(")[0].replace("float", "f32")
        t = re.match(r"(conv_\w+)<(.*)>$", short)
        if t:
            args = [x.strip() for x in t.group(2).split(",")]
            bools = [x for x in args if x in ("true", "false")]
            args = [x for x in args if x not in ("true", "false")] + (["cat" if bools[0] == "true" else "plain"] if bools else []) + (["split"] if len(bools) > 1 and bools[1] == "true" else [])
            short = "%s<%s>" % (t.group(1), ", ".join(args))
    else:
        continue
    d = agg.setdefault(short, {"n": set(), "ns": 0})
    if r["Dispatch_Id"] not in d["n"]:
        d["n"].add(r["Dispatch_Id"]); d["ns"] += dur[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
lines = ["%-46s %6s %10s %9s %9s" % ("kernel", "calls", "total ms", "MFMA busy", "clock GHz")]
tot = {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "SQ_BUSY_CU_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "ns": 0}
for k, d in sorted(agg.items()):
    lines.append("%-46s %6d %10.3f %9.3f %9.3f" % (k, len(d["n"]), d["ns"] / 1e6, d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"]),
                                                  d["GRBM_GUI_ACTIVE"] / 8.0 / d["ns"]))
    for c in tot:
        tot[c] += d[c]
lines.append("%-46s %6s %10.3f %9.3f %9.3f" % ("all", "", tot["ns"] / 1e6, tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * tot["SQ_BUSY_CU_CYCLES"]), tot["GRBM_GUI_ACTIVE"] / 8.0 / tot["ns"]))
txt = "\n".join(lines)
print(txt)
open(out + ".txt", "w").write(txt + "\n")
