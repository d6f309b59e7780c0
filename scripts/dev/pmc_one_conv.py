#!/usr/bin/env python3
"""SQ counters of ONE convolution layer shape (dev tool; GPU box).
usage: python3 scripts/dev/pmc_one_conv.py cin cout k stride hw [--half] [--batch 64]   (under rocprofv3 it is its own driver: run without arguments to profile)"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--child" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
    import torch
    from image_detection.model import yolov5s
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    cin, cout, k, s, hw = (int(v) for v in a[:5])
    half = "--half" in sys.argv
    dt = torch.float16 if half else torch.float32
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2).cuda().to(dt)
    hc = yolov5s.HipConv(conv, True)
    x = torch.randn((64, cin, hw, hw), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    for _ in range(6):
        hc(x)
    torch.cuda.synchronize()
    sys.exit(0)
PASSES = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
          "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
          "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE"]
tot = {}
for i, counters in enumerate(PASSES):
    out = os.path.join(ROOT, "gpurun_out", "pmc_one_%d" % i)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "r", "--output-format", "csv", "--", "python3", os.path.abspath(__file__), "--child"] + sys.argv[1:]
    rc = subprocess.call(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if rc != 0:
        sys.exit("pass %d failed" % i)
    f = sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True))[-1]
    kt = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Kernel_Name"]:
            v, n = tot.get(r["Counter_Name"], (0.0, 0))
            tot[r["Counter_Name"]] = (v + float(r["Counter_Value"]), n + 1)
            tot["ns"] = (tot.get("ns", (0.0, 0))[0] + dur[r["Dispatch_Id"]], tot.get("ns", (0.0, 0))[1] + 1)
print(" ".join(sys.argv[1:]))
for k, (v, n) in tot.items():
    print("  %-26s %16.0f per launch" % (k, v / n))
