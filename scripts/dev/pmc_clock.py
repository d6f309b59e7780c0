"""Average shader clock during the delay-and-sum kernel (dev tool; run on the GPU box): GRBM_GUI_ACTIVE (GPU clocks while busy) over the
kernel's duration from the same run's kernel trace.  usage: python3 scripts/dev/pmc_clock.py [bench args...]"""
import csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out", "pmc_clock")
cmd = ["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "-d", out, "-o", "r", "--output-format", "csv", "--",
       "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "3", "--warmup", "1"] + sys.argv[1:]
rc = subprocess.call(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=500, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if rc != 0:
    sys.exit("rocprofv3 failed (%d)" % rc)
cc = sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True))[-1]
kt = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
for r in csv.DictReader(open(cc)):
    if "das_" in r["Kernel_Name"] and "digest" not in r["Kernel_Name"]:
        ns, name = dur[r["Dispatch_Id"]]
        print("%-40s %10.3f ms  GRBM_GUI_ACTIVE %12.0f  -> %.3f GHz" % (name[:40], ns / 1e6, float(r["Counter_Value"]), float(r["Counter_Value"]) / ns))
