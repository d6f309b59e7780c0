"""SQ counter passes over one bench launch configuration (dev tool; run on the GPU box).
usage: python3 scripts/dev/pmc_sq.py [bench args...]   -> prints per-launch averages for the delay-and-sum kernel"""
import csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU",
    "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU",
]
env = dict(os.environ, TMPDIR="/tmp")
tot = {}
for i, counters in enumerate(PASSES):
    out = os.path.join(ROOT, "gpurun_out", "pmc_sq_%d" % i)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "r", "--output-format", "csv", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "3", "--warmup", "1"] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env, cwd="/tmp", timeout=500, stdout=subprocess.DEVNULL)
    if rc != 0:
        sys.exit("pass %d failed (%d)" % (i, rc))
    f = sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "das_" in r["Kernel_Name"] and "digest" not in r["Kernel_Name"]:
            k = r["Counter_Name"]
            v, n = tot.get(k, (0.0, 0))
            tot[k] = (v + float(r["Counter_Value"]), n + 1)
for k, (v, n) in tot.items():
    print("%-28s %16.0f per launch (%d launches)" % (k, v / n, n))
