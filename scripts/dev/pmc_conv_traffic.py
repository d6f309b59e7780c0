#!/usr/bin/env python3
"""HBM traffic of the detector's 60 convolutions against their algorithmic bytes (dev tool; GPU box).

  python3 scripts/dev/pmc_conv_traffic.py [--half] [--f32-mode native]   -> gpurun_out/r03_conv_traffic_<kind>.txt

Two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md asks) around scripts/conv_layer_table.py --reps 2; the per-layer
section of that script issues 6 launches per layer (2 warm-up, then two replays of a graph of 2), the last 360 convolution dispatches.  Traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB
(gfx950: FETCH_SIZE counts half of a 16 B/lane read), per launch; algorithmic bytes = inputs + weights + output of the layer (the table's MB column)."""
import argparse, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--half", action="store_true")
ap.add_argument("--f32-mode", choices=["split", "native"], default="split")
a = ap.parse_args()
kind = "f16" if a.half else ("f32" if a.f32_mode == "split" else "f32_native")
vals = {}
table = os.path.join(ROOT, "gpurun_out", "traffic_layers_%s.csv" % kind)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    out = os.path.join(ROOT, "gpurun_out", "conv_traffic_%s_%s" % (kind, ctr))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "r", "--output-format", "csv", "--", "python3", os.path.join(ROOT, "scripts", "conv_layer_table.py"),
           "--no-miopen", "--reps", "2", "--f32-mode", a.f32_mode, "--csv", table] + (["--half"] if a.half else [])
    if subprocess.call(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=500, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
        sys.exit("rocprofv3 failed")
    rows = [r for r in csv.DictReader(open(sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True))[-1])) if "conv_" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals[ctr] = [float(r["Counter_Value"]) for r in rows][-360:]
layers = [r for r in csv.DictReader(open(table)) if not r["layer"].startswith("TOTAL")]
lines = ["%-18s %9s %9s %7s" % ("layer", "algo MB", "HBM MB", "ratio")]
tot_a = tot_t = 0.0
for i, l in enumerate(layers):
    f = sum(vals["FETCH_SIZE"][6 * i + 2:6 * i + 6]) / 4.0
    w = sum(vals["WRITE_SIZE"][6 * i + 2:6 * i + 6]) / 4.0
    t = (2.0 * f + w) * 1024 / 1e6
    al = float(l["mbytes"])
    tot_a += al; tot_t += t
    lines.append("%-18s %9.1f %9.1f %7.2f" % (l["layer"], al, t, t / al))
lines.append("%-18s %9.1f %9.1f %7.2f" % ("all 60 layers", tot_a, tot_t, tot_t / tot_a))
txt = "\n".join(lines)
open(os.path.join(ROOT, "gpurun_out", "r03_conv_traffic_%s.txt" % kind), "w").write(txt + "\n")
print(txt)
