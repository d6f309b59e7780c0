#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's roofline block cites, for one (workload, algo).

On the GPU box (three separate profiler passes, as MI355X_MICROARCH.md prescribes for HBM counters):

    python3 scripts/profile_round.py run  --tag r01 --workload cfg2 --algo lerp      # writes gpurun_out/<tag>_*/

Back in the repo (parses what gpurun merged into gpurun_out/ and writes the tracked summaries):

    python3 scripts/profile_round.py fold --tag r01 --workload cfg2 --algo lerp      # -> profiles/<tag>_*.csv, profiles/traffic.json

`fold` keeps only the rows of this repository's kernels plus the header (the raw traces stay in gpurun_out/).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE counts
half of a 16 B/lane coalesced read (MI355X_MICROARCH.md, HBM section)."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_cmd(a, steps):
    return ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", str(steps), "--warmup", "2",
            "--workload", a.workload, "--algo", a.algo, "--frames", str(a.frames)]


def run(a):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    passes = [("stats", ["--kernel-trace", "--stats"], 20), ("pmc_fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"], 3),
              ("pmc_write", ["--kernel-trace", "--pmc", "WRITE_SIZE"], 3)]
    for name, flags, steps in passes:
        out = os.path.join(ROOT, "gpurun_out", "%s_%s_%s_%s" % (a.tag, name, a.workload, a.algo))
        cmd = ["rocprofv3"] + flags + ["-d", out, "-o", "r", "--output-format", "csv", "--"] + bench_cmd(a, steps)
        print("[profile]", " ".join(cmd), flush=True)
        rc = subprocess.call(cmd, env=env, cwd="/tmp", timeout=600)
        if rc != 0:
            sys.exit("profiler pass %s failed with %d" % (name, rc))


def _one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        sys.exit("nothing matches " + pattern)
    return hits[-1]


def fold(a):
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    key = "%s_%s" % (a.workload, a.algo)
    base = os.path.join(ROOT, "gpurun_out", "%s_%%s_%s" % (a.tag, key))
    ours = lambda name: "bf::" in name
    # kernel statistics
    src = _one(base % "stats" + "/**/*kernel_stats.csv")
    rows = list(csv.reader(open(src)))
    dst = os.path.join(prof, "%s_kernel_stats_%s.csv" % (a.tag, key))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0])
        for r in rows[1:]:
            if ours(r[0]):
                w.writerow(r)
    avg_ns = [float(r[3]) for r in rows[1:] if "das_" in r[0] and "digest" not in r[0]]
    # counters
    sums = {}
    for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        src = _one(base % which + "/**/*counter_collection.csv")
        rd = csv.DictReader(open(src))
        keep = [r for r in rd if "das_" in r["Kernel_Name"] and "digest" not in r["Kernel_Name"] and r["Counter_Name"] == counter]
        dst = os.path.join(prof, "%s_%s_%s.csv" % (a.tag, which, key))
        with open(dst, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=rd.fieldnames)
            w.writeheader()
            w.writerows(keep)
        vals = [float(r["Counter_Value"]) for r in keep]
        sums[counter] = sum(vals) / len(vals)
    tpath = os.path.join(prof, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic["%s_f%d_n1" % (key, a.frames)] = {
        "hbm_bytes_per_launch": (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0,
        "fetch_size_kib": sums["FETCH_SIZE"], "write_size_kib": sums["WRITE_SIZE"],
        "kernel_avg_ms_rocprof": (avg_ns[0] / 1e6) if avg_ns else None,
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts half of a 16 B/lane coalesced read; MI355X_MICROARCH.md, HBM)",
        "source": "profiles/%s_pmc_fetch_%s.csv, profiles/%s_pmc_write_%s.csv (separate --pmc passes)" % (a.tag, key, a.tag, key),
    }
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(traffic["%s_f%d_n1" % (key, a.frames)], indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "fold"])
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--algo", default="lerp")
    ap.add_argument("--frames", type=int, default=190)
    a = ap.parse_args()
    run(a) if a.mode == "run" else fold(a)
