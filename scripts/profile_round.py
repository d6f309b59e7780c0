#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's roofline block cites, for one (workload, algo).

On the GPU box, in ONE lease so that the numbers nest (a plain bench line first, then three separate profiler passes, as
MI355X_MICROARCH.md prescribes for HBM counters):

    python3 scripts/profile_round.py run  --tag r01 --workload cfg2 --algo lerp      # writes gpurun_out/<tag>_*/
    python3 scripts/profile_round.py extras --tag r03                                 # the default bench with its side measurements: kernel
                                                                                      # statistics + MFMA-busy counters of the config-3 / config-4 kernels

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
    # the un-profiled line of the same box, same lease: the profile's average kernel time has to sit at or under its step time
    line = subprocess.run(bench_cmd(a, 20), env=env, cwd=ROOT, timeout=600, capture_output=True, text=True)
    js = [l for l in line.stdout.splitlines() if l.startswith("{")]
    if line.returncode != 0 or not js:
        sys.exit("plain bench line failed: " + line.stderr[-2000:])
    open(os.path.join(ROOT, "gpurun_out", "%s_bench_%s_%s.json" % (a.tag, a.workload, a.algo)), "w").write(js[-1] + "\n")
    passes = [("stats", ["--kernel-trace", "--stats"], 20), ("pmc_fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"], 3),
              ("pmc_write", ["--kernel-trace", "--pmc", "WRITE_SIZE"], 3)]
    for name, flags, steps in passes:
        out = os.path.join(ROOT, "gpurun_out", "%s_%s_%s_%s" % (a.tag, name, a.workload, a.algo))
        cmd = ["rocprofv3"] + flags + ["-d", out, "-o", "r", "--output-format", "csv", "--"] + bench_cmd(a, steps)
        print("[profile]", " ".join(cmd), flush=True)
        rc = subprocess.call(cmd, env=env, cwd="/tmp", timeout=600)
        if rc != 0:
            sys.exit("profiler pass %s failed with %d" % (name, rc))


def extras(a):
    """Kernel statistics and MFMA-pipe counters of everything the default bench line runs (config 2 headline + config 3 MVDR /
    frequency-domain DAS + config 4 fused step and detector), plus the plain line itself, same lease."""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "2"]
    line = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    js = [l for l in line.stdout.splitlines() if l.startswith("{")]
    if line.returncode != 0 or not js:
        sys.exit("plain bench line failed: " + line.stderr[-2000:])
    open(os.path.join(ROOT, "gpurun_out", "%s_bench_default_with_extras.json" % a.tag), "w").write(js[-1] + "\n")
    for name, flags in (("stats", ["--kernel-trace", "--stats"]),
                        ("pmc_mfma", ["--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"])):
        out = os.path.join(ROOT, "gpurun_out", "%s_%s_default_with_extras" % (a.tag, name))
        full = ["rocprofv3"] + flags + ["-d", out, "-o", "r", "--output-format", "csv", "--"] + cmd
        print("[profile]", " ".join(full), flush=True)
        rc = subprocess.call(full, env=env, cwd="/tmp", timeout=900, stdout=subprocess.DEVNULL)
        if rc != 0:
            sys.exit("profiler pass %s failed with %d" % (name, rc))


def fold_extras(a):
    prof = os.path.join(ROOT, "profiles")
    base = os.path.join(ROOT, "gpurun_out", "%s_%%s_default_with_extras" % a.tag)
    rows = list(csv.reader(open(_one(base % "stats" + "/**/*kernel_stats.csv"))))
    with open(os.path.join(prof, "%s_kernel_stats_cfg2_default_with_extras.csv" % a.tag), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow(r)          # every kernel of the process: what is left of torch's own shows here too
    kt = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(_one(base % "pmc_mfma" + "/**/*kernel_trace.csv")))}
    agg = {}
    for r in csv.DictReader(open(_one(base % "pmc_mfma" + "/**/*counter_collection.csv"))):
        name = r["Kernel_Name"]
        if "bf::" not in name and "_ZN2bf" not in name:
            continue
        short = short_name(name)
        d = agg.setdefault(short, {"ids": set(), "ns": 0})
        if r["Dispatch_Id"] not in d["ids"]:
            d["ids"].add(r["Dispatch_Id"]); d["ns"] += kt[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    with open(os.path.join(prof, "%s_pmc_mfma_default_with_extras.csv" % a.tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "mfma_busy = MFMA_BUSY / (4 * BUSY_CU)",
                    "clock_ghz = GUI_ACTIVE / 8 / ns"])
        for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
            mb, bc, ga = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("SQ_BUSY_CU_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
            w.writerow([k, len(d["ids"]), "%.4f" % (d["ns"] / 1e6), "%.2f" % (d["ns"] / 1e3 / len(d["ids"])), "%.0f" % mb, "%.0f" % bc, "%.0f" % ga,
                        "%.4f" % (mb / (4.0 * bc) if bc else 0.0), "%.3f" % (ga / 8.0 / d["ns"] if d["ns"] else 0.0)])
    src = os.path.join(ROOT, "gpurun_out", "%s_bench_default_with_extras.json" % a.tag)
    if os.path.exists(src):
        open(os.path.join(prof, "%s_bench_default_with_extras.json" % a.tag), "w").write(open(src).read())
    print("folded extras of", a.tag)


def short_name(name):
    """A readable kernel name: the template instance without namespaces and argument lists (rocprofv3 leaves _Float16 instantiations mangled)."""
    import re
    m = re.search(r"conv_(dma|igemm)_kernelI(DF16_|f)((?:Li\d+E)+)Lb(\d)", name)
    if m:
        return "conv_%s_kernel<%s, %s, %s>" % (m.group(1), "f16" if m.group(2) != "f" else "f32", ", ".join(re.findall(r"Li(\d+)E", m.group(3))), "cat" if m.group(4) == "1" else "plain")
    m = re.search(r"_ZN2bf(?:12_GLOBAL__N_1)?(\d+)", name)
    if name.startswith("_Z") and m:
        i = name.index(m.group(1), m.start(1)) + len(m.group(1))
        return name[i:i + int(m.group(1))] + "<f16 instantiation>"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[name.index("bf::") + 4:].split("(")[0] if "bf::" in name else name.split("(")[0]


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
    # the un-profiled line of the same lease: keep what the fractions are recomputed from
    src = os.path.join(ROOT, "gpurun_out", "%s_bench_%s.json" % (a.tag, key))
    if os.path.exists(src):
        d = json.loads(open(src).read())
        r = d["roofline"]
        keep = {"value_frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "kernel": r["kernel"],
                "kernel_ms_mean": r["kernel_ms"], "kernel_ms_min": r.get("kernel_ms_min"), "kernel_ms_median": r.get("kernel_ms_median"),
                "kernel_ms_max": r.get("kernel_ms_max"), "frac": r["frac"], "rocprof_kernel_avg_ms_same_lease": (avg_ns[0] / 1e6) if avg_ns else None,
                "note": "plain bench.py line and the rocprofv3 --stats pass of ONE gpurun lease (same box, back to back); the profiled pass runs under the profiler's "
                        "overhead and clock (MI355X_MICROARCH.md, DVFS item 2), so its average may exceed the plain step by a few percent"}
        json.dump(keep, open(os.path.join(prof, "%s_bench_%s.json" % (a.tag, key)), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "fold", "extras", "fold_extras"])
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--algo", default="lerp")
    ap.add_argument("--frames", type=int, default=190)
    a = ap.parse_args()
    {"run": run, "fold": fold, "extras": extras, "fold_extras": fold_extras}[a.mode](a)
