#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference in a /tmp scratch directory.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference, gcc, Cython):

    python oracle/gen_golden.py [cfg1 cfg2 shipped cfg5]

What it does, per size in oracle/configs.py (recipe of SURVEY.md section 8(c)):
  1. copies PC/src and PC/interface of the read-only reference into a fresh directory under /tmp
     (never into this repository),
  2. edits the scratch `src/config.json` size keys (N_MICROPHONES, N_SAMPLES, MAX_RES_X, MAX_RES_Y, N_TAPS)
     and, for the 64-mic sizes, the two hard-coded constants `_N_MICS`/`_ACTIVE_MICS` of the scratch
     `src/directions.pyx:15-16`,
  3. runs the reference's own generator `src/build_config.py` there (writes src/config.h, interface/config.py),
  4. cythonizes ONLY the `directions` and `tests` extensions with the reference's flags (PC/setup.py:15),
  5. imports them in a child process (cwd = scratch), feeds the synthetic blocks S1-S3 and stores
     tables / images as plain data in tests/golden/<cfg>.npz.

Only inputs and expected outputs are stored; no reference source, bytecode or binary enters the repo.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/PC"
sys.path.insert(0, os.path.join(REPO, "oracle"))
from configs import CONFIGS  # noqa: E402

SETUP = textwrap.dedent('''
    from setuptools import Extension, setup
    from Cython.Build import cythonize
    import numpy
    CFLAGS = "-finline-functions -O3 -march=native -mavx2"   # PC/setup.py:15
    ext = [
        Extension("directions", ["src/directions.pyx"], include_dirs=["src/", numpy.get_include()]),
        Extension("tests", ["src/benchmark.pyx"], include_dirs=["src/", "src/algorithms/", numpy.get_include()],
                  extra_compile_args=CFLAGS.split()),
    ]
    setup(ext_modules=cythonize(ext, language_level=3), script_args=["build_ext", "--inplace"])
''')

WORKER = textwrap.dedent('''
    import sys, os, hashlib, time, importlib.util
    import numpy as np
    sys.path.insert(0, "")
    cfg_name, out_path, synth_path = sys.argv[1:4]
    spec = importlib.util.spec_from_file_location("synth", synth_path); synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    from interface import config
    from lib.directions import (calc_r_prime, active_microphones, calculate_delays, calculate_coefficients,
                                compute_convolve_h, get_h, get_h2)
    from lib import tests as T

    M, N, X, Y = config.N_MICROPHONES, config.N_SAMPLES, config.MAX_RES_X, config.MAX_RES_Y
    big = X * Y * M > 200000          # cfg1 is the only "small" size
    huge = X * Y * M > 2000000        # cfg5
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    out = {}
    t0 = time.time()
    log = lambda *a: print("[%6.1fs]" % (time.time() - t0), *a, flush=True)

    active, n_active = active_microphones()
    out["active_mics"] = np.asarray(active, dtype=np.int64)
    d32 = float(np.float32(config.ELEMENT_DISTANCE))           # what calculate_delays passes (C float -> double)
    out["r_prime"] = np.asarray(calc_r_prime(d32), dtype=np.float64)
    delays = calculate_delays()
    assert delays.shape == (X, Y, n_active) and delays.dtype == np.float64
    flat = delays.reshape(X * Y, n_active)
    rows = np.sort(np.random.default_rng(7).choice(X * Y, size=min(256, X * Y), replace=False))
    out["delay_sha256"] = np.array(sha(delays))
    out["delay_rows_idx"] = rows
    out["delay_rows"] = flat[rows]
    out["delay_minmax"] = np.array([delays.min(), delays.max()])
    whole = delays.astype(int).astype(np.int32)                 # directions.pyx:265 + benchmark.pyx:102
    out["whole_sha256"] = np.array(sha(whole))
    out["whole_rows"] = whole.reshape(X * Y, n_active)[rows]
    out["delay_f32_sha256"] = np.array(sha(np.float32(delays)))  # benchmark.pyx:154
    if not big:
        out["delay"] = delays
        w2, h = calculate_coefficients()                        # get_h taps (unused by the pad path)
        assert np.array_equal(w2.astype(np.int32), whole)
        out["taps_get_h"] = h
        out["taps_get_h2"] = compute_convolve_h()
    # a few scalar known answers for the tap generators
    probe = np.array([0.0, 0.25, 0.5, 0.9990234375, 3.75, 17.125, 46.62])
    out["tap_probe_delay"] = probe
    out["tap_probe_get_h"] = np.stack([get_h(float(x)) for x in probe])
    out["tap_probe_get_h2"] = np.stack([get_h2(float(x), N=config.N_TAPS) for x in probe])
    log("tables done", delays.shape)

    assert n_active == M and np.array_equal(active, np.arange(M))
    x0, y0 = X // 3, (2 * Y) // 3
    inputs = {"s1": synth.s1_tone(M, N), "s2": synth.s2_noise(M, N), "s3": synth.s3_plane_wave(delays[x0, y0], N)}
    out["s3_dir"] = np.array([x0, y0])
    out["s1_row"] = inputs["s1"][0]
    for k, v in inputs.items():
        out["in_sha256_" + k] = np.array(sha(v))
    if not huge:
        out["in_s2"] = inputs["s2"]; out["in_s3"] = inputs["s3"]

    plan = {"lerp": ["s1", "s2", "s3"], "hybrid": ["s1", "s2", "s3"], "pad": ["s1", "s2", "s3"], "convolve": ["s1", "s2"]}
    if huge:
        plan = {"lerp": ["s1", "s2"], "hybrid": ["s2"]}          # pad/convolve wrappers need hours of Python tap loops
    fn = {"pad": T.mimo_pad_wrapper, "lerp": T.mimo_lerp_wrapper, "hybrid": T.mimo_hybrid_convolve_wrapper,
          "convolve": T.mimo_convolve_wrapper}
    for algo, names in plan.items():
        for name in names:
            img = np.array(fn[algo](inputs[name]), dtype=np.float32, copy=True)
            assert img.shape == (X, Y)
            out["img_%s_%s" % (algo, name)] = img
            log(algo, name, "argmax", np.unravel_index(np.argmax(img), img.shape), "max", float(img.max()))
    np.savez_compressed(out_path, **out)
    log("wrote", out_path)
''')


def build_scratch(name, cfg):
    scratch = tempfile.mkdtemp(prefix="refgold_%s_" % name, dir="/tmp")
    for sub in ("src", "interface"):
        shutil.copytree(os.path.join(REF, sub), os.path.join(scratch, sub))
    subprocess.check_call(["chmod", "-R", "u+w", scratch])
    cj = os.path.join(scratch, "src", "config.json")
    data = json.load(open(cj))
    g = data["general"]
    g["N_MICROPHONES"], g["N_SAMPLES"] = cfg["M"], cfg["N"]
    g["MAX_RES_X"], g["MAX_RES_Y"], g["N_TAPS"] = cfg["X"], cfg["Y"], cfg["T"]
    json.dump(data, open(cj, "w"), indent=4)
    pyx = os.path.join(scratch, "src", "directions.pyx")
    src = open(pyx).read()
    src, n1 = re.subn(r"^_N_MICS = 256$", "_N_MICS = %d" % cfg["M"], src, flags=re.M)
    src, n2 = re.subn(r"^_ACTIVE_MICS = 4$", "_ACTIVE_MICS = %d" % cfg["arrays"], src, flags=re.M)
    assert n1 == 1 and n2 == 1
    open(pyx, "w").write(src)
    subprocess.check_call([sys.executable, "src/build_config.py"], cwd=scratch)
    shutil.copy(os.path.join(scratch, "src", "config.h"), os.path.join(scratch, "src", "algorithms", "config.h"))
    open(os.path.join(scratch, "setup_scratch.py"), "w").write(SETUP)
    with open(os.path.join(scratch, "build.log"), "w") as lf:
        subprocess.check_call([sys.executable, "setup_scratch.py"], cwd=scratch, stdout=lf, stderr=subprocess.STDOUT)
    os.makedirs(os.path.join(scratch, "lib"), exist_ok=True)
    for f in os.listdir(scratch):
        if f.endswith(".so"):
            shutil.move(os.path.join(scratch, f), os.path.join(scratch, "lib", f))
    open(os.path.join(scratch, "golden_worker.py"), "w").write(WORKER)
    return scratch


def main():
    names = sys.argv[1:] or list(CONFIGS)
    os.makedirs(os.path.join(REPO, "tests", "golden"), exist_ok=True)
    synth = os.path.join(REPO, "zybo-rt-sampler-image-detection_amd", "synth.py")
    for name in names:
        cfg = CONFIGS[name]
        print("== %s %s" % (name, cfg), flush=True)
        scratch = build_scratch(name, cfg)
        out = os.path.join(REPO, "tests", "golden", name + ".npz")
        subprocess.check_call([sys.executable, "golden_worker.py", name, out, synth], cwd=scratch)
        shutil.rmtree(scratch)


if __name__ == "__main__":
    main()
