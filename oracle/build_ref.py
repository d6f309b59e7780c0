#!/usr/bin/env python3
"""Compile the reference's own delay-and-sum C sources, where they lie, into oracle/_ref/libref_<cfg>.so.

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (build container); the GPU box only uses the prebuilt .so
files (oracle/_ref/ is git-ignored but travels with gpurun).

The four translation units PC/src/algorithms/{pad,lerp,convolve,hybrid_convolve}_and_sum.c need nothing but
libc/libm and `config.h`, a list of #defines that the reference generates from src/config.json with its own
`src/build_config.py`.  We run THAT generator on a copy of the reference's config.json whose only edits are
the size keys, in a /tmp scratch directory, and pass the scratch directory with -I.  No reference source is
copied into the repository and nothing is written for the reference by hand.

Flags: the reference's `-O3 -march=native -mavx2 -finline-functions` (PC/setup.py:15, PC/Makefile:6) with
`-march=native` replaced by `-march=x86-64-v3` (AVX2+FMA, what `native` means on the AVX2 desktop the
reference targets) so that the .so built in this container also runs on the GPU box's host CPU.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/PC"
sys.path.insert(0, HERE)
from configs import CONFIGS  # noqa: E402

UNITS = ["pad_and_sum.c", "lerp_and_sum.c", "convolve_and_sum.c", "hybrid_convolve_and_sum.c"]
CFLAGS = ["-O3", "-march=x86-64-v3", "-mavx2", "-finline-functions", "-fPIC", "-shared", "-w"]


def build(name, cfg, outdir):
    scratch = tempfile.mkdtemp(prefix="refcfg_%s_" % name, dir="/tmp")
    try:
        os.makedirs(os.path.join(scratch, "src"))
        os.makedirs(os.path.join(scratch, "interface"))
        data = json.load(open(os.path.join(REF, "src", "config.json")))
        g = data["general"]
        g["N_MICROPHONES"], g["N_SAMPLES"] = cfg["M"], cfg["N"]
        g["MAX_RES_X"], g["MAX_RES_Y"], g["N_TAPS"] = cfg["X"], cfg["Y"], cfg["T"]
        json.dump(data, open(os.path.join(scratch, "src", "config.json"), "w"), indent=4)
        subprocess.check_call([sys.executable, os.path.join(REF, "src", "build_config.py")], cwd=scratch)
        out = os.path.join(outdir, "libref_%s.so" % name)
        srcs = [os.path.join(REF, "src", "algorithms", u) for u in UNITS]
        subprocess.check_call(["gcc"] + CFLAGS + ["-I", os.path.join(scratch, "src")] + srcs + ["-lm", "-o", out])
        # the receiver's datagram -> frame conversion (receiver.c:94-151); needs only libc sockets
        rcv = os.path.join(outdir, "libref_receiver_%s.so" % name)
        subprocess.check_call(["gcc"] + CFLAGS + ["-I", os.path.join(scratch, "src"), os.path.join(REF, "src", "receiver.c"), "-o", rcv])
        return out
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def main():
    if not os.path.isdir(REF):
        print("build_ref: %s absent, keeping prebuilt oracle/_ref" % REF)
        return 0
    outdir = os.path.join(HERE, "_ref")
    os.makedirs(outdir, exist_ok=True)
    for name in (sys.argv[1:] or list(CONFIGS)):
        print("build_ref:", build(name, CONFIGS[name], outdir))
    return 0


if __name__ == "__main__":
    sys.exit(main())
