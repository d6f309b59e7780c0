#!/usr/bin/env python3
"""Build a variant of the library with extra -D flags into scripts/dev/bin/ (dev tool; for same-box A/B runs through BF_NATIVE_LIB).
usage: python scripts/dev/build_variant.py <name> -DFOO=1 [-DBAR ...]   -> scripts/dev/bin/libbeamformer_hip_<name>.so"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
name, defs = sys.argv[1], sys.argv[2:]
srcs = sorted(glob.glob(os.path.join(ge.CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(ge.CSRC, "*.cpp")))
out = os.path.join(ROOT, "scripts", "dev", "bin", "libbeamformer_hip_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
objs, jobs = [], []
tmp = os.path.join(ROOT, "build", "variant_" + name)
os.makedirs(tmp, exist_ok=True)
for s in srcs:
    o = os.path.join(tmp, os.path.basename(s) + ".o")
    objs.append(o)
    jobs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + ge.HIPCC_FLAGS + defs + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]))
if any(j.wait() for j in jobs):
    sys.exit("compile failed")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
