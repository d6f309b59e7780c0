"""ISA check for the asm-managed LDS reads of das_kernels.hip.

issue_quads / steps4 / pair_steps request LDS reads into registers that later asm statements consume after an s_waitcnt of
their own; the compiler does not know those registers have a read in flight.  That is sound only while it never COPIES such
a register between the request and the wait (a v_mov would carry the old value, and the late-landing data would hit a
register the compiler already considers free).  It does insert such copies at C++-level branches and merges, so no read may
be in flight across one.  This script scans the gfx950 assembly of every das_copies_kernel / das_pair_kernel instantiation
for a v_mov that reads a register between a ds_read into it and the next `s_waitcnt lgkmcnt(0)`.

usage: python scripts/dev/check_inflight_copies.py [file.s]     (without a file: compiles das_kernels.hip to assembly first)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compile_asm(out):
    src = os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd", "csrc", "das_kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-jump-tables",
           "--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def scan(path):
    """-> (kernels seen, [(kernel, line number, instruction), ...])"""
    name, pending, bad, kernels = None, {}, [], 0
    for i, line in enumerate(open(path)):
        m = re.match(r"^(_ZN2bf\S*(das_copies_kernel|das_pair_kernel)\S*):", line)
        if m:
            name, pending, kernels = m.group(1), {}, kernels + 1
            continue
        if name is None:
            continue
        l = line.strip()
        if "s_endpgm" in l:
            name = None
            continue
        m = re.match(r"ds_read_b(64|128) v\[(\d+):(\d+)\]", l)
        if m:
            for r in range(int(m.group(2)), int(m.group(3)) + 1):
                pending[r] = i
            continue
        if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
            pending = {}
            continue
        m = re.match(r"v_mov_b(32|64)_e32 (v\[?\d+(?::\d+)?\]?), (v\[?\d+(?::\d+)?\]?)", l)
        if m and pending:
            nums = [int(x) for x in re.findall(r"\d+", m.group(3))]
            if any(r in pending for r in range(nums[0], nums[-1] + 1)):
                bad.append((name, i + 1, l))
    return kernels, bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "das_kernels.s")
        compile_asm(path)
    kernels, bad = scan(path)
    for b in bad[:20]:
        print("COPY OF A REGISTER WITH A READ IN FLIGHT: %s line %d: %s" % b)
    print("%d kernels scanned, %d offending copies" % (kernels, len(bad)))
    sys.exit(1 if bad or kernels == 0 else 0)
