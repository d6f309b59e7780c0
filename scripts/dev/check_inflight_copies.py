"""ISA check for the asm-managed LDS reads of das_kernels.hip.

issue_quads / steps4 / pair_steps request LDS reads into registers that later asm statements consume after an s_waitcnt of
their own; the compiler does not know those registers have a read in flight.  That is sound only while it never COPIES such
a register between the request and the wait (a v_mov would carry the old value, and the late-landing data would hit a
register the compiler already considers free).  It does insert such copies at C++-level branches and merges, so no read may
be in flight across one.  This script scans the gfx950 assembly of every das_copies_kernel / das_pair_kernel instantiation
for a vector instruction that reads or writes a register while a ds_read into it is outstanding (lgkmcnt accounting:
LDS reads and writes and scalar loads in issue order, `s_waitcnt lgkmcnt(k)` retires all but the k youngest).

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
    name, queue, bad, kernels = None, [], [], 0     # queue: registers of the outstanding LGKM operations, oldest first

    def pending():
        regs = set()
        for q in queue:
            regs |= q
        return regs

    for i, line in enumerate(open(path)):
        m = re.match(r"^(_ZN2bf\S*(das_copies_kernel|das_pair_kernel)\S*):", line)
        if m:
            name, queue, kernels = m.group(1), [], kernels + 1
            continue
        if name is None:
            continue
        l = line.strip()
        if "s_endpgm" in l:
            name = None
            continue
        m = re.match(r"ds_read\w* (v\[(\d+):(\d+)\]|v(\d+))", l)
        if m:
            lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
            queue.append(set(range(lo, hi + 1)))
            continue
        if re.match(r"(ds_\w+|s_load_\w+|s_buffer_load_\w+) ", l):
            queue.append(set())                      # LDS writes and scalar loads count in lgkmcnt too
            continue
        if l.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:                                    # (LDS returns in order: all but the k youngest operations are done)
                k = int(m.group(1))
                queue = queue[len(queue) - k:] if k else []
            continue
        inflight = pending()
        if not inflight:
            continue
        if re.match(r"(v_|ds_write|global_store|scratch_store|buffer_store)", l):
            # any vector instruction READING such a register (a v_mov copy at a branch or merge is the case that was found)
            ops = l.split(None, 1)[1] if " " in l else ""
            first_is_dest = l.startswith("v_") and not re.match(r"v_(cmp|cmpx)", l)
            toks = re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", ops)
            for n, t in enumerate(toks):
                if n == 0 and first_is_dest:
                    continue
                lo, hi = (int(t[0]), int(t[1])) if t[0] else (int(t[2]), int(t[2]))
                if any(r in inflight for r in range(lo, hi + 1)):
                    bad.append((name, i + 1, "READ  " + l))
                    break
        # ... and nothing else may WRITE such a register before the wait (the compiler reusing it for another value)
        m = re.match(r"(v_(?!cmp|readlane|readfirstlane)\w+|global_load_\w+|scratch_load_\w+|buffer_load_\w+) (v\[(\d+):(\d+)\]|v(\d+))[, ]", l)
        if m:
            lo, hi = (int(m.group(3)), int(m.group(4))) if m.group(3) else (int(m.group(5)), int(m.group(5)))
            if any(r in inflight for r in range(lo, hi + 1)):
                bad.append((name, i + 1, "WRITE " + l))
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
