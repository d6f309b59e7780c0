"""ISA check for the asm-managed LDS reads of das_kernels.hip.

issue_quads / steps4 / pair_steps request LDS reads into registers that later asm statements consume after an s_waitcnt of
their own; the compiler does not know those registers have a read in flight.  That is sound only while NOTHING touches such
a register between the request and the wait that covers it: a v_mov copy would carry the old value (hipcc inserts such
copies at C++-level branches and merges), a write would be overwritten by the late-landing data, a store would store stale
bytes.  This script checks that on the gfx950 assembly of every das_copies_kernel / das_pair_kernel instantiation:

  * the kernel's text is split into its subsections (the out-of-line re-read stubs live in `.subsection 1`) and turned into a
    control-flow graph: fall-through inside a subsection, s_branch / s_cbranch_* edges to their labels;
  * a forward data-flow pass carries the queue of outstanding LGKM operations (LDS reads with their destination registers,
    LDS writes and scalar loads as empty entries; `s_waitcnt lgkmcnt(k)` retires all but the k youngest -- LDS returns in
    order) along EVERY path, including the not-taken stubs and loop back-edges (states are memoised per instruction);
  * any VALU / LDS / vector-memory instruction that reads or writes a VGPR with a read outstanding is reported;
  * from the code-object metadata: no instantiation that keeps reads in flight may spill (`.vgpr_spill_count`,
    `.private_segment_fixed_size` must be 0).

usage: python scripts/dev/check_inflight_copies.py [file.s]     (without a file: compiles das_kernels.hip to assembly first)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL_RE = re.compile(r"^(_ZN2bf\S*(das_copies_kernel|das_pair_kernel|das_pair2_kernel|das_long_kernel|das_hybrid_pair_kernel)\S*):")
MAX_QUEUE = 24          # lgkmcnt is a 4-bit counter; older entries than this cannot be told apart by any wait


def compile_asm(out):
    src = os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd", "csrc", "das_kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-jump-tables",
           "--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def demangle(names):
    out = subprocess.run(["c++filt"] + list(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o.replace("(anonymous namespace)::", "").replace("void ", "")) for o in out[:len(names)]]


def vregs(tok):
    """'v[8:11]' or 'v13' -> set of register numbers"""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(text):
    regs = set()
    for a, b, c in re.findall(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        regs |= set(range(int(a), int(b) + 1)) if a else {int(c)}
    return regs


def parse_kernels(path):
    """-> {mangled name: [instruction dict ...]} with `next` (fall-through index or None) and `targets` (label names)."""
    kernels, name, subs, cur = {}, None, None, 0
    for lineno, line in enumerate(open(path), 1):
        m = KERNEL_RE.match(line)
        if m:
            name, subs, cur = m.group(1), {0: [], 1: []}, 0
            kernels[name] = subs
            continue
        if name is None:
            continue
        l = line.split(";")[0].strip()
        if not l:
            continue
        m = re.match(r"\.subsection\s+(\d+)", l)
        if m:
            cur = int(m.group(1))
            subs.setdefault(cur, [])
            continue
        if l.startswith(".Lfunc_end") or l.startswith(".section") or l.startswith(".rodata"):
            name = None
            continue
        if l.startswith("."):
            m = re.match(r"(\.[\w$.]+):", l)
            if m:
                subs[cur].append({"label": m.group(1), "line": lineno})
            continue
        subs[cur].append({"text": l, "line": lineno})
    out = {}
    for name, subs in kernels.items():
        insts, labels = [], {}
        for sub in sorted(subs):
            pending_labels = []
            for item in subs[sub]:
                if "label" in item:
                    pending_labels.append(item["label"])
                    continue
                for lab in pending_labels:
                    labels[lab] = len(insts)
                pending_labels = []
                insts.append({"text": item["text"], "line": item["line"], "sub": sub})
            for lab in pending_labels:            # a label at the very end of a subsection: falls off -> no instruction
                labels[lab] = None
        for i, ins in enumerate(insts):
            t = ins["text"]
            nxt = i + 1 if i + 1 < len(insts) and insts[i + 1]["sub"] == ins["sub"] else None
            op = t.split()[0]
            targets = []
            if op == "s_branch":
                targets, nxt = [t.split()[1]], None
            elif op.startswith("s_cbranch"):
                targets = [t.split()[1]]
            elif op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                nxt = None
            ins["next"], ins["targets"] = nxt, targets
        out[name] = (insts, labels)
    return out


def transfer(queue, text):
    """-> (new queue, violation or None).  queue: tuple of frozensets (VGPRs each outstanding LGKM operation will write)."""
    op = text.split()[0]
    inflight = set().union(*queue) if queue else set()
    bad = None
    operands = text[len(op):]
    if op.startswith("ds_read") or op.startswith("ds_load"):
        toks = [x.strip() for x in operands.split(",")]
        dest = vregs(toks[0])
        addr = all_vregs(",".join(toks[1:]))
        if addr & inflight:
            bad = "READ  " + text
        queue = queue + (frozenset(dest),)
    elif op.startswith("ds_") or op.startswith("s_load_") or op.startswith("s_buffer_load_") or op in ("s_memtime", "s_memrealtime", "s_dcache_inv"):
        if op.startswith("ds_") and (all_vregs(operands) & inflight):
            bad = "READ  " + text                      # an LDS write (or atomic) whose address / data has a read in flight
        queue = queue + (frozenset(),)
    elif op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", text)
        if m:
            k = int(m.group(1))
            queue = queue[len(queue) - k:] if k else ()
        elif re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", text):
            imm = int(text.split()[1], 0)
            k = (imm >> 8) & 0xF
            queue = queue[len(queue) - k:] if k < len(queue) else queue
            if k == 0:
                queue = ()
    elif op.startswith(("v_", "global_", "scratch_", "buffer_", "flat_")):
        if all_vregs(operands) & inflight:
            kind = "WRITE " if (op.startswith("v_") and not re.match(r"v_(cmp|cmpx|readlane|readfirstlane)", op) and
                                vregs(operands.split(",")[0].strip()) & inflight) else "READ  "
            bad = kind + text
    if len(queue) > MAX_QUEUE:
        queue = queue[len(queue) - MAX_QUEUE:]
    return queue, bad


def scan_kernel(insts, labels):
    """Forward data-flow over every path from the kernel's first instruction.  -> [(line, description)]"""
    if not insts:
        return []
    seen = [set() for _ in insts]
    bad = {}
    work = [(0, ())]
    while work:
        pc, queue = work.pop()
        while pc is not None:
            if queue in seen[pc]:
                break
            seen[pc].add(queue)
            ins = insts[pc]
            queue, v = transfer(queue, ins["text"])
            if v:
                bad[ins["line"]] = v
            for lab in ins["targets"]:
                t = labels.get(lab)
                if t is not None:
                    work.append((t, queue))
            pc = ins["next"]
    return sorted(bad.items())


def hardwired_gaps(path):
    """das_pair2_kernel (and das_long_kernel for lerp: v90..v127) keeps a mic's quads in hard-wired registers (v96..v120, named as clobbers) from its statement S1 (reads, step 0)
    to its statement S2 (steps 1..7); the compiler is free to use those registers in between.  -> [(line, instruction)] for every
    instruction between a ';BF_S1_END' and the next ';BF_S2_BEGIN' that names one of v96..v120 (only table requests belong there)."""
    bad, inside, pairs, regs = [], False, 0, set(range(96, 121))
    for lineno, line in enumerate(open(path), 1):
        if "BF_S1_END" in line:
            inside = True
            m = re.search(r"BF_S1_END\s+(\d+)\s+(\d+)", line)      # das_long_kernel names its own range: ';BF_S1_END 90 127'
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set(range(96, 121))
            continue
        if "BF_S2_BEGIN" in line:
            inside, pairs = False, pairs + 1
            continue
        if not inside:
            continue
        l = line.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        if all_vregs(l) & regs:           # (other vector work, e.g. a compare on loop state, is harmless)
            bad.append((lineno, l))
    return pairs, bad


def metadata(path):
    """{mangled kernel name: dict(spill=, scratch=, vgprs=)} from the .amdgpu_metadata block of the assembly."""
    txt = open(path).read()
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
        out[name.group(1)] = dict(spill=g("vgpr_spill_count"), scratch=g("private_segment_fixed_size"), vgprs=g("vgpr_count"))
    return out


def scan(path):
    """-> (kernels seen, [(kernel, line number, instruction), ...])"""
    bad = []
    ks = parse_kernels(path)
    for name, (insts, labels) in ks.items():
        for line, desc in scan_kernel(insts, labels):
            bad.append((name, line, desc))
    return len(ks), bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "das_kernels.s")
        compile_asm(path)
    kernels, bad = scan(path)
    for b in bad[:20]:
        print("REGISTER WITH A READ IN FLIGHT TOUCHED: %s line %d: %s" % b)
    pairs, gaps = hardwired_gaps(path)
    for g in gaps[:10]:
        print("VECTOR INSTRUCTION BETWEEN S1 AND S2 OF A HARD-WIRED MIC: line %d: %s" % g)
    print("%d S1/S2 gaps checked" % pairs)
    bad = bad + [("gap", g[0], g[1]) for g in gaps]
    md = metadata(path)
    names = [n for n in md if any(k in n for k in ("das_copies_kernel", "das_pair_kernel", "das_pair2_kernel", "das_long_kernel", "das_hybrid_pair_kernel"))]
    for n, d in zip(names, demangle(names)):
        if md[n]["spill"] or md[n]["scratch"]:
            print("SPILLS: %-60s vgprs %3d spill %d scratch %d" % (d, md[n]["vgprs"], md[n]["spill"], md[n]["scratch"]))
    print("%d kernels scanned, %d offending instructions" % (kernels, len(bad)))
    sys.exit(1 if bad or kernels == 0 else 0)
