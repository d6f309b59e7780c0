"""The hand-scheduled LDS reads of das_kernels.hip stay sound only while nothing touches a register that has a read in
flight (scripts/dev/check_inflight_copies.py explains).  CPU-only: compiles the kernels to gfx950 assembly, builds the control-
flow graph of every das_copies_kernel / das_pair_kernel instantiation (out-of-line `.subsection 1` stubs and loop back-edges
included), runs the in-flight data-flow over every path, and reads spill / scratch sizes from the code-object metadata."""
import importlib.util
import os

import pytest

import util


@pytest.fixture(scope="module")
def chk():
    spec = importlib.util.spec_from_file_location("check_inflight_copies", os.path.join(util.ROOT, "scripts", "dev", "check_inflight_copies.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def asm(chk, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("isa") / "das_kernels.s")
    chk.compile_asm(path)
    return path


FAKE = """
_ZN2bf12_GLOBAL__N_16copies15das_pair_kernelILi9EEEvPKfPfPKiS7_S4_S4_NS0_5KArgsE:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	s_waitcnt lgkmcnt(0)
	ds_read_b64 v[10:11], v1 offset:0
%s
	s_endpgm
.Lfunc_end0:
"""


@pytest.mark.parametrize("body,n_bad", [
    ("\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 v[2:3], v[2:3], v[10:11]", 0),                              # consumed after the wait
    ("\tv_mov_b32_e32 v20, v11\n\ts_waitcnt lgkmcnt(0)", 1),                                              # copied while in flight
    ("\tv_mov_b32_e32 v10, 0\n\ts_waitcnt lgkmcnt(0)", 1),                                                # overwritten while in flight
    ("\tds_write_b64 v1, v[10:11]\n\ts_waitcnt lgkmcnt(0)", 1),                                           # stored while in flight
    ("\tds_read_b64 v[12:13], v1 offset:8\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v20, v10\n\tv_mov_b32_e32 v21, v12\n\ts_waitcnt lgkmcnt(0)", 1),   # counted wait: the older read has landed, the younger not
    # the hazard sits behind a branch, in an out-of-line stub that the straight-line text never reaches
    ("\ts_cmp_lg_u32 s2, s3\n\ts_cbranch_scc1 .Lstub\n.Lback:\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 v[2:3], v[2:3], v[10:11]\n"
     "\t.subsection 1\n.Lstub:\n\tv_add_u32_e32 v20, v10, v1\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .Lback\n\t.subsection 0", 1),
    # a wait inside the not-taken stub must not hide a read that is still in flight on the fall-through path
    ("\ts_cbranch_scc1 .Lstub\n.Lback:\n\tv_mov_b32_e32 v20, v10\n\ts_waitcnt lgkmcnt(0)\n"
     "\t.subsection 1\n.Lstub:\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .Lback\n\t.subsection 0", 1),
    # carried around a loop: in flight at the back-edge, touched at the loop head on the second trip
    (".Lloop:\n\tv_mov_b32_e32 v20, v30\n\tds_read_b64 v[30:31], v1 offset:16\n\ts_add_u32 s2, s2, 1\n\ts_cmp_lt_u32 s2, 4\n\ts_cbranch_scc1 .Lloop\n\ts_waitcnt lgkmcnt(0)", 1),
])
def test_scanner_finds_what_it_should(chk, tmp_path, body, n_bad):
    p = tmp_path / "fake.s"
    p.write_text(FAKE % body)
    kernels, bad = chk.scan(str(p))
    assert kernels == 1 and len(bad) == n_bad, bad


def test_no_register_with_a_read_in_flight_is_touched(chk, asm):
    kernels, bad = chk.scan(asm)
    assert kernels >= 43
    assert not bad, bad[:5]


def test_every_launchable_instantiation_is_scanned_and_none_that_pipelines_reads_spills(chk, asm):
    """launch_nc (das_kernels.hip) can pick: the pair kernel (pad, lerp); the one-frame sweep for 1 / 2 / 4 segments with the fixed or
    the run-time row stride, 16 waves -- one segment also with 8 waves -- and its direction-outer (DIRECT) twin; the three 8-tap FIR
    flavours.  The instantiations that keep LDS reads in flight across asm statements (pair kernel; one-segment sweep; the long-row
    kernel, whose lerp sweep issues a mic's reads behind its predecessor's last step) must not use scratch: a spilled register with a read in flight is reloaded before the data lands."""
    md = chk.metadata(asm)
    names = [n for n in md if any(k in n for k in ("das_copies_kernel", "das_pair_kernel", "das_pair2_kernel", "das_long_kernel", "das_hybrid_pair_kernel"))]
    short = dict(zip(chk.demangle(names), names))
    want = ["bf::copies::das_pair_kernel<%d>" % a for a in (0, 1)]
    for a in (0, 1):
        want += ["bf::copies::das_copies_kernel<%d, 1, %d, %d, %s>" % (a, rs, w, d) for rs in (312, 0) for w, d in ((16, "false"), (8, "false"), (16, "true"))]
        want += ["bf::copies::das_copies_kernel<%d, %d, %d, 16, %s>" % (a, seg, rs, d) for seg, fixed in ((2, 576), (4, 1088)) for rs in (fixed, 0) for d in ("false", "true")]
    want += ["bf::copies::das_copies_kernel<%d, 1, %d, 16, false>" % (a, rs) for a in (2, 3, 4) for rs in (320, 0)]
    want += ["bf::copies::das_long_kernel<%d, %d, %d>" % (a, seg, rs) for a in (0, 1) for seg, fixed in ((2, 576), (4, 1088)) for rs in (fixed, 0)]
    want += ["bf::copies::das_hybrid_pair_kernel<%d>" % a for a in (2, 3, 4)] + ["bf::copies::das_pair2_kernel<0>", "bf::copies::das_pair2_kernel<1>"]
    missing = [w for w in want if w not in short]
    assert not missing, missing
    kernels, _ = chk.scan(asm)
    assert kernels == len(names)                       # the scan covered every one of them
    pipelined = [w for w in want if "bf::copies::das_pair_kernel" in w or "bf::copies::das_long_kernel" in w or "bf::copies::das_pair2_kernel" in w or
                 "bf::copies::das_hybrid_pair_kernel" in w or
                 (w.startswith("bf::copies::das_copies_kernel<0, 1,") or w.startswith("bf::copies::das_copies_kernel<1, 1,")) and w.endswith("false>")]
    assert len(pipelined) == 2 + 8 + 2 * 4 + 2 + 3     # (das_kernels.hip refuses to launch any of these from a build that uses scratch)
    for w in pipelined:
        m = md[short[w]]
        assert m["spill"] == 0 and m["scratch"] == 0, (w, m)
        assert m["vgprs"] <= 128                        # 16 waves per CU


def test_hardwired_quads_survive_between_the_two_statements_of_a_mic(chk, asm):
    """das_pair2_kernel reads a mic's quads into v96..v111 in its statement S1 and consumes them in S2; both name those registers
    as clobbers, so the compiler may use them in between -- where only the scalar table requests belong.  No instruction in any
    S1 -> S2 gap of the generated code may name one of those registers."""
    pairs, bad = chk.hardwired_gaps(asm)
    assert pairs >= 2 * 5          # pad and lerp, five mic bodies each (three in the trip loop, two after it)
    assert not bad, bad[:5]


def test_convolution_and_split_kernels_use_no_scratch():
    """Every detector convolution kernel (LDS-DMA in float32 native / split and float16, register-staged, the stem's patch kernel), the split MVDR quadratic
    form and the tiled DFT compile without scratch memory: a spill in these loops is a silent 2-5x (hipcc's kernel-resource-usage remarks, gfx950).
    (The phase-steer power GEMM's 2- to 4-row-tile instantiations do spill a few registers, measured and accepted in DESIGN.md section 7: not asserted.)"""
    import re
    import subprocess
    import __graft_entry__ as ge
    CSRC = ge.CSRC
    seen = {}
    for name in ("conv_kernels.hip", "freq_kernels.hip"):
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + [f for f in ge.HIPCC_FLAGS if f != "-fPIC"] + [
            "--cuda-device-only", "-c", os.path.join(CSRC, name), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, timeout=900).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: +Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
                continue
            m = re.search(r"remark: +ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and cur:
                seen[cur] = int(m.group(1))
    conv = {k: v for k, v in seen.items() if "conv_dma_kernel" in k or "conv_igemm_kernel" in k or "conv_patch_kernel" in k}
    assert len(conv) >= 40, sorted(seen)[:5]                   # float32 native + split, float16, three kernels, all tile shapes
    assert not {k: v for k, v in conv.items() if v}, {k: v for k, v in conv.items() if v}
    others = {k: v for k, v in seen.items() if "dft_tile_kernel" in k or "cholesky_inverse_reg_kernel" in k or re.search(r"cgemm_bins_kernelILi2ELi1ELb[01]E", k)}
    assert len(others) >= 4, [k for k in seen if "cgemm_bins" in k][:4]
    assert not {k: v for k, v in others.items() if v}, {k: v for k, v in others.items() if v}
