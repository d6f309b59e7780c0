"""The hand-scheduled LDS reads of das_kernels.hip stay sound only while hipcc never copies a register that has a read in
flight (scripts/dev/check_inflight_copies.py explains).  Compiles the kernels to gfx950 assembly (no GPU needed) and scans it."""
import importlib.util
import os

import util


def test_no_register_with_a_read_in_flight_is_copied(tmp_path):
    spec = importlib.util.spec_from_file_location("check_inflight_copies", os.path.join(util.ROOT, "scripts", "dev", "check_inflight_copies.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    asm = str(tmp_path / "das_kernels.s")
    chk.compile_asm(asm)
    kernels, bad = chk.scan(asm)
    assert kernels >= 30           # every das_copies_kernel / das_pair_kernel instantiation
    assert not bad, bad[:5]
