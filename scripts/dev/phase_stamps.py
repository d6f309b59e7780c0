#!/usr/bin/env python3
"""Dev tool: where the waves of the batched pad / lerp kernel spend their time.

  build (here or on the GPU box):  python3 scripts/dev/phase_stamps.py build     -> scripts/dev/bin/libbeamformer_hip_stamps.so
  run (GPU box):                   python3 scripts/dev/phase_stamps.py run [lerp|pad] [workload] [frames] [BF_DEBUG]

The profiling library is the production source compiled with -DBF_STAMPS: every wave of das_pair_kernel reads s_memtime at
its phase boundaries (all next to barriers) and the per-phase totals are summed over the launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "scripts", "dev", "bin", "libbeamformer_hip_stamps.so")
sys.path.insert(0, ROOT)


def build():
    import __graft_entry__ as ge
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(ge.CSRC, s) for s in ge.SOURCES]
    cmd = ["/opt/rocm/bin/hipcc"] + ge.HIPCC_FLAGS + ["-shared", "-DBF_STAMPS"] + srcs + ["-o", OUT]
    subprocess.check_call(cmd)
    print("built", OUT)


def run(algo="lerp", workload="cfg2", frames=190, debug=0):
    os.environ["BF_NATIVE_LIB"] = OUT
    os.environ["BF_DEBUG"] = str(debug)
    import ctypes as C
    import numpy as np
    import torch
    import bench
    sys.path.insert(0, bench.PKG)
    from interface import config
    from lib import _native as nat, directions
    import synth
    M, tiles, N, X, Y, T = bench.WORKLOADS[workload]
    config.configure(N_MICROPHONES=M, ACTIVE_TILES=tiles, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)
    delays = directions.calculate_delays()
    if algo == "pad":
        t = np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel()
        nat.lib.load_coefficients_pad(nat.iptr(t), t.size)
    else:
        t = np.ascontiguousarray(np.float32(delays)).ravel()
        nat.lib.load_coefficients_lerp(nat.fptr(t), t.size)
    nat.check()
    D = X * Y
    mics = np.arange(M, dtype=np.int32)
    sig = torch.from_numpy(synth.frame_batch(M, N, frames)).cuda()
    img = torch.empty((frames, D), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    go = lambda: nat.lib.bf_das_device(nat.ALGOS[algo] if hasattr(nat, "ALGOS") else {"pad": 0, "lerp": 1}[algo], sig.data_ptr(), M, img.data_ptr(), D, frames,
                                       nat.iptr(mics), M, 0, D, stream)
    for _ in range(3):
        assert go() == 0, nat.check()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert nat.lib.bf_read_phase_stamps(buf, 1) == 0
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    assert nat.lib.bf_read_phase_stamps(buf, 1) == 0
    v = [int(x) for x in buf]
    names = ["sweep", "wait (chunk free)", "staging", "wait (chunk staged)", "wait (rows free)", "parking", "wait (rows parked)", "ordered sum (+ its idle)"]
    tot = sum(v[:8])
    print("%s %s %d frames BF_DEBUG=%d: %.4f ms per launch (%.0f frames/s), %d waves stamped per launch" % (algo, workload, frames, debug, ms, frames / ms * 1e3, v[8] // reps))
    for n, x in zip(names, v[:8]):
        print("  %-26s %6.2f %%" % (n, 100.0 * x / max(tot, 1)))
    print("  mean wave lifetime %.1f ticks; ticks per ms of launch per wave slot: %.0f" % (tot / max(v[8], 1), tot / reps / ms / 4096.0))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        a = sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:]
        run(a[0] if a else "lerp", a[1] if len(a) > 1 else "cfg2", int(a[2]) if len(a) > 2 else 190, int(a[3]) if len(a) > 3 else 0)
