#!/usr/bin/env python3
"""bench.py -- beam-steered frames/s of the delay-and-sum hot path on MI355X (BASELINE.json metric).

One STEP = one batched launch of the hot path over `--frames` synthetic 64-mic x 256-sample windows
(190 windows = one second of audio at 48828 Hz, PC/src/config.json:4,16), every window steered to all
101 x 101 directions (BASELINE.json configs[1]).  Inputs and tables are resident in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]      (N > 1: launched by torch.distributed.run, one rank per GPU)

N > 1 (north_star): the direction grid is sharded over the ranks, every rank steers ALL
frames of the (N x larger) global batch to its own contiguous shard of directions -- per-GPU work is fixed, i.e.
weak scaling -- and one RCCL all-gather per step assembles the full heat-maps on every rank.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd")
sys.path.insert(0, PKG)

WORKLOADS = {
    # name: (N_MICROPHONES, tiles, N_SAMPLES, MAX_RES_X, MAX_RES_Y, N_TAPS)
    "cfg1": (64, 1, 256, 11, 11, 8),
    "cfg2": (64, 1, 256, 101, 101, 8),
    "shipped": (256, 4, 256, 57, 32, 8),
    "cfg5": (256, 4, 1024, 361, 361, 8),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_TLANEOPS = 78.65     # MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector (spec) = 256 CUs x 4 SIMD-32 x 2.4 GHz lane-ops/s


def algorithmic_bytes_per_frame(algo, M, N, D, T):
    """signals + tables + image."""
    sig, img = 4 * M * N, 4 * D
    return {"pad": sig + 4 * D * M + img, "lerp": sig + 8 * D * M + img, "hybrid": sig + 4 * D * M * (1 + T) + img,
            "fir_vec": sig + 4 * D * M * T + img, "fir_naive": sig + 4 * D * M * T + img}[algo]


def cpu_baseline(workload, algo, budget_s=12.0):
    """The reference's own C (oracle/_ref, built by oracle/build_ref.py) -- or, if absent, the oracle port -- timed
    on ONE host core (the reference runs a single producer process, PC/src/main.pyx:699) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import das_oracle
    import directions_np as D
    import synth
    M, tiles, N, X, Y, T = WORKLOADS[workload]
    if das_oracle.RefLib.available(workload):
        eng = das_oracle.RefLib(workload)
    else:
        eng = das_oracle.Oracle(N, X, Y, T)
    delays = D.calculate_delays(X, Y, arrays=tiles)
    mics = np.arange(M, dtype=np.int32)
    table = D.whole_samples(delays) if algo == "pad" else np.float32(delays)
    run = {"pad": eng.mimo_pad, "lerp": eng.mimo_lerp, "hybrid": eng.mimo_hybrid}[algo]
    sig = synth.s2_noise(M, N, seed=100)
    run(sig, table, mics)                      # warm-up (also loads the table)
    # time the kernel only: tables stay loaded, as in the live loop (PC/src/main.pyx:172-199)
    fn = getattr(eng.lib, eng.pre + {"pad": "mimo_pad", "lerp": "mimo_lerp", "hybrid": "mimo_convolve_hybrid"}[algo])
    img = np.zeros(X * Y, dtype=np.float32)
    args = (das_oracle._f(sig), das_oracle._f(img), das_oracle._i(mics), C.c_int(M))
    if hasattr(eng, "_fn"):
        eng._fn("mimo_pad")
    t0 = time.perf_counter()
    frames = 0
    while True:
        fn(*args)
        frames += 1
        el = time.perf_counter() - t0
        if el >= budget_s or frames >= 400:
            break
    out = {"value": frames / el, "unit": "frames/s", "cores": 1, "kind": eng.kind,
           "sample": "%d frames of %s (%s, S2 noise), kernel only, tables preloaded, %.1f s" % (frames, workload, algo, el)}
    # SURVEY.md section 8d also asks for the box's whole CPU: one independent replica of the same loop per host core
    # (forked here, before anything has touched the GPU; the loaded tables are inherited)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("BF_BENCH_CPU_REPLICAS", "16"))))   # a one-GPU box's CPU share is 16 cores
    per = max(8, min(50000, int(frames * 4.0 / max(el, 1e-3))))          # about 4 s per replica
    pipes = []
    t0 = time.perf_counter()
    for _ in range(cores):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            t1 = time.perf_counter()
            for _k in range(per):
                fn(*args)
            os.write(w, ("%d %.6f" % (per, time.perf_counter() - t1)).encode())
            os._exit(0)
        os.close(w)
        pipes.append((pid, r))
    done = 0
    for pid, r in pipes:
        msg = os.read(r, 64).decode().split()
        os.close(r)
        os.waitpid(pid, 0)
        done += int(msg[0]) if msg else 0
    wall = time.perf_counter() - t0
    out["all_cores"] = {"value": done / wall, "unit": "frames/s", "cores": cores,
                        "sample": "%d replicas x %d frames, wall %.1f s (includes fork)" % (cores, per, wall)}
    return out


def timed(fn, torch, iters, warm=2, windows=3):
    """Seconds per call: the median of `windows` timed runs of `iters` calls each (one run alone is at the mercy of a single allocator or clock hiccup:
    the extras build and drop several pipelines in one process)."""
    for _ in range(warm):
        fn()
    took = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        took.append((time.perf_counter() - t0) / iters)
    return sorted(took)[len(took) // 2]


def config_dirs():
    from interface import config
    return config.MAX_RES_X * config.MAX_RES_Y


def extras(torch, nat, delays, mics, dev):
    """The other single-GPU BASELINE configs, measured after the headline run (each a few hundred ms):
    config 3 MVDR (64 mics, 101x101), config 4 fused heat-map -> 640x640 overlay -> YOLOv5s detection, plus the
    frequency-domain delay-and-sum and the detector alone.  Parity for MVDR / detector is unpinned."""
    import synth
    from pipeline import FusedPipeline
    from realtime_scripts import beam_forming_algorithm as B, config as C
    out = {}
    M, N = 64, 256
    # the other time-domain flavours on the headline workload (same launch shape: 190 frames resident in HBM)
    Dn = config_dirs()
    win190 = torch.from_numpy(synth.frame_batch(M, N, 64)).to(dev).repeat(3, 1, 1)[:190].contiguous()
    img190 = torch.empty((190, Dn), dtype=torch.float32, device=dev)
    for name, algo_id, loader, table in (
            ("time_domain_pad", nat.PAD, nat.lib.load_coefficients_pad, np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel()),
            ("time_domain_hybrid", nat.HYBRID, nat.lib.load_coefficients_convolve_hybrid, np.ascontiguousarray(np.float32(delays)).ravel())):
        loader(nat.iptr(table) if table.dtype == np.int32 else nat.fptr(table), table.size)
        nat.check()
        s0 = torch.cuda.current_stream().cuda_stream
        dt = timed(lambda: nat.lib.bf_das_device(algo_id, win190.data_ptr(), M, img190.data_ptr(), Dn, 190, nat.iptr(mics), M, 0, Dn, s0), torch, 5)
        nat.check()
        out[name] = {"frames_per_s": 190 / dt, "ms_per_launch": dt * 1e3}
    # what a drop-in user of the reference's one-frame API sees (PC/src/api.c:951-958 -> mimo_lerp with HOST pointers: H2D of the
    # 64 KB block, launch, D2H of the 40 KB map, sync) -- PCIe-inclusive, never the headline value
    lerp_tab = np.ascontiguousarray(np.float32(delays)).ravel()
    nat.lib.load_coefficients_lerp(nat.fptr(lerp_tab), lerp_tab.size)
    nat.check()
    one = synth.s2_noise(M, N)
    img1 = np.zeros(Dn, dtype=np.float32)
    for _ in range(30):
        nat.lib.mimo_lerp(nat.fptr(one), nat.fptr(img1), nat.iptr(mics), M)
    nat.check()
    t0 = time.perf_counter()
    for _ in range(300):
        nat.lib.mimo_lerp(nat.fptr(one), nat.fptr(img1), nat.iptr(mics), M)
    dt = (time.perf_counter() - t0) / 300
    nat.check()
    out["host_pointer_mimo_lerp"] = {"calls_per_s": 1.0 / dt, "us_per_call": dt * 1e6, "note": "one frame per call, host pointers, PCIe both ways, timed inside this process (torch loaded, other streams alive; the call polls its stream before it blocks); "
                                                                                                    "scripts/host_path_latency.py measures the same call at 58 us standalone; real time needs 190.7 windows/s"}
    # config 4: float32 first -- the precision the reference's detector call runs at (ultralytics' predict default,
    # image-detection/src/yolo_smooth_tracking.py:13-23) -- then the float16 fast mode, labelled as such
    Bf = 64
    win = torch.from_numpy(synth.frame_batch(M, N, Bf)).to(dev)
    cam = torch.randint(0, 256, (Bf, 640, 640, 3), dtype=torch.uint8, device=dev)
    # float32 twice: the library's default float32 mode (every operand split exactly into three bfloat16 parts, six products on the bfloat16 matrix
    # pipes, float32 accumulation: float32 accuracy -- tests/test_detector.py holds it to the same bounds against float64 as the float32 instruction)
    # and the float32 matrix instruction itself
    for half, mode, suffix in ((False, 1, ""), (False, 0, "_f32_mfma_instruction"), (True, 1, "_fp16_fast_mode")):
        initial = nat.lib.bf_conv2d_f32_mode(mode)
        try:
            pipe = FusedPipeline("lerp", 640, dev, half=half)
            pipe.load_tables(delays, mics)
            dt = timed(lambda: pipe.step(win, cam), torch, 10)
            if half:
                prec, how, peak = "fp16 (f32 accumulation; narrower than the reference's fp32 predict)", "v_mfma_f32_32x32x16_f16", 2500.0
            elif mode == 1:
                prec, how, peak = ("fp32 (= ultralytics' default predict precision): float32 operands, each split exactly into 3 bfloat16 parts, 6 part products per "
                                   "product, float32 accumulation", "6 x v_mfma_f32_32x32x16_bf16 per 16 values of K", 2500.0 / 6.0)
            else:
                prec, how, peak = "fp32 (= ultralytics' default predict precision), the float32 matrix instruction", "v_mfma_f32_32x32x2_f32", 157.3
            out["fused_heatmap_overlay_yolo" + suffix] = {"frames_per_s": Bf / dt, "batch": Bf, "image": "640x640x3 uint8", "dtype": "fp16" if half else "fp32",
                                                          "detector": "YOLOv5s-shaped, %s, random init, 1 class" % prec, "conv_backend": pipe.detector.conv_backend}
            x = pipe.detector.preprocess(cam)
            dt = timed(lambda: pipe.detector.postprocess(pipe.detector.raw(x)), torch, 10)
            out["yolo_only" + suffix] = {"detections_per_s": Bf / dt, "batch": Bf, "gflop_per_frame": 15.8, "dtype": "fp16" if half else "fp32",
                                         "conv_backend": pipe.detector.conv_backend, "matrix_instruction": how, "mfma_tflops": 15.8e9 * Bf / dt / 1e12,
                                         "mfma_peak_tflops": peak}
            del pipe, x
        finally:
            nat.lib.bf_conv2d_f32_mode(initial)
    # config 3 + frequency-domain DAS: same 64-mic array and 101x101 grid through the frequency-domain geometry
    old = (C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y)
    C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 64, 1, 101, 101
    try:
        fb = B.FrequencyBeamformer()
        F = 190
        frames = torch.from_numpy(synth.frame_batch(M, N, 64)).to(dev).repeat(3, 1, 1)[:F].contiguous()
        dt = timed(lambda: fb.mvdr_power(frames, 1e-2), torch, 5)
        dt_stream = timed(lambda: fb.mvdr_power(frames, 1e-2, defer_check=True), torch, 5)     # the same maps without a host read-back per map
        fb.check_deferred()
        # executed matrix flops per bin: covariance 8 M^2 F; quadratic form over the lower-triangular L^-1, whose all-zero 32 x 32 blocks are not issued
        tri = sum(32 * min(M, 32 * (t + 1)) for t in range((M + 31) // 32))
        flop = fb.K * (8.0 * M * M * F + 8.0 * tri * fb.D)
        how = ("float32 operands; the two bin-reducing GEMMs on 6 x v_mfma_f32_32x32x16_bf16 per 16 values of K (exact 3-way bfloat16 split, float32 sums), "
               "DFT / covariance on v_mfma_f32_32x32x2_f32") if nat.lib.bf_fd_gemm_f32_mode(-1) == 1 else "v_mfma_f32_32x32x2_f32"
        out["mvdr"] = {"maps_per_s": 1.0 / dt, "frames_per_s": F / dt, "windows_per_map": F, "bins": fb.K, "ms_per_map": dt * 1e3,
                       "maps_per_s_streamed": 1.0 / dt_stream, "streamed_note": "status of the factorisations checked once after the run (defer_check) instead of read back per map",
                       "mfma_tflops": flop / dt / 1e12, "mfma_peak_tflops_f32": 157.3, "matrix_instruction": how,
                       "flop_note": "executed matrix flops of the float32 products (dense count of the quadratic form x %.2f: zero blocks of the triangular factor skipped)" % (tri / float(M * M))}
        dt = timed(lambda: fb.das_power(frames), torch, 5)
        out["freq_domain_das"] = {"frames_per_s": F / dt, "ms_per_step": dt * 1e3, "mfma_tflops": fb.K * 8.0 * M * F * fb.D / dt / 1e12, "matrix_instruction": how}
        initial = nat.lib.bf_fd_gemm_f32_mode(0)               # the same two on the float32 matrix instruction, for comparison
        try:
            dt = timed(lambda: fb.mvdr_power(frames, 1e-2), torch, 5)
            out["mvdr_f32_mfma_instruction"] = {"maps_per_s": 1.0 / dt, "ms_per_map": dt * 1e3, "mfma_tflops": flop / dt / 1e12, "mfma_peak_tflops_f32": 157.3}
            dt = timed(lambda: fb.das_power(frames), torch, 5)
            out["freq_domain_das_f32_mfma_instruction"] = {"frames_per_s": F / dt, "ms_per_step": dt * 1e3, "mfma_tflops": fb.K * 8.0 * M * F * fb.D / dt / 1e12}
        finally:
            nat.lib.bf_fd_gemm_f32_mode(initial)
        # BASELINE config 5's array: 4 tiles = 256 mics (two-block Cholesky), the as-shipped 57 x 32 grid, 320 windows per map
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 256, 4, 57, 32
        fb4 = B.FrequencyBeamformer()
        F4 = 320
        frames4 = torch.from_numpy(synth.frame_batch(256, N, 64)).to(dev).repeat(5, 1, 1)[:F4].contiguous()
        frames4 = frames4 + 0.05 * torch.randn_like(frames4)          # 320 distinct windows: a full-rank covariance
        dt = timed(lambda: fb4.mvdr_power(frames4, 1e-2), torch, 3)
        out["mvdr_256_mics"] = {"maps_per_s": 1.0 / dt, "windows_per_map": F4, "bins": fb4.K, "grid": "57x32", "ms_per_map": dt * 1e3}
    finally:
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = old
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=190, help="windows per GPU per step")
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--algo", default="lerp", choices=["pad", "lerp", "hybrid", "fir_vec", "fir_naive"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the MVDR / fused-detector side measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    # the CPU baseline forks replicas: run it before anything initialises the GPU in this process
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline and args.algo in ("pad", "lerp", "hybrid"):
        cpu_line = cpu_baseline(args.workload, args.algo)

    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # $BF_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the cards and the
    # heat-map gather is staged through the host.  The measured configuration is always nccl (= RCCL over xGMI).
    backend = os.environ.get("BF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from interface import config
    from lib import _native as nat
    from lib import directions
    import synth
    import multi_gpu

    M, tiles, N, X, Y, T = WORKLOADS[args.workload]
    D = X * Y
    if nat.lib.bf_set_device(local_rank) != 0:
        nat.check()
    config.configure(N_MICROPHONES=M, ACTIVE_TILES=tiles, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)

    # ---- tables (once, outside the timed region -- as the reference's producer loops do)
    delays = directions.calculate_delays()
    mics = np.ascontiguousarray(directions.active_microphones()[0].astype(np.int32))
    algo_id = {"pad": nat.PAD, "lerp": nat.LERP, "hybrid": nat.HYBRID, "fir_vec": nat.FIR_VEC, "fir_naive": nat.FIR_NAIVE}[args.algo]
    if args.algo == "pad":
        t = np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel()
        nat.lib.load_coefficients_pad(nat.iptr(t), t.size)
    elif args.algo == "lerp":
        t = np.ascontiguousarray(np.float32(delays)).ravel()
        nat.lib.load_coefficients_lerp(nat.fptr(t), t.size)
    elif args.algo == "hybrid":
        t = np.ascontiguousarray(np.float32(delays)).ravel()
        nat.lib.load_coefficients_convolve_hybrid(nat.fptr(t), t.size)
    else:
        t = directions.compute_convolve_h().ravel()
        nat.lib.load_coefficients_convolve(nat.fptr(t), t.size)
    nat.check()

    # ---- synthetic input resident in HBM: the GLOBAL batch on every rank (direction sharding broadcasts frames)
    frames_global = args.frames * world
    host = synth.frame_batch(M, N, min(frames_global, 64))          # 64 distinct windows, tiled to the batch size
    reps = (frames_global + host.shape[0] - 1) // host.shape[0]
    d_sig = torch.from_numpy(np.tile(host, (reps, 1, 1))[:frames_global]).to(dev)
    lo, hi = multi_gpu.shard_range(D, world, rank)
    shard_cap = multi_gpu.shard_capacity(D, world)
    # two heat-map buffers: the all-gather of step k (RCCL's own stream) overlaps the beamforming kernel of step k + 1
    nbuf = 2 if world > 1 else 1
    d_part = [torch.zeros((frames_global, shard_cap), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    d_full = [torch.zeros((world * frames_global, shard_cap), dtype=torch.float32, device=dev) for _ in range(nbuf)] if world > 1 else None
    pending = [None] * nbuf
    stream = torch.cuda.current_stream()

    def gather(k):
        b = k % nbuf
        if backend == "nccl":
            pending[b] = dist.all_gather_into_tensor(d_full[b], d_part[b], async_op=True)
        else:
            out = torch.empty(d_full[b].shape, dtype=torch.float32)
            dist.all_gather_into_tensor(out, d_part[b].cpu())
            d_full[b].copy_(out)

    def release(k):
        b = k % nbuf
        if pending[b] is not None:
            pending[b].wait()          # the gather that last read this buffer (stream-level wait, the host runs on)
            pending[b] = None

    def beamform(k):
        b = k % nbuf
        release(k)
        rc = nat.lib.bf_das_device(algo_id, d_sig.data_ptr(), M, d_part[b].data_ptr(), shard_cap, frames_global, nat.iptr(mics), M,
                                   lo, hi, stream.cuda_stream)
        if rc != 0:
            nat.check()

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for k in range(args.warmup):
        beamform(k)
        if world > 1:
            gather(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        # HIP events on the launch stream bracket the beamforming kernel only: neither the collective nor the wait for
        # the previous gather of this buffer
        release(k)
        ev[k][0].record(stream)
        beamform(k)
        ev[k][1].record(stream)
        if world > 1:
            gather(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_all = np.asarray([a.elapsed_time(b) for a, b in ev], dtype=np.float64)
    kernel_ms = float(np.mean(kernel_all))

    assert torch.isfinite(d_part[0][:, : hi - lo]).all(), "non-finite beam power"
    if world > 1:
        # every rank now holds every shard: check the assembled map against this rank's own shard of the last step
        last = (args.steps - 1) % nbuf
        mine = d_full[last][rank * frames_global:(rank + 1) * frames_global, : hi - lo]
        assert torch.equal(mine, d_part[last][:, : hi - lo]), "all-gather returned a different shard"

    if rank == 0:
        fps = frames_global * args.steps / elapsed
        # per-launch algorithmic bytes of THIS rank's kernel: all frames, its direction shard
        share = (hi - lo) / D
        bytes_frame = algorithmic_bytes_per_frame(args.algo, M, N, D, T)
        sig_bytes = 4 * M * N
        alg_bytes = frames_global * (sig_bytes + (bytes_frame - sig_bytes) * share)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        macs = frames_global * (hi - lo) * M * N
        # The binding resource is the fp32 VALU, not HBM: per (direction, mic, sample) the kernels execute
        # 1 lane-op (pad: add), 2 (lerp: fma + add), 8 (8-tap FIRs: fma chain; 15 in the AVX summation order).  Peak: the guide's
        # 157.3 TFLOP/s FP32 vector = 78.65 T lane-ops/s at 2.4 GHz (an FMA lane-op counts two flops there).
        lane_ops = macs * {"pad": 1, "lerp": 2, "hybrid": 8, "fir_naive": 8, "fir_vec": 15}[args.algo]
        valu_achieved = lane_ops / (kernel_ms * 1e-3) / 1e12
        # ... and against what the instruction mix alone sustains on the chip (scripts/dev/valu_probe.hip / gen_issue_probe.py,
        # 16 waves per CU, ns per (direction, mic, 256-sample segment) step per CU)
        steps = frames_global * (hi - lo) * M * ((N + 255) // 256)
        floor_ns = {"pad": 1.12, "lerp": 2.12}.get(args.algo, 32 * 1.18 / 4)
        step_ns = kernel_ms * 1e6 * 256 / steps              # 256 CUs
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            rec = json.load(open(tpath)).get("%s_%s_f%d_n%d" % (args.workload, args.algo, args.frames, world))
            if rec:
                traffic = rec["hbm_bytes_per_launch"]
                traffic_source = "stored, not measured in this run: profiles/traffic.json <- " + rec.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
        line = {
            "metric": "beam-steered frames/sec (64 mics x 256 samples x 101x101 angles)" if args.workload == "cfg2" else "beam-steered frames/sec",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d mics x %d samples x %dx%d directions, delay-and-sum (%s), %d frames/GPU/step"
                                   % (args.workload, M, N, X, Y, args.algo, args.frames),
                       "frames_per_step_global": frames_global, "directions_per_gpu": hi - lo,
                       "parallelism": "directions sharded over %d GPU(s) + RCCL all-gather of the heat-maps" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "achieved": valu_achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s (fp32; peak = 157.3 TFLOP/s / 2)",
                         "frac": valu_achieved / VALU_PEAK_TLANEOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": {5: "copies::das_pair_kernel<%s>", 2: "copies::das_copies_kernel<%s>", 3: "copies::das_copies_kernel<%s, DIRECT>",
                                    4: "copies::das_copies_kernel<%s>", 6: "copies::das_long_kernel<%s>", 7: "copies::das_hybrid_pair_kernel<%s>", 8: "copies::das_pair2_kernel<%s>"}.get(nat.lib.bf_last_das_variant(), "das_mimo_kernel<%s>") % args.algo,
                         "kernel_ms": kernel_ms, "kernel_ms_min": float(kernel_all.min()), "kernel_ms_median": float(np.median(kernel_all)),
                         "kernel_ms_max": float(kernel_all.max()), "gmacs_per_s": macs / (kernel_ms * 1e-3) / 1e9,
                         "note": "gather-accumulate kernel, no MFMA: tables stay L2-resident across the frames of a launch and sample quads are "
                                 "re-read from LDS only when a direction's delay differs from its neighbour's, so HBM is nearly idle and the "
                                 "fp32 VALU binds (DESIGN.md section 5)",
                         "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": alg_bytes, "traffic": traffic, "traffic_source": traffic_source},
                         "valu_probe": {"ns_per_step_per_cu": step_ns, "floor_ns_per_step_per_cu": floor_ns, "frac_of_probe_floor": floor_ns / step_ns,
                                        "floor": "the step's packed instructions alone at 16 waves/CU and the clock the chip holds under that load"}},
        }
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        if world == 1 and not args.no_extras and args.workload == "cfg2":
            try:
                line["extra"] = extras(torch, nat, delays, mics, dev)
            except Exception as e:      # side measurements must never take the headline line down
                line["extra"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
