#!/usr/bin/env python3
"""HIP-graph capture and replay of the fused heat-map -> overlay -> detector step (pipeline.FusedPipeline.step).

  python3 scripts/graph_fused.py [batch ...]        (GPU box; default batches 1 8 64)

Every device entry point of the C-ABI only enqueues on the caller's stream with by-value arguments, so a step can be
captured once its first (eager) call has uploaded the adaptive array and built the table digest.  The script warms up,
captures one step into a torch.cuda.CUDAGraph (HIP graph on ROCm), replays it and compares every output with the eager
call on the same inputs; it also times both.  tests/test_detector.py::test_gpu_fused_step_graph_replay runs the same
check once at batch 64."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))


def build_pipeline(half=False):
    from interface import config
    from lib import directions
    from pipeline import FusedPipeline
    config.configure(N_MICROPHONES=64, ACTIVE_TILES=1, N_SAMPLES=256, MAX_RES_X=101, MAX_RES_Y=101, N_TAPS=8)
    pipe = FusedPipeline("lerp", 640, half=half)
    pipe.load_tables(directions.calculate_delays(), np.arange(64))
    return pipe


def graph_vs_eager(pipe, B, seed=0, time_it=False):
    """-> dict(equal=bool per output, eager_ms, graph_ms).  The camera frames and windows live in static tensors (graph inputs)."""
    import torch
    import synth
    rng = np.random.default_rng(seed)
    win = torch.from_numpy(synth.frame_batch(64, 256, B)).cuda()
    cam = torch.from_numpy(rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    prev0 = pipe.stream_state.prev.clone()

    def reset():
        pipe.stream_state.prev.copy_(prev0)           # the temporal blend carries state from batch to batch

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # warm-up on a side stream, as torch.cuda.graph asks
        for _ in range(2):
            reset()
            eager = pipe.step(win, cam)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = [t.clone() for t in eager]
    reset()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = pipe.step(win, cam)
    res = {}
    for rep in range(2):                              # two replays: the second one runs on memory the first one left behind
        reset()
        g.replay()
        torch.cuda.synchronize()
        for name, a, b in zip(("power", "frames", "boxes", "counts"), eager, out):
            res[name] = res.get(name, True) and bool(torch.equal(a, b))
    if time_it:
        def timed(fn, n=20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        res["eager_ms"] = timed(lambda: pipe.step(win, cam))
        res["graph_ms"] = timed(g.replay)
    return res


if __name__ == "__main__":
    half = "--half" in sys.argv
    pipe = build_pipeline(half)
    for B in [int(a) for a in sys.argv[1:] if a != "--half"] or [1, 8, 64]:
        r = graph_vs_eager(pipe, B, time_it=True)
        print("B=%d eager %.3f ms  graph %.3f ms  (%.2fx)  equal: %s" % (B, r["eager_ms"], r["graph_ms"], r["eager_ms"] / r["graph_ms"],
                                                                       {k: v for k, v in r.items() if isinstance(v, bool)}), flush=True)
