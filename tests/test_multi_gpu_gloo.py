"""CPU, world_size 2 over gloo: the N > 1 path (direction sharding + one all-gather) with the oracle standing in
for the per-shard GPU compute.  Checks that the assembled maps equal the single-process maps."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util

ROOT = util.ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import das_oracle
    import multi_gpu
    import synth
    import util as U
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = U.CONFIGS["cfg1"]
    M, N, D = c["M"], c["N"], c["X"] * c["Y"]
    frames = synth.frame_batch(M, N, 3)
    orc = das_oracle.Oracle(N, c["X"], c["Y"], c["T"])
    orc.load(1, U.table_for("lerp", "cfg1"))
    mics = np.arange(M, dtype=np.int32)

    def compute(lo, hi):
        return torch.from_numpy(np.stack([orc.mimo_range(1, frames[f], mics, lo, hi) for f in range(frames.shape[0])]))

    full = multi_gpu.sharded_heatmaps(compute, frames.shape[0], D)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direction_sharding_gloo(tmp_path, oracle_lib, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import synth
    c = util.CONFIGS["cfg1"]
    M, N, D = c["M"], c["N"], c["X"] * c["Y"]
    frames = synth.frame_batch(M, N, 3)
    orc = oracle_lib.Oracle(N, c["X"], c["Y"], c["T"])
    orc.load(1, util.table_for("lerp", "cfg1"))
    mics = np.arange(M, dtype=np.int32)
    want = np.stack([orc.mimo_range(1, frames[f], mics, 0, D) for f in range(3)])
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got.shape == (3, D) and np.array_equal(got, want)


def test_shard_ranges_cover_the_grid():
    import multi_gpu
    for D in (1, 7, 121, 10201, 130321):
        for world in (1, 2, 3, 4, 8):
            cap = multi_gpu.shard_capacity(D, world)
            spans = [multi_gpu.shard_range(D, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == D
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= cap for lo, hi in spans)


@pytest.mark.gpu
def test_bench_two_ranks_over_gloo_keeps_the_json_contract():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE card with
    BF_BENCH_BACKEND=gloo (host-staged gather): the JSON line's multi-GPU bookkeeping -- n_gpus, the global batch (weak
    scaling: per-GPU frames fixed), the direction shard per GPU -- and the shard self-check bench.py runs after its timed region.
    The RCCL path itself needs two GPUs; this box has one."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "16"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["frames_per_step_global"] == 32 and d["config"]["directions_per_gpu"] == 5101
    assert d["unit"] == "frames/s" and d["value"] > 0 and abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["hbm"]["peak"] == 8000.0
    assert "cpu_baseline" not in d                                  # rank 0 at N = 1 only
