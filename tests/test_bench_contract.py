"""bench.py's bookkeeping, without a GPU: the algorithmic work figures are SURVEY.md section 8(d)'s, the workloads are the
BASELINE.json configs, and the CPU-baseline leg (the only place outside tests/ and smoke() that may touch oracle/) returns
what the JSON line needs."""
import importlib.util
import json
import os

import util


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(util.ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_the_survey():
    b = _bench()
    M, tiles, N, X, Y, T = b.WORKLOADS["cfg2"]
    assert (M, N, X, Y) == (64, 256, 101, 101)
    D = X * Y
    assert D * M * N == 167_133_184                                   # MAC per frame
    assert b.algorithmic_bytes_per_frame("lerp", M, N, D, T) == 5_329_252
    assert b.algorithmic_bytes_per_frame("pad", M, N, D, T) == 2_717_796
    M, tiles, N, X, Y, T = b.WORKLOADS["cfg1"]
    assert b.algorithmic_bytes_per_frame("lerp", M, N, X * Y, T) == 127_972 and b.algorithmic_bytes_per_frame("pad", M, N, X * Y, T) == 96_996
    M, tiles, N, X, Y, T = b.WORKLOADS["cfg5"]
    assert X * Y * M * N == 34_162_868_224
    assert b.algorithmic_bytes_per_frame("lerp", M, N, X * Y, T) == 268_467_268 and b.algorithmic_bytes_per_frame("pad", M, N, X * Y, T) == 135_018_564


def test_metric_is_the_baseline_metric():
    base = json.load(open(os.path.join(util.ROOT, "BASELINE.json")))
    src = open(os.path.join(util.ROOT, "bench.py")).read()
    assert "beam-steered frames/sec (64 mics x 256 samples x 101x101 angles)" in src
    assert base["metric"].startswith("beam-steered frames/sec (64 mics")


def test_cpu_baseline_leg(monkeypatch):
    b = _bench()
    monkeypatch.setenv("BF_BENCH_CPU_REPLICAS", "2")
    out = b.cpu_baseline("cfg1", "lerp", budget_s=0.5)
    assert out["unit"] == "frames/s" and out["cores"] == 1 and out["kind"] in ("reference", "port") and out["value"] > 0
    assert out["all_cores"]["cores"] == 2 and out["all_cores"]["value"] > 0
