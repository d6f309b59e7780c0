#!/bin/bash
# Dev tool (GPU box): phase ablation of the batched pad / lerp kernel through BF_DEBUG profiling bits (wrong images by design).
#   bit 9 (512) no staging writes, bit 10 (1024) no ordered power sum, bit 11 (2048) no sweep
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for algo in lerp pad; do
  for dbg in 0 512 1024 2048 1536 2560 3072 3584; do
    BF_DEBUG=$dbg python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --algo $algo "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$algo BF_DEBUG=%5d  %9.0f frames/s  kernel %.4f ms' % ($dbg, d['value'], d['roofline']['kernel_ms']))"
  done
done
