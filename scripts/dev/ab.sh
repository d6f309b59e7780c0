#!/bin/bash
# Dev tool (GPU box): bench.py A/B over BF_DEBUG values.  usage: ab.sh "0 4096 ..." [bench args]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
vals="$1"; shift
for algo in lerp pad; do
  for dbg in $vals; do
    BF_DEBUG=$dbg python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --algo $algo "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$algo BF_DEBUG=%5d  %9.0f frames/s  kernel %.4f ms' % ($dbg, d['value'], d['roofline']['kernel_ms']))"
  done
done
