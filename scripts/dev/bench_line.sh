#!/bin/bash
# Dev tool (GPU box): one short bench.py line per (workload, algo) pair.  usage: bench_line.sh "cfg5:lerp cfg5:pad cfg2:lerp" [bench args]
# BF_NATIVE_LIB=<other build of the library> for a same-box A/B; a run that fails says so (a silent one once passed for a measurement).
cd "$(dirname "$0")/../.."
pairs="$1"; shift
for p in $pairs; do
  w=${p%%:*}; a=${p##*:}
  out=$(python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload $w --algo $a "$@" 2>gpurun_out/.bench_line.err | grep '^{')
  if [ -z "$out" ]; then echo "$w $a FAILED (lib ${BF_NATIVE_LIB:-default}): $(tail -n 1 gpurun_out/.bench_line.err)"; continue; fi
  echo "$out" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('$w $a %10.1f frames/s  kernel %.4f ms  frac %.3f  %s  [lib ${BF_NATIVE_LIB:-default}]' % (d['value'], r['kernel_ms'], r['frac'], r['kernel']))"
done
