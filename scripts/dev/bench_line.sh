#!/bin/bash
# Dev tool (GPU box): one short bench.py line per (workload, algo) pair.  usage: bench_line.sh "cfg5:lerp cfg5:pad cfg2:lerp" [bench args]
cd "$(dirname "$0")/../.."
pairs="$1"; shift
for p in $pairs; do
  w=${p%%:*}; a=${p##*:}
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload $w --algo $a "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$w $a %10.1f frames/s  kernel %.4f ms  frac %.3f  %s' % (d['value'], r['kernel_ms'], r['frac'], r['kernel']))"
done
