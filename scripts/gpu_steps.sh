#!/bin/bash
# Run GPU steps one after another on the gpurun box, each under its own timeout, logging to gpurun_out/.
# A step that is killed by its timeout (exit 124/137) stops the sequence (no further GPU work after a hang);
# an ordinary failure (non-zero exit) is recorded and the next step still runs.
# usage: scripts/gpu_steps.sh "<name>|<timeout_s>|<command>" ...
mkdir -p gpurun_out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
overall=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] timeout ${tmo}s: $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/${name}.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s; tail:"
  tail -n 15 "gpurun_out/${name}.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== [$name] was killed by its timeout: stopping here"; exit $rc; fi
  [ $rc -ne 0 ] && overall=$rc
done
exit $overall
