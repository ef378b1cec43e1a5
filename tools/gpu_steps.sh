#!/bin/bash
# Helper for multi-step gpurun calls: `step <seconds> <logfile> <command...>` runs one GPU step under its own timeout and
# logs to gpurun_out/; a step that is killed at its limit (124 / 137) ends the whole call (no further GPU step is started
# after a hang), an ordinary failure is recorded and the call goes on.
step() {
  local limit=$1 log=$2; shift 2
  echo "=== $(date +%T) step: $* (limit ${limit}s) -> $log"
  timeout -k 10 "$limit" "$@" > "$log" 2>&1
  local rc=$?
  echo "    rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "step timed out or was killed: stopping the call"; tail -5 "$log"; exit $rc
  fi
  return 0
}
