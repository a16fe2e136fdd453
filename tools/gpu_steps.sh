#!/bin/bash
# Run GPU steps one after another on the box: each under its own `timeout -k 10`, logs under gpurun_out/.  A step that
# FAILS (tests red, a bench error) does not stop the next one; a step that is KILLED at its limit (124 / 137) does -- after a
# hang nothing more goes to the GPU in this call.  Usage: source tools/gpu_steps.sh; step <name> <seconds> <command...>
mkdir -p gpurun_out
STOPPED=0
step() {
  local name=$1 limit=$2; shift 2
  if [ "$STOPPED" != 0 ]; then echo "[steps] skipping $name: an earlier step was killed at its limit"; return 0; fi
  echo "[steps] $name: $*"
  local t0=$SECONDS
  timeout -k 10 "$limit" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "[steps] $name rc=$rc in $((SECONDS - t0)) s"
  if [ $rc = 124 ] || [ $rc = 137 ]; then STOPPED=1; fi
  return 0
}
