#!/bin/bash
# usage: tools/gpu_step.sh LIMIT_SECONDS command...   -- runs the command under timeout; exit 0 unless it was killed at the limit
# (a failing test run must not stop the profile steps behind it, a hung GPU step must)
lim=$1; shift
timeout -k 10 "$lim" "$@"
rc=$?
echo "[gpu_step] rc=$rc: $*" >&2
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
exit 0
