#!/usr/bin/env bash
# Development aid: sample rocm-smi (socket power, sclk, temperature) every 0.25 s while a command runs.
#   bash tools/power_trace.sh out.txt python bench.py ...
OUT=$1; shift
rocm-smi --showmaxpower --showpower --showclocks > "$OUT.before" 2>&1 || true
( while true; do rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ';' ; echo; sleep 0.25; done ) > "$OUT" &
SPID=$!
"$@"
RC=$?
kill $SPID 2>/dev/null
exit $RC
