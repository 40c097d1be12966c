#!/bin/bash
# Developer tool (GPU box): sample sclk / power with rocm-smi while a command runs.
# Usage: scripts/clock_probe.sh <command...>
"$@" > /tmp/probe_cmd.out 2>&1 &
PID=$!
sleep ${PROBE_DELAY:-1.5}
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | head -6
  echo "--"
  sleep 0.4
done
wait $PID
tail -5 /tmp/probe_cmd.out
