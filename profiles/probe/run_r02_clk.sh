#!/bin/bash
export TMPDIR=/tmp
ls /sys/class/drm/card*/device/pp_dpm_* 2>/dev/null | head
(python profiles/probe/fir_alloc_probe.py 16384 4 10 54 2>&1 | grep round | while read l; do echo "$(date +%s.%N | cut -c1-14) $l"; done) &
P=$!
for i in $(seq 1 40); do
  echo "$(date +%s.%N | cut -c1-14) SMI $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|power\|junction\|memory" | sed 's/GPU\[0\]\s*: //' | tr '\n' '|' | cut -c1-420)"
  sleep 0.5
done
wait $P
