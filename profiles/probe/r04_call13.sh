#!/bin/bash
# round 4, GPU call 13: the placement diagnostic (four contexts alive, acg_placement_trial on each after a warm-up round, the FIRST
# kept) at 4096 and 16 384 channels -- what rounds 2-3 selected on
R=$(pwd); O=$R/gpurun_out/r04_call13; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
timeout 300 python bench.py --config stress --also wide,cs16 --placements 4 --no-cpu-baseline --no-ref-leg --sustain 2 --steps 10 --warmup 2 > $O/bench_placements4_stdout.txt 2> $O/bench_placements4_stderr.txt
tail -n 1 $O/bench_placements4_stdout.txt > $O/bench_line_placements4.json
python - $O/bench_line_placements4.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("stress", d["value"], d["config"].get("placement_ms_per_call"))
for k, v in d.get("also", {}).items():
    print(k, v.get("value"), v.get("placement_ms_per_call"))
PY
