#!/bin/bash
# round 6, GPU call 27: where the bit log stands -- share8 (4 lanes per channel) and the headline without the log, lean kernel on / off in the same process
R=$(pwd); O=$R/gpurun_out/r06_call27; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for spec in "share8 ACG_MSK_LEAN4=0,1 0" "share8 ACG_MSK_LEAN4=0,1 1" "throughput ACG_MSK_NOLEAN=1,0 0" "throughput ACG_MSK_NOLEAN=1,0 1"; do
  set -- $spec
  ( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 --config $1 --also none --no-cpu-baseline --no-live-traffic --bitlog $3 --ab $2 --detail-file $O/$1_log$3.json ) > $O/$1_log$3.txt 2>&1
  python - $O/$1_log$3.json $1 $3 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); ab = d.get("ab_same_process") or {}
    print(sys.argv[2], "bitlog", sys.argv[3], d["value"], {k: v for k, v in ab.items() if "telemetry" not in k})
except Exception as e:
    print("failed", e)
PY
done
