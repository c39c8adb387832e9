#!/bin/bash
# round 6, GPU call 28: 4 lanes per channel through the lean kernel at 16 384 channels beside the streaming down-converter (wide), same process
R=$(pwd); O=$R/gpurun_out/r06_call28; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for spec in "wide ACG_MSK_LEAN4=0,1 1" "wide ACG_MSK_LEAN4=1,0 1"; do
  set -- $spec
  ( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 --config $1 --also none --no-cpu-baseline --no-live-traffic --sustain 5 --bitlog $3 --ab $2 --detail-file $O/$1_$2.json ) > $O/$1_$2.txt 2>&1
  python - $O/$1_$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); ab = d.get("ab_same_process") or {}
    print(sys.argv[2], sys.argv[3], d["value"], {k: v for k, v in ab.items() if "telemetry" not in k}, [ (v[0] or {}).get("sclk") for k, v in ab.items() if "telemetry" in k])
except Exception as e:
    print("failed", e)
PY
done
