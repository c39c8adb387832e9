#!/bin/bash
# round 6, GPU call 25: A/B builds of msk_lean.hip on one box (the wave-wide test as one compare; the phase wrap one level shorter; the
# recipe's -mllvm switches dropped one at a time), the demodulator alone with the bit log as bench.py runs it; 4 lanes per channel through
# the lean kernel in the same process (share8)
R=$(pwd); O=$R/gpurun_out/r06_call25; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8 ) > $O/msk_lean_builds_ab.txt 2>&1
cat $O/msk_lean_builds_ab.txt | cut -c 1-140
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 --config share8 --also none --no-cpu-baseline --no-live-traffic --ab ACG_MSK_LEAN4=0,1 --detail-file $O/share8_ab.json ) > $O/share8_ab.txt 2>&1
python - $O/share8_ab.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("share8", d["value"], d.get("ab_same_process"))
except Exception as e:
    print("failed", e)
PY
