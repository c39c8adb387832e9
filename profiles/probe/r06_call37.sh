#!/bin/bash
# round 6, GPU call 37: the bench gate (reference-builds leg included) with fir_u8_mm1_kernel as the down-converter of the <= 2048-channel cases
R=$(pwd); O=$R/gpurun_out/r06_call37; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
export ACG_ALLOW_TUNING=1 ACG_FIR_MM1=1
for c in throughput shard2048; do
  ( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 --config $c --also none --no-cpu-baseline --no-live-traffic --detail-file $O/$c.json ) > $O/$c.txt 2>&1
  tail -n 3 $O/$c.txt | cut -c 1-600
  python - $O/$c.json $c <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["roofline"]["kernel"], json.dumps(d["parity"])[:900])
except Exception as e:
    print("failed", e)
PY
done
