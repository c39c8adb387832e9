#!/bin/bash
# round 6, GPU call 14: the unrolled demodulator -- the recipe's -mllvm switches one at a time (they were tuned on the rolled loop),
# then the whole GPU suite and the bench's headline / share8 / shard2048 with the product build
R=$(pwd); O=$R/gpurun_out/r06_call14; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8; bash profiles/probe/run_ab.sh 16384 4 ) > $O/msk_flags_ab.txt 2>&1
cat $O/msk_flags_ab.txt | cut -c 1-140
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 5 $O/pytest_gpu.txt | cut -c 1-300
for c in throughput shard2048 share8; do
  ( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config $c --also none --no-cpu-baseline --no-live-traffic --detail-file $O/$c.json ) > $O/$c.txt 2>&1
  python - $O/$c.json $c <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], (d.get("roofline_msk") or {}).get("us_per_bit"), d["parity"]["end_to_end"]["blocks_differing"], d["parity"]["end_to_end"]["gpu_vs_ref_ofast"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
