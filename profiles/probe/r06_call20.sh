#!/bin/bash
# round 6, GPU call 20: msk_lean.hip with the bit log (every in_callback-shaped launch now takes the lean kernel): its parity
# tests, the whole GPU suite, the same-process A/B alone with and without the log, the bench A/B
R=$(pwd); O=$R/gpurun_out/r06_call20; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 900 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean.txt 2>&1
tail -n 30 $O/pytest_lean.txt | cut -c 1-400
( timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 acars 1
  timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 acars 0
  timeout 200 python profiles/probe/msk_lean_ab.py 2048 8 acars 1
  timeout 200 python profiles/probe/msk_lean_ab.py 16384 4 acars 1 ) > $O/msk_lean_ab.txt 2>&1
grep -v amdgpu.ids $O/msk_lean_ab.txt | cut -c 1-200
for l in lean inline; do
  for c in throughput share8 shard2048; do
  ( if [ $l = inline ]; then export ACG_ALLOW_TUNING=1 ACG_MSK_NOLEAN=1; fi; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config $c --also none --no-cpu-baseline --no-live-traffic --detail-file $O/${c}_$l.json ) > $O/${c}_$l.txt 2>&1
  python - $O/${c}_$l.json $c $l <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d["value"], d["whole_job_frac_of_hbm"], (d.get("roofline_msk") or {}).get("us_per_bit"), d["parity"]["blocks"], d["parity"]["blocks_exact_given_gpu_dm"], d["parity"]["end_to_end"]["blocks_differing"], d["parity"]["end_to_end"]["gpu_vs_ref_ofast"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 8 $O/pytest_gpu.txt | cut -c 1-400
