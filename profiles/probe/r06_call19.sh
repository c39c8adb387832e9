#!/bin/bash
# round 6, GPU call 19: msk_lean.hip (the framing state machine off the per-bit path) -- parity against msk.hip's kernel and the
# oracle, then the same-process A/B alone and in the bench
R=$(pwd); O=$R/gpurun_out/r06_call19; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 900 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean.txt 2>&1
tail -n 30 $O/pytest_lean.txt | cut -c 1-400
( for t in acars noise mixed; do timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 $t; done
  timeout 200 python profiles/probe/msk_lean_ab.py 2048 8 acars
  timeout 200 python profiles/probe/msk_lean_ab.py 16384 4 acars ) > $O/msk_lean_ab.txt 2>&1
grep -v amdgpu.ids $O/msk_lean_ab.txt | cut -c 1-200
for l in lean inline; do
  for c in throughput share8; do
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
