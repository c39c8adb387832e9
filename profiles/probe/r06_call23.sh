#!/bin/bash
# round 6, GPU call 23: 16 384 channels with 8 lanes per channel (two lean demodulator waves per SIMD) against the table's 4 lanes
# (msk.hip's kernel) and 4 lanes through the lean kernel: alone and in the bench (share8 is the demodulator-bound case there)
R=$(pwd); O=$R/gpurun_out/r06_call23; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 200 python profiles/probe/msk_lean_ab.py 16384 4 acars 1
  ACG_ALLOW_TUNING=1 ACG_MSK_LPC=8 timeout 200 python profiles/probe/msk_lean_ab.py 16384 4 acars 1
  ACG_ALLOW_TUNING=1 ACG_MSK_LEAN4=1 timeout 200 python profiles/probe/msk_lean_ab.py 16384 4 acars 1 ) > $O/msk_lean_ab.txt 2>&1
grep -v amdgpu.ids $O/msk_lean_ab.txt | cut -c 1-200
for l in table lpc8 lean4; do
  for c in share8 wide; do
  ( if [ $l = lpc8 ]; then export ACG_ALLOW_TUNING=1 ACG_MSK_LPC=8; fi; if [ $l = lean4 ]; then export ACG_ALLOW_TUNING=1 ACG_MSK_LEAN4=1; fi
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config $c --also none --no-cpu-baseline --no-live-traffic --detail-file $O/${c}_$l.json ) > $O/${c}_$l.txt 2>&1
  python - $O/${c}_$l.json $c $l <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d["value"], d["whole_job_frac_of_hbm"], (d.get("roofline_msk") or {}).get("us_per_bit"), d["parity"]["blocks"], d["parity"]["blocks_exact_given_gpu_dm"], d["parity"]["end_to_end"]["blocks_differing"], d["parity"]["end_to_end"]["gpu_vs_ref_ofast"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
