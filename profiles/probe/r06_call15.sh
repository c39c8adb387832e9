#!/bin/bash
# round 6, GPU call 15: the demodulator's bit-decision part without a branch around it (one basic block per pass, side effects of a
# pass without a bit masked): A/B build against the product's, alone and in the bench; its parity through the demodulator tests
R=$(pwd); O=$R/gpurun_out/r06_call15; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8; bash profiles/probe/run_ab.sh 2048 8; bash profiles/probe/run_ab.sh 16384 4 ) > $O/msk_flat_ab.txt 2>&1
cat $O/msk_flat_ab.txt | cut -c 1-140
( ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libflat.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "not lab and not poly and not compat and not stamp" ) > $O/pytest_flat.txt 2>&1
tail -n 6 $O/pytest_flat.txt | cut -c 1-300
for l in base flat; do
  for c in throughput share8; do
  ( ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config $c --also none --no-cpu-baseline --no-live-traffic --detail-file $O/${c}_$l.json ) > $O/${c}_$l.txt 2>&1
  python - $O/${c}_$l.json $c $l <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], d["value"], d["whole_job_frac_of_hbm"], (d.get("roofline_msk") or {}).get("us_per_bit"), d["parity"]["blocks"], d["parity"]["blocks_exact_given_gpu_dm"], d["parity"]["end_to_end"]["blocks_differing"], d["parity"]["end_to_end"]["gpu_vs_ref_ofast"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
