#!/bin/bash
# round 6, GPU call 13: the demodulator's counted loop unrolled 3 x / 9 x (A/B builds, same box): may the scheduler run period n's
# decision / framing beside period n + 1's VCO steps?  + the full-width tests incl. rtlMult 160 / 192
R=$(pwd); O=$R/gpurun_out/r06_call13; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8; bash profiles/probe/run_ab.sh 2048 8; bash profiles/probe/run_ab.sh 16384 4 ) > $O/msk_unroll_ab.txt 2>&1
cat $O/msk_unroll_ab.txt | cut -c 1-200
for l in base u3 u9; do
  ( ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --config throughput --also none --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 16 --detail-file $O/head_$l.json ) > $O/head_$l.txt 2>&1
  python - $O/head_$l.json $l <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], "headline", d["value"], d["roofline_msk"]["us_per_bit"], d["parity"]["blocks"], d["parity"]["blocks_exact_given_gpu_dm"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_fullsize.txt 2>&1
tail -n 4 $O/pytest_fullsize.txt | cut -c 1-300
