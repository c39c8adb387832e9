#!/bin/bash
# wide / stress with a CU partition (demodulator on a few CUs with fewer lanes per channel) against the shared-CU default, same box
O=$1; mkdir -p $O
run() { # label env...
  l=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ref-leg --also none --steps 20 --warmup 3 --sustain 2 --check-channels 8 --placements 1 --config $CASE > $O/$l.json 2>/dev/null
  python - $O/$l.json $l <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s value %9.0f fir %.3f whole %.3f  %s" % (sys.argv[2], d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], {k: round(v, 2) for k, v in d["kernels"].items() if k != "note"}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for CASE in wide stress; do
  run ${CASE}_default_a A=1
  run ${CASE}_cus32_lpc1 ACG_MSK_CUS=32 ACG_MSK_LPC=1
  run ${CASE}_cus64_lpc2 ACG_MSK_CUS=64 ACG_MSK_LPC=2
  run ${CASE}_cus64_lpc1 ACG_MSK_CUS=64 ACG_MSK_LPC=1
  run ${CASE}_cus96_lpc4 ACG_MSK_CUS=96 ACG_MSK_LPC=4
  run ${CASE}_default_b A=1
done
