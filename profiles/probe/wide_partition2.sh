#!/bin/bash
O=$1; mkdir -p $O
run() { l=$1; shift
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
CASE=wide
for r in 1 2 3; do
  run wide_default_$r A=1
  run wide_cus48_lpc1_$r ACG_MSK_CUS=48 ACG_MSK_LPC=1
  run wide_cus32_lpc1_$r ACG_MSK_CUS=32 ACG_MSK_LPC=1
  run wide_cus64_lpc2_$r ACG_MSK_CUS=64 ACG_MSK_LPC=2
done
CASE=stress
for r in 1 2; do
  run stress_default_$r A=1
  run stress_cus128_lpc8_$r ACG_MSK_CUS=128 ACG_MSK_LPC=8
  run stress_cus96_lpc8_$r ACG_MSK_CUS=96 ACG_MSK_LPC=8
  run stress_cus80_lpc4_$r ACG_MSK_CUS=80 ACG_MSK_LPC=4
  run stress_cus64_lpc2_$r ACG_MSK_CUS=64 ACG_MSK_LPC=2
done
