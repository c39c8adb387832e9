#!/bin/bash
# sustained whole job: waves per CU x staging depth for the register-resident taps (7, 70..73) against the default (5), two rounds
O=gpurun_out/r02ct3
mkdir -p $O
export TMPDIR=/tmp
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 40 --warmup 5 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02ct3/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02ct3/%s.err" % l).read()[-300:])
PY
}
for r in a b; do
  for c in stress wide; do
    run ${c}_5_w7_$r ACG_FIR_VARIANT=5 -- --config $c
    for w in 3 4 5 6 7; do run ${c}_7_w${w}_$r ACG_FIR_VARIANT=7 ACG_FIR_WG_PER_CU=$w -- --config $c; done
    for w in 2 3 4; do run ${c}_71_w${w}_$r ACG_FIR_VARIANT=71 ACG_FIR_WG_PER_CU=$w -- --config $c; done
    run ${c}_72_w3_$r ACG_FIR_VARIANT=72 ACG_FIR_WG_PER_CU=3 -- --config $c
    run ${c}_73_w10_$r ACG_FIR_VARIANT=73 ACG_FIR_WG_PER_CU=10 -- --config $c
  done
done
