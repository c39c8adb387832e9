#!/bin/bash
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 10 --warmup 2 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02d/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02d/%s.err" % l).read()[-300:])
PY
}
E="ACG_FIR_WAVES_PER_WG=1 ACG_FIR_WG_PER_CU=7"
run o_wide_lpc4 $E -- --config wide
run o_wide_lpc2 $E ACG_MSK_LPC=2 -- --config wide
run o_wide_lpc1 $E ACG_MSK_LPC=1 -- --config wide
run o_wide_lpc8 $E ACG_MSK_LPC=8 -- --config wide
run o_wide_lpc2_pipe2 $E ACG_MSK_LPC=2 ACG_PIPE_BLOCKS=2 -- --config wide
run o_wide_lpc2_pipe8 $E ACG_MSK_LPC=2 ACG_PIPE_BLOCKS=8 -- --config wide
run o_stress_w1cu7 $E -- --config stress
run o_stress_w1cu7_lpc4 $E ACG_MSK_LPC=4 -- --config stress
run o_stress_w1cu7_prio $E ACG_FIR_PRIO=1 ACG_MSK_LPC=4 -- --config stress
