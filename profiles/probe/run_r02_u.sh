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
run u_stress_default X=1 -- --config stress
run u_stress_mskprio ACG_MSK_PRIO=1 -- --config stress
run u_wide_default X=1 -- --config wide
run u_wide_mskprio ACG_MSK_PRIO=1 -- --config wide
run u_stress_pipe2 ACG_PIPE_BLOCKS=2 -- --config stress
run u_stress_pipe8 ACG_PIPE_BLOCKS=8 -- --config stress
run u_8192 X=1 -- --channels 8192 --blocks 16
run u_8192_lpc4 ACG_MSK_LPC=4 -- --channels 8192 --blocks 16
run u_8192_mskprio ACG_MSK_PRIO=1 -- --channels 8192 --blocks 16
