#!/bin/bash
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 20 --warmup 3 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
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
run v_4096_m0 ACG_MSK_PRIO=0 -- --config stress
run v_4096_m1 ACG_MSK_PRIO=1 -- --config stress
run v_4096_m1f1 ACG_MSK_PRIO=1 ACG_FIR_PRIO=1 -- --config stress
run v_8192_m0f1 ACG_MSK_PRIO=0 -- --channels 8192 --blocks 16
run v_8192_m1f1 ACG_MSK_PRIO=1 -- --channels 8192 --blocks 16
run v_8192_m1f0 ACG_MSK_PRIO=1 ACG_FIR_PRIO=0 -- --channels 8192 --blocks 16
run v_8192_m0f0 ACG_MSK_PRIO=0 ACG_FIR_PRIO=0 -- --channels 8192 --blocks 16
run v_16384_m0f1 ACG_MSK_PRIO=0 -- --config wide
run v_16384_m1f1 ACG_MSK_PRIO=1 -- --config wide
run v_16384_m1f0 ACG_MSK_PRIO=1 ACG_FIR_PRIO=0 -- --config wide
run v_32768_m0f1 ACG_MSK_PRIO=0 -- --channels 32768 --blocks 4
run v_32768_m1f1 ACG_MSK_PRIO=1 -- --channels 32768 --blocks 4
run v_2048_m1 ACG_MSK_PRIO=1 -- --config shard2048
run v_3072_m0 ACG_MSK_PRIO=0 -- --channels 3072 --blocks 32
run v_3072_m1 ACG_MSK_PRIO=1 -- --channels 3072 --blocks 32
