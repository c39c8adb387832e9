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
run p_wide_auto -- --config wide
run p_wide_p1 ACG_FIR_RUN_PAIRS=1 -- --config wide
run p_wide_p4 ACG_FIR_RUN_PAIRS=4 -- --config wide
run p_stress_auto -- --config stress
run p_stress_p1 ACG_FIR_RUN_PAIRS=1 -- --config stress
run p_stress_p4 ACG_FIR_RUN_PAIRS=4 -- --config stress
run p_head_auto -- --config throughput
run p_head_p2 ACG_FIR_RUN_PAIRS=2 -- --config throughput
run p_head_p4 ACG_FIR_RUN_PAIRS=4 -- --config throughput
run p_shard2048 -- --config shard2048
