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
run x_head_cb8 X=1 -- --config throughput
run x_head_cb8_novec ACG_MSK_NOVEC=1 -- --config throughput
run x_head_cb16 X=1 -- --config throughput --call-blocks 16
run x_head_cb36 X=1 -- --config throughput --call-blocks 36
run x_head_cb72 X=1 -- --config throughput --call-blocks 72
run x_stress_a X=1 -- --config stress
run x_stress_b X=1 -- --config stress
run x_wide_a X=1 -- --config wide
run x_wide_b X=1 -- --config wide
timeout 300 python profiles/probe/msk_only.py 1024 8 2>&1 | tail -1
timeout 300 python profiles/probe/msk_only.py 1024 36 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
