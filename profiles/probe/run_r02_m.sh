#!/bin/bash
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
timeout 100 python profiles/probe/fir_direct_debug.py 2>&1 | grep -v amdgpu.ids | head -3
for w in 1 2 4; do ACG_FIR_WAVES_PER_WG=$w timeout 300 python profiles/probe/fir_only_sweep.py 16384:200:4:200:5 1024:200:8:200:5 4096:200:4:192:5 2>&1 | grep fir_only | sed "s/^/wpg=$w /"; done
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
for w in 1 2 4; do
  run w${w}_wide ACG_FIR_WAVES_PER_WG=$w -- --config wide
  run w${w}_stress ACG_FIR_WAVES_PER_WG=$w -- --config stress
  run w${w}_head ACG_FIR_WAVES_PER_WG=$w -- --config throughput
done
