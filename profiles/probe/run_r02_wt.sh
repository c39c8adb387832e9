#!/bin/bash
O=gpurun_out/r02wt
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python profiles/probe/fir_variant_check.py 55 96 2 200 2>&1 | tail -1
timeout 300 python profiles/probe/placement_probe.py 5 8 2>&1 | grep -v "amdgpu.ids" > $O/placement.txt; grep "variants" $O/placement.txt | cut -c1-20,60-200
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 40 --warmup 5 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02wt/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02wt/%s.err" % l).read()[-300:])
PY
}
for r in a b; do
  for c in stress wide; do
    for v in 5 55; do run ${c}_v${v}_$r ACG_FIR_VARIANT=$v -- --config $c; done
  done
  for v in 5 55; do run head_v${v}_$r ACG_FIR_VARIANT=$v -- --config throughput --steps 20; done
done
