#!/bin/bash
# staged write bursts (ACG_FIR_VARIANT=8): parity, placement sensitivity alone, sustained whole job against 5 and 7
O=gpurun_out/r02st
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python profiles/probe/fir_variant_check.py 8 96 2 200 2>&1 | tail -2
timeout 300 python profiles/probe/fir_variant_check.py 8 96 2 192 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "scrambled or variants_all_match_oracle and 8" 2>&1 | tail -3
timeout 300 python profiles/probe/placement_probe.py 5 8 2>&1 | grep -v "amdgpu.ids" > $O/placement.txt; grep "variants" $O/placement.txt
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 40 --warmup 5 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02st/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02st/%s.err" % l).read()[-300:])
PY
}
for r in a b; do
  for c in stress wide; do
    for v in 5 7 8; do run ${c}_${v}_$r ACG_FIR_VARIANT=$v -- --config $c; done
  done
done
for v in 5 8; do run head_${v} ACG_FIR_VARIANT=$v -- --config throughput --steps 20; done
