#!/bin/bash
# register-resident taps (ACG_FIR_VARIANT=7) against the default kernel (5): parity, the launch alone, sustained whole job;
# plus the per-instruction cost table of one wave (valu_rate_probe).
O=gpurun_out/r02ct
mkdir -p $O
export TMPDIR=/tmp
timeout 60 ./profiles/probe/valu_rate_probe > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
timeout 300 python profiles/probe/fir_variant_check.py 7 96 2 200 2>&1 | tail -4
timeout 300 python profiles/probe/fir_variant_check.py 7 96 2 192 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "scrambled or variants_all_match_oracle and 7" 2>&1 | tail -3
timeout 600 python profiles/probe/fir_only_sweep.py 16384:200:4:200:5 16384:200:4:200:7 16384:200:4:200:7:2 16384:200:4:200:5 16384:200:4:200:7 \
   4096:200:4:192:5 4096:200:4:192:7 1024:200:8:200:5 1024:200:8:200:7 1024:200:144:200:5 1024:200:144:200:7 1024:200:144:200:7:2 > $O/fir_only.txt 2>&1; cat $O/fir_only.txt
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 40 --warmup 5 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02ct/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02ct/%s.err" % l).read()[-300:])
PY
}
run ct_stress_5 ACG_FIR_VARIANT=5 -- --config stress
run ct_stress_7 ACG_FIR_VARIANT=7 -- --config stress
run ct_stress_5b ACG_FIR_VARIANT=5 -- --config stress
run ct_stress_7b ACG_FIR_VARIANT=7 -- --config stress
run ct_stress_7w8 ACG_FIR_VARIANT=7 ACG_FIR_WG_PER_CU=7 -- --config stress
run ct_wide_5 ACG_FIR_VARIANT=5 -- --config wide
run ct_wide_7 ACG_FIR_VARIANT=7 -- --config wide
run ct_head_5 ACG_FIR_VARIANT=5 -- --config throughput --steps 20
run ct_head_7 ACG_FIR_VARIANT=7 -- --config throughput --steps 20
