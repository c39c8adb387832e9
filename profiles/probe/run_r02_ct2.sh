#!/bin/bash
# occupancy sweep: waves per CU for the default kernel (5) and the register-resident taps (7), alone and sustained
O=gpurun_out/r02ct2
mkdir -p $O
export TMPDIR=/tmp
S=""
for w in 4 5 6 7 8 10 12; do S="$S 16384:200:4:200:7:$w"; done
for w in 5 6 7 8; do S="$S 16384:200:4:200:5:$w"; done
for w in 4 6 8 10; do S="$S 1024:200:144:200:7:$w"; done
for w in 6 8; do S="$S 1024:200:144:200:5:$w"; done
ACG_FIR_WAVES_PER_WG=1 timeout 600 python profiles/probe/fir_only_sweep.py $S > $O/fir_only_waves.txt 2>&1; grep fir_only $O/fir_only_waves.txt | cut -c1-150
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 40 --warmup 5 --check-channels 8 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02ct2/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02ct2/%s.err" % l).read()[-300:])
PY
}
for w in 5 6 7 8; do run stress_7_w$w ACG_FIR_VARIANT=7 ACG_FIR_WG_PER_CU=$w -- --config stress; done
for w in 5 6 7; do run stress_5_w$w ACG_FIR_VARIANT=5 ACG_FIR_WG_PER_CU=$w -- --config stress; done
for w in 5 6 7 8; do run wide_7_w$w ACG_FIR_VARIANT=7 ACG_FIR_WG_PER_CU=$w -- --config wide; done
for w in 6 7; do run wide_5_w$w ACG_FIR_VARIANT=5 ACG_FIR_WG_PER_CU=$w -- --config wide; done
run head_7_w2 ACG_FIR_VARIANT=7 ACG_FIR_WG_PER_CU=2 -- --config throughput --steps 20
run head_5 ACG_FIR_VARIANT=5 -- --config throughput --steps 20
