#!/bin/bash
O=gpurun_out/r02i
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "soapy or sdrplay or air or cs16 or split or f32 or fmt or format or feed or samples" > $O/pytest_fmt.log 2>&1; tail -4 $O/pytest_fmt.log
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --also none --steps 10 --warmup 2 --check-channels 16 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open("gpurun_out/r02i/%s.json" % l) if x.startswith("{")][-1])
    print("%-28s value %9.0f ms/step %8.3f fir_frac %.3f whole %.3f fir_ms %.3f msk_ms %.3f %s parity %s" % (l, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["kernels"]["fir_ms_per_step"], d["kernels"]["msk_ms_per_step"], d["roofline"]["kernel"], d["parity"]["dm_within_1e5_rel"]))
except Exception as e:
    print(l, "FAILED", e, open("gpurun_out/r02i/%s.err" % l).read()[-400:])
PY
}
for f in cs16 f32 split16; do
  dec=""; [ $f = split16 ] && dec="--decim 160"
  run ${f}_4096_new -- --format $f --channels 4096 --blocks 16 $dec
  run ${f}_4096_old ACG_FIR_VARIANT=3 -- --format $f --channels 4096 --blocks 16 $dec
done
run f32_800_new -- --format f32 --channels 1024 --blocks 16 --decim 800
run f32_800_old ACG_FIR_VARIANT=3 -- --format f32 --channels 1024 --blocks 16 --decim 800
run f32_480_new -- --format f32 --channels 2048 --blocks 16 --decim 480
run cs16_1024_new -- --format cs16 --channels 1024 --blocks 32
