#!/bin/bash
# round 6, GPU call 21: the default bench line with msk_lean.hip as the demodulator of the 8-lanes-per-channel launches
R=$(pwd); O=$R/gpurun_out/r06_call21; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 3 $O/bench_stderr.txt | cut -c 1-300
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"), d["parity"]["gpu_vs_ref_ofast"], d["roofline_msk"]["us_per_bit"])
    for k, v in d.get("also", {}).items():
        print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("gpu_vs_ref_ofast"), v.get("b5"), v.get("sclk"))
except Exception as e:
    print("no bench line:", e)
PY
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
( timeout 300 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean.txt 2>&1
tail -n 3 $O/pytest_lean.txt | cut -c 1-300
