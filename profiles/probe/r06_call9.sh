#!/bin/bash
# round 6, GPU call 9: the north-star-width group-independence test of the matrix-pipe kernel, and the default bench line once more
# on another box (run-to-run / box-to-box spread of the final tree)
R=$(pwd); O=$R/gpurun_out/r06_call9; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_fullsize.txt 2>&1
tail -n 5 $O/pytest_fullsize.txt | cut -c 1-400
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 3 $O/bench_stderr.txt | cut -c 1-400
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"), d["parity"]["gpu_vs_ref_ofast"])
    for k, v in d.get("also", {}).items():
        print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("gpu_vs_ref_ofast"), v.get("b5"), v.get("sclk"), v.get("ch8"), v.get("ch16"))
except Exception as e:
    print("no bench line:", e)
PY
