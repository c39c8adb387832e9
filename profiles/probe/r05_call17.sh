#!/bin/bash
# round 5, GPU call 17: roofline.traffic of the headline shape measured live (two rocprofv3 --pmc child runs inside bench.py) -- the bench tests and the default line
R=$(pwd); O=$R/gpurun_out/r05_call17; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider ) > $O/pytest_bench.txt 2>&1
tail -n 4 $O/pytest_bench.txt | cut -c 1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 3 $O/bench_stderr.txt | cut -c 1-300
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"), d["parity"])
    for k, v in d.get("also", {}).items():
        print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("traffic"), v.get("parity_ok"), v.get("gpu_vs_ref_ofast"), v.get("ch8"), v.get("ch16"))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
