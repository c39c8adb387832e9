#!/bin/bash
# round 6, GPU call 3: the bench with its eleven "also" cases (m160 / m192 / split16 / share8 new, the HBM-saturating cases at >= 20 s),
# the bench tests, the round-6 kernel tests
R=$(pwd); O=$R/gpurun_out/r06_call3; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_round6.txt 2>&1
tail -n 6 $O/pytest_round6.txt | cut -c 1-300
( time timeout 1200 python -m pytest tests/test_gpu_bench.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_bench.txt 2>&1
tail -n 25 $O/pytest_bench.txt | cut -c 1-600
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 5 $O/bench_stderr.txt | cut -c 1-600
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"), d["roofline"].get("traffic_src"), d["parity"])
    print("msk", d.get("roofline_msk"))
    for k, v in d.get("also", {}).items():
        print("  ", k, json.dumps(v))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
