#!/bin/bash
# round 5, GPU call 7: counted inner loop in the product -- all GPU tests, the default bench line, demodulator alone
R=$(pwd); O=$R/gpurun_out/r05_call7; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
tail -n 8 $O/pytest_gpu.txt | cut -c 1-300
for ch in 1024 2048; do timeout 100 python profiles/probe/msk_only.py $ch 8 2>&1 | tail -1; done | tee $O/msk_only.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt | cut -c 1-300
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["parity"])
    for k, v in d.get("also", {}).items():
        print("  ", k, json.dumps(v)[:330])
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
