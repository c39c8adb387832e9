#!/bin/bash
# round 4, GPU call 2: the GPU suite, the default bench line, the demodulator A/B builds.  Every command has its own timeout
# and no stdin (call 1 lost 22 GPU-minutes to a `head` that waited on stdin).
R=$(pwd); O=$R/gpurun_out/r04_call2; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 25 $O/pytest_gpu.txt
( time timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt
cp bench_detail.json $O/ 2>/dev/null
timeout 240 bash profiles/probe/run_ab.sh 1024 8 > $O/msk_ab.txt 2>&1
cat $O/msk_ab.txt
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["roofline"]["frac"], d["parity"])
for k, v in d.get("also", {}).items():
    print(k, v["value"], v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("hostfed"))
print(d.get("cpu_baseline"))
PY
