#!/bin/bash
# round 4, GPU call 3: the tests that failed in call 2, the default bench line, then the profiling passes (r04_call3.sh)
R=$(pwd); O=$R/gpurun_out/r04_call3; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 420 python -m pytest tests/test_gpu_bench.py tests/test_gpu_multidev.py "tests/test_gpu_parity.py::test_message_drain_keeps_what_does_not_fit_and_records_are_fully_defined" -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_rerun.txt 2>&1
tail -n 12 $O/pytest_rerun.txt | cut -c 1-300
( time timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["parity"])
    for k, v in d.get("also", {}).items():
        print(k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("hostfed"), v.get("error"))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
cat gpurun_out/multidev_rates.txt 2>/dev/null | tail -4
bash profiles/probe/r04_call3.sh
