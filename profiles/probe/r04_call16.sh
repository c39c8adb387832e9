#!/bin/bash
# round 4, GPU call 16: the default bench line with GPU_MAX_HW_QUEUES=16 set by bench.py itself
R=$(pwd); O=$R/gpurun_out/r04_call16; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d["value"], d["roofline"]["frac"], d["parity"])
for k, v in d.get("also", {}).items():
    print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("hostfed"), v.get("error"))
PY
