#!/bin/bash
# round 4, GPU call 11: the whole GPU suite and the default bench line at the final commit
R=$(pwd); O=$R/gpurun_out/r04_call11; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
rm -f gpurun_out/multidev_rates.txt
( time timeout 540 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 8 $O/pytest_gpu.txt | cut -c 1-260
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
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cat gpurun_out/multidev_rates.txt
