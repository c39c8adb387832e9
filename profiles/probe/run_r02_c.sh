#!/bin/bash
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02c/bench_line.json") if l.startswith("{")][-1])
def show(n,x): print(n, x["value"], x["ms_per_step"], "fir frac", x["roofline"]["frac"], "whole", x["whole_job_frac_of_hbm"], x["time_dominant_kernel"], x["kernels"], x["parity"])
show("head", d)
for k,v in d.get("also",{}).items(): show(k, v)
print(d["roofline"].get("pure_reader_GBs_measured_this_run"), d.get("cpu_baseline",{}).get("value"))
PY
tail -5 $O/bench_err.txt
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
