#!/bin/bash
O=gpurun_out/r02q
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02q/bench_line.json") if l.startswith("{")][-1])
def show(n,x): print(n, x["value"], x["ms_per_step"], "fir frac", x["roofline"]["frac"], "whole", x["whole_job_frac_of_hbm"], x["time_dominant_kernel"], x["kernels"]["fir_ms_per_step"], x["kernels"]["msk_ms_per_step"], x["parity"]["bit_exact"])
show("head", d)
for k,v in d.get("also",{}).items(): show(k, v)
PY
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
