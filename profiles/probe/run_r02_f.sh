#!/bin/bash
O=gpurun_out/r02f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "msk or wav or golden or soak or lane or compat or stream or refill or reciprocal" > $O/pytest_msk.log 2>&1; tail -3 $O/pytest_msk.log
timeout 300 python profiles/probe/msk_only.py 1024 8 2>&1 | grep -v amdgpu.ids | tee $O/msk_only.txt
timeout 300 python profiles/probe/msk_phase_stamps.py 1024 8 2>&1 | grep -v amdgpu.ids | tee $O/msk_stamps.txt
timeout 600 python bench.py --no-cpu-baseline --also none --steps 20 --warmup 3 --check-channels 16 > $O/bench_head.json 2> $O/bench_head.err; python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r02f/bench_head.json") if x.startswith("{")][-1])
print("head value %.0f ms/step %.3f fir_frac %.3f whole %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"]), d["kernels"], d["parity"])
PY
tail -3 $O/bench_head.err
