#!/bin/bash
# round 6, GPU call 38: the power-aware A/B of call 4 once more at 2048 channels, now that the demodulator is the longer stage by a wider
# margin (msk_lean.hip): down-converter waves per CU 8 / 7 / 6 / 5 / 4 (waves per workgroup : workgroups per CU), same process
R=$(pwd); O=$R/gpurun_out/r06_call38; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --also none --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 16 --config shard2048 \
    --ab "ACG_FIR_WAVES_PER_WG+ACG_FIR_WG_PER_CU=4:2,1:7,2:3,1:5,4:1" --detail-file $O/power_ab_detail.json ) > $O/power_ab_stdout.txt 2> $O/power_ab_stderr.txt
python - $O/power_ab_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["value"], d["whole_job_frac_of_hbm"])
    for k, v in d["ab_same_process"].items():
        print("   ", k, v if "tele" not in k else [(t or {}).get("sclk") for t in v] + [(t or {}).get("power_w") for t in v])
except Exception as e:
    print("no ab:", e)
PY
