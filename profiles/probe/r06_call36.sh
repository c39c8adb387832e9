#!/bin/bash
# round 6, GPU call 36 (the same A/B with msk_lean.hip as the demodulator): fir_u8_mm1_kernel with more waves per CU (13 KiB of LDS and 160 VGPRs each: up to 12 fit where the demodulator
# has CUs of its own) -- does the lighter kernel win where the down-converter is co-critical (2048 channels) once it has the bytes in flight?
R=$(pwd); O=$R/gpurun_out/r06_call36; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
ab() { # tag, ab spec, bench args...
  tag=$1; spec=$2; shift; shift
  ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --also none --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 16 \
      --ab "$spec" --detail-file $O/${tag}_detail.json "$@" ) > $O/${tag}_stdout.txt 2> $O/${tag}_stderr.txt
  tail -n 2 $O/${tag}_stderr.txt | cut -c 1-300
  python - $O/${tag}_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["config"]["case"], d["config"]["channels_per_gpu"], "value", d["value"], "job", d["whole_job_frac_of_hbm"], "fir", d["roofline"]["frac"])
    for k, v in d["ab_same_process"].items():
        print("   ", k, v)
except Exception as e:
    print("no ab:", e)
PY
}
ab shard2048 "ACG_FIR_MM1+ACG_FIR_MM1_WAVES=0:8,1:8,1:10,1:12" --config shard2048
ab throughput "ACG_FIR_MM1+ACG_FIR_MM1_WAVES=0:8,1:12" --config throughput

ab m160 "ACG_FIR_MM1+ACG_FIR_MM1_WAVES=0:7,1:8,1:9,1:10" --config m160
