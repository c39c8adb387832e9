#!/bin/bash
# round 6, GPU call 7: fir_u8_mm1_kernel where the shader clock matters (2048 channels: the demodulator sets the step and follows the
# clock the down-converter's power leaves it), and its waves per CU at 4096 channels; the round-6 tests again
R=$(pwd); O=$R/gpurun_out/r06_call7; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -n 5 $O/pytest.txt | cut -c 1-400
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
ab shard2048 "ACG_FIR_MM1=0,1" --config shard2048
ab stress "ACG_FIR_MM1+ACG_FIR_MM1_WAVES=0:7,1:7,1:8,1:6,1:5" --config stress
ab ch8192 "ACG_FIR_MM1=0,1" --config throughput --channels 8192 --blocks 16
