#!/bin/bash
# round 6, GPU call 2: first run of the matrix-pipe shared-stream kernel (fir_mm.hip): its tests, the shared-stream tests, share8 line + stats
R=$(pwd); O=$R/gpurun_out/r06_call2; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_round6.py "tests/test_gpu_parity.py::test_fir_shared_stream_and_stream_map" "tests/test_gpu_parity.py::test_fir_shared_stream_kernel_ragged_groups" "tests/test_gpu_parity.py::test_config2_shared_stream_8ch" -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -n 30 $O/pytest.txt | cut -c 1-400
for sh in 8 16; do
( time timeout 400 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --detail-file $O/share${sh}_detail.json --share $sh --channels 16384 --blocks 16 ) > $O/share${sh}_stdout.txt 2> $O/share${sh}_stderr.txt
tail -n 1 $O/share${sh}_stdout.txt > $O/share${sh}_line.json; tail -n 3 $O/share${sh}_stderr.txt | cut -c 1-300
done
cd /tmp
D=$O/x_stats
timeout 300 rocprofv3 --kernel-trace --stats -d $D -- python $R/bench.py --gpus 1 --steps 4 --warmup 1 --sustain 0 --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 8 --share 8 --channels 16384 --blocks 16 --detail-file /tmp/d.json > $O/prof_stdout.txt 2>&1
db=$(find $D -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/share8_stats.txt 2>&1; fi
rm -rf $D
head -8 $O/share8_stats.txt | cut -c 1-200
cd $R
python - $O <<'PY'
import json, sys, os
for n in ("share8", "share16"):
    try:
        d = json.load(open(os.path.join(sys.argv[1], n + "_line.json")))
        print(n, d["value"], d["ms_per_step"], d["roofline"].get("frac"), d.get("whole_job_frac_of_hbm"), d["roofline"].get("kernel"), d.get("parity"))
    except Exception as e:
        print(n, "no line:", e)
PY
