#!/bin/bash
# round 4, GPU call 8: the whole GPU suite, the default bench line, rocprofv3 --stats per case (after the block-repair fix) and of
# the full-width default-kernel pytest.  Every command has its own timeout and no stdin.
R=$(pwd); O=$R/gpurun_out/r04_call8; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
rm -f gpurun_out/multidev_rates.txt
( time timeout 540 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 14 $O/pytest_gpu.txt | cut -c 1-260
( time timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["parity"])
    for k, v in d.get("also", {}).items():
        print(k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("traffic"), v.get("parity_ok"), v.get("hostfed"), v.get("error"))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
args() { case $1 in split16) echo "--format split16 --channels 4096 --decim 160 --blocks 16" ;; *) echo "--config $1" ;; esac; }
for c in throughput wide stress shard2048 cs16 f32 split16; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none $(args $c)"
  D=$O/x_stats_$c
  timeout 200 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -5
done
D=$O/x_fullsize
timeout 400 rocprofv3 --kernel-trace --stats -d $D -- python -m pytest $R/tests/test_gpu_fullsize.py -q -k default_kernels -p no:cacheprovider > $O/pytest_fullsize_under_rocprof.txt 2> $O/fullsize.err
db=$(find $D -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/pytest_fullsize_stats.txt 2>&1; fi
rm -rf $D
tail -n 3 $O/pytest_fullsize_under_rocprof.txt; grep -h "fir_\|msk_demod" $O/pytest_fullsize_stats.txt | cut -c1-64,66-150 | head -8
cd $R; cat gpurun_out/multidev_rates.txt
