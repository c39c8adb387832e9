#!/bin/bash
# round 5, GPU call 11 (final code): the repair pass's fixed round trips overlapped -- all GPU tests, the default bench line, stats
# of the three cases the pass weighs most in
R=$(pwd); O=$R/gpurun_out/r05_call11; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
tail -n 6 $O/pytest_gpu.txt | cut -c 1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 3 $O/bench_stderr.txt | cut -c 1-300
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"))
    for k, v in d.get("also", {}).items():
        print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("traffic"), v.get("parity_ok"), v.get("gpu_vs_ref_ofast"), v.get("ch8"), v.get("ch16"))
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
for c in throughput shard2048 stress wide; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c"
  D=$O/x_stats_$c
  timeout 200 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
done
