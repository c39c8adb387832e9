#!/bin/bash
# round 4, GPU call 6: after the small-footprint block kernels -- the tests that touch them, the default bench line, stats per case
R=$(pwd); O=$R/gpurun_out/r04_call6; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 400 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "message or repair or msgs or bench_line_contract or multidev or soapy or host_fed or compat" ) > $O/pytest_subset.txt 2>&1
tail -n 8 $O/pytest_subset.txt | cut -c 1-260
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
