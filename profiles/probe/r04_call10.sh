#!/bin/bash
# round 4, GPU call 10: the repair pass at one wave per 8 channels, table in LDS -- the tests that touch it, the default bench line (+ one with --collect-lag 1), stats
R=$(pwd); O=$R/gpurun_out/r04_call10; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 300 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "message or repair or msgs or bench_line_contract or soapy or collect or compat" ) > $O/pytest_subset.txt 2>&1
tail -n 5 $O/pytest_subset.txt | cut -c 1-260
( time timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.load(open(p))
        print(p.split("/")[-1], "headline", d["value"], d["roofline"]["frac"], d["config"].get("collect_lag"), d["parity"])
        for k, v in d.get("also", {}).items():
            print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("hostfed"), v.get("error"))
    except Exception as e:
        print("no bench line:", p, e)
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
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
done
