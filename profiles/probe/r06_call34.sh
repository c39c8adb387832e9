#!/bin/bash
# round 6, GPU call 34: smoke(), the whole GPU suite and the default bench line at the final tree (4 and 8 lanes per channel on msk_lean.hip);
# rocprofv3 stats of the cases whose demodulator kernel changed with it (share8, wide)
R=$(pwd); O=$R/gpurun_out/r06_call34; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -n 3 $O/smoke.txt | cut -c 1-300
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 6 $O/pytest_gpu.txt | cut -c 1-400
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 4 $O/bench_stderr.txt | cut -c 1-400
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["roofline"].get("traffic"), d["roofline"].get("traffic_src"), d["parity"])
    print("msk", d.get("roofline_msk"))
    for k, v in d.get("also", {}).items():
        print("  ", k, json.dumps(v))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
for c in share8 wide; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic --sustain 0 --check-channels 16 --also none --config $c --detail-file /tmp/d_$c.json"
  D=$O/x_stats_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  echo "== $c"; grep -h "fir_\|msk_demod\|msk_lean\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
done
