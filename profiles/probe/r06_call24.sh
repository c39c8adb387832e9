#!/bin/bash
# round 6, GPU call 24 (evidence with msk_lean.hip as the demodulator of the 8-lane launches): the whole GPU suite; the default bench line exactly as the driver invokes it; rocprofv3 --kernel-trace
# --stats per case; PMC FETCH / WRITE passes for the headline; RCCL with a world of one
R=$(pwd); O=$R/gpurun_out/r06_call24; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
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
    print("msk", d.get("roofline_msk"), "traffic_src", d.get("traffic_src"))
    for k, v in d.get("also", {}).items():
        print("  ", k, json.dumps(v))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
for c in throughput shard2048 share8 stress m160; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic --sustain 0 --check-channels 16 --also none --config $c --detail-file /tmp/d_$c.json"
  D=$O/x_stats_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  echo "== $c"; grep -h "fir_\|msk_demod\|msk_lean\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
done
for p in FETCH_SIZE WRITE_SIZE; do
  c=throughput; D=$O/x_pmc_${c}_$p; n=$( [ $p = FETCH_SIZE ] && echo fetch || echo write )
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic --sustain 0 --check-channels 16 --also none --config $c --detail-file /tmp/d_$c.json"
  timeout 300 rocprofv3 --kernel-trace --pmc $p -d $D -- $B > $O/pmc_line_${c}_$n.json 2> $O/pmc_${c}_$n.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_$n.txt 2>&1; fi
  rm -rf $D
  grep -h "fir_u8" $O/${c}_$n.txt | grep SIZE | cut -c1-48,64-140
done
cd $R
( time timeout 300 python bench.py --gpus 1 --rccl-selftest --config shard2048 --also none --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic ) > $O/rccl_selftest.txt 2>&1
tail -n 1 $O/rccl_selftest.txt | cut -c 1-300
