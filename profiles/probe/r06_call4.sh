#!/bin/bash
# round 6, GPU call 4: the whole GPU suite at the current tree; rocprofv3 --stats + PMC FETCH/WRITE passes for the launch shapes that
# are new on the line (m160, m192, share8; split16 again); the power-aware A/B at 2048 channels
R=$(pwd); O=$R/gpurun_out/r06_call4; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 15 $O/pytest_gpu.txt | cut -c 1-400
# ---- power-aware A/B, same process / decoder / buffers: down-converter waves per CU 8 (default) / 6 / 4 / 2 at 2048 channels
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --config shard2048 --also none --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 8 \
    --ab "ACG_FIR_WAVES_PER_WG+ACG_FIR_WG_PER_CU=4:2,2:3,4:1,2:1" --detail-file $O/power_ab_detail.json ) > $O/power_ab_stdout.txt 2> $O/power_ab_stderr.txt
python - $O/power_ab_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("shard2048 value", d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"])
    for k, v in d["ab_same_process"].items():
        print("   ", k, v)
except Exception as e:
    print("no ab:", e)
PY
cd /tmp
args() { case $1 in m160|m192|split16|share8) echo "--config $1" ;; esac; }
for c in m160 m192 split16 share8; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic --sustain 0 --check-channels 16 --also none --config $c --detail-file /tmp/d_$c.json"
  D=$O/x_stats_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
  for p in FETCH_SIZE WRITE_SIZE; do
    D=$O/x_pmc_${c}_$p; n=$( [ $p = FETCH_SIZE ] && echo fetch || echo write )
    timeout 300 rocprofv3 --kernel-trace --pmc $p -d $D -- $B > $O/pmc_line_${c}_$n.json 2> $O/pmc_${c}_$n.err
    db=$(find $D -name "*.db" 2>/dev/null | head -1)
    if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_$n.txt 2>&1; fi
    rm -rf $D
    grep -h "fir_" $O/${c}_$n.txt | grep SIZE | cut -c1-48,64-140
  done
done
