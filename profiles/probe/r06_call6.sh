#!/bin/bash
# round 6, GPU call 6: the one-stream-per-channel down-converter on the matrix pipe (fir_u8_mm1_kernel) -- its tests, then same-process
# A/B against the wave-private vector kernel on the headline, 2048 channels, wide, stress and rtlMult 160
R=$(pwd); O=$R/gpurun_out/r06_call6; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -n 8 $O/pytest.txt | cut -c 1-400
for cfg in throughput shard2048 wide stress m160; do
  ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --config $cfg --also none --no-cpu-baseline --no-live-traffic --no-ref-leg \
      --ab "ACG_FIR_MM1=0,1" --detail-file $O/${cfg}_detail.json ) > $O/${cfg}_stdout.txt 2> $O/${cfg}_stderr.txt
  tail -n 2 $O/${cfg}_stderr.txt | cut -c 1-300
  python - $O/${cfg}_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["config"]["case"], "value", d["value"], "job", d["whole_job_frac_of_hbm"], "fir", d["roofline"]["frac"])
    for k, v in d["ab_same_process"].items():
        print("   ", k, v)
except Exception as e:
    print("no ab:", e)
PY
done
cd /tmp
for c in wide throughput; do
D=$O/x_stats_$c
ACG_ALLOW_TUNING=1 ACG_FIR_MM1=1 timeout 300 rocprofv3 --kernel-trace --stats -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --no-live-traffic --sustain 0 --check-channels 16 --also none --config $c --detail-file /tmp/d_$c.json > $O/bench_line_${c}_mm1_under_rocprof.json 2> $O/stats_$c.err
db=$(find $D -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_mm1_stats.txt 2>&1; fi
rm -rf $D
grep -h "fir_\|msk_demod\|blk_repair" $O/${c}_mm1_stats.txt | cut -c1-64,66-150 | head -4
done
