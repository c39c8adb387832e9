#!/bin/bash
# round 5, GPU call 12: the repair pass's stream on the down-converter's side of the CU partition (<= 2048 channels): same-box
# A/B against no mask, bench + stats
R=$(pwd); O=$R/gpurun_out/r05_call12; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
export ACG_ALLOW_TUNING=1
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f" % (sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for c in throughput shard2048; do for m in 1 0 1 0; do
  ACG_POST_MASK=$m timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/${c}_mask$m.json 2> $O/${c}_mask$m.err; pt $O/${c}_mask$m.json ${c}_mask$m
done; done | tee $O/bench_ab.txt
cd /tmp
for c in throughput shard2048; do for m in 1 0; do
  D=$O/x_stats_${c}_$m
  ACG_POST_MASK=$m timeout 200 rocprofv3 --kernel-trace --stats -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c > $O/line_${c}_$m.json 2> $O/stats_${c}_$m.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_mask${m}_stats.txt 2>&1; fi
  rm -rf $D
  echo "$c mask=$m"; grep -h "fir_u8_direct\|msk_demod\|blk_repair\|msg_split" $O/${c}_mask${m}_stats.txt | cut -c1-64,66-150 | head -4
done; done
cd $R
unset ACG_ALLOW_TUNING
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "repair or msgs or message or ring_across or streaming_past or collect or bench_line_contract" ) > $O/pytest_subset.txt 2>&1
tail -n 3 $O/pytest_subset.txt | cut -c 1-300
