#!/bin/bash
# round 5, GPU call 19: the repair pass's waves at the demodulator's priority for their short life (-DACG_BLK_AB_PRIO): same-box A/B
R=$(pwd); O=$R/gpurun_out/r05_call19; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f" % (sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for c in throughput shard2048; do for l in base blkprio base blkprio; do
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --no-live-traffic --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/${c}_$l.json 2> $O/${c}_$l.err; pt $O/${c}_$l.json ${c}_$l
done; done | tee $O/bench_ab.txt
cd /tmp
for c in throughput shard2048; do for l in base blkprio; do
  D=$O/x_stats_${c}_$l
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 200 rocprofv3 --kernel-trace --stats -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c > $O/line_${c}_$l.json 2> $O/stats_${c}_$l.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_${l}_stats.txt 2>&1; fi
  rm -rf $D
  echo "$c $l"; grep -h "msk_demod\|blk_repair" $O/${c}_${l}_stats.txt | cut -c1-64,66-150 | head -3
done; done | tee $O/stats_ab.txt
