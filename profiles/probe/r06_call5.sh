#!/bin/bash
# round 6, GPU call 5: the two-tiles-in-flight variant of the matrix-pipe kernel (one wave per SIMD, so that a demodulator wave
# fits the same SIMD's registers) against the one-tile variant, same process; its stats + PMC passes; the round-6 tests again
R=$(pwd); O=$R/gpurun_out/r06_call5; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_bench.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -n 8 $O/pytest.txt | cut -c 1-400
for cfg in "share8" "share8 --decim 160" ; do
  tag=$(echo $cfg | tr -d ' -'); 
  ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --config $cfg --also none --no-cpu-baseline --no-live-traffic \
      --ab "ACG_FIR_MM_STAGES=2,1" --detail-file $O/${tag}_detail.json ) > $O/${tag}_stdout.txt 2> $O/${tag}_stderr.txt
  tail -n 2 $O/${tag}_stderr.txt | cut -c 1-300
  python - $O/${tag}_detail.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d["config"]["case"], d["config"]["decim"], "value", d["value"], "job", d["whole_job_frac_of_hbm"], "fir", d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"].get("valu_equivalent", {}).get("frac"), "parity", d["parity"]["end_to_end"])
    for k, v in d["ab_same_process"].items():
        print("   ", k, v)
except Exception as e:
    print("no ab:", e)
PY
done
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --share 16 --channels 16384 --blocks 16 --no-cpu-baseline --no-live-traffic --detail-file $O/share16_detail.json ) > $O/share16_stdout.txt 2> $O/share16_stderr.txt
tail -n 1 $O/share16_stdout.txt | cut -c 1-300
cd /tmp
c=share8
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
