#!/bin/bash
# round 5, GPU call 8 (evidence): the default bench line exactly as the driver invokes it; rocprofv3 --kernel-trace --stats per
# case; separate --pmc FETCH_SIZE / WRITE_SIZE passes for the launch shapes that are new (2048 channels x 8 callbacks, host-fed
# 10 000 channels x 2 callbacks) and for the headline; 8-GPU readiness on one GPU (RCCL world of one, eight contexts on device 0)
R=$(pwd); O=$R/gpurun_out/r05_call8; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 3 $O/bench_stderr.txt | cut -c 1-300
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"])
    for k, v in d.get("also", {}).items():
        print("  ", k, v.get("value"), v.get("whole_job_frac"), v.get("roofline_frac"), v.get("parity_ok"), v.get("ch8"), v.get("ch16"))
except Exception as e:
    print("no bench line:", e)
PY
cd /tmp
args() { case $1 in split16) echo "--format split16 --channels 4096 --decim 160 --blocks 16" ;; *) echo "--config $1" ;; esac; }
for c in throughput shard2048 stress wide cs16 f32; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none $(args $c)"
  D=$O/x_stats_$c
  timeout 200 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/stats_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; else echo "no db" > $O/${c}_stats.txt; fi
  rm -rf $D
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/${c}_stats.txt | cut -c1-64,66-150 | head -4
done
for c in throughput shard2048; do
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c"
  for p in FETCH_SIZE WRITE_SIZE; do
    D=$O/x_pmc_${c}_$p; n=$( [ $p = FETCH_SIZE ] && echo fetch || echo write )
    timeout 300 rocprofv3 --kernel-trace --pmc $p -d $D -- $B > $O/pmc_line_${c}_$n.json 2> $O/pmc_${c}_$n.err
    db=$(find $D -name "*.db" 2>/dev/null | head -1)
    if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/${c}_$n.txt 2>&1; fi
    rm -rf $D
    grep -h "fir_u8" $O/${c}_$n.txt | grep SIZE | cut -c1-40,64-130
  done
done
# the host-fed shape: the child process of the hostfed case under the counters
for p in FETCH_SIZE WRITE_SIZE; do
  D=$O/x_pmc_hostfed_$p; n=$( [ $p = FETCH_SIZE ] && echo fetch || echo write )
  timeout 300 rocprofv3 --kernel-trace --pmc $p -d $D -- python $R/bench.py --hostfed-child --steps 6 --warmup 2 --sustain 0 --check-channels 16 > $O/pmc_line_hostfed_$n.txt 2> $O/pmc_hostfed_$n.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/hostfed_$n.txt 2>&1; fi
  rm -rf $D
  grep -h "fir_u8" $O/hostfed_$n.txt | grep SIZE | cut -c1-40,64-130
done
cd $R
# 8-GPU readiness without an 8-GPU node: the collectives with a world of one, and eight contexts (the per-GPU shards of BASELINE
# configs[3]: 8 x 2048 channels) driven by the C host on device 0
( time timeout 300 python bench.py --gpus 1 --rccl-selftest --config shard2048 --also none --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg ) > $O/rccl_selftest.txt 2>&1
tail -n 1 $O/rccl_selftest.txt | cut -c 1-400
( time GPU_MAX_HW_QUEUES=16 timeout 600 acarsdec_amd/lib/host_multidev random rtl 16384 200 8 8 8 --msgs --time 20 ) > $O/multidev8_stdout.txt 2> $O/multidev8.txt
wc -l $O/multidev8_stdout.txt; tail -n 14 $O/multidev8.txt; rm -f $O/multidev8_stdout.txt
