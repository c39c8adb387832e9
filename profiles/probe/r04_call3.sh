#!/bin/bash
# round 4, GPU call 3: rocprofv3 --kernel-trace --stats per case, separate --pmc FETCH_SIZE / WRITE_SIZE passes for the launch
# shapes that have no traffic entry yet (and the headline again).  Every command has its own timeout and no stdin.
R=$(pwd); O=$R/gpurun_out/r04_call3; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
cd /tmp
args() { case $1 in
  split16) echo "--format split16 --channels 4096 --decim 160 --blocks 16" ;;
  *) echo "--config $1" ;;
esac; }
run() { # $1 = kind (stats|fetch|write), $2 = case
  local B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none $(args $2)"
  local D=$O/x_$1_$2
  case $1 in
    stats) timeout 200 rocprofv3 --kernel-trace --stats -d $D -- $B > $O/bench_line_$2_under_rocprof.json 2> $O/$1_$2.err ;;
    fetch) timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D -- $B > /dev/null 2> $O/$1_$2.err ;;
    write) timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D -- $B > /dev/null 2> $O/$1_$2.err ;;
  esac
  local db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 60 python $R/profiles/summarize_rocpd.py $db > $O/$2_$1.txt 2>&1; else echo "no db for $1 $2" > $O/$2_$1.txt; tail -3 $O/$1_$2.err; fi
  rm -rf $D
  grep -h "fir_\|msk_demod\|blk_repair\|msg_split" $O/$2_$1.txt | cut -c1-64,66-150 | head -8
}
for c in throughput wide stress shard2048 cs16 f32 split16; do run stats $c; done
for c in cs16 f32 shard2048 split16 throughput; do run fetch $c; run write $c; done
cd $R
ls $O | head -60
