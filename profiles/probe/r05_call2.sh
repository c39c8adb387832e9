#!/bin/bash
# round 5, GPU call 2: the repair pass searched 64 candidates at a time, the field split a wave per block, the ring a power of two,
# soh_sample, batched legacy state -- the GPU tests, then shard2048 / stress / throughput (bench + traces for tails.py)
R=$(pwd); O=$R/gpurun_out/r05_call2; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 12 $O/pytest_gpu.txt | cut -c 1-300
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f fir_ms/launch %.4f  e2e %s" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"],
        d.get("parity", {}).get("end_to_end_differing")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
export ACG_ALLOW_TUNING=1
run() { n=$1; c=$2; shift; shift
  ( env "$@" timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/$n.json 2> $O/$n.err ); pt $O/$n.json $n; }
run s2048_base shard2048
run s2048_pipe8 shard2048 ACG_PIPE_BLOCKS=8
run s2048_cus96 shard2048 ACG_MSK_CUS=96
run s2048_pipe8_cus96 shard2048 ACG_PIPE_BLOCKS=8 ACG_MSK_CUS=96
run s2048_pipe8_cus112 shard2048 ACG_PIPE_BLOCKS=8 ACG_MSK_CUS=112
run s2048_pipe8_cus80 shard2048 ACG_PIPE_BLOCKS=8 ACG_MSK_CUS=80
run thr_base throughput
run stress_base stress
unset ACG_ALLOW_TUNING
cd /tmp
for c in shard2048 stress throughput; do
  D=$O/x_trace_$c
  timeout 200 rocprofv3 --kernel-trace -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c > $O/trace_$c.json 2> $O/trace_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 120 python $R/profiles/probe/tails.py $db > $O/tails_$c.txt 2>&1; fi
  rm -rf $D
  head -14 $O/tails_$c.txt
done
