#!/bin/bash
# round 5, GPU call 1 (round 4's kernels): (a) the <= 2048-channel regime -- lanes per channel x CU partition x chunk size, one
# process per point (the partition is a create-time decision); (b) kernel traces WITH TIMESTAMPS of wide / stress / shard2048 /
# throughput for profiles/probe/tails.py (who runs beside the slow launches)
R=$(pwd); O=$R/gpurun_out/r05_call1; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
export ACG_ALLOW_TUNING=1
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    k = d.get("kernels", {})
    print("%-34s value %9.0f whole %.4f fir_frac %.4f fir_ms/launch %.4f  fir/msk ms per step %s %s  e2e %s" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"],
        k.get("fir_ms_per_step"), k.get("msk_ms_per_step"), d.get("parity", {}).get("end_to_end_differing")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="python bench.py --config shard2048 --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8"
run() { # name, env...
  n=$1; shift
  ( env "$@" timeout 150 $B > $O/$n.json 2> $O/$n.err ); pt $O/$n.json $n
}
run base
for cus in 64 96 160; do run lpc8_cus$cus ACG_MSK_CUS=$cus; done
for cus in 32 48 64 80; do run lpc4_cus$cus ACG_MSK_LPC=4 ACG_MSK_CUS=$cus; done
run lpc8_nopart ACG_MSK_CUS=0
run lpc4_nopart ACG_MSK_LPC=4 ACG_MSK_CUS=0
run lpc8_pipe2 ACG_PIPE_BLOCKS=2
run lpc8_pipe8 ACG_PIPE_BLOCKS=8
unset ACG_ALLOW_TUNING
cd /tmp
for c in wide stress shard2048 throughput; do
  D=$O/x_trace_$c
  timeout 200 rocprofv3 --kernel-trace -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c > $O/trace_$c.json 2> $O/trace_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 120 python $R/profiles/probe/tails.py $db > $O/tails_$c.txt 2>&1; ls -la $db; fi
  rm -rf $D
  head -12 $O/tails_$c.txt
done
