#!/bin/bash
# round 5, GPU call 5: the SOH stamp out of the demodulator's registers (the per-bit loop had started spilling): demodulator alone,
# the fixed tests, stress / throughput / shard2048 / wide quick benches
R=$(pwd); O=$R/gpurun_out/r05_call5; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for ch in 1024 2048; do timeout 100 python profiles/probe/msk_only.py $ch 8 2>&1 | tail -1; done | tee $O/msk_only.txt
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bench_line_contract or also_cases or state_of_n or time_stamps or testwav or lane_layouts or two_wave or precise or ring_across or streaming_past" ) > $O/pytest_subset.txt 2>&1
tail -n 4 $O/pytest_subset.txt | cut -c 1-300
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f fir_ms/launch %.4f" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { n=$1; c=$2; shift; shift
  ( env "$@" timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/$n.json 2> $O/$n.err ); pt $O/$n.json $n; }
run thr throughput; run stress stress; run s2048 shard2048; run wide wide; run thr_b throughput; run stress_b stress; run s2048_b shard2048
