#!/bin/bash
# round 5, GPU call 18: bench.py under a profiler must not start a profiler of its own (live traffic guard); and once plain
R=$(pwd); O=$R/gpurun_out/r05_call18; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
cd /tmp
( time timeout 300 rocprofv3 --kernel-trace --stats -d $O/x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config throughput --detail-file $O/detail_profiled.json ) > $O/profiled.json 2> $O/profiled.err
rm -rf $O/x
cd $R
( time timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config throughput --detail-file $O/detail_plain.json ) > $O/plain.json 2> $O/plain.err
python - $O <<'PY'
import json, sys
for n in ("detail_profiled.json", "detail_plain.json"):
    try:
        r = json.load(open(sys.argv[1] + "/" + n))["roofline"]
        print(n, r.get("traffic"), r.get("traffic_live"), r.get("traffic_live_note"), (r.get("traffic_source") or "")[:80])
    except Exception as e:
        print(n, "FAILED", e)
PY
grep real $O/profiled.err $O/plain.err
