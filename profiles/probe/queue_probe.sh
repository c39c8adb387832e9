#!/bin/bash
# which hardware queue the demodulator stream lands on: the trial times of four contexts (stress case) under stream-creation variations
O=$1; mkdir -p $O
run() { l=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ref-leg --also none --steps 10 --warmup 2 --sustain 0 --check-channels 8 --placements 4 --config stress > $O/$l.json 2>/dev/null
  python - $O/$l.json $l <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s value %9.0f  trial ms per call %s" % (sys.argv[2], d["value"], d["config"]["placement"]["ms_per_call"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run skip0 A=1
run skip1 ACG_MSK_STREAM_SKIP=1
run skip2 ACG_MSK_STREAM_SKIP=2
run skip3 ACG_MSK_STREAM_SKIP=3
run prio_normal ACG_MSK_STREAM_PRIO_NORMAL=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
