#!/bin/bash
# trial times of four contexts under stream-creation shifts, stress and wide
O=$1; mkdir -p $O
run() { l=$1; c=$2; shift; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ref-leg --also none --steps 6 --warmup 1 --sustain 0 --check-channels 8 --placements 4 --config $c > $O/$l.json 2>/dev/null
  python - $O/$l.json $l <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-22s value %9.0f  trial ms per call %s" % (sys.argv[2], d["value"], d["config"]["placement"]["ms_per_call"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for c in stress wide; do
  for k in 0 1 2 3; do run ${c}_skipn$k $c ACG_STREAM_SKIP_NORMAL=$k; done
  run ${c}_skiph1 $c ACG_MSK_STREAM_SKIP=1
done
