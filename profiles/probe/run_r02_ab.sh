#!/bin/bash
# same-process A/B of the down-converter variants inside the bench (same decoder, buffers, placement), several processes
O=gpurun_out/r02ab
mkdir -p $O
export TMPDIR=/tmp
for r in 1 2 3 4; do
  for c in stress wide; do
    timeout 600 python bench.py --no-cpu-baseline --also none --steps 30 --warmup 3 --check-channels 8 --config $c --ab 5,55,8,7 > $O/${c}_$r.json 2> $O/${c}_$r.err
    python - $O/${c}_$r.json $c $r <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-7s %s  value %9.0f  fir_frac %.3f  ab %s" % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["frac"], json.dumps(d.get("ab_same_process"))))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
timeout 600 python bench.py --no-cpu-baseline --also none --steps 15 --warmup 3 --check-channels 8 --config throughput --ab 5,55,8 > $O/head.json 2> $O/head.err
python - <<'PY'
import json
d = json.loads([x for x in open("gpurun_out/r02ab/head.json") if x.startswith("{")][-1])
print("head value %9.0f fir_frac %.3f ab %s" % (d["value"], d["roofline"]["frac"], json.dumps(d.get("ab_same_process"))))
PY
