#!/bin/bash
# round 6, GPU call 12: soak -- the cases that are new on the line sustained for 40 s each (step time min / median / max, clocks, power)
R=$(pwd); O=$R/gpurun_out/r06_call12; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for c in share8 m160 m192 split16; do
  ( time timeout 400 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --no-live-traffic --steps 20 --warmup 3 --sustain 40 --check-channels 16 --detail-file $O/detail_$c.json ) > $O/$c.json 2> $O/$c.err
  python - $O/detail_$c.json $c <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["sustain"]
    print("%-8s value %9.0f whole %.4f fir %.4f timed %.1f s  passes/step %d  step ms min/median/max %s  sclk %s  power %s  blocks decoded %d" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["timed_region_s"], s["passes_per_step"], s["step_ms_min_median_max"],
        s["shader_clock_mhz_start_mid_end"], [t and t.get("power_w") for t in s["telemetry_start_mid_end"]], d["config"]["blocks_decoded_timed"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done | tee $O/soak.txt
