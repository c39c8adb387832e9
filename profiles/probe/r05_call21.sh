#!/bin/bash
# round 5, GPU call 21: soak -- the headline and the 16 384-channel case sustained for 40 s each (step time min / median / max, clocks)
R=$(pwd); O=$R/gpurun_out/r05_call21; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for c in throughput wide; do
  ( time timeout 400 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --no-live-traffic --steps 20 --warmup 3 --sustain 40 --check-channels 16 --detail-file $O/detail_$c.json ) > $O/$c.json 2> $O/$c.err
  python - $O/detail_$c.json $c <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["sustain"]
    print("%-11s value %9.0f whole %.4f fir %.4f timed %.1f s  passes/step %d  step ms min/median/max %s  sclk %s  %s" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["timed_region_s"], s["passes_per_step"], s["step_ms_min_median_max"],
        s["shader_clock_mhz_start_mid_end"], s["telemetry_mid_run"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done | tee $O/soak.txt
