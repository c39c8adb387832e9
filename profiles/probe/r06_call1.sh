#!/bin/bash
# round 6, GPU call 1: where the never-timed product configurations stand (VERDICT r05 missing 2-4): rtlMult 160 / 192 one stream per
# channel, split int16, and the shared-stream kernel at 16 384 channels on 2 048 streams -- bench line + rocprofv3 --stats each
R=$(pwd); O=$R/gpurun_out/r06_call1; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
run() { # name, args...
  n=$1; shift
  ( time timeout 400 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --detail-file $O/${n}_detail.json "$@" ) > $O/${n}_stdout.txt 2> $O/${n}_stderr.txt
  tail -n 1 $O/${n}_stdout.txt > $O/${n}_line.json; tail -n 3 $O/${n}_stderr.txt | cut -c 1-300
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/${n}_prof -o ${n} -- python $R/bench.py --gpus 1 --steps 4 --warmup 1 --sustain 0 --no-cpu-baseline --no-live-traffic --no-ref-leg --check-channels 8 --detail-file /tmp/${n}_d.json "$@" ) > $O/${n}_prof_stdout.txt 2>&1
  f=$(find $O/${n}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c 1-260 > $O/${n}_stats.txt; cat $O/${n}_stats.txt
  find $O/${n}_prof -name "*.db" -delete; find $O/${n}_prof -name "*trace.csv" -delete
}
run m160 --channels 4096 --decim 160 --blocks 16
run m192 --channels 4096 --decim 192 --blocks 16
run split16 --channels 4096 --decim 160 --format split16 --blocks 16
run share8 --share 8 --channels 16384 --blocks 16
python - $O <<'PY'
import json, sys, os
for n in ("m160", "m192", "split16", "share8"):
    try:
        d = json.load(open(os.path.join(sys.argv[1], n + "_line.json")))
        print(n, d["value"], d["ms_per_step"], d["roofline"].get("frac"), d.get("whole_job_frac_of_hbm"), d["roofline"].get("kernel"), d.get("parity"))
    except Exception as e:
        print(n, "no line:", e)
PY
