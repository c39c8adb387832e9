#!/bin/bash
# round 5, GPU call 3: new <= 2048-channel defaults (whole-call launches, 80 CUs for the demodulator, fewer barrier packets on
# its chain), the tightened gate with reference legs on every format, the rtl8 case -- all GPU tests, the default bench line,
# shard2048 three times, traces of shard2048 / throughput with the demodulator's launch gaps
R=$(pwd); O=$R/gpurun_out/r05_call3; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
tail -n 15 $O/pytest_gpu.txt | cut -c 1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json; tail -n 6 $O/bench_stderr.txt | cut -c 1-400
cp bench_detail.json $O/ 2>/dev/null
python - $O/bench_line.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("headline", d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d["parity"])
    for k, v in d.get("also", {}).items():
        print("  ", k, json.dumps(v)[:400])
except Exception as e:
    print("no bench line:", e)
PY
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
for i in 1 2 3; do
  timeout 150 python bench.py --config shard2048 --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/s2048_$i.json 2> $O/s2048_$i.err; pt $O/s2048_$i.json s2048_$i
done
cd /tmp
for c in shard2048 throughput; do
  D=$O/x_trace_$c
  timeout 200 rocprofv3 --kernel-trace -d $D -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ref-leg --sustain 0 --check-channels 16 --also none --config $c > $O/trace_$c.json 2> $O/trace_$c.err
  db=$(find $D -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then timeout 120 python $R/profiles/probe/tails.py $db > $O/tails_$c.txt 2>&1; timeout 120 python $R/profiles/probe/tails.py $db msk_demod > $O/tails_msk_$c.txt 2>&1; fi
  rm -rf $D
  head -8 $O/tails_$c.txt; head -6 $O/tails_msk_$c.txt
done
