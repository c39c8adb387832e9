#!/bin/bash
# First GPU call of the next round: what round 2 left unmeasured when its GPU budget ran out.
#   (on the build host first:  python profiles/probe/build_ab.py table poly:-DACG_MSK_SINCOS_POLY )
#   1 the whole GPU suite at the round-2 head (the last change, the table sin/cos, only saw the demodulator's tests)
#   2 the default bench line
#   3 same-process A/B (bench.py --ab: same decoder, buffers, placement): lanes per channel at 16 384 and 4096 channels,
#     down-converter variants 5 (write-through) / 55 (write-back) / 8 (parked results) once more
#   4 the demodulator alone, table against polynomial sin/cos (lib/ab/*.so from build_ab.py)
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; tail -c 400 $O/bench_line.json; echo
ab() { # label, case, --ab spec
  timeout 400 python bench.py --no-cpu-baseline --also none --steps 30 --warmup 3 --check-channels 8 --config $2 --ab "$3" > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-22s value %9.0f  fir_frac %.3f  ab %s" % (sys.argv[2], d["value"], d["roofline"]["frac"], json.dumps(d.get("ab_same_process"))))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for r in 1 2; do
  ab wide_lpc_$r wide ACG_MSK_LPC_LIVE=4,2,1
  ab stress_lpc_$r stress ACG_MSK_LPC_LIVE=8,4,2
  ab wide_fir_$r wide 5,55,8
  ab stress_fir_$r stress 5,55,8
done
ab head_lpc throughput ACG_MSK_LPC_LIVE=8,4
[ -d acarsdec_amd/lib/ab ] && bash profiles/probe/run_ab.sh 1024 8 | tee $O/msk_ab.txt
