#!/bin/bash
# round 5, GPU call 15: branch weights on the demodulator's three per-bit branches (block placement: the common path fall-through)
R=$(pwd); O=$R/gpurun_out/r05_call15; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
bash profiles/probe/run_ab.sh 1024 8 2>&1 | tee $O/msk_ab.txt
ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libexpect.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "testwav_blocks_bits_state or many_channels or ragged or lane_layouts or noise" 2>&1 | tail -2 | tee $O/ab_exact.txt
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f" % (sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for c in throughput stress shard2048 wide; do for l in base expect base expect; do
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/${c}_$l.json 2> $O/${c}_$l.err; pt $O/${c}_$l.json ${c}_$l
done; done | tee $O/bench_ab.txt
