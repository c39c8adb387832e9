#!/bin/bash
# round 5, GPU call 4: why is the demodulator 16 % slower per bit at 2048 channels than at 1024?  (a) alone on its partition at
# both widths; (b) the call size (dm footprint: 2048 x 8 callbacks x 2 buffers = 134 MB, beyond what stays in the Infinity
# Cache beside the streaming input); (c) its prefetch distance (refill blocks of 64 vs 128 samples); + the three updated tests
R=$(pwd); O=$R/gpurun_out/r05_call4; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bench_line_contract or also_cases or state_of_n" ) > $O/pytest_subset.txt 2>&1
tail -n 4 $O/pytest_subset.txt | cut -c 1-300
for ch in 1024 2048; do for l in wb64 wb128; do
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 100 python profiles/probe/msk_only.py $ch 8 2>&1 | tail -1 | sed "s/^/$l: /"
done; done | tee $O/msk_only.txt
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    k = json.load(open("bench_detail.json")).get("kernels", {})
    print("%-34s value %9.0f whole %.4f fir_frac %.4f fir_ms/launch %.4f  msk ms/pass %s" % (
        sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], k.get("msk_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run() { n=$1; c=$2; shift; shift
  ( cd $R; env "$@" timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 $EXTRA > $O/$n.json 2> $O/$n.err ); (cd $R; pt $O/$n.json $n); }
W64=ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libwb64.so; W128=ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libwb128.so
EXTRA="--call-blocks 8"; run s2048_cb8_wb64 shard2048 $W64; run s2048_cb8_wb128 shard2048 $W128
EXTRA="--call-blocks 4"; run s2048_cb4_wb64 shard2048 $W64; run s2048_cb4_wb128 shard2048 $W128
EXTRA="--call-blocks 2"; run s2048_cb2_wb64 shard2048 $W64
EXTRA="--call-blocks 8"; run s2048_cb8_wb64_b shard2048 $W64
EXTRA="--call-blocks 4"; run s2048_cb4_wb64_b shard2048 $W64
EXTRA="--call-blocks 8"; run thr_cb8_wb64 throughput $W64; run thr_cb8_wb128 throughput $W128
EXTRA="--call-blocks 4"; run thr_cb4_wb64 throughput $W64
EXTRA="--call-blocks 16"; run thr_cb16_wb128 throughput $W128
