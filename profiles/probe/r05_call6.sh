#!/bin/bash
# round 5, GPU call 6 (the one structural try on the demodulator): same-box A/B of the register budget of one
# wave per SIMD and of a counted inner loop of 9 bit periods between two looks at the dm window; exactness of each on the golden
# recording; + the updated bench tests
R=$(pwd); O=$R/gpurun_out/r05_call6; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
bash profiles/probe/run_ab.sh 1024 8 2>&1 | tee $O/msk_ab.txt
for l in cnt both; do
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "testwav_blocks_bits_state or many_channels or ragged" 2>&1 | tail -2 | sed "s/^/$l: /"
done | tee $O/ab_exact.txt
pt() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print("%-34s value %9.0f whole %.4f fir_frac %.4f" % (sys.argv[2], d["value"], d["whole_job_frac_of_hbm"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for c in throughput stress shard2048; do for l in base eu1 cnt both base; do
  ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/lib$l.so timeout 150 python bench.py --config $c --also none --no-cpu-baseline --no-ref-leg --steps 10 --warmup 2 --sustain 2 --check-channels 8 > $O/${c}_$l.json 2> $O/${c}_$l.err; pt $O/${c}_$l.json ${c}_$l
done; done | tee $O/bench_ab.txt
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "bench_line_contract or also_cases or state_of_n" ) > $O/pytest_subset.txt 2>&1
tail -n 4 $O/pytest_subset.txt | cut -c 1-300
