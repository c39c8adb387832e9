#!/bin/bash
# round 6, GPU call 33: the wave-wide test of a period inside a segment right behind the VCO / clock steps (mixer .. loop filter one basic block)
R=$(pwd); O=$R/gpurun_out/r06_call33; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8 ) > $O/msk_lean_early_ab.txt 2>&1
cat $O/msk_lean_early_ab.txt | cut -c 1-140
( ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libearly.so timeout 600 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean_early.txt 2>&1
tail -n 4 $O/pytest_lean_early.txt | cut -c 1-300
