#!/bin/bash
# round 6, GPU call 30: two more A/B builds of msk_lean.hip (three-address Horner steps in the mixer's sin/cos; h[] stored by tap phase: three
# b128 reads instead of six), alone with the bit log; their parity through the lean tests
R=$(pwd); O=$R/gpurun_out/r06_call30; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8 ) > $O/msk_lean_builds_ab2.txt 2>&1
cat $O/msk_lean_builds_ab2.txt | cut -c 1-140
( ACARSDEC_AMD_LIB=$R/acarsdec_amd/lib/ab/libboth.so timeout 600 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean_both.txt 2>&1
tail -n 4 $O/pytest_lean_both.txt | cut -c 1-300
