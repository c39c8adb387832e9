#!/bin/bash
# round 6, GPU call 32: the lean tests incl. a host-written bit count / block length, the demodulator tests of the older files
R=$(pwd); O=$R/gpurun_out/r06_call32; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 900 python -m pytest tests/test_gpu_lean.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1
tail -n 12 $O/pytest.txt | cut -c 1-300
