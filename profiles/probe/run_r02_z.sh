#!/bin/bash
export TMPDIR=/tmp
bash profiles/probe/run_ab.sh 2>&1 | cut -c1-80
L=$(ls $(pwd)/acarsdec_amd/lib/ab/lib*.so | tail -1); echo "tests on $L"
ACARSDEC_AMD_LIB=$L timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
