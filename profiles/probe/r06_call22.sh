#!/bin/bash
# round 6, GPU call 22: msk_lean.hip, a period inside a segment as one basic block (six steps uncommitted, one wave-wide test): parity, A/B
R=$(pwd); O=$R/gpurun_out/r06_call22; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 900 python -m pytest tests/test_gpu_lean.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest_lean.txt 2>&1
tail -n 30 $O/pytest_lean.txt | cut -c 1-400
( timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 acars 1
  timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 noise 1
  timeout 200 python profiles/probe/msk_lean_ab.py 1024 8 acars 0
  timeout 200 python profiles/probe/msk_lean_ab.py 2048 8 acars 1 ) > $O/msk_lean_ab.txt 2>&1
grep -v amdgpu.ids $O/msk_lean_ab.txt | cut -c 1-200
