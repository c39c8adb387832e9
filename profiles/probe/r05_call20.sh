#!/bin/bash
# round 5, GPU call 20: the complete GPU suite at the final tree (smoke included)
R=$(pwd); O=$R/gpurun_out/r05_call20; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=8 ) > $O/pytest_gpu.txt 2>&1
tail -n 6 $O/pytest_gpu.txt | cut -c 1-300
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -n 4 $O/smoke.txt
