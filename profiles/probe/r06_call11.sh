#!/bin/bash
# round 6, GPU call 11: smoke() and the whole GPU suite at the final tree
R=$(pwd); O=$R/gpurun_out/r06_call11; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -n 3 $O/smoke.txt | cut -c 1-300
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 6 $O/pytest_gpu.txt | cut -c 1-400
