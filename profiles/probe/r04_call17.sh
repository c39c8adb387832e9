#!/bin/bash
# round 4, GPU call 17: the whole GPU suite at the final commit
R=$(pwd); O=$R/gpurun_out/r04_call17; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
rm -f gpurun_out/multidev_rates.txt
( time timeout 560 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -n 8 $O/pytest_gpu.txt | cut -c 1-260
cat gpurun_out/multidev_rates.txt
