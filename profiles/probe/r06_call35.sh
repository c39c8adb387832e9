#!/bin/bash
# round 6, GPU call 35: segment length 4 / 6 / 8 and the history arrays without their zero fill, alone with the bit log
R=$(pwd); O=$R/gpurun_out/r06_call35; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8 ) > $O/msk_lean_seg_ab.txt 2>&1
cat $O/msk_lean_seg_ab.txt | cut -c 1-140
