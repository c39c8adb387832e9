#!/bin/bash
# round 6, GPU call 31: the instruction scheduler of msk_lean.hip -- other strategies / directions / no scheduler at all, alone with the bit log
R=$(pwd); O=$R/gpurun_out/r06_call31; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( bash profiles/probe/run_ab.sh 1024 8 ) > $O/msk_lean_sched_ab.txt 2>&1
cat $O/msk_lean_sched_ab.txt | cut -c 1-140
