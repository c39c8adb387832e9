#!/bin/bash
# round 5, GPU call 16: the demodulator's per-bit cycle breakdown with the counted loop (the -DACG_MSK_STAMP build), next to the
# un-instrumented time of the product kernel on the same box
R=$(pwd); O=$R/gpurun_out/r05_call16; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
timeout 200 python profiles/probe/msk_phase_stamps.py 1024 8 2>&1 | tee $O/stamps.txt
timeout 100 python profiles/probe/msk_only.py 1024 8 2>&1 | tail -1 | tee -a $O/stamps.txt
