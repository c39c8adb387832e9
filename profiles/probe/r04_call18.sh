#!/bin/bash
# round 4, GPU call 18: the demodulator's per-phase s_memtime stamps (measurement build) next to its un-instrumented time per bit
R=$(pwd); O=$R/gpurun_out/r04_call18; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
timeout 200 python profiles/probe/msk_phase_stamps.py 1024 8 > $O/msk_phase_stamps.txt 2>&1
timeout 200 python profiles/probe/msk_only.py 1024 8 > $O/msk_only.txt 2>&1
cat $O/msk_phase_stamps.txt | tail -14; tail -2 $O/msk_only.txt
