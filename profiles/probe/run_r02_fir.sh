#!/bin/bash
export TMPDIR=/tmp
S="16384:200:4:200:5"
echo default; ACG_FIR_DEBUG_SHAPE=1 timeout 200 python profiles/probe/fir_only_sweep.py $S 2>&1 | grep "fir_only\|fir_u8_direct" | sort | uniq -c
echo pairs8; ACG_FIR_DEBUG_SHAPE=1 ACG_FIR_RUN_PAIRS=8 timeout 200 python profiles/probe/fir_only_sweep.py $S 2>&1 | grep "fir_only\|fir_u8_direct" | sort | uniq -c
echo pairs4; ACG_FIR_DEBUG_SHAPE=1 ACG_FIR_RUN_PAIRS=4 timeout 200 python profiles/probe/fir_only_sweep.py $S 2>&1 | grep "fir_only\|fir_u8_direct" | sort | uniq -c
