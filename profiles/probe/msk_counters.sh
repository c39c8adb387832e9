#!/bin/bash
# SQ counters of the demodulator alone: one-wave kernel (msk.hip) and the two-wave split (msk2.hip), same process
O=$1; mkdir -p $O; R=$(pwd)
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/c1 -- python $R/profiles/probe/msk_only.py 1024 8 0,1 > $O/c1.log 2>&1
db=$(find $O/c1 -name "*.db" | head -1); [ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/msk_sq_counters.txt 2>&1; rm -rf $O/c1
grep -h "msk_demod" $O/msk_sq_counters.txt | cut -c1-50,60-140
