#!/bin/bash
# SQ counter passes over the demodulator alone
R=$(pwd)
O=$R/gpurun_out/r02g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/pmc_sq1 -- python $R/profiles/probe/msk_only.py 1024 8 > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES -d $O/pmc_sq2 -- python $R/profiles/probe/msk_only.py 1024 8 > $O/pmc_sq2.log 2>&1
cd $R
for d in $O/pmc_sq1 $O/pmc_sq2; do
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py $db > $d.txt 2>&1
done
grep -h "msk_only" $O/pmc_sq1.log $O/pmc_sq2.log
grep -h "msk_demod" $O/pmc_sq1.txt $O/pmc_sq2.txt | cut -c1-30,60-140
