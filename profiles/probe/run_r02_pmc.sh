#!/bin/bash
# SQ counter passes over the down-converter alone (wave-private kernel vs workgroup-granular kernel)
R=$(pwd)
O=$R/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
SPEC="16384:200:8:200:5 16384:200:8:200:3"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/pmc_sq1 -- python $R/profiles/probe/fir_only_sweep.py $SPEC > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $O/pmc_sq2 -- python $R/profiles/probe/fir_only_sweep.py $SPEC > $O/pmc_sq2.log 2>&1
cd $R
for d in $O/pmc_sq1 $O/pmc_sq2; do
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py $db > $d.txt 2>&1
done
tail -3 $O/pmc_sq1.log $O/pmc_sq2.log
grep -h "fir_u8" $O/pmc_sq1.txt $O/pmc_sq2.txt | cut -c1-40,60-140
