#!/bin/bash
# round 4, GPU call 14: does the spread between equal contexts in the C host follow the number of hardware queues the HIP runtime
# maps its streams onto (GPU_MAX_HW_QUEUES, default 4)?
R=$(pwd); O=$R/gpurun_out/r04_call14; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for q in 1 2 4 8 16 24; do
  for N in 4 2; do
    echo "== GPU_MAX_HW_QUEUES=$q N=$N" >> $O/multidev_hw_queues.txt
    GPU_MAX_HW_QUEUES=$q timeout 120 acarsdec_amd/lib/host_multidev random rtl 8192 200 8 8 $N --msgs --time 150 2>&1 >/dev/null | grep "alone\|together" >> $O/multidev_hw_queues.txt
  done
done
cat $O/multidev_hw_queues.txt
