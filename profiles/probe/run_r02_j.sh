#!/bin/bash
R=$(pwd)
O=$R/gpurun_out/r02j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -- python $R/profiles/probe/fir_only_sweep.py 16384:200:8:200:5 16384:200:8:200:62 16384:200:8:200:63 > $O/f.log 2>&1
cd $R
db=$(find $O/f -name "*.db" | head -1); python profiles/summarize_rocpd.py $db | grep -i "direct" | cut -c1-60,64-150
grep fir_only $O/f.log
