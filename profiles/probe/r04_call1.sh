#!/bin/bash
# round 4, GPU call 1: the new bench line as the driver runs it, the GPU suite, and the context probe (plain + traced)
R=$(pwd); O=$R/gpurun_out/r04_call1; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -n 1 $O/bench_stdout.txt > $O/bench_line.json; wc -c $O/bench_line.json
cp bench_detail.json $O/ 2>/dev/null
for cfg in "4096 192" "16384 200"; do
  set -- $cfg
  timeout 300 python profiles/probe/context_probe.py $1 $2 4 0.5 2 > $O/ctx_$1.txt 2>&1
  ACG_STREAMS_DEDICATED=1 timeout 300 python profiles/probe/context_probe.py $1 $2 4 0.5 2 > $O/ctx_$1_ded1.txt 2>&1
  ACG_STREAMS_DEDICATED=2 timeout 300 python profiles/probe/context_probe.py $1 $2 4 0.5 2 > $O/ctx_$1_ded2.txt 2>&1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace_4096 -o t -- python $R/profiles/probe/context_probe.py 4096 192 4 0.25 1 > $O/ctx_4096_traced.txt 2>&1
python $R/profiles/probe/context_trace_summary.py $O/trace_4096 4 > $O/trace_4096_summary.txt 2>&1
head -c 300000 $(find $O/trace_4096 -name '*kernel_trace.csv' | head -1) > $O/trace_4096_head.csv
rm -rf $O/trace_4096
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
grep -h "round [01]:" $O/ctx_*.txt
cat $O/trace_4096_summary.txt | cut -c 1-400
