#!/bin/bash
# evidence pass: GPU suite, default bench line, rocprofv3 kernel stats and the two PMC traffic passes of the same command
R=$(pwd)
O=$R/gpurun_out/r02p
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; tail -c 600 $O/bench_line.json; echo
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -- $BENCH > $O/bench_line_under_rocprof.json 2> $O/stats.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- $BENCH > $O/fetch.out 2> $O/fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- $BENCH > $O/write.out 2> $O/write.err
cd $R
for d in stats fetch write; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py $db > $O/$d.txt 2>&1 && rm -rf $O/$d
done
head -12 $O/stats.txt | cut -c1-150
grep -h "fir_u8\|fill_random" $O/fetch.txt $O/write.txt | cut -c1-60,64-140
