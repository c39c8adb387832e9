#!/bin/bash
# end-of-round evidence at the final kernels (short form: the GPU budget of the round is nearly spent)
R=$(pwd)
O=$R/gpurun_out/r02final
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 3 > $O/r02_bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/r02_bench_line.json; echo
cd /tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --config throughput --also none"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/s_throughput -- $B > $O/r02_bench_line_throughput_under_rocprof.json 2> $O/s_throughput.err
db=$(find $O/s_throughput -name "*.db" | head -1)
[ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/r02_throughput_stats.txt 2>&1
rm -rf $O/s_throughput
grep -h "fir_u8\|msk_demod" $O/r02_throughput_stats.txt | cut -c1-64,66-140
cd $R
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r02_pytest_gpu.txt; cat $O/r02_pytest_gpu.txt
