#!/bin/bash
# round-2 evidence: the default bench line; per case rocprofv3 kernel stats and the FETCH_SIZE / WRITE_SIZE passes of the
# same command (one case per invocation, so that a kernel name means one launch shape); probes.
R=$(pwd)
O=$R/gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > $O/r02_bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/r02_bench_line.json; echo
cd /tmp
for c in throughput wide stress; do
  B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --config $c --also none"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/s_$c -- $B > $O/r02_bench_line_${c}_under_rocprof.json 2> $O/s_$c.err
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$c -- $B > /dev/null 2> $O/f_$c.err
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w_$c -- $B > /dev/null 2> $O/w_$c.err
  for k in s f w; do
    db=$(find $O/${k}_$c -name "*.db" | head -1)
    n=$( [ $k = s ] && echo stats || ( [ $k = f ] && echo fetch || echo write ) )
    [ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/r02_${c}_$n.txt 2>&1
    rm -rf $O/${k}_$c
  done
done
cd $R
timeout 300 python profiles/probe/fir_only_sweep.py 1024:200:8:200:5 1024:200:8:200:3 16384:200:4:200:5 16384:200:4:200:3 4096:200:4:192:5 4096:200:4:192:3 1024:200:144:200:5 > $O/r02_probe_fir_only.txt 2>&1
timeout 300 python profiles/probe/msk_only.py 1024 8 > $O/r02_probe_msk_only.txt 2>&1
timeout 300 python profiles/probe/msk_phase_stamps.py 1024 8 > $O/r02_probe_msk_phase_stamps.txt 2>&1
timeout 120 ./profiles/probe/front_probe 16 > $O/r02_probe_front.txt 2>&1
grep -h "fir_u8\|msk_demod" $O/r02_*_stats.txt | cut -c1-64,66-140
grep -h "fir_u8" $O/r02_*_fetch.txt $O/r02_*_write.txt | grep SIZE | cut -c1-40,64-120
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r02_pytest_gpu.txt; cat $O/r02_pytest_gpu.txt
[ -x ./profiles/probe/valu_beside_loads_probe ] && timeout 120 ./profiles/probe/valu_beside_loads_probe > $O/r02_probe_valu_beside_loads.txt 2>&1
