#!/bin/bash
# first GPU pass of round 2: correctness of the new down-converter, kernel sweeps, demodulator phase stamps,
# one default bench line, then the whole GPU test suite.  Everything lands in gpurun_out/r02a/.
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
echo "== fir tests" ; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fir" > $O/pytest_fir.log 2>&1; tail -3 $O/pytest_fir.log
echo "== fir sweep"; timeout 600 python profiles/probe/fir_only_sweep.py > $O/fir_only.txt 2>&1; cat $O/fir_only.txt
echo "== msk only"; timeout 300 python profiles/probe/msk_only.py 1024 8 > $O/msk_only.txt 2>&1; cat $O/msk_only.txt
echo "== msk stamps"; timeout 300 python profiles/probe/msk_phase_stamps.py 1024 8 > $O/msk_stamps.txt 2>&1; cat $O/msk_stamps.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; tail -c 3000 $O/bench_line.json; tail -5 $O/bench_err.txt
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
