#!/bin/bash
export TMPDIR=/tmp
timeout 100 python profiles/probe/fir_direct_debug.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fir" 2>&1 | tail -2
for p in 1 2 4 8; do
  ACG_FIR_RUN_PAIRS=$p timeout 300 python profiles/probe/fir_only_sweep.py 16384:200:8:200:5 4096:200:32:192:5 1024:200:144:200:5 2>&1 | grep fir_only | sed "s/^/pairs=$p /"
done
timeout 300 python profiles/probe/fir_only_sweep.py 16384:200:8:200:5 16384:200:4:200:5 4096:200:4:192:5 1024:200:8:200:5 16384:200:8:200:3 2>&1 | grep fir_only | sed "s/^/auto /"
