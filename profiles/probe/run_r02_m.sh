#!/bin/bash
export TMPDIR=/tmp
python profiles/probe/fir_variant_check.py 6 2>&1 | tail -1
for w in 4 2 1; do ACG_FIR_DEBUG_SHAPE=1 ACG_FIR_WAVES_PER_WG=$w python profiles/probe/fir_alloc_probe.py 16384 4 3 6 2>&1 | grep "round\|mfma" | sort | uniq | cut -c1-150 | sed "s/^/wpg=$w /"; done
