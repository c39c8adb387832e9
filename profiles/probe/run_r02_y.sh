#!/bin/bash
# A/B on one box: lib/ab/libA.so (previous demodulator) against the in-tree build, alternating
export TMPDIR=/tmp
A=$(pwd)/acarsdec_amd/lib/ab/libA.so
for i in 1 2 3; do
  [ -f $A ] && ACARSDEC_AMD_LIB=$A timeout 300 python profiles/probe/msk_only.py 1024 8 2>&1 | tail -1 | sed 's/^/A: /'
  timeout 300 python profiles/probe/msk_only.py 1024 8 2>&1 | tail -1 | sed 's/^/B: /'
done
