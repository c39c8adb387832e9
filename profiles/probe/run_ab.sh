#!/bin/bash
# same-box A/B of the demodulator builds under acarsdec_amd/lib/ab/ (profiles/probe/build_ab.py), two alternating rounds
export TMPDIR=/tmp
CH=${1:-1024}; BL=${2:-8}
for i in 1 2; do
  for l in $(pwd)/acarsdec_amd/lib/ab/lib*.so; do
    ACARSDEC_AMD_LIB=$l timeout 100 python profiles/probe/msk_only.py $CH $BL 2>&1 | tail -1 | sed "s/^/$(basename $l .so | sed s/^lib//): /"
  done
done
