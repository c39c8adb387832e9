#!/bin/bash
# round 4, GPU call 12: the Airspy / SDRplay / SoapySDR demos against their CPU twins; the C multi-device host with every context
# reading ONE shared input buffer (is the 5-18 % between equal contexts the context or the (input, output) buffer pair?)
R=$(pwd); O=$R/gpurun_out/r04_call12; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( time timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "callback_front_end or soapy or compat" ) > $O/pytest_demos.txt 2>&1
tail -n 6 $O/pytest_demos.txt | cut -c 1-250
for mode in "" "--shared-input"; do
  for N in 2 4; do
    echo "== N=$N $mode" >> $O/multidev_shared_input.txt
    timeout 200 acarsdec_amd/lib/host_multidev random rtl 8192 200 8 8 $N --msgs --time 150 $mode 2>&1 >/dev/null | grep "alone\|together" >> $O/multidev_shared_input.txt
  done
done
cat $O/multidev_shared_input.txt
