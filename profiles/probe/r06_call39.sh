#!/bin/bash
# round 6, GPU call 39: last look at the final tree -- smoke, the lean tests, the headline alone
R=$(pwd); O=$R/gpurun_out/r06_call39; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
( timeout 600 python -m pytest tests/test_gpu_lean.py tests/test_gpu_bench.py -m gpu -q -x -p no:cacheprovider ) > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt | cut -c 1-200
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --also none --no-cpu-baseline --detail-file $O/headline.json ) > $O/headline.txt 2>&1
python - $O/headline.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline_msk"], d["parity"]["end_to_end"]["blocks_differing"], d["parity"]["end_to_end"]["gpu_vs_ref_ofast"])
PY
