#!/bin/bash
# round 4, GPU call 15: does a SINGLE context gain from more hardware queues? (bench cases under GPU_MAX_HW_QUEUES = 4 / 16)
R=$(pwd); O=$R/gpurun_out/r04_call15; mkdir -p $O; export TMPDIR=/tmp
exec </dev/null
for q in 4 16 4 16; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --config shard2048 --also throughput,stress --no-cpu-baseline --no-ref-leg --sustain 2 --steps 10 --warmup 2 --check-channels 16 > $O/b_$q.txt 2> $O/b_$q.err
  tail -n 1 $O/b_$q.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GPU_MAX_HW_QUEUES=$q', 'shard2048', d['value'], {k:v['value'] for k,v in d.get('also',{}).items()})" | tee -a $O/summary.txt
done
