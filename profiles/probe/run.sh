#!/bin/bash
# One parameterised measurement script (replaces round 2's forty run_r02_*.sh lab notes; they are in the history).
#   usage (as ONE gpurun call):  bash profiles/probe/run.sh <outdir-tag> <stage> [<stage> ...]
# stages:
#   tests            pytest tests -m gpu
#   bench            the default bench line                       -> bench_line.json
#   stats            rocprofv3 --kernel-trace --stats per case    -> <case>_stats.txt, bench_line_<case>_under_rocprof.json
#   pmc              rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per case (separate runs) -> <case>_fetch.txt / _write.txt
#   ab:<case>:<spec> same-process A/B of a per-launch switch (bench.py --ab), e.g. ab:wide:ACG_MSK_LPC_LIVE=4,2,1  ab:stress:5,55,8
#   decoders:<case>:<n>  n separately allocated decoders in one process, --placements 1 (the placement spread)
#   fironly          the down-converter alone per launch shape (profiles/probe/fir_only_sweep.py)
#   mskonly          the demodulator alone (profiles/probe/msk_only.py 1024 8)
#   dmfoot           the down-converter alone by dm footprint (profiles/probe/dm_footprint_probe.py)
#   cmd:<file>       bash <file> (anything else; the file is part of the snapshot)
R=$(pwd)
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
CASES=${CASES:-"throughput wide stress"}
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    a = d.get("also", {})
    print("%-28s value %9.0f frac %.3f whole %.3f  burst %s | %s | ab %s trials %s" % (
        sys.argv[2], d["value"], d["roofline"]["frac"], d["whole_job_frac_of_hbm"], d.get("burst", {}).get("value"),
        " ".join("%s %.0f/%.3f/%.3f" % (k, v["value"], v["roofline"]["frac"], v["whole_job_frac_of_hbm"]) for k, v in a.items()),
        json.dumps(d.get("ab_same_process")), json.dumps(d.get("placement_trials"))))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for st in "$@"; do
  case $st in
    tests) timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/pytest_gpu.txt ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; line $O/bench_line.json bench; tail -5 $O/bench_err.txt ;;
    stats|pmc)
      cd /tmp
      for c in $CASES; do
        B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ref-leg --sustain 0 --config $c --also none"
        if [ $st = stats ]; then
          timeout 600 rocprofv3 --kernel-trace --stats -d $O/s_$c -- $B > $O/bench_line_${c}_under_rocprof.json 2> $O/s_$c.err
          db=$(find $O/s_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/${c}_stats.txt 2>&1; rm -rf $O/s_$c
          grep -h "fir_u8\|msk_demod" $O/${c}_stats.txt | cut -c1-64,66-140
        else
          timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$c -- $B > /dev/null 2> $O/f_$c.err
          db=$(find $O/f_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/${c}_fetch.txt 2>&1; rm -rf $O/f_$c
          timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w_$c -- $B > /dev/null 2> $O/w_$c.err
          db=$(find $O/w_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/profiles/summarize_rocpd.py $db > $O/${c}_write.txt 2>&1; rm -rf $O/w_$c
          grep -h "fir_u8" $O/${c}_fetch.txt $O/${c}_write.txt | grep SIZE | cut -c1-40,64-120
        fi
      done
      cd $R ;;
    ab:*) IFS=: read -r _ c spec <<< "$st"; n=ab_${c}_$(echo $spec | tr '=,' '__')
      timeout 500 python bench.py --no-cpu-baseline --no-ref-leg --also none --steps 30 --warmup 3 --sustain 0 --check-channels 8 --config $c --ab "$spec" > $O/$n.json 2> $O/$n.err; line $O/$n.json $n ;;
    decoders:*) IFS=: read -r _ c k <<< "$st"; n=decoders_${c}_$k
      timeout 600 python bench.py --no-cpu-baseline --no-ref-leg --also none --steps 20 --warmup 3 --sustain 0 --check-channels 8 --config $c --placements 1 --decoders $k > $O/$n.json 2> $O/$n.err; line $O/$n.json $n ;;
    fironly) timeout 300 python profiles/probe/fir_only_sweep.py 1024:200:8:200:5 16384:200:4:200:5 4096:200:4:192:5 1024:200:144:200:5 > $O/probe_fir_only.txt 2>&1; tail -6 $O/probe_fir_only.txt ;;
    mskonly) timeout 300 python profiles/probe/msk_only.py 1024 8 > $O/probe_msk_only.txt 2>&1; tail -3 $O/probe_msk_only.txt ;;
    dmfoot) timeout 400 python profiles/probe/dm_footprint_probe.py 16384 4 1 2 4 > $O/probe_dm_footprint.txt 2>&1; cat $O/probe_dm_footprint.txt | tail -14 ;;
    cmd:*) bash ${st#cmd:} $O ;;
    *) echo "unknown stage $st" ;;
  esac
done
