"""roofline.traffic: HBM bytes of one launch from PMC counters -- measured in the invocation (two rocprofv3 child runs) or looked up
in the committed passes."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter


def lookup_traffic(kernel, nch, M, ntaps, blocks_per_launch):
    """HBM bytes of one launch of this shape from the committed PMC passes (profiles/pmc_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of the same command; rocprofv3 cannot run inside the timed process): (bytes, source) or (None, None)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["channels"], e["decim"], e["ntaps"]) == (nch, M, ntaps) and abs(e["blocks_per_launch"] - blocks_per_launch) < 1e-9 \
                        and e["kernel"] == kernel:
                    return e["traffic_bytes"], e.get("source", "profiles/pmc_traffic.json")
    except Exception:
        pass
    return None, None


def live_traffic(case_name, kernel, nch, blocks_per_launch, timeout_s=100):
    """HBM bytes of ONE launch of `kernel` measured in THIS invocation: two child runs of this script under rocprofv3
    (--kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE: separate passes, never combined with another trace domain, as
    MI355X_MICROARCH.md's HBM section prescribes), a short burst of the same launch shape each; traffic = 2 x FETCH_SIZE x 1024
    + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes; profiles/pmc_traffic.json _about).
    Returns (bytes, source text) or (None, reason): any failure leaves the committed look-up in place."""
    import shutil
    import sqlite3
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    # (this process may itself be running under a profiler -- `rocprofv3 --stats -- python bench.py ...`: no profiler inside a profiler)
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler: no nested PMC pass"
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [prof, "--kernel-trace", "--pmc", ctr, "-d", td, "--", sys.executable, BENCH, "--config", case_name,
                       "--also", "none", "--blocks", str(int(2 * blocks_per_launch)), "--channels", str(nch), "--steps", "3", "--warmup", "1", "--sustain", "0",
                       "--no-cpu-baseline", "--no-ref-leg", "--check-channels", "8", "--no-live-traffic", "--detail-file", os.path.join(td, "detail.json")]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(td) for f in fs if f.endswith(".db")]
                if r.returncode != 0 or not dbs:
                    return None, "rocprofv3 --pmc %s child failed (%d)" % (ctr, r.returncode)
                con = sqlite3.connect(dbs[0])
                rows = con.execute("select k.name, count(*), avg(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                                   "where p.counter_name = ? group by k.name", (ctr,)).fetchall()
                con.close()
                hit = [(n, c, v) for n, c, v in rows if n.replace("void ", "").startswith(kernel.split("(")[0])]
                if not hit:
                    return None, "kernel %s not in the %s pass" % (kernel, ctr)
                vals[ctr] = (hit[0][2], hit[0][1])
    except Exception as ex:                      # (timeouts included: the line must not depend on a profiler)
        return None, "live PMC pass failed: %r" % (ex,)
    traffic = int(round(2 * vals["FETCH_SIZE"][0] * 1024 + vals["WRITE_SIZE"][0] * 1024))
    return traffic, ("measured in this invocation: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of this script at "
                     "the same launch shape, %d / %d launches): 2 x FETCH_SIZE + WRITE_SIZE" % (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]))
