"""bench.py end to end on the GPU box: the JSON contract, and a rehearsal of the N > 1 launch path
(two ranks under torch.distributed.run; with one GPU they share it and the few collectives go over gloo --
the sharding, the scatter, the barrier-bracketed timing and the rank-0 report are the code the 8-GPU run uses)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--channels", "256", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    assert d["parity"]["bit_exact"] is True and d["parity"]["blocks"] > 0
    assert abs(d["value"] - 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, ACG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--channels", "256", "--steps", "4",
           "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["channels_per_gpu"] == 256 and d["config"]["channels_total"] == 512
    assert d["parity"]["bit_exact"] is True
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1          # rank 0 alone reports
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(d["value"] - 2 * 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]
