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
    assert d["parity"]["bit_exact"] is True and d["parity"]["blocks"] > 0 and d["parity"]["dm_within_1e5_rel"] is True
    assert d["parity"]["channels_checked"] == 64
    assert abs(d["value"] - 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]
    assert d["time_dominant_kernel"] in ("msk_demod_kernel", d["roofline"]["kernel"]) and 0 < d["whole_job_frac_of_hbm"] < 1
    assert d["roofline"]["pure_reader_GBs_measured_this_run"] > 1000


def test_bench_also_cases_in_one_line():
    """one invocation = the headline case plus the cases under "also", each with its own parity gate, down-converter
    roofline and whole-job fraction of HBM bandwidth (here at reduced sizes through the case table)."""
    code = ("import sys, bench; bench.CASES['throughput'].update(channels=128, blocks=4); "
            "bench.CASES['wide'].update(channels=512, blocks=2); bench.CASES['stress'].update(channels=256, blocks=2); "
            "sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']; bench.main()")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert set(d["also"]) == {"wide", "stress"}
    for name, a in d["also"].items():
        assert a["parity"]["dm_within_1e5_rel"] is True and a["parity"]["bit_exact"] is True and a["parity"]["channels_checked"] == 64
        assert 0 < a["roofline"]["frac"] < 1 and 0 < a["whole_job_frac_of_hbm"] < 1 and a["value"] > 0
    assert d["also"]["wide"]["parity"]["blocks"] >= 0 and "filter" in d["also"]["stress"]["config"]


def test_bench_gpus_flag_self_launch():
    """`python bench.py --gpus 2` with NO torchrun around it: the flag itself launches the ranks (here two ranks sharing
    the box's GPU over gloo, because RCCL wants one GPU per rank), rank 0 alone prints the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ACG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--channels", "256", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["channels_total"] == 512 and len(d["per_gpu"]) == 2
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1
    assert d["parity"]["bit_exact"] is True and d["parity"]["channels_checked"] == 64
    # without the rehearsal backend a 1-GPU box must refuse 2 ranks instead of running one
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("ACG_BENCH_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                           timeout=300, cwd=ROOT, env=env)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr


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
