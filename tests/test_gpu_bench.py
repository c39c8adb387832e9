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


def compact_json(text):
    """the driver-facing result: the LAST stdout line, compact (< 4 KB)"""
    lines = [l for l in text.splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), text[-2000:]
    assert len(lines[-1]) < 4096, len(lines[-1])
    assert len([l for l in lines if l.startswith("{")]) == 1          # one JSON-looking line per run
    return json.loads(lines[-1])


def last_json(text):
    """the full detail: the earlier '# bench_detail: {...}' line (also written to bench_detail.json)"""
    lines = [l for l in text.splitlines() if l.startswith("# bench_detail: ")]
    assert len(lines) == 1, text[-2000:]
    compact_json(text)
    return json.loads(lines[0][len("# bench_detail: "):])


def par_msgs(d):
    m = d["parity"]["msgs"]
    assert m["exact"] is True and m["delivered"] >= m["records"]
    return m["records"]


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--channels", "256", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--sustain", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    c = compact_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d and k in c, k
    assert c["value"] == d["value"] and c["ms_per_step"] == d["ms_per_step"] and c["roofline"]["frac"] == d["roofline"]["frac"]
    assert c["parity"]["exact_given_gpu_dm"] is True and c["parity"]["msgs_exact"] is True and c["parity"]["exact_order_identical"] is True
    assert c["config"]["channels_per_gpu"] == 256 and "workload" in c["config"] and "acg_msg" in c["config"]["delivered"]
    assert c["roofline"]["launches_per_step"] == d["roofline"]["launches_per_pass"] * d["sustain"]["passes_per_step"]
    # the delivered path is what is timed: repaired blocks -> acg_msg records, and the gate compared every field of them
    assert par_msgs(d) > 0 and "ACG_F_REPAIR" in d["config"]["delivered"]
    with open(os.path.join(ROOT, "bench_detail.json")) as f:
        assert json.load(f)["value"] == d["value"]
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    par = d["parity"]
    assert par["blocks_exact_given_gpu_dm"] is True and par["blocks"] > 0 and par["dm_within_1e5_rel"] is True
    assert par["channels_checked"] == 64 and "bit_exact" not in par
    # the exact-order mode closes the argument: same GPU demodulator, dm bit-identical to the oracle -> end to end identical
    assert par["exact_order_mode"]["dm_bit_identical_to_oracle"] is True and par["exact_order_mode"]["blocks_identical_end_to_end"] is True
    assert (par["end_to_end"]["allowed"] is None) == (par["reference_builds"] is None)     # (without the builds: reported, not gated)
    if par["reference_builds"] is not None:
        assert par["end_to_end"]["blocks_differing"] <= par["end_to_end"]["allowed"]           # oracle/_ref travelled: the reference's two builds on the same bytes
        assert par["reference_builds"]["oracle_vs_ref_o2_blocks_differing"] == 0
        # no slack: what the reference's own builds differ by, and nothing at all against the reference as shipped (-Ofast)
        assert par["end_to_end"]["allowed"] == par["reference_builds"]["ref_fast_vs_ref_o2_blocks_differing"]
        assert par["reference_builds"]["gpu_vs_ref_ofast_blocks_differing"] == 0 and par["end_to_end"]["gpu_vs_ref_ofast"] == 0
    assert d["sustain"]["passes_per_step"] == 1 and d["burst"]["value"] > 0
    assert abs(d["value"] - 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]
    assert d["time_dominant_kernel"] in ("msk_demod_kernel", "msk_lean_kernel", d["roofline"]["kernel"]) and 0 < d["whole_job_frac_of_hbm"] < 1
    assert d["roofline"]["pure_reader_GBs_measured_this_run"] > 1000


def test_bench_also_cases_in_one_line():
    """one invocation = the headline case plus the cases under "also", each with its own parity gate, down-converter
    roofline and whole-job fraction of HBM bandwidth (here at reduced sizes through the case table)."""
    code = ("import sys, bench; bench.CASES['throughput'].update(channels=128, blocks=12); "
            "bench.CASES['wide'].update(channels=512, blocks=2); bench.CASES['stress'].update(channels=256, blocks=2); "
            "bench.CASES['cs16'].update(channels=256, blocks=2); bench.CASES['f32'].update(channels=256, blocks=2); "
            "bench.CASES['shard2048'].update(channels=192, blocks=4); "
            "bench.CASES['m160'].update(channels=256, blocks=4); bench.CASES['m192'].update(channels=256, blocks=4); "
            "bench.CASES['split16'].update(channels=256, blocks=4); bench.CASES['share8'].update(channels=512, blocks=4); "
            "sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--sustain', '0.5', '--sustain-hbm', '0.5', "
            "'--hostfed-channels', '320']; bench.main()")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert set(d["also"]) == {"wide", "stress", "shard2048", "cs16", "f32", "hostfed", "rtl8", "m160", "m192", "split16", "share8"}
    hf = d["also"].pop("hostfed")
    # BASELINE configs[1]: one dongle's 8 (and 16) channels on one 2.0 Msps stream, a callback at a time from host memory --
    # the legacy view (the reference program on compat_msk.c), the batched API with nstreams = 1, the CPU reference; ms per
    # callback against the 81.92 ms a callback's signal lasts, and the three printed outputs identical
    r8 = d["also"].pop("rtl8")
    assert r8["budget_ms_per_callback"] == 81.92
    for k in ("ch8", "ch16"):
        e = r8[k]
        assert e["messages"] > 0 and e["parity"]["batched_equals_cpu_program"] in (True, None)
        assert 0 < e["batched_ms_per_callback"] < 81.92
        if e["parity"]["legacy_program_equals_cpu_program"] is not None:       # the demo binaries travelled
            assert e["parity"]["legacy_program_equals_cpu_program"] is True and e["parity"]["batched_equals_legacy_program"] is True
            assert 0 < e["legacy_ms_per_callback"] < 10.0 and e["cpu_reference_ms_per_callback"] > 0
    # the host-fed case: the same records as the _dev entry point on the same bytes, the oracle on the gate channels, and a
    # rate that is a fraction (<= 1, within timer noise) of what a bare host-to-device copy reaches on this box
    assert hf["parity"]["same_records_as_dev_entry_point"] is True and hf["parity"]["blocks_exact_given_gpu_dm"] is True and hf["parity"]["blocks"] > 0
    assert hf["hostfed"]["h2d_GBs_measured"] > 1 and 0 < hf["hostfed"]["frac_of_h2d"] < 1.1 and hf["value"] > 0
    assert hf["config"]["channels_per_gpu"] == 320 and hf["hostfed"]["realtime_needs"] == 320 * 2.5
    c = compact_json(r.stdout)
    assert set(c["also"]) == set(d["also"]) | {"hostfed", "rtl8"} and all(a["parity_ok"] is True for k_, a in c["also"].items() if k_ != "rtl8")
    assert c["also"]["rtl8"]["ch8"]["parity_ok"] is True and c["also"]["rtl8"]["ch16"]["batched_ms"] > 0
    # every format's gate has its reference-builds leg now (rtl.c / soapy.c / air.c from oracle/_ref) where the builds travelled
    # (round 6: sdrplay.c for the split planes, and rtl.c's in_callback channel by channel for the 8-channels-per-dongle case)
    for name in ("cs16", "f32", "split16", "share8", "m160", "m192"):
        rb = d["also"][name]["parity"]["reference_builds"]
        if rb is not None:
            assert rb["oracle_vs_ref_o2_blocks_differing"] == 0 and rb["gpu_vs_ref_ofast_blocks_differing"] == 0
    assert c["also"]["hostfed"]["hostfed"]["realtime"] in (True, False)
    assert c["also"]["shard2048"]["ch"] == 192 and "u8" not in d["also"]["cs16"]["config"]["arithmetic"]
    # rtl.c's own shape on the matrix pipe: 8 channels per dongle stream, the kernel named, its HBM fraction and what a VALU kernel would need
    s8 = d["also"]["share8"]
    assert s8["config"]["channels_per_stream"] == 8 and s8["roofline"]["kernel"] == "fir_u8_mm_kernel<25, 1>" and s8["roofline"]["bound"] == "hbm"
    assert s8["roofline"]["valu_equivalent"]["frac"] > 0 and 0 < s8["roofline"]["mfma_i8"]["frac"] < 1 and "ACARS" in s8["data"]
    assert c["also"]["share8"]["ch_per_stream"] == 8 and c["also"]["split16"]["fmt"] == "split16" and c["also"]["m160"]["M"] == 160
    assert d["also"]["m160"]["roofline"]["kernel"].startswith("fir_u8_direct_kernel<20,") and d["also"]["m192"]["roofline"]["kernel"].startswith("fir_u8_direct_kernel<24,")
    assert d["also"]["split16"]["roofline"]["kernel"] == "fir_fmt_direct_kernel<2, 20, 64>" and "ACARS" in d["also"]["split16"]["data"]
    assert d["roofline_msk"]["bound"] == "issue" and d["roofline_msk"]["us_per_bit"] > 0 and c["roofline_msk"]["instr_per_bit"] == 275 and d["roofline_msk"]["kernel"] == "msk_lean_kernel"
    for name, a in d["also"].items():
        assert a["parity"]["dm_within_1e5_rel"] is True and a["parity"]["blocks_exact_given_gpu_dm"] is True
        assert a["parity"]["channels_checked"] == (256 if name == "wide" else 64)        # (wide: half a block per channel, so four times the channels)
        if name not in ("cs16", "f32", "split16"):        # (the exact-order mode restates rtl.c's u8 loop)
            assert a["parity"]["exact_order_mode"]["blocks_identical_end_to_end"] is True
        assert 0 < a["roofline"]["frac"] < 1 and 0 < a["whole_job_frac_of_hbm"] < 1 and a["value"] > 0
        # sustained timing: a step is several passes, the timed region lasts what --sustain asked for, value follows from it
        su = a["sustain"]
        assert su["passes_per_step"] >= 1 and a["timed_region_s"] >= 0.25 and len(su["step_ms_min_median_max"]) == 3     # (reps come from the burst rate)
        c = a["config"]
        want = c["channels_per_gpu"] * c["blocks_per_step"] * 1024 * c["decim"] * 3 / a["timed_region_s"] / 1e6       # (every channel counts, shared stream or not)
        assert abs(a["value"] - want) < 2e-3 * want and c["blocks_per_step"] == c["blocks_per_pass"] * su["passes_per_step"]
    # the stress and CS16 cases' gate channels carry ACARS (their block comparison is not vacuous at the real sizes; at this
    # test's 0.16 s of signal a block may or may not complete)
    assert "ACARS" in d["also"]["stress"]["data"] and "filter" in d["also"]["stress"]["config"]
    assert "ACARS" in d["also"]["cs16"]["data"] and d["also"]["cs16"]["config"]["input_format"] == "cs16"
    assert "ACARS" in d["also"]["f32"]["data"] and d["also"]["f32"]["config"]["input_format"] == "f32"
    assert d["parity"]["blocks"] > 0


def test_bench_gpus_flag_self_launch():
    """`python bench.py --gpus 2` with NO torchrun around it: the flag itself launches the ranks (here two ranks sharing
    the box's GPU over gloo, because RCCL wants one GPU per rank), rank 0 alone prints the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ACG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--channels", "256", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline", "--sustain", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["channels_total"] == 512 and len(d["per_gpu"]) == 2
    assert d["parity"]["blocks_exact_given_gpu_dm"] is True and d["parity"]["channels_checked"] == 64
    assert len(compact_json(r.stdout)["per_gpu"]) == 2
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
           "--warmup", "1", "--no-cpu-baseline", "--sustain", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["channels_per_gpu"] == 256 and d["config"]["channels_total"] == 512
    assert d["parity"]["blocks_exact_given_gpu_dm"] is True
    compact_json(r.stdout)                                                             # rank 0 alone reports: one result line
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(d["value"] - 2 * 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]


def test_bench_rccl_path_runs_with_world_size_one():
    """The RCCL code path on a one-GPU box: torch.distributed initialised with backend nccl and world size 1, and the channel
    scatter (two broadcasts of device tensors), the barrier(device_ids=...) pairs around the timed regions, the MAX / SUM
    all-reduces and the all-gather of the per-rank times all go through it instead of the world == 1 short-cuts -- the calls
    the 8-GPU run makes, executed once on hardware."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ACG_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rccl-selftest", "--channels", "256", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline", "--sustain", "0", "--no-ref-leg"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and "nccl" in d["config"]["collectives"] and d["parity"]["blocks_exact_given_gpu_dm"] is True
    assert abs(d["value"] - 256 * 8 * 1024 * 200 * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 1e-3 * d["value"]


def test_shard_collectives_over_rccl_world_one():
    """acarsdec_amd/shard.py's collectives on device tensors over nccl (= RCCL) with a world of one, in a child process."""
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from acarsdec_amd import shard\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "cfg = np.arange(40, dtype=np.float64).reshape(10, 4)\n"
        "mine = shard.scatter_channel_config(cfg, 1, 0, dist, device=dev, force=True)\n"
        "assert np.array_equal(mine, cfg)\n"
        "dist.barrier(device_ids=[0])\n"
        "t, c = shard.reduce_timing(1.25, 7, 1, dist, dev)\n"
        "assert (t, c) == (1.25, 7.0)\n"
        "assert shard.gather_scalars(2.5, 1, dist, dev) == [2.5]\n"
        "assert shard.gather_blocks([(0, 5, 0, b'ab', b'xyz', 9)], [3], 1, 0, dist) == [(3, 5, 0, b'ab', b'xyz', 9)]\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-800:], r.stderr[-2500:])
