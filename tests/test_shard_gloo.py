"""The N>1 path on CPU: world_size-2 gloo processes exercise the channel partition, the config
scatter, the block gather and the timing reduction that bench.py uses with RCCL.  The per-rank
"decoder" here is the oracle (test infrastructure) -- what is under test is the sharding logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from acarsdec_amd import shard


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_a_bijection():
    for nch, world in ((16384, 8), (1024, 2), (10, 4), (3, 8)):
        seen = np.concatenate([shard.owned_channels(nch, r, world) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(nch))
        for c in (0, nch - 1, nch // 2):
            r = shard.owner_of(c, world)
            assert shard.owned_channels(nch, r, world)[shard.local_index(c, world)] == c


def _worker(rank, world, port, nch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from acarsdec_amd import synth as S
    cfg = None
    if rank == 0:
        cfg = np.stack([np.arange(nch) * 1.0, 1000.0 + np.arange(nch)], axis=1)     # [seed, tag]
    mine = shard.scatter_channel_config(cfg, world, rank, dist)
    own = shard.owned_channels(nch, rank, world)
    assert mine.shape == (len(own), 2) and np.array_equal(mine[:, 0], own.astype(np.float64))
    blocks = []
    for li, row in enumerate(mine):
        rng = np.random.default_rng(int(row[0]))
        a, _ = S.channel_audio(rng, 12000, nframes=1, gap=(2000, 3000), text_len=(5, 20))
        ch = O.Channel(li)
        ch.demod(S.envelope(a))
        for f in ch.frames:
            blocks.append((li, int(f.len), bytes(f.txt[: f.len]), int(f.end_bit)))
    merged = shard.gather_blocks(blocks, own, world, rank, dist)
    t, c = shard.reduce_timing(0.5 + rank, len(blocks), world, dist)
    per = shard.gather_scalars(0.5 + rank, world, dist)
    if rank == 0:
        q.put((merged, t, c, per))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_shard_scatter_gather():
    world, nch = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nch, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, t, c, per = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process ground truth
    from oracle import oracle as O
    from acarsdec_amd import synth as S
    want = []
    for g in range(nch):
        rng = np.random.default_rng(g)
        a, _ = S.channel_audio(rng, 12000, nframes=1, gap=(2000, 3000), text_len=(5, 20))
        ch = O.Channel(0)
        ch.demod(S.envelope(a))
        want += [(g, int(f.len), bytes(f.txt[: f.len]), int(f.end_bit)) for f in ch.frames]
    assert merged == sorted(want, key=lambda b: (b[0], b[-1])) and len(merged) >= nch - 1
    assert t == 1.5 and c == len(merged)          # max over ranks, sum over ranks
    assert per == [0.5, 1.5]                      # every rank's own time, in rank order


def test_bench_gpus_flag_launches_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must start 2 ranks by itself -- the driver invokes
    exactly that.  Without a GPU the launcher refuses loudly (no silent single-rank run); with the gloo
    rehearsal backend it starts torch.distributed.run, whose ranks then stop at "needs a GPU"."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr, r.stderr[-800:]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=dict(env, ACG_BENCH_BACKEND="gloo"), timeout=600)
    # the ranks were started (torch.distributed.run reports a failed child) and got as far as the GPU check; the elastic agent may
    # kill the second rank before it has printed its own message, so one is enough
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") >= 1 and "ChildFailedError" in r.stderr, r.stderr[-1500:]
    # a rank count that contradicts the flag is an error, not a silent override
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _solo_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    cfg = np.arange(24, dtype=np.float64).reshape(6, 4)
    mine = shard.scatter_channel_config(cfg, 1, 0, dist, force=True)
    t, c = shard.reduce_timing(0.75, 3, 1, dist)
    per = shard.gather_scalars(0.75, 1, dist)
    merged = shard.gather_blocks([(1, 7, b"t", 11), (0, 5, b"s", 4)], [10, 20], 1, 0, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((mine.tolist(), t, c, per, merged))


def test_collectives_run_with_a_world_of_one():
    """bench.py --rccl-selftest sends the scatter / reductions / gathers through torch.distributed even with one rank
    (that is how the RCCL calls get executed on a one-GPU box); the same path over gloo here."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(free_port(), q))
    p.start()
    mine, t, c, per, merged = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert mine == np.arange(24, dtype=np.float64).reshape(6, 4).tolist() and (t, c, per) == (0.75, 3.0, [0.75])
    assert merged == [(10, 5, b"s", 4), (20, 7, b"t", 11)]


def _bench_dry(args, extra_env=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_eight_ranks_rehearsal_without_gpus():
    """the launcher's 8-rank path had never run with 8 processes.  `bench.py --gpus 8 --dry-run` is the
    real thing minus the device: bench.py launches 8 ranks itself (torch.distributed.run, rendezvous on 127.0.0.1), every rank gets
    its rows of the channel table (shard.scatter_channel_config over gloo), the REAL timed region runs (benchlib.timing.timed_region:
    barriers, the passes-per-step agreement, max-over-ranks clock, the gather of every rank's time) around a stub that sleeps
    -- rank r 5 % x r longer than rank 0 -- and rank 0 prints the REAL compact line.  Checked: 8 per_gpu entries in rank order and
    falling with the rank's sleep, value = all ranks' work / the slowest rank's time, every rank on the same passes per step, the
    "also" case (BASELINE configs[3]'s per-GPU share) with its own 8 entries, one line under 4 KB that says it measured nothing."""
    import json
    r = _bench_dry(["--gpus", "8", "--dry-run", "--steps", "4", "--warmup", "1", "--sustain", "0.4"])
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])
    detail = json.loads([l for l in lines if l.startswith("# bench_detail: ")][0][len("# bench_detail: "):])
    assert len(lines[-1]) < 4096 and line["dry_run"] is True and line["n_gpus"] == 8 and line["scaling"] == "weak"
    for rec, nch in ((detail, 1024), (detail["also"]["shard2048"], 2048)):
        pg, cfgd = rec["per_gpu"], rec["config"]
        assert len(pg) == 8 and cfgd["channels_total"] == 8 * nch and len(cfgd["per_rank_seconds"]) == 8
        secs = cfgd["per_rank_seconds"]
        # rank r sleeps 1 + 0.05 r times what rank 0 sleeps: the slowest is the last, and per_gpu says so
        assert secs.index(max(secs)) == 7 and pg.index(min(pg)) == 7 and pg[0] > pg[7]
        assert 1.25 < secs[7] / secs[0] < 1.45
        # value = the work of all ranks / the slowest rank's time (timed_region_s is that maximum)
        work = 8 * cfgd["channels_per_gpu"] * cfgd["blocks_per_step"] * 1024 * cfgd["decim"] * 4
        assert abs(rec["value"] - work / rec["timed_region_s"] / 1e6) < 2e-3 * rec["value"]
        assert 0 <= rec["timed_region_s"] - max(secs) < 0.05          # (the job's clock stops behind the barrier, the slowest rank's own just before it)
        # every rank ran the same number of passes: each rank's own rate x its own time is the same work
        assert all(abs(p * t - pg[0] * secs[0]) < 2e-3 * pg[0] * secs[0] for p, t in zip(pg, secs))
        assert cfgd["blocks_per_step"] == cfgd["blocks_per_pass"] * cfgd["passes_per_step"] and cfgd["passes_per_step"] >= 1
    assert line["per_gpu"] == detail["per_gpu"] or [int(round(x)) for x in detail["per_gpu"]] == line["per_gpu"]
    assert len(line["also"]["shard2048"]["per_gpu"]) == 8


def test_bench_rank_that_dies_takes_the_launch_down():
    """... and a rank that dies in the middle of the timed region (os._exit inside its third pass) must not leave seven ranks
    waiting at a barrier: the launcher exits non-zero, quickly, and prints no result line."""
    import time
    t0 = time.time()
    r = _bench_dry(["--gpus", "8", "--dry-run", "--steps", "4", "--warmup", "1", "--sustain", "0.4"], {"ACG_BENCH_DRY_DIE_RANK": "5"}, timeout=240)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
