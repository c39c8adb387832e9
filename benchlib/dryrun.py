"""bench.py --dry-run: the N-rank launch path WITHOUT a GPU -- the real launcher (torch.distributed.run), the real channel
scatter / barriers / reductions (acarsdec_amd.shard over gloo), the real timed region (benchlib.timing.timed_region) and the real
result line (benchlib.line.compact_line), with a stub that sleeps where the decoder would run.  It measures nothing: the line
says "dry_run": true and carries no roofline numbers.  What it rehearses is what an 8-GPU node runs for the first time at
round end: rendezvous, every rank on the same passes per step, value = all ranks' work / the slowest rank's time, per_gpu in
rank order, and a rank that dies taking the whole launch down (tests/test_shard_gloo.py)."""
import os
import time


class StubDecoder:
    """what timed_region() needs of a Decoder: the event sums of the library (here: of the sleeps)"""
    def __init__(self):
        self.fir_ms, self.msk_ms, self.launches = 0.0, 0.0, 0

    def ran(self, seconds):
        self.fir_ms += 0.4e3 * seconds
        self.msk_ms += 0.9e3 * seconds
        self.launches += 1

    def set_timing(self, mode):
        pass

    def timing(self):
        out = dict(fir_ms=self.fir_ms, fir_launches=self.launches, msk_ms=self.msk_ms, msk_launches=self.launches)
        self.fir_ms, self.msk_ms, self.launches = 0.0, 0.0, 0
        return out


def run_case_dry(J, name, case, args, steps, warmup, headline):
    import numpy as np
    from acarsdec_amd import shard
    from .timing import timed_region
    dist, world, rank = J.dist, J.world, J.rank
    nch, M, ntaps, nblk = case["channels"], case["decim"], case["ntaps"], case["blocks"]
    nout = nblk * 1024
    # the per-channel configuration: made on rank 0 for ALL channels of the job, broadcast, every rank keeps its rows (as run_case)
    nch_total = nch * world
    cfg_rows = None
    if rank == 0:
        r0 = np.random.default_rng(0xACA25)
        off = r0.integers(-48, 49, size=nch_total) * 25000.0
        cfg_rows = np.stack([off, r0.uniform(0, 2 * np.pi, nch_total), np.zeros(nch_total), np.arange(nch_total, dtype=np.float64)], axis=1)
    mine = shard.scatter_channel_config(cfg_rows, world, rank, J.coll, device=None, force=J.coll is not None)
    own = shard.owned_channels(nch_total, rank, world)
    assert mine.shape[0] == nch and np.array_equal(mine[:, 3].astype(np.int64), own)
    dec = StubDecoder()
    # a pass "takes" 4 ms on rank 0 and 5 % more per rank: the slowest rank is the last one, by a known factor
    pass_s = float(os.environ.get("ACG_BENCH_DRY_PASS_MS", "4")) * 1e-3 * (1.0 + 0.05 * rank)
    die = int(os.environ.get("ACG_BENCH_DRY_DIE_RANK", "-1"))
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        if rank == die and calls["n"] == warmup + 2:          # (inside the burst: the other ranks are waiting at a barrier soon)
            os._exit(3)
        time.sleep(pass_s)
        dec.ran(pass_s)
        return nch // 64

    def drain():
        return 0

    def barrier():
        if J.coll is not None:
            dist.barrier()

    T = timed_region(step, drain, barrier, dec, steps, warmup, args.sustain, J)
    if rank != 0:
        return None
    reps, dt, per_rank = T["reps"], T["dt"], T["per_rank"]
    samples_per_step = nch * nout * M * reps
    value = world * samples_per_step * steps / dt / 1e6
    out = {"value": round(value, 1), "ms_per_step": round(dt / steps * 1e3, 4), "timed_region_s": round(dt, 4),
           "sustain": {"passes_per_step": reps, "step_ms_min_median_max": [round(T["step_ms"][0], 3), round(T["step_ms"][len(T["step_ms"]) // 2], 3),
                                                                           round(T["step_ms"][-1], 3)]},
           "data": "none: launcher rehearsal (--dry-run), a stub sleeps where the decoder would run",
           "config": {"workload": "DRY RUN of %s: %d channels/rank, no GPU work; rank r's pass takes %.1f ms x (1 + 0.05 r)" % (case["tag"], nch, pass_s * 1e3 / (1.0 + 0.05 * rank)),
                      "case": name, "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "blocks_per_pass": nblk, "passes_per_step": reps,
                      "blocks_per_step": nblk * reps, "channels_total": nch_total, "input_format": "u8", "callbacks_per_call": min(8, nblk),
                      "collect_lag": args.collect_lag, "delivered": "nothing (dry run)", "contexts": "none (dry run)",
                      "per_rank_seconds": [round(t, 4) for t in per_rank], "records_counted": int(T["nfr_total"])},
           "roofline": {"bound": "hbm", "kernel": "none (dry run)", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                        "traffic_src": None, "bytes_per_launch": None, "avg_launch_ms": None, "launches_per_step": None},
           "whole_job_frac_of_hbm": None, "time_dominant_kernel": "none (dry run)", "parity": None}
    if world > 1:
        out["per_gpu"] = [round(samples_per_step * steps / t / 1e6, 1) for t in per_rank]
    return out
