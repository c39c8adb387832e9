"""The timed region of a bench case: warm-up, the burst, the sustained steps -- barrier + synchronize on both sides, the slowest
rank's clock, every rank on the same number of passes per step."""
import time

from .telemetry import gpu_clock_mhz, gpu_telemetry


def timed_region(step, drain, barrier, dec, steps, warmup, sustain_s, J, sync=None):
    """warm-up, the burst of `steps` single passes, then the reported region: `steps` steps of `reps` passes each, bracketed
    by barrier + synchronize on both sides; everything a step does is inside step() / drain() above: the process call(s) of
    the hot path and the collect of the delivered records.  Returns the raw clocks and counters; no probe, no switch.
    step() -> records delivered by one pass; drain() -> records still queued; barrier(): device synchronize + rank barrier;
    dec.timing() / dec.set_timing(): the library's event sums.  (bench.py --dry-run rehearses exactly this function, with a
    stub in place of the decoder, on 8 gloo ranks without a GPU: tests/test_shard_gloo.py.)"""
    import numpy as np
    from acarsdec_amd import shard
    world, cdev = J.world, J.cdev
    for _ in range(warmup):
        step()
    drain()                           # flush: the timed region starts with empty queues
    warm = dec.timing()               # event sums of warm-up: the demodulator's launches are timed here only --
    dec.set_timing(2)                 # in the timed region only the down-converter (roofline) is bracketed,
                                      # event records on the demodulator stream sit on its serial launch chain
    barrier()
    t0 = time.perf_counter()
    nfr_b = 0
    for _ in range(steps):
        nfr_b += step()
    nfr_b += drain()
    barrier()
    dt_burst = time.perf_counter() - t0
    tim_b = dec.timing()
    dt_burst, _ = shard.reduce_timing(dt_burst, nfr_b, world, J.coll, cdev)
    reps = 1
    if sustain_s > 0:
        reps = max(1, int(np.ceil(sustain_s / max(dt_burst, 1e-6))))
        if world > 1 or J.coll is not None:           # every rank must use the same reps
            reps = int(shard.reduce_timing(float(reps), 0.0, world, J.coll, cdev)[0])
    clk0 = gpu_clock_mhz(J.local)
    barrier()
    t0 = time.perf_counter()
    nfr = 0
    marks = [t0]
    clk_mid, tele_mid = None, None
    tele0 = gpu_telemetry(J.local)
    for k_ in range(steps):
        for _ in range(reps):
            nfr += step()
        marks.append(time.perf_counter())
        if k_ == steps // 2:
            clk_mid = gpu_clock_mhz(J.local)
            tele_mid = gpu_telemetry(J.local)
    clk1 = gpu_clock_mhz(J.local)              # the last call(s) are still running
    tele1 = gpu_telemetry(J.local)
    nfr += drain()                             # the last call's results: all K steps fully delivered inside the timed region
    if sync is not None:
        sync()                                 # this rank's device is idle: its OWN time (per_gpu), before it waits for the others
    dt_own = time.perf_counter() - t0
    barrier()
    dt_local = time.perf_counter() - t0        # ... and the job's time: behind the barrier, the slowest rank's
    tim = dec.timing()
    step_ms = sorted((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
    dt, nfr_total = shard.reduce_timing(dt_local, nfr, world, J.coll, cdev)
    per_rank = shard.gather_scalars(dt_own, world, J.coll, cdev)       # (round 5 gathered the post-barrier time: eight equal numbers)
    return dict(warm=warm, tim_b=tim_b, dt_burst=dt_burst, reps=reps, clk0=clk0, clk_mid=clk_mid, clk1=clk1, tele_mid=tele_mid, tele0=tele0, tele1=tele1, marks=marks,
                dt_local=dt_local, dt_own=dt_own, tim=tim, step_ms=step_ms, dt=dt, nfr_total=nfr_total, per_rank=per_rank)
