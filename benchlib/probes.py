"""Measurement aids of bench.py (--ab, --decoders): dead in the default run, kept out of the timed path."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter

from .telemetry import gpu_telemetry


def _probe_ab(args, J, step, drain, steps, nch, nout, M):
    """measurement aid (--ab): the same decoder, buffers and placement, timed again under each value of a per-launch switch in
    turn (acg_tune: ACG_FIR_VARIANT, ACG_MSK_LPC_LIVE, ...), two rounds -- not part of the reported value"""
    import torch
    from acarsdec_amd import _capi as K
    ab = {}
    ab_name, _, ab_vals = args.ab.rpartition("=")                # "5,55,8" or "ACG_MSK_LPC_LIVE=2,4"
    ab_name = ab_name or "ACG_FIR_VARIANT"
    names = ab_name.split("+")                                   # "A+B=a1:b1,a2:b2": two switches set together
    for rnd in range(2):
        for v in ab_vals.split(","):
            for n_, v_ in zip(names, v.split(":")):
                K.tune(n_, v_)
            step()
            drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            clk_ab = None
            for i_ in range(steps):
                step()
                if i_ == steps - 2:
                    clk_ab = gpu_telemetry(J.local)
            drain()
            torch.cuda.synchronize()
            ab.setdefault(v, []).append(round(nch * nout * M * steps / (time.perf_counter() - t1) / 1e6, 0))
            ab.setdefault(v + " telemetry", []).append(clk_ab)
    for n_ in names:
        K.tune(n_, os.environ.get(n_))
    return ab


def _probe_decoders(args, J, make_decoder, step, drain, steps, reps, dt_local, nch, nout, M, dev):
    """measurement aid (--decoders N): further decoders in the same process (each with its own allocations, all kept alive), the
    same input, timed the same way -- how much of the run-to-run spread is where the decoder's buffers happen to lie.  Probe
    switches: ACG_BENCH_SPACER_MB changes where the next decoder's buffers land without touching its streams;
    ACG_BENCH_DUMMY_STREAMS creates streams in between, which shifts the decoder's streams to other hardware queues without
    touching its memory; ACG_BENCH_DECODERS_ALT times each decoder once more under another FIR variant"""
    import torch
    from acarsdec_amd import _capi as K
    trials = [round(nch * nout * M * steps * reps / dt_local / 1e6, 0)]
    others, spacers, dummies, trials_alt = [], [], [], []
    for k in range(1, args.decoders):
        sp = int(os.environ.get("ACG_BENCH_SPACER_MB", "53"))
        if sp:
            spacers.append(torch.empty((((k * sp) << 20) + 4096 * k,), dtype=torch.uint8, device=dev))
        for _ in range(int(os.environ.get("ACG_BENCH_DUMMY_STREAMS", "0"))):
            dummies.append(torch.cuda.Stream(priority=-1))
            dummies.append(torch.cuda.Stream())
        d2 = make_decoder()
        others.append(d2)

        def timed():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                step(dec=d2)
            drain(d2)
            torch.cuda.synchronize()
            return round(nch * nout * M * steps / (time.perf_counter() - t1) / 1e6, 0)
        for _ in range(2):
            step(dec=d2)
        drain(d2)
        trials.append(timed())
        alt = os.environ.get("ACG_BENCH_DECODERS_ALT")
        if alt:
            K.tune("ACG_FIR_VARIANT", alt)
            step(dec=d2)
            drain(d2)
            trials_alt.append(timed())
            K.tune("ACG_FIR_VARIANT", os.environ.get("ACG_FIR_VARIANT"))
    for d2 in others:
        d2.close()
    return dict(default=trials, alt_variant=trials_alt) if trials_alt else trials
