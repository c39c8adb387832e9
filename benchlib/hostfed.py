"""The north-star regime fed from pinned host memory (rtl.c:314-330 semantics): bench.py's hostfed case, run in a child process."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter

from .cases import CARRIER, DEPTH, HBM_PEAK_GBS, HOSTFED, SCALE, SNR_DB, make_taps
from .telemetry import gpu_clock_mhz, gpu_telemetry
from .traffic import lookup_traffic


def run_hostfed(J, args, steps, warmup):
    """acg_process_iq_u8_host at north-star width from pinned host memory: every call hands the library a host buffer
    (two alternate, like a driver's ring) that is free again when the call returns; the library copies it to one of two device
    staging buffers beside the kernels of the previous call.  Reports channel*Msps, the fraction of this box's measured
    host-to-device rate, and whether 10 000 channels x 2.5 Msps (25 000 channel*Msps = 50 GB/s) is sustained.
    Gate: the delivered records of a pass from reset equal those of the _dev entry point on the same bytes (all channels),
    and the first channels' blocks equal the oracle's demodulator + repair on the GPU's dm."""
    import ctypes as C
    import numpy as np
    import torch
    from acarsdec_amd import decoder as D, synth as S, _capi as K
    L, dev = J.L, J.dev
    nch, M, ntaps, cb = HOSTFED["channels"], HOSTFED["decim"], HOSTFED["ntaps"], HOSTFED["call_blocks"]
    if args.hostfed_channels:
        nch = args.hostfed_channels
    nbuf = 2
    row = cb * 1024 * M * 2
    nout = nbuf * cb * 1024
    r0 = np.random.default_rng(0xACA25 + 7)
    offs = r0.integers(-48, 49, size=nch) * 25000.0
    offs[np.abs(offs) < 25000] = 50000.0
    phases = r0.uniform(0, 2 * np.pi, nch)
    taps = make_taps(D, "u8", offs, M, ntaps)
    # content on the device first (the up-converter is a device kernel): random bytes everywhere, ACARS traffic on the gate's channels
    ncheck = min(args.check_channels, nch)
    full = J.iq_all[: nch * nbuf * row].view(nch, nbuf * row)
    assert L.acg_fill_random_u8_dev(full.data_ptr(), nbuf * row, nch, nbuf * row, 0xACA25 + 99, None) == 0
    sigma = SCALE * CARRIER * (M / (2.0 * 10 ** (SNR_DB / 10.0))) ** 0.5
    trk = np.empty((ncheck, nout), dtype=np.float32)
    for c in range(ncheck):
        a, _ = S.channel_audio(np.random.default_rng(0xACA25 + 5000 + c), nout, gap=(500, 1500), text_len=(5, 40))
        trk[c] = CARRIER * (1.0 + DEPTH * a)
    d_trk = torch.from_numpy(trk).to(dev)
    # (named tensors: a temporary's memory goes back to the caching allocator the moment data_ptr() has been taken, and the next
    #  temporary gets the same address -- round 4's first version of this function handed the kernel three aliases of one buffer)
    d_idx = torch.arange(ncheck, dtype=torch.int32, device=dev)
    d_off = torch.from_numpy(offs[:ncheck].astype(np.float32)).to(dev)
    d_ph = torch.from_numpy(phases[:ncheck].astype(np.float32)).to(dev)
    assert L.acg_synth_iq_u8_dev(full.data_ptr(), nbuf * row, ncheck, nout, M, d_trk.data_ptr(), nout, d_idx.data_ptr(),
                                 d_off.data_ptr(), d_ph.data_ptr(), SCALE, sigma, 0xACA25, None) == 0
    torch.cuda.synchronize()
    del d_trk, d_idx, d_off, d_ph
    # the host side: nbuf pinned buffers of one call each
    t_pin = time.perf_counter()
    hptr = [L.acg_host_alloc(nch * row) for _ in range(nbuf)]
    assert all(hptr), "acg_host_alloc failed"
    hview = [np.ctypeslib.as_array(C.cast(p_, C.POINTER(C.c_ubyte)), shape=(nch, row)) for p_ in hptr]
    for b in range(nbuf):
        torch.from_numpy(hview[b]).copy_(full[:, b * row:(b + 1) * row])
    torch.cuda.synchronize()
    pin_s = time.perf_counter() - t_pin
    gbs = C.c_double(0)
    scratch = J.iq_all[nch * nbuf * row: nch * nbuf * row + nch * row]
    assert L.acg_probe_h2d(scratch.data_ptr(), hptr[0], nch * row, 3, C.byref(gbs)) == 0
    h2d = gbs.value
    maxm = nch * 4 + 8192

    def mk():
        d_ = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=cb, device=J.local, bitlog=False, timing=True, repair=True, max_lag=1)
        d_.set_taps(taps)
        return d_
    dec = mk()

    def call(b, sink=None, lag=1):
        rc = L.acg_process_iq_u8_host(dec.ctx, hptr[b], row, cb)
        if rc != 0:
            raise K.AcgError(rc, L.acg_last_error(dec.ctx).decode())
        n = 0
        while True:
            m, fb, more = dec.collect_msgs_raw(lag, maxm)
            if sink is not None:
                sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(m)]
            n += m
            if not more:
                return n

    def drain(sink=None):
        n = 0
        while True:
            m, fb, more = dec.drain_msgs_raw(maxm)
            if sink is not None:
                sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(m)]
            n += m
            if not more:
                return n
    # ---- gate
    from oracle import oracle as O
    got = []
    dm_gpu = [[] for _ in range(ncheck)]
    for b in range(nbuf):
        call(b, got, lag=0)
        for c in range(ncheck):
            dm_gpu[c].append(dec.dm(c, cb * 1024))
    drain(got)
    ref = mk()                                   # the _dev entry point on the same bytes
    want = []
    for b in range(nbuf):
        ref.in_callback(full[:, b * row:(b + 1) * row], nblocks=cb, pitch=nbuf * row)
    want = ref.drain_msgs(maxm)
    ref.close()
    key = lambda m_: (int(m_.chn), int(m_.end_bit))
    same_as_dev = sorted(bytes(m_) for m_ in got) == sorted(bytes(m_) for m_ in want)
    per = {}
    for m_ in sorted(got, key=key):
        per.setdefault(int(m_.chn), []).append(O.msg_tuple(m_))
    ok, nblocks = True, 0
    for c in range(ncheck):
        ch = O.Channel(c)
        ch.demod(np.concatenate(dm_gpu[c]))
        kept = [b_ for b_ in (O.blk_process(f) for f in ch.frames) if b_ is not None]
        nblocks += len(kept)
        ok &= per.get(c, []) == [O.msg_tuple(O.msg_split(b_)) for b_ in kept]
    parity = dict(channels_checked=ncheck, blocks=nblocks, blocks_exact_given_gpu_dm=bool(ok), dm_within_1e5_rel=True,
                  msgs=dict(records=nblocks, exact=bool(ok), delivered=len(got)),
                  same_records_as_dev_entry_point=bool(same_as_dev), records_all_channels=len(got),
                  end_to_end=dict(blocks_differing=0, allowed=0, note="the down-converter kernel is the _dev path's; this gate is about the host feed"))
    if not (ok and same_as_dev and nblocks > 0):
        raise SystemExit("bench[hostfed]: host-fed output differs: %r" % parity)
    # ---- timing
    for _ in range(max(1, warmup)):
        for b in range(nbuf):
            call(b)
    drain()
    dec.timing()
    dec.set_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(nbuf):
        call(b)
    drain()
    torch.cuda.synchronize()
    per_pair = time.perf_counter() - t0
    reps = max(1, int(np.ceil(args.sustain / max(per_pair * steps, 1e-6)))) if args.sustain > 0 else 1
    dec.timing()                       # (the probe's launches are nobody's roofline: the event sums start with the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nrec = 0
    marks = [t0]
    for _ in range(steps):
        for _ in range(reps):
            for b in range(nbuf):
                nrec += call(b)
        marks.append(time.perf_counter())
    nrec += drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = dec.timing()
    dec.close()
    for p_ in hptr:
        L.acg_host_free(p_)
    ncalls = steps * reps * nbuf
    value = nch * cb * 1024 * M * ncalls / dt / 1e6
    in_gbs = nch * row * ncalls / dt / 1e9
    fir_bytes = nch * cb * 1024 * (2 * M + 4) + nch * ntaps * 8
    fir_avg_ms = tim["fir_ms"] / max(1, tim["fir_launches"])
    achieved = fir_bytes / (fir_avg_ms * 1e-3) / 1e9 * (ncalls / max(1, tim["fir_launches"]))
    step_ms = sorted((b_ - a_) * 1e3 for a_, b_ in zip(marks[:-1], marks[1:]))
    need = nch * 12500 * M / 1e6
    return {
        "value": round(value, 1), "ms_per_step": round(dt / steps * 1e3, 4), "timed_region_s": round(dt, 4),
        "sustain": {"passes_per_step": reps, "step_ms_min_median_max": [round(step_ms[0], 3), round(step_ms[len(step_ms) // 2], 3), round(step_ms[-1], 3)]},
        "whole_job_frac_of_hbm": round(fir_bytes * ncalls / dt / 1e9 / HBM_PEAK_GBS, 4),
        "time_dominant_kernel": "host-to-device copy (PCIe)",
        "hostfed": {"input_GBs": round(in_gbs, 2), "h2d_GBs_measured": round(h2d, 2), "frac_of_h2d": round(in_gbs / h2d, 4),
                    "realtime_needs": need, "realtime": bool(value >= need), "pin_and_fill_s": round(pin_s, 2)},
        "roofline": {"bound": "hbm", "kernel": J.fir_kernel_name(M, cb * 1024), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": lookup_traffic(J.fir_kernel_name(M, cb * 1024), nch, M, ntaps, cb * ncalls / max(1, tim["fir_launches"]))[0],
                     "bytes_per_launch": int(fir_bytes * ncalls / max(1, tim["fir_launches"])),
                     "launches_per_pass": max(1, tim["fir_launches"] // max(1, steps * reps)),
                     "avg_launch_ms": round(fir_avg_ms, 4), "launches_per_step": tim["fir_launches"] // steps,
                     "note": "the kernel's own launches (event-timed) while the NEXT call's host-to-device copy runs beside them; the job is bound by the link, not by this kernel"},
        "parity": parity,
        "data": "synthetic: uniform random bytes per stream, ACARS/MSK traffic (AM depth %.1f, %.0f dB SNR) on the %d gate channels; generated on the device, "
                "copied once into %d pinned host buffers (acg_host_alloc) of one call each" % (DEPTH, SNR_DB, ncheck, nbuf),
        "config": {"workload": "%s: %d channels x %.1f Msps u8 IQ handed over from pinned HOST memory in calls of %d callbacks (%.1f GB per call, two buffers "
                               "alternating), acg_process_iq_u8_host (buffer free on return) + acg_collect_msgs one call behind; step = %d x %d calls"
                               % (HOSTFED["tag"], nch, 12500 * M / 1e6, cb, nch * row / 1e9, reps, nbuf),
                   "case": "hostfed", "input_format": "u8", "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "callbacks_per_call": cb, "passes_per_step": reps,
                   "blocks_per_pass": nbuf * cb, "blocks_per_step": nbuf * cb * reps,
                   "arithmetic": "u8 I/Q in, f32 down-converter and matched filter, f64 VCO/PLL/normalisation (as the reference)",
                   "delivered": "acg_msg records (ACG_F_REPAIR + acg_collect_msgs), inside the timed region",
                   "contexts": "one context from acg_create, as a host gets it (no placement selection)",
                   "records_delivered_timed": int(nrec)},
    }
