"""One case of bench.py: input synthesis, the parity gate on the first pass, the timed region, the result record."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter

from .cases import CARRIER, COPY_CEILING_GBS, DEPTH, HBM_PEAK_GBS, SCALE, SNR_DB, make_taps
from .probes import _probe_ab, _probe_decoders
from .telemetry import gpu_clock_mhz, gpu_telemetry
from .traffic import lookup_traffic


def run_case(J, name, case, args, steps, warmup, headline):
    """Builds the input of one workload in J.iq, checks the first pass against the oracle, times `steps`
    steps.  Returns the dict that goes into the JSON line (rank 0) or None."""
    import numpy as np
    import torch
    from acarsdec_amd import decoder as D, synth as S, _capi as K, shard
    L, dist, world, rank, dev, cdev = J.L, J.dist, J.world, J.rank, J.dev, J.cdev
    nch, M, ntaps, nblk, content = case["channels"], case["decim"], case["ntaps"], case["blocks"], case["content"]
    fmt_name = case.get("format") or (args.format if headline else "u8")
    fmt = {"u8": 0, "cs16": K.FMT_CS16, "split16": K.FMT_S16_SPLIT, "f32": K.FMT_F32_REAL}[fmt_name]
    bps = 2 if fmt == 0 else 4
    share = max(1, args.share) if headline else 1
    if share > 1:
        assert fmt == 0 and nch % share == 0, "--share needs the u8 format and a channel count divisible by it"
        content = "random"
    if fmt != 0 and content != "format+acars":
        content = "format"
    nstreams = nch // share
    nout = nblk * 1024
    row = nout * M * bps
    assert nstreams * row <= J.iq_all.numel(), "input buffer too small for this case"
    iq = J.iq_all[: nstreams * row].view(nstreams, row)

    # ---- per-channel configuration: made on rank 0 for ALL channels of the job, scattered over RCCL
    # (the only data that ever crosses xGMI: 32 B per channel; inputs are generated where they are used)
    nch_total = nch * world
    cfg_rows = None
    if rank == 0:
        r0 = np.random.default_rng(0xACA25)
        off = r0.integers(-48, 49, size=nch_total) * 25000.0           # multiples of 12.5 kHz within +-1.2 MHz
        off[np.abs(off) < 25000] = 50000.0                              # >= 25 kHz from DC like chooseFc enforces
        cfg_rows = np.stack([off, r0.uniform(0, 2 * np.pi, nch_total), np.zeros(nch_total),
                             np.arange(nch_total, dtype=np.float64)], axis=1)
    mine = shard.scatter_channel_config(cfg_rows, world, rank, J.coll, device=cdev, force=J.coll is not None)
    own = shard.owned_channels(nch_total, rank, world)
    assert mine.shape[0] == nch and np.array_equal(mine[:, 3].astype(np.int64), own)
    offs, phases = mine[:, 0], mine[:, 1]
    taps = make_taps(D, fmt_name, offs, M, ntaps)

    # ---- input, resident in HBM: distinct content per channel, working set >> 256 MiB Infinity Cache
    sigma = SCALE * CARRIER * (M / (2.0 * 10 ** (SNR_DB / 10.0))) ** 0.5

    def synth_acars(n_first):
        """channels [0, n_first) of this rank: ACARS/MSK traffic, seeded 0xACA25 + global channel
        id: random printable frames of 20-220 characters every 0.25-1 s, AM depth 0.5, own carrier offset and phase, AWGN at
        20 dB SNR in the 12.5 kHz channel; modulated on the host (numpy), up-converted and quantised on the device."""
        trk = torch.empty((n_first, nout), dtype=torch.float32, device=dev)
        GEN = 512
        for c0 in range(0, n_first, GEN):
            n = min(GEN, n_first - c0)
            buf = np.empty((n, nout), dtype=np.float32)
            for i in range(n):
                a, _ = S.channel_audio(np.random.default_rng(0xACA25 + int(own[c0 + i])), nout, gap=(3125, 12500), text_len=(20, 220))
                buf[i] = CARRIER * (1.0 + DEPTH * a)
            trk[c0:c0 + n] = torch.from_numpy(buf).to(dev)
        d_idx = torch.arange(n_first, dtype=torch.int32, device=dev)
        d_off = torch.from_numpy(offs[:n_first].astype(np.float32)).to(dev)
        d_ph = torch.from_numpy(phases[:n_first].astype(np.float32)).to(dev)
        rc = L.acg_synth_iq_u8_dev(iq.data_ptr(), row, n_first, nout, M, trk.data_ptr(), nout, d_idx.data_ptr(),
                                   d_off.data_ptr(), d_ph.data_ptr(), SCALE, sigma, 0xACA25 + rank, None)
        assert rc == 0, rc
        torch.cuda.synchronize()

    if content == "acars":
        synth_acars(nch)
        data_desc = ("ACARS/MSK traffic on every channel, content seeded 0xACA25 + channel id (frames of 20-220 characters every "
                     "0.25-1 s), AM depth %.1f, carrier offset and phase per channel, AWGN at %.0f dB SNR in the 12.5 kHz channel "
                     "(sigma %.4f per I/Q sample); MSK modulator on the host, up-converter + u8 quantiser on the device" % (DEPTH, SNR_DB, sigma))
    elif content == "random+acars":
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
        nacars = min(nch, max(64, args.check_channels))
        synth_acars(nacars)
        data_desc = ("uniform random bytes, seeded per stream (SURVEY 8d config 5: the value distribution is irrelevant to bandwidth); "
                     "the first %d channels (the ones the parity gate looks at) carry ACARS/MSK traffic as in the other cases, so "
                     "that the gate compares decoded blocks and not only magnitudes" % nacars)
    elif content == "format+acars":
        assert fmt in (K.FMT_CS16, K.FMT_F32_REAL)
        if fmt == K.FMT_CS16:
            assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
            iq.view(torch.int16).bitwise_and_(0x0FFF)
        else:
            iq.view(torch.float32).normal_(0.0, 0.1)
        nacars = min(nch, max(64, args.check_channels))
        # the gate's channels: ACARS/MSK traffic as in the u8 cases, up-converted with torch on the device, a few channels at a time:
        # CS16 = complex baseband quantised to int16 (rint(32767 * 0.9 x), synth.iq_s16_from_envelopes); real f32 = 2 x env x cos
        # at the channel's offset from 0 Hz of the real spectrum (synth.real_f32_from_envelopes)
        tt = torch.arange(nout * M, dtype=torch.float64, device=dev) * (2.0 * np.pi / (12500.0 * M))
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xACA25 + rank)
        v16 = iq.view(torch.int16).view(nstreams, -1)
        v32 = iq.view(torch.float32).view(nstreams, -1)
        for c in range(nacars):
            a, _ = S.channel_audio(np.random.default_rng(0xACA25 + int(own[c])), nout, gap=(3125, 12500), text_len=(20, 220))
            env = torch.from_numpy((SCALE * CARRIER * (1.0 + DEPTH * a)).astype(np.float32)).to(dev).repeat_interleave(M)
            # (real f32: air.c mixes with Fc - Fr + rate / 4, air.c:278, i.e. the channel sits at its offset + a quarter of the rate)
            f_c = float(offs[c]) + (12500.0 * M / 4.0 if fmt == K.FMT_F32_REAL else 0.0)
            ph = torch.remainder(tt * f_c + float(phases[c]), 2.0 * np.pi).to(torch.float32)
            if fmt == K.FMT_CS16:
                xi = env * torch.cos(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
                xq = env * torch.sin(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
                v16[c, 0::2] = torch.round(32767.0 * 0.9 * xi).clamp_(-32768, 32767).to(torch.int16)
                v16[c, 1::2] = torch.round(32767.0 * 0.9 * xq).clamp_(-32768, 32767).to(torch.int16)
            else:
                v32[c] = 2.0 * env * torch.cos(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
        del tt, env, ph
        data_desc = ("%s; the first %d channels (the ones the parity gate looks at) carry ACARS/MSK traffic "
                     "as in the u8 cases (AM depth %.1f, %.0f dB SNR in the channel), generated on the device"
                     % ("uniform random 12-bit int16 samples" if fmt == K.FMT_CS16 else "gaussian float32 samples", nacars, DEPTH, SNR_DB))
    elif content == "random":
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
        data_desc = "uniform random bytes, seeded per stream (SURVEY 8d config 5: the value distribution is irrelevant to bandwidth)"
    else:
        if fmt == K.FMT_F32_REAL:
            iq.view(torch.float32).normal_(0.0, 0.1)
            data_desc = "gaussian float32 samples (format throughput run; blocks of this format are covered by tests/)"
        else:
            assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
            iq.view(torch.int16).bitwise_and_(0x0FFF)
            data_desc = "uniform random 12-bit int16 samples (format throughput run; blocks of this format are covered by tests/)"
    torch.cuda.synchronize()

    # The batch is streamed through the library in calls of `cb` callbacks (the reference hands over ONE callback at a
    # time, rtl.c:314; 8 keeps the 12.5 kHz intermediate of a call inside the Infinity Cache at 1024 channels).
    cb = min(args.call_blocks, nblk)
    while nblk % cb:
        cb -= 1
    ncall = nblk // cb
    if fmt == K.FMT_S16_SPLIT:
        cb, ncall = nblk, 1                                # (plane layout: one call)
    repair = not args.raw_blocks
    def make_decoder():
        # max_lag = --collect-lag: this host collects that many calls behind, and the block queue holds that many + 1 calls' worth
        d_ = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=nstreams, max_blocks=cb, device=J.local, bitlog=bool(args.bitlog), timing=True,
                       repair=repair, max_lag=max(1, args.collect_lag))
        d_.set_taps(taps)
        if share > 1:
            d_.set_channel_streams(np.arange(nch) // share)
        return d_
    stream = torch.cuda.current_stream().cuda_stream
    # The context a host gets from acg_create is the one that is timed (--placements 1, the default).  --placements N is a
    # DIAGNOSTIC: N contexts alive at once, acg_placement_trial on each, their times reported under config.placement; the
    # FIRST is still the one timed unless --placement-keep best (rounds 2-3 kept the fastest of four: selection).
    ntrial = args.placements if share == 1 else 1
    dec, trial_ms, trial_best = D.best_placed(make_decoder, ntrial, iq, cb, row, repeats=8 if nch > 2048 else 24, stream=stream, fmt=fmt,
                                              plane=row // 2 if fmt == K.FMT_S16_SPLIT else 0, keep=args.placement_keep)
    dec0 = dec
    maxfr = max(8192, int(nch * (cb / 3.0 + 2)))
    cb_bytes = cb * 1024 * M * bps

    def step(lag=None, sink=None, dec=None, dm_sink=None, frames=False):
        """one pass of the hot path over the batch; the results (acg_msg records; blocks with frames=True or --raw-blocks) are
        delivered to the host --collect-lag calls behind (default 2: the host then never waits for the block repair of the call
        before the newest one before it may hand over the next -- with a lag of 1 that wait sat between every two calls)"""
        lag = args.collect_lag if lag is None else lag
        n = 0
        dec = dec or dec0
        for k in range(ncall):
            part = iq[:, k * cb_bytes:(k + 1) * cb_bytes]
            if fmt == 0:
                dec.in_callback(part, nblocks=cb, pitch=row, stream=stream)
            else:
                dec.process_samples(fmt, part, cb, pitch=row, plane=row // 2 if fmt == K.FMT_S16_SPLIT else 0, stream=stream)
            if repair and not frames:
                # the delivered path: repaired blocks through outputmsg()'s field split, as acg_msg records
                m = 0
                while True:
                    mm, fb, more = dec.collect_msgs_raw(lag, maxfr)
                    if sink is not None:
                        sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(mm)]
                    m += mm
                    if not more:
                        break
            else:
                m, fb = dec.collect_frames_raw(lag, maxfr)
                if sink is not None:
                    sink += [K.Frame.from_buffer_copy(fb[i]) for i in range(m)]
            if dm_sink is not None:                     # (gate only) the 12.5 kHz samples this call's demodulator consumed
                for c in dm_sink:
                    dm_sink[c].append(dec.dm(c, cb * 1024))
            n += m
        return n

    def drain(dec=None):
        """everything still queued, through the delivered path; returns the count"""
        dec = dec or dec0
        if not repair:
            return dec.drain_frames_raw(maxfr)[0]
        m = 0
        while True:
            mm, _, more = dec.drain_msgs_raw(maxfr)
            m += mm
            if not more:
                return m

    def barrier():
        torch.cuda.synchronize()
        if J.coll is not None:
            dist.barrier(device_ids=[J.local]) if J.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness gate on the first pass (state starts from reset): a subset of this rank's channels goes through
    # the CPU checkers on the very bytes the GPU consumed.  SURVEY 8c's parity statement has two halves, and the gate
    # checks each of them and then closes the argument between them:
    #   (1) the 12.5 kHz magnitudes of EVERY call against the oracle's down-converter: |d dm| <= 1e-5 |dm| + 1e-6 full scale
    #       (the streaming kernel re-associates the sum; so does the reference's own -Ofast build);
    #   (2) the blocks against the oracle's demodulator + framing fed with the dm the GPU's demodulator consumed: BIT-EXACT
    #       (`blocks_exact_given_gpu_dm`: the demodulator and the framing are exact);
    #   (3) the same channels once more through the library in its exact-order mode (ACG_F_EXACT_FIR: rtl.c:335-353 in the
    #       reference's own order of operations): dm BIT-IDENTICAL to the oracle's, blocks identical END TO END -- so the only
    #       thing that can differ between the product path and the reference is the rounding of (1);
    #   (4) end to end with the streaming kernel (oracle down-converter -> oracle demodulator): a 1e-7 difference in dm can
    #       flip a soft decision that sits at |vo| < 1e-3 in a noise-only stretch, after which the two loops wander apart until
    #       the next preamble and one of them may lock a block late.  How often the reference's own builds do that to each
    #       other is MEASURED here: the same bytes and taps through the unmodified reference compiled -O2 (IEEE) and with its
    #       own flags (-Ofast -march=native), both from oracle/_ref.  The streaming path may differ from the oracle in no more
    #       blocks than those two builds differ from each other, plus one.
    #   (5) the DELIVERED records: the pass once more from reset, collected as acg_msg (ACG_F_REPAIR + acg_collect_msgs), against
    #       orc_blk_process + orc_msg_split of the oracle's blocks of (2): every field of every message, and no message of a
    #       block that the reference's block thread drops (acars.c:124-207).
    # With ACG_F_REPAIR (the default) "blocks" are what outputmsg() receives: checked / repaired, parity stripped, the dropped
    # ones omitted -- on both sides (oracle: orc_blk_process; reference builds: what their blk_thread handed to outputmsg()).
    def gate_first_pass():
        """the first pass from reset through the CPU checkers (the comment above); returns the parity record (rank 0) or None;
        raises SystemExit when the GPU output differs.  Nothing in here is timed."""
        parity = None
        first = []
        ncheck = min(args.check_channels, nch) if rank == 0 else 0
        dm_gpu = {c: [] for c in range(ncheck)}
        step(lag=0, sink=first, dm_sink=dm_gpu if ncheck else None, frames=True)
        msgs_first = []
        if repair:
            dec.reset()
            step(lag=0, sink=msgs_first)
        parity = None
        if rank == 0:
            from oracle import oracle as O

            def processed(frames):
                """the oracle's block thread on raw blocks: kept ones as OrcFrame (ACG_F_REPAIR), or the raw blocks themselves"""
                if not repair:
                    return list(frames)
                return [b for b in (O.blk_process(f) for f in frames) if b is not None]
            got = {}
            got_end = {}
            for f in first:
                got.setdefault(int(f.chn), []).append(D.frame_tuple(f))
                got_end.setdefault(int(f.chn), []).append(int(f.end_bit))
            got_msgs = {}
            for m_ in msgs_first:
                got_msgs.setdefault(int(m_.chn), []).append(O.msg_tuple(m_))
            ok, nblocks, dm_err, dm_ok = True, 0, 0.0, True
            msgs_ok, nmsgs, nraw, first_bad_msg = True, 0, 0, None
            e2e_blocks_off, e2e_channels_off = 0, []
            first_bad = None
            # absolute floor of the dm tolerance: 1e-6 of the largest term of the sum.  u8: |x - 127.37| / 127.5 <= 1; CS16:
            # 4095 / 32768; split planes (random 12-bit samples, |D| / 4): 4095 / 4; real f32: ~0.5
            dm_fullscale = {0: 1.0, K.FMT_CS16: 1.0, K.FMT_S16_SPLIT: 1024.0, K.FMT_F32_REAL: 1.0}[fmt]
            host_rows = iq[:(ncheck + share - 1) // share].cpu().numpy()
            dm_orc, e2e_want = [], []
            for c in range(ncheck):
                r = host_rows[c // share]
                if fmt == 0:
                    dm = O.fir_u8(r, M, taps[c], ntaps=ntaps)
                elif fmt == K.FMT_CS16:
                    dm = O.fir_cs16(r.view(np.int16), M, taps[c])
                elif fmt == K.FMT_S16_SPLIT:
                    h = r.view(np.int16)
                    dm = O.fir_split16(h[: h.size // 2], h[h.size // 2:], M, taps[c])
                else:
                    dm = O.fir_f32r(r.view(np.float32), M, taps[c])
                dm_orc.append(dm)
                g = np.concatenate(dm_gpu[c])
                e = np.abs(g - dm[: g.size])
                dm_ok &= bool(g.size == dm.size and np.all(e <= 1e-5 * np.abs(dm) + 1e-6 * dm_fullscale))
                dm_err = max(dm_err, float(e.max()))
                ch = O.Channel(c)
                ch.demod(g)                                         # (2): the oracle's demodulator on the GPU's dm
                nraw += len(ch.frames)
                kept = processed(ch.frames)
                want = [O.frame_tuple(f) for f in kept]
                nblocks += len(want)
                mine = got.get(c, [])
                if mine != want and first_bad is None:
                    k_ = next((i for i in range(min(len(mine), len(want))) if mine[i] != want[i]), min(len(mine), len(want)))
                    first_bad = dict(channel=c, gpu_blocks=len(mine), oracle_blocks=len(want), first_difference_at=k_,
                                     gpu_end_bits=got_end.get(c, []), oracle_end_bits=[int(f.end_bit) for f in ch.frames],
                                     gpu=repr(mine[k_])[:300] if k_ < len(mine) else None, oracle=repr(want[k_])[:300] if k_ < len(want) else None)
                ok &= mine == want
                if repair:                                          # (5): the delivered records, field for field
                    want_m = [O.msg_tuple(O.msg_split(b)) for b in kept]
                    nmsgs += len(want_m)
                    mine_m = got_msgs.get(c, [])
                    if mine_m != want_m and first_bad_msg is None:
                        first_bad_msg = dict(channel=c, gpu_msgs=len(mine_m), oracle_msgs=len(want_m))
                    msgs_ok &= mine_m == want_m
                ch2 = O.Channel(c)
                ch2.demod(dm)                                       # (4): oracle down-converter -> oracle demodulator
                want2 = [O.frame_tuple(f) for f in processed(ch2.frames)]
                e2e_want.append(want2)
                if mine != want2:
                    e2e_channels_off.append(c)
                    e2e_blocks_off += len(set(mine) ^ set(want2))
            # (3) the exact-order mode of the library on the same channels
            exact = None
            if fmt == 0 and share == 1 and ncheck:
                dx = D.Decoder(ncheck, decim=M, ntaps=ntaps, nstreams=ncheck, max_blocks=cb, device=J.local, bitlog=False, exact_fir=True, repair=repair)
                dx.set_taps(taps[:ncheck])
                xfr, xdm_same = [], True
                for k in range(ncall):
                    dx.in_callback(iq[:ncheck, k * cb_bytes:(k + 1) * cb_bytes], nblocks=cb, pitch=row, stream=stream)
                    for c in range(ncheck):
                        xdm_same &= bool(np.array_equal(dx.dm(c, cb * 1024).view(np.uint32), dm_orc[c][k * cb * 1024:(k + 1) * cb * 1024].view(np.uint32)))
                xgot = {}
                for f in dx.drain_frames(maxfr):
                    xgot.setdefault(int(f.chn), []).append(D.frame_tuple(f))
                dx.close()
                xoff = sum(len(set(xgot.get(c, [])) ^ set(e2e_want[c])) for c in range(ncheck))
                xsame = all(xgot.get(c, []) == e2e_want[c] for c in range(ncheck))
                exact = dict(dm_bit_identical_to_oracle=bool(xdm_same), blocks=sum(len(w) for w in e2e_want),
                             blocks_differing_end_to_end=int(xoff), blocks_identical_end_to_end=bool(xsame),
                             means="the library in ACG_F_EXACT_FIR mode (rtl.c:335-353 in the reference's order) -> the same GPU demodulator: "
                                   "everything identical to oracle down-converter -> oracle demodulator, so the streaming path's only deviation is "
                                   "the re-associated sum of its down-converter")
            # (4b) the reference's own builds against each other on the same bytes and taps: rtl.c in_callback for u8, soapy.c's reader
            # loop for CS16, air.c rx_callback for real f32 (oracle/_ref: the unmodified sources, -O2 and the reference's -Ofast)
            refs = None
            front = {0: "rtl", K.FMT_CS16: "soapy", K.FMT_F32_REAL: "air"}.get(fmt)
            if front and share == 1 and ncheck and not args.no_ref_leg:
                rows_ = [host_rows[c] for c in range(ncheck)]
                wf_ = [taps[c] for c in range(ncheck)]
                t_ref = time.perf_counter()
                which = "out" if repair else "raw"
                pick = lambda d: None if d is None else d[which]
                if front == "rtl":
                    b_o2 = pick(O.ref_blocks("", rows_, M, wf_))
                    b_fast, fast_label = pick(O.ref_blocks("_fast", rows_, M, wf_)), "-Ofast -march=native"
                    if b_fast is None:
                        b_fast, fast_label = pick(O.ref_blocks("_v3", rows_, M, wf_)), "-Ofast -march=x86-64-v3"
                else:
                    b_o2 = pick(O.ref_blocks("_" + front, rows_, M, wf_, front=front))
                    b_fast, fast_label = pick(O.ref_blocks("_%s_fast" % front, rows_, M, wf_, front=front)), "-Ofast -march=x86-64-v3"
                if b_o2 is not None and b_fast is not None:
                    strip = lambda lst: [t[1:] for t in lst]
                    refs = dict(o2_blocks=sum(len(x) for x in b_o2), ofast_blocks=sum(len(x) for x in b_fast),
                                ref_fast_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(y)) for x, y in zip(b_o2, b_fast)),
                                oracle_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(strip(y))) for x, y in zip(b_o2, e2e_want)),
                                gpu_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(strip(got.get(c, [])))) for c, x in enumerate(b_o2)),
                                gpu_vs_ref_ofast_blocks_differing=sum(len(set(x) ^ set(strip(got.get(c, [])))) for c, x in enumerate(b_fast)),
                                builds="oracle/_ref (-O2, IEEE) vs the reference's own flags (%s): unmodified %s + msk.c + "
                                       "acars.c (%s) on the GPU's input bytes and tap tables, one channel per pass, each build in a child interpreter"
                                       % (fast_label, {"rtl": "rtl.c in_callback", "soapy": "soapy.c reader loop", "air": "air.c rx_callback"}[front],
                                          "blocks as its blk_thread hands them to outputmsg()" if repair else "blocks as decodeAcars queues them"),
                                cpu_seconds=round(time.perf_counter() - t_ref, 1))
            # What the streaming path may differ from the IEEE oracle by: exactly what the reference's own -O2 and -Ofast builds differ
            # from each other on these bytes (MEASURED above; no slack on top of it -- VERDICT r04), and, where the -Ofast leg ran, NOT
            # AT ALL from the reference as shipped (its -Ofast build).  Without a reference leg (split planes, shared streams,
            # --no-ref-leg, oracle/_ref absent) that yardstick is missing: (4) is then reported, not enforced -- (1)-(3) are, and they
            # already pin the only deviation of the streaming path to the rounding of (1).
            allowed = refs["ref_fast_vs_ref_o2_blocks_differing"] if refs else None
            parity = dict(channels_checked=ncheck, blocks=nblocks, blocks_exact_given_gpu_dm=bool(ok),
                          blocks_are=("what outputmsg() receives: checked / repaired by the device (ACG_F_REPAIR, acars.c:93-215), parity stripped, "
                                      "dropped blocks omitted" if repair else "as decodeAcars queues them (pre-repair, --raw-blocks)"),
                          raw_blocks_before_repair=nraw,
                          blocks_exact_given_gpu_dm_means="blocks identical to the oracle's demodulator + framing (+ block repair) fed with the dm the GPU's demodulator consumed",
                          msgs=(dict(records=nmsgs, exact=bool(msgs_ok), delivered=len(msgs_first),
                                     means="acg_msg records of acg_collect_msgs (a second pass from reset) == orc_msg_split(orc_blk_process(block)) field for field "
                                           "(output.c:486-560)") if repair else None),
                          dm_within_1e5_rel=bool(dm_ok), dm_max_abs_err=dm_err, dm_samples_per_channel=nout,
                          exact_order_mode=exact,
                          end_to_end=dict(blocks_differing=e2e_blocks_off, channels=e2e_channels_off, exact=bool(e2e_blocks_off == 0),
                                          allowed=allowed, allowed_means="what the reference's -O2 and -Ofast builds differ by on this input (measured in this run); "
                                                                         "and zero against the reference's -Ofast build",
                                          gpu_vs_ref_ofast=(refs["gpu_vs_ref_ofast_blocks_differing"] if refs else None),
                                          note="streaming down-converter -> GPU demodulator against oracle down-converter -> oracle demodulator; a differing "
                                               "block = a razor-edge soft decision (|vo| < 1e-3 in noise) flipped by the 1e-7 re-association of dm"),
                          reference_builds=refs,
                          blocks_first_pass_all_channels=len(first))
            bad = (not (ok and dm_ok and msgs_ok) or (allowed is not None and e2e_blocks_off > allowed) or
                   (refs is not None and (refs["gpu_vs_ref_ofast_blocks_differing"] != 0 or refs["oracle_vs_ref_o2_blocks_differing"] != 0)) or
                   (exact is not None and not (exact["dm_bit_identical_to_oracle"] and exact["blocks_identical_end_to_end"])))
            if bad:
                raise SystemExit("bench[%s]: GPU output differs from the oracle: %r; first mismatch: %r %r" % (name, parity, first_bad, first_bad_msg))
        return parity

    parity = gate_first_pass()

    # ---- timing.  A "pass" = the hot path once over the resident batch (the step of rounds 1-2).  `burst`: `steps` single
    # passes, timed as before (about half a second at the headline case: too short to be seen by an outside observer, and
    # inside the window in which the shader clock has not settled).  The reported `value` is SUSTAINED: a step is `reps`
    # passes, reps chosen from the burst rate so that `steps` steps take >= --sustain seconds; per-step times (host clock at
    # the step boundaries, no extra synchronisation: the host runs at most one call ahead of the device) give min / median /
    # max, the shader clock is read from sysfs while the device is still busy.
    def timed_region():
        """warm-up, the burst of `steps` single passes, then the reported region: `steps` steps of `reps` passes each, bracketed
        by barrier + synchronize on both sides; everything a step does is inside step() / drain() above: the process call(s) of
        the hot path and the collect of the delivered records.  Returns the raw clocks and counters; no probe, no switch."""
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
        if args.sustain > 0:
            reps = max(1, int(np.ceil(args.sustain / max(dt_burst, 1e-6))))
            if world > 1 or J.coll is not None:           # every rank must use the same reps
                reps = int(shard.reduce_timing(float(reps), 0.0, world, J.coll, cdev)[0])
        clk0 = gpu_clock_mhz(J.local)
        barrier()
        t0 = time.perf_counter()
        nfr = 0
        marks = [t0]
        clk_mid, tele_mid = None, None
        for k_ in range(steps):
            for _ in range(reps):
                nfr += step()
            marks.append(time.perf_counter())
            if k_ == steps // 2:
                clk_mid = gpu_clock_mhz(J.local)
                tele_mid = gpu_telemetry(J.local)
        clk1 = gpu_clock_mhz(J.local)              # the last call(s) are still running
        nfr += drain()                             # the last call's results: all K steps fully delivered inside the timed region
        barrier()
        dt_local = time.perf_counter() - t0
        tim = dec.timing()
        step_ms = sorted((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
        dt, nfr_total = shard.reduce_timing(dt_local, nfr, world, J.coll, cdev)
        per_rank = shard.gather_scalars(dt_local, world, J.coll, cdev)
        return dict(warm=warm, tim_b=tim_b, dt_burst=dt_burst, reps=reps, clk0=clk0, clk_mid=clk_mid, clk1=clk1, tele_mid=tele_mid,
                    dt_local=dt_local, tim=tim, step_ms=step_ms, dt=dt, nfr_total=nfr_total, per_rank=per_rank)

    T = timed_region()
    warm, tim_b, dt_burst, reps, clk0, clk_mid, clk1, tele_mid = (T[k] for k in ('warm', 'tim_b', 'dt_burst', 'reps', 'clk0', 'clk_mid', 'clk1', 'tele_mid'))
    dt_local, tim, step_ms, dt, nfr_total, per_rank = (T[k] for k in ('dt_local', 'tim', 'step_ms', 'dt', 'nfr_total', 'per_rank'))
    # measurement aids (--ab, --decoders: same-process A/B of a per-launch switch, further decoders in the same process); not part
    # of the reported value, dead in the default run, and kept out of this function (the timed path must be auditable)
    ab = _probe_ab(args, J, step, drain, steps, nch, nout, M) if (args.ab and world == 1) else None
    trials = (_probe_decoders(args, J, make_decoder, step, drain, steps, reps, dt_local, nch, nout, M, dev)
              if (args.decoders > 1 and world == 1) else None)
    dec.close()
    if rank != 0:
        return None

    samples_per_pass = nch * nout * M                               # complex input samples per GPU per pass over the batch
    samples_per_step = samples_per_pass * reps
    value = world * samples_per_step * steps / dt / 1e6             # channel * Msamples/s
    # algorithmic bytes: 2 B per input sample per channel read (bps for the other formats), 4 B per
    # 12.5 kHz output written, taps (8 B each) read once per launch.  A step is `lps` pipelined FIR launches.
    lps = max(1, round(tim["fir_launches"] / (steps * reps)))       # launches per PASS
    pass_bytes = nstreams * nout * bps * M + nch * nout * 4 + lps * nch * ntaps * 8      # shared-stream mode: a stream's bytes count once
    step_bytes = pass_bytes * reps
    fir_bytes = pass_bytes / lps
    fir_avg_ms = tim["fir_ms"] / max(1, tim["fir_launches"])
    achieved = fir_bytes / (fir_avg_ms * 1e-3) / 1e9
    fir_ms_step = tim["fir_ms"] / steps
    burst = {"value": round(world * samples_per_pass * steps / dt_burst / 1e6, 1), "ms_per_pass": round(dt_burst / steps * 1e3, 4),
             "timed_region_s": round(dt_burst, 4), "whole_job_frac_of_hbm": round(pass_bytes * steps / dt_burst / 1e9 / HBM_PEAK_GBS, 4),
             "roofline_frac": round(fir_bytes / (tim_b["fir_ms"] / max(1, tim_b["fir_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "note": "`steps` single passes over the batch from a cold-ish device, as rounds 1-2 timed them; not the reported value"}
    msk_ms_step = warm["msk_ms"] / (warmup + (2 if repair else 1)) * reps      # (the gate's one or two passes are in the sum)
    if fmt == 0:
        kname = "fir_u8_shared_kernel" if share > 1 else J.fir_kernel_name(M, nout)
    else:
        # (mirrors acg_launch_fir_fmt: the wave-private kernel <FMT, 16-byte chunks per window (per plane), windows per tile> where it is
        #  instantiated for the window length, else round 1's workgroup-granular kernel)
        fid = {"cs16": 1, "split16": 2, "f32": 3}[fmt_name]
        shape = {("cs16", 160): (40, 32), ("cs16", 192): (48, 32), ("cs16", 200): (50, 32), ("f32", 200): (50, 32), ("f32", 240): (60, 16),
                 ("f32", 480): (120, 8), ("f32", 800): (200, 8), ("split16", 160): (20, 64)}.get((fmt_name, M))
        kname = ("fir_fmt_direct_kernel<%d, %d, %d>" % ((fid,) + shape)) if shape else "fir_fmt_kernel<%d>" % fid
    # HBM traffic of this launch shape from the committed PMC passes (rocprofv3 cannot run inside the timed
    # process): looked up by the full kernel signature and launch shape, not measured in this run -- the source is named next to the number
    traffic, traffic_src = lookup_traffic(kname, nch, M, ntaps, nblk / lps) if share == 1 else (None, None)
    whole = step_bytes * steps / dt / 1e9                            # per GPU
    out = {
        "value": round(value, 1),
        "ms_per_step": round(dt / steps * 1e3, 4),
        "timed_region_s": round(dt, 4),
        "sustain": {"passes_per_step": reps, "step_ms_min_median_max": [round(step_ms[0], 3), round(step_ms[len(step_ms) // 2], 3), round(step_ms[-1], 3)],
                    "shader_clock_mhz_start_mid_end": [clk0, clk_mid, clk1], "telemetry_mid_run": tele_mid,
                    "note": "a step = passes_per_step passes over the resident batch (chosen from the burst rate so that the timed region lasts "
                            ">= --sustain seconds); step times from host time stamps at the step boundaries (the host runs at most one call ahead "
                            "of the device); clocks from sysfs while the device is busy (null where the box does not expose them)"},
        "burst": burst,
        "data": "synthetic: " + data_desc,
        "config": {"workload": "%s: %d channels/GPU x %.1f Msps %s, one stream per channel, rtlMult=%d, ntaps=%d; step = %d pass(es) over a resident batch of "
                               "%d callbacks/channel in calls of %d; FIR decimate + MSK demod + framing%s, delivered to the host %d call(s) behind"
                               % (case["tag"], nch, 12500 * M / 1e6, {"u8": "u8 IQ", "cs16": "CS16 IQ", "split16": "split int16 I/Q", "f32": "real f32"}[fmt_name],
                                  M, ntaps, reps, nblk, cb, " + block repair + message split" if repair else "", args.collect_lag),
                   "signal_seconds_per_pass": round(nblk * 0.08192, 3),
                   "callbacks_per_call": cb, "collect_lag": args.collect_lag, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "case": name, "input_format": fmt_name, "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "blocks_per_step": nblk * reps, "blocks_per_pass": nblk, "passes_per_step": reps,
                   "input_bytes_per_gpu": int(nstreams * row),
                   "realtime_channels_equiv": int(value / (12500 * M / 1e6)),
                   "arithmetic": "%s in, f32 down-converter and matched filter, f64 VCO/PLL/normalisation (as the reference)"
                                 % {"u8": "u8 I/Q", "cs16": "int16 I/Q", "split16": "split int16 I/Q", "f32": "real f32"}[fmt_name],
                   "delivered": ("acg_msg records: blocks checked / repaired on the device (ACG_F_REPAIR, acars.c:93-215) and split into outputmsg()'s fields "
                                 "(output.c:486-560) by acg_collect_msgs, inside the timed region" if repair else
                                 "pre-repair blocks (acg_collect_frames, --raw-blocks)"),
                   "channels_total": nch_total, "blocks_decoded_timed": int(nfr_total),
                   "contexts": "one context from acg_create, as a host gets it (no placement selection)" if not trial_ms or trial_best == 0 else "best of %d contexts (--placement-keep best)" % len(trial_ms),
                   "placement": ({"contexts_tried": len(trial_ms), "ms_per_call": [round(x, 3) for x in trial_ms], "kept": trial_best,
                                  "spread": round(max(trial_ms) / min(trial_ms) - 1.0, 4),
                                  "fir_ms_per_launch": ([round(x, 4) for x in D.best_placed.last_fir_ms] if getattr(D.best_placed, "last_fir_ms", None) else None),
                                  "note": "diagnostic (--placements N), untimed: N contexts alive at once, acg_placement_trial on each after a warm-up round; "
                                          "`kept` is the one timed (0 = the first, unless --placement-keep best)"}
                                 if trial_ms else None)},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": (traffic_src + " (rocprofv3 PMC passes of the same launch shape: 2 x FETCH_SIZE + WRITE_SIZE; "
                                        "looked up by the full kernel signature, not collected in this run)") if traffic else None,
                     "bytes_per_launch": int(fir_bytes), "avg_launch_ms": round(fir_avg_ms, 4), "launches_per_step": lps * reps,
                     "launches_per_pass": lps,
                     "timing": "HIP events around every launch of the kernel on its own stream, inside the timed region "
                               "(the demodulator of the previous call / chunk runs beside it)",
                     "frac_of_measured_copy_ceiling_6290": round(achieved / COPY_CEILING_GBS, 4)},
        "whole_job_frac_of_hbm": round(whole / HBM_PEAK_GBS, 4),
        "whole_job_GBs_per_gpu": round(whole, 1),
        "time_dominant_kernel": "msk_demod_kernel" if msk_ms_step > fir_ms_step else kname,
        "kernels": {"fir_ms_per_step": round(fir_ms_step, 4), "msk_ms_per_step": round(msk_ms_step, 4),
                    "note": "per-step sums of event-timed launches; the stages overlap (down-converter of call/chunk i+1 beside the "
                            "demodulator of i); the demodulator figure is taken during warm-up (its events are off in the timed region)"},
        "parity": parity,
    }
    if ab:
        out["ab_same_process"] = ab
    if trials:
        out["placement_trials"] = trials
    if ntaps != M:
        out["config"]["filter"] = ("%d-tap low-pass = the channel's NCO taps (rtl.c:283-286) x Hamming window, unit DC gain; the reference "
                                   "only has the boxcar, so the oracle for this filter is the same sum(vb*wf) formula with these taps" % ntaps)
    if world > 1:
        out["per_gpu"] = [round(samples_per_step * steps / t / 1e6, 1) for t in per_rank]
    if J.coll is not None and world == 1:
        out["config"]["collectives"] = "forced through torch.distributed/%s with world size 1 (--rccl-selftest)" % J.backend
    if share > 1:
        out["config"]["channels_per_stream"] = share
        out["roofline"]["note"] = ("shared-stream mode: %d channels reuse each stream's bytes, the down-converter is VALU-bound "
                                   "(8*K flop per 2 B); achieved counts each stream once and is NOT the HBM roofline figure" % share)
        keff = min(share, 8)
        ops = nch * nout * M * (2.0 + 3.0 / keff) * steps / (tim["fir_ms"] * 1e-3)
        out["valu"] = {"kernel": "fir_u8_shared_kernel", "lane_ops_per_s": round(ops, 0), "peak": 256 * 4 * 16 * 2.4e9,
                       "frac": round(ops / (256 * 4 * 16 * 2.4e9), 4), "lane_ops_per_channel_sample": round(2.0 + 3.0 / keff, 3)}
    return out
