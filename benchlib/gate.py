"""The parity gate of a bench case: the first pass from reset through the CPU checkers.  Nothing in here is timed; only this module
(and the cpu_baseline / rtl8 legs) touches oracle/."""
import time

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
#       blocks than those two builds differ from each other (no slack on top), and in NONE from the -Ofast build.
#   (5) the DELIVERED records: the pass once more from reset, collected as acg_msg (ACG_F_REPAIR + acg_collect_msgs), against
#       orc_blk_process + orc_msg_split of the oracle's blocks of (2): every field of every message, and no message of a
#       block that the reference's block thread drops (acars.c:124-207).
# With ACG_F_REPAIR (the default) "blocks" are what outputmsg() receives: checked / repaired, parity stripped, the dropped
# ones omitted -- on both sides (oracle: orc_blk_process; reference builds: what their blk_thread handed to outputmsg()).


def first_pass(cx):
    """the first pass from reset through the CPU checkers (the comment above); returns the parity record (rank 0) or None;
    raises SystemExit when the GPU output differs.  Nothing in here is timed."""
    import numpy as np
    from acarsdec_amd import decoder as D, _capi as K
    (args, J, name, rank, nch, share, fmt, M, taps, ntaps, iq, row, nout, cb, ncall, cb_bytes, stream, maxfr, repair, dec, step) = (
        getattr(cx, k) for k in ("args", "J", "name", "rank", "nch", "share", "fmt", "M", "taps", "ntaps", "iq", "row", "nout", "cb", "ncall", "cb_bytes",
                                 "stream", "maxfr", "repair", "dec", "step"))
    parity = None
    first = []
    ncheck = min(cx.check, nch) if rank == 0 else 0
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
        if fmt == 0 and ncheck and ncheck % share == 0:
            nsx = ncheck // share
            dx = D.Decoder(ncheck, decim=M, ntaps=ntaps, nstreams=nsx, max_blocks=cb, device=J.local, bitlog=False, exact_fir=True, repair=repair)
            dx.set_taps(taps[:ncheck])
            if share > 1:
                dx.set_channel_streams(np.arange(ncheck) // share)
            xfr, xdm_same = [], True
            for k in range(ncall):
                dx.in_callback(iq[:nsx, k * cb_bytes:(k + 1) * cb_bytes], nblocks=cb, pitch=row, stream=stream)
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
        front = {0: "rtl", K.FMT_CS16: "soapy", K.FMT_F32_REAL: "air", K.FMT_S16_SPLIT: "sdrplay" if M == 160 else None}.get(fmt)
        if front and ncheck and not args.no_ref_leg:
            rows_ = [host_rows[s_] for s_ in range((ncheck + share - 1) // share)]      # (several channels per dongle: each row once)
            row_of = [c // share for c in range(ncheck)]
            wf_ = [taps[c] for c in range(ncheck)]
            t_ref = time.perf_counter()
            which = "out" if repair else "raw"
            pick = lambda d: None if d is None else d[which]
            if front == "rtl":
                b_o2 = pick(O.ref_blocks("", rows_, M, wf_, row_of=row_of))
                b_fast, fast_label = pick(O.ref_blocks("_fast", rows_, M, wf_, row_of=row_of)), "-Ofast -march=native"
                if b_fast is None:
                    b_fast, fast_label = pick(O.ref_blocks("_v3", rows_, M, wf_, row_of=row_of)), "-Ofast -march=x86-64-v3"
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
                                   % (fast_label, {"rtl": "rtl.c in_callback", "soapy": "soapy.c reader loop", "air": "air.c rx_callback", "sdrplay": "sdrplay.c stream callback"}[front],
                                      "blocks as its blk_thread hands them to outputmsg()" if repair else "blocks as decodeAcars queues them"),
                            cpu_seconds=round(time.perf_counter() - t_ref, 1))
        # What the streaming path may differ from the IEEE oracle by: exactly what the reference's own -O2 and -Ofast builds differ
        # from each other on these bytes (MEASURED above; no slack on top of it -- VERDICT r04), and, where the -Ofast leg ran, NOT
        # AT ALL from the reference as shipped (its -Ofast build).  Without a reference leg (--no-ref-leg, oracle/_ref absent, a
        # window length the front end does not have) that yardstick is missing: a fixed bound of 1 % of the blocks (at least two:
        # what the reference's builds differ by at 2048 channels) then keeps (4) from ever being off -- (1)-(3)
        # already pin the only deviation to the rounding of (1).
        allowed = refs["ref_fast_vs_ref_o2_blocks_differing"] if refs else max(2, nblocks // 100)
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
