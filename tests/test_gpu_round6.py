"""Round 6: the shared-stream down-converter on the matrix pipe (fir_mm.hip: rtl.c:344-354 with K channels per dongle as an
exact int8 contraction), the legacy view's context made at initMsk() time, and the configurations that were never timed before
(rtlMult 160 / 192, SDRplay planes) through the bench gate.  All through the C ABI; the oracle is the checker."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    from acarsdec_amd import decoder
    from acarsdec_amd import _capi as K
    assert K.load().acg_device_count() > 0, "GPU tests need a GPU; the library has no CPU fallback"
    return decoder


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def S():
    from acarsdec_amd import synth
    return synth


def exact_dm(iq_row, M, taps, nout):
    """|sum (u8 - 127.37f) w| of rtl.c:335-353 with every product and the whole sum in f64: what an infinitely precise
    evaluation of the reference's expression gives (its f32 operands taken as they are)."""
    x = iq_row[: nout * M * 2].astype(np.float64).reshape(nout, M, 2) - np.float64(np.float32(127.37))
    w = np.zeros((M, 2), dtype=np.float64)
    w[: taps.shape[0]] = taps.astype(np.float64)
    re = x[:, :, 0] @ w[:, 0] - x[:, :, 1] @ w[:, 1]
    im = x[:, :, 0] @ w[:, 1] + x[:, :, 1] @ w[:, 0]
    return np.hypot(re, im)


@pytest.mark.parametrize("M,ntaps", [(200, 200), (160, 160), (192, 192), (200, 192), (160, 37)])
def test_matrix_pipe_shared_stream_kernel(D, O, M, ntaps, tune):
    """fir_u8_mm_kernel: streams feeding 1, 3, 8, 11 and 16 channels in scrambled channel order (groups of <= 8, ragged), the
    extremes of the u8 range, tap tables replaced between calls.  dm within the 1e-5 bar of the oracle, within 2e-7 of the
    f64-exact value of the reference's expression (the kernel rounds the sum ONCE), and the vector-pipe kernel it replaces
    (ACG_FIR_MM=0) agrees with it within the bar."""
    rng = np.random.default_rng(1000 * M + ntaps)
    sizes = [1, 3, 8, 11, 16, 1]
    smap = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(smap)
    nch, nblk = int(smap.size), 2
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(len(sizes), nout * M * 2), dtype=np.uint8)
    iq[0, : 4 * M] = 0
    iq[1, : 4 * M] = 255
    iq[2, : 2 * M] = 128
    iq[4, 2 * M: 6 * M: 2] = 255
    iq[4, 2 * M + 1: 6 * M: 2] = 0
    taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
    for c in range(nch):
        taps[c] = O.rtl_taps(131000000 + 25000 * int(rng.integers(-40, 41)), 131000000, M)[:ntaps]
    taps[3] *= np.float32(1.0 / 512)                 # a table 2^9 below the others: its own scale
    taps[5, 1::2] = 0                                # exact zeros inside a table

    def run():
        dec = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=len(sizes), max_blocks=nblk)
        dec.set_taps(taps)
        dec.set_channel_streams(smap)
        dec.in_callback(iq)
        out = np.stack([dec.dm(c, nout) for c in range(nch)])
        dec.set_taps(taps[::-1].copy())             # the digit images must follow the tap tables
        dec.in_callback(iq)
        out2 = np.stack([dec.dm(c, nout) for c in range(nch)])
        dec.close()
        return out, out2

    mm, mm2 = run()
    tune("ACG_FIR_MM", "0")
    valu, valu2 = run()
    tune("ACG_FIR_MM", None)
    assert not np.array_equal(mm, valu)              # (two different kernels ran)
    worst, worst_valu = 0.0, 0.0
    for c in range(nch):
        for got, tp, other in ((mm[c], taps[c], valu[c]), (mm2[c], taps[nch - 1 - c], valu2[c])):
            want = O.fir_u8(iq[smap[c]], M, tp, nout=nout, ntaps=ntaps)
            assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6), c
            assert np.all(np.abs(got - other) <= 1e-5 * np.abs(other) + 1e-6), c
            ex = exact_dm(iq[smap[c]], M, tp, nout)
            err = np.abs(got.astype(np.float64) - ex)
            assert np.all(err <= 2e-7 * ex + 1e-9), (c, float((err / (ex + 1e-9)).max()))
            worst = max(worst, float((err / (ex + 1e-3)).max()))
            worst_valu = max(worst_valu, float((np.abs(other.astype(np.float64) - ex) / (ex + 1e-3)).max()))
    # the one-rounding kernel sits closer to the exact value than the f32 evaluation order of the vector-pipe kernel
    assert worst < 2e-7 and worst < worst_valu, (worst, worst_valu)


def test_matrix_pipe_blocks_one_dongle_sixteen_channels(D, O, S):
    """one 2.0 Msps stream, 16 channels (two groups), six callbacks handed over one at a time and as one call: blocks per channel
    identical to oracle down-converter -> oracle demodulator wherever the dm agree to 1e-5 -- and identical between the two
    chunkings bit for bit (the kernel has no chunk-dependent arithmetic)."""
    rng = np.random.default_rng(66)
    M, nch, nblk = 160, 16, 6
    nout = nblk * 1024
    fr = [131.0e6 + 25000.0 * k for k in (-20, -14, -9, -6, -4, -2, 2, 3, 5, 7, 9, 12, 15, 18, 21, 24)]
    fc = 131.0e6
    env = []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, nframes=2, gap=(800, 1500), text_len=(10, 40))
        env.append(0.5 * (1 + 0.5 * a))
    iq = S.iq_u8_from_envelopes(np.array(env), M, [f - fc for f in fr], phases=list(rng.uniform(0, 6.28, nch)), noise=0.004, rng=rng,
                                scale=0.06)
    taps = np.stack([O.rtl_taps(int(f), int(fc), M) for f in fr])

    def run(chunks):
        dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
        dec.set_taps(taps)
        dms = [[] for _ in range(nch)]
        per = nblk // chunks
        for k in range(chunks):
            dec.in_callback(iq.reshape(1, -1)[:, k * per * 1024 * M * 2:(k + 1) * per * 1024 * M * 2], nblocks=per)
            for c in range(nch):
                dms[c].append(dec.dm(c, per * 1024))
        got = {}
        for f in dec.drain_frames():
            got.setdefault(int(f.chn), []).append(D.frame_tuple(f))
        dec.close()
        return got, [np.concatenate(d) for d in dms]

    got1, dm1 = run(1)
    got6, dm6 = run(6)
    assert got1 == got6 and all(np.array_equal(a, b) for a, b in zip(dm1, dm6))
    total = 0
    for c in range(nch):
        want_dm = O.fir_u8(iq, M, taps[c])
        assert np.all(np.abs(dm1[c] - want_dm) <= 1e-5 * np.abs(want_dm) + 1e-6)
        ch = O.Channel(c)
        ch.demod(dm1[c])                                    # the oracle's demodulator on the GPU's dm: exact
        want = [O.frame_tuple(f) for f in ch.frames]
        assert got1.get(c, []) == want, c
        total += len(want)
    assert total >= nch


@pytest.mark.parametrize("fe", ["rtl", "soapy", "air", "sdrplay", "file"])
def test_legacy_view_first_callback_fits_its_transfer_period(fe, S, O, tmp_path):
    """the legacy view made its GPU context inside the FIRST callback (0.14-0.27 s; librtlsdr's four-buffer
    ring absorbs that, libairspy / SDRplay callbacks do not).  compat_msk.c now makes the context -- and runs one throw-away call
    through it -- when initMsk() sees the last channel (acarsdec.c:445-454).  Every front end's program: the first entry-point
    call costs what the others cost (far below the shortest transfer period of its radio: an Airspy transfer of 20 000 samples
    at 2.5 Msps lasts 8 ms, an rtl.c callback 81.92 ms), the context's cost is reported separately, the output is unchanged
    (the four test_compat_* tests), and the printed messages arrive."""
    import os
    import re
    import subprocess
    from conftest import ROOT, GOLDEN
    exe = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu" + ("" if fe == "file" else "_" + fe))
    if not os.path.exists(exe):
        pytest.skip("demo binary not built (needs the reference tree at build time)")
    z = np.load(os.path.join(GOLDEN, "testwav_pcm16.npz"))
    wav = (z["pcm"].astype(np.float32) / np.float32(32768.0))
    freqs = ["131.525", "131.725", "131.825", "131.550"]
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    env = S.pad_blocks(0.5 + 0.5 * wav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * 3), 0.5)], axis=1)
    ph = [0.3, 1.1, 2.2, 0.7]
    envv = dict(os.environ, ACARSDEC_AMD_STATS="1")
    if fe == "file":
        import wave
        src = str(tmp_path / "t.wav")                       # the golden recording (tests/golden/testwav_pcm16.npz) as a PCM16 file again
        with wave.open(src, "wb") as wv:
            wv.setnchannels(z["pcm"].shape[1])
            wv.setsampwidth(2)
            wv.setframerate(12500)
            wv.writeframes(np.ascontiguousarray(z["pcm"], dtype="<i2").tobytes())
        args = ["-o", "1", "-f", src]
    else:
        if fe == "rtl":
            fc = O.choose_fc(fr, 160)
            data = S.iq_u8_from_envelopes(env, 160, [f - fc for f in fr], phases=ph)
            args = ["-o", "1", "-r", "0"] + freqs
        elif fe == "soapy":
            data = S.iq_s16_from_envelopes(env, 160, [f - 131850000 for f in fr], phases=ph)
            args = ["-o", "1", "-m", "160", "-d", "file"] + freqs
        elif fe == "air":
            rate = 2500000
            fc = O.air_choose_fc(fr)
            data = S.real_f32_from_envelopes(env, rate // 12500, [fc - f + rate / 4 for f in fr], phases=ph, scale=0.15)
            args = ["-o", "1", "-s", "0"] + freqs
        else:
            data = S.iq_s16_from_envelopes(env, 160, [f - 131850000 for f in fr], phases=ph, full_scale=0.06)
            args = ["-o", "1", "-s"] + freqs
        path = tmp_path / ("t." + fe)
        path.write_bytes(data.tobytes())
        envv["ACARSDEC_IQ_FILE"] = str(path)
    r = subprocess.run([exe] + args, env=envv, capture_output=True, timeout=300)
    err = r.stderr.decode("latin-1")
    assert r.returncode == 0 or fe == "file", err[-800:]       # (the reference's main() ends a sound-file run with the reader's -1, acarsdec.c:480-489)
    m = re.search(r"acarsdec_amd compat: (\d+) calls, .*first call ([0-9.]+) ms; the others ([0-9.]+) ms per call; context made at initMsk\(\) time in ([0-9.]+) ms", err)
    assert m, err[-600:]
    calls, first, others, ctx = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))
    assert calls > 10 and ctx > 5.0                              # the context was paid for before the radio started
    assert first < 5.0 and first < 8.0, (first, others, ctx)     # < 5 ms: inside every front end's shortest transfer period
    assert len([l for l in r.stdout.decode("latin-1").splitlines() if l.startswith("#")]) == 7


def test_channel_moved_mid_block_keeps_its_soh_stamp(D, O, S):
    """the SOH stamp (where acars.c:290 stamps blk->tv) was not part of acg_chan_state, so a channel moved to another
    slot / context with acg_get_state -> acg_set_state while a block was being assembled delivered that block with a stale stamp.
    acg_chan_state.soh_back carries it as a distance now: cut a stream in the middle of a block, move the channel to slot 1 of a
    second context whose sample counter stands elsewhere, finish there -- the block's end - SOH distance is the oracle's.  (The
    text assembled so far travels with acg_get_block_text / acg_set_block_text: it is the one part of the block state that is not
    in acg_chan_state.)"""
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(606)
    n = 12 * 1024
    a, _ = S.channel_audio(rng, n, nframes=1, gap=(1500, 1600), text_len=(100, 110))
    x = S.envelope(a, noise=0.003, rng=rng).astype(np.float32)
    ch = O.Channel(0)
    ch.demod(x)
    assert len(ch.frames) == 1
    f = ch.frames[0]
    cut = (int(f.soh_sample) + int(f.end_sample)) // 2 // 1024 * 1024          # a call boundary strictly inside the block
    assert int(f.soh_sample) < cut < int(f.end_sample)
    d1 = D.Decoder(1, decim=8, ntaps=8, nstreams=1, max_blocks=12)
    d1.demod_msk(x[:cut].reshape(1, -1))
    s = K.ChanState()
    d1._chk(d1.L.acg_get_state(d1.ctx, 0, C.byref(s)))
    txt = (C.c_ubyte * 250)()
    d1._chk(d1.L.acg_get_block_text(d1.ctx, 0, txt))           # blk->txt so far: the part of the state that is not a scalar
    assert s.Acarsstate in (3, 4, 5) and s.soh_back == cut - int(f.soh_sample) and 0 < s.blk_len < int(f.len)
    assert bytes(txt[: s.blk_len]) != bytes(s.blk_len)
    assert d1.drain_frames() == []
    d2 = D.Decoder(2, decim=8, ntaps=8, nstreams=2, max_blocks=12)
    skew = np.full((2, 3 * 1024), 0.5, dtype=np.float32)                       # the destination has consumed 3072 samples already
    d2.demod_msk(skew)
    d2._chk(d2.L.acg_set_state(d2.ctx, 1, C.byref(s)))
    d2._chk(d2.L.acg_set_block_text(d2.ctx, 1, txt))
    rest = np.full((2, n - cut), 0.5, dtype=np.float32)
    rest[1] = x[cut:]
    d2.demod_msk(rest)
    got = [g for g in d2.drain_frames() if int(g.chn) == 1]
    assert len(got) == 1 and D.frame_tuple(got[0])[1:] == O.frame_tuple(f)[1:]
    assert int(got[0].end_sample) - int(got[0].soh_sample) == int(f.end_sample) - int(f.soh_sample)
    assert int(got[0].end_sample) == 3 * 1024 + (int(f.end_sample) - cut)      # (the destination's own sample index)
    d1.close()
    d2.close()


@pytest.mark.parametrize("M,ntaps,nblk", [(200, 200, 2), (160, 160, 1), (192, 192, 1), (200, 192, 1), (160, 37, 1)])
def test_matrix_pipe_one_stream_per_channel_kernel(D, O, M, ntaps, nblk, tune):
    """fir_u8_mm1_kernel (ACG_FIR_MM1=1): the exact int8 contraction with ONE channel per stream -- not for its arithmetic rate but
    to take the down-converter's arithmetic off the vector pipe.  Channels on scrambled streams (an explicit channel -> stream map
    and the identity), the extremes of the u8 range, a tap table replaced between calls: dm within the 1e-5 bar of the oracle,
    within 2e-7 of the f64-exact value, and the wave-private vector kernel agrees within the bar."""
    rng = np.random.default_rng(7000 + M + ntaps)
    nch = 37
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(nch, nout * M * 2), dtype=np.uint8)
    iq[0, : 4 * M] = 0
    iq[1, : 4 * M] = 255
    iq[2, : 4 * M] = 128
    taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
    for c in range(nch):
        taps[c] = O.rtl_taps(131000000 + 25000 * int(rng.integers(-40, 41)), 131000000, M)[:ntaps]
    taps[3] *= np.float32(1.0 / 1024)
    taps[5, ::2] = 0

    def run(smap):
        dec = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=nch, max_blocks=nblk)
        dec.set_taps(taps)
        if smap is not None:
            dec.set_channel_streams(smap)
        dec.in_callback(iq)
        out = np.stack([dec.dm(c, nout) for c in range(nch)])
        dec.set_taps(taps[::-1].copy())
        dec.in_callback(iq)
        out2 = np.stack([dec.dm(c, nout) for c in range(nch)])
        dec.close()
        return out, out2

    perm = rng.permutation(nch)
    tune("ACG_FIR_MM1", "1")
    mm_id, mm_id2 = run(None)
    mm_pm, mm_pm2 = run(perm)
    tune("ACG_FIR_MM1", "0")
    va_id, va_id2 = run(None)
    assert not np.array_equal(mm_id, va_id)
    worst, worst_v = 0.0, 0.0
    for c in range(nch):
        for got, other, row, tp in ((mm_id[c], va_id[c], c, taps[c]), (mm_id2[c], va_id2[c], c, taps[nch - 1 - c]),
                                    (mm_pm[c], None, perm[c], taps[c]), (mm_pm2[c], None, perm[c], taps[nch - 1 - c])):
            want = O.fir_u8(iq[row], M, tp, nout=nout, ntaps=ntaps)
            assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6), c
            ex = exact_dm(iq[row], M, tp, nout)
            err = np.abs(got.astype(np.float64) - ex)
            assert np.all(err <= 2e-7 * ex + 1e-9), (c, float((err / (ex + 1e-9)).max()))
            worst = max(worst, float((err / (ex + 1e-3)).max()))
            if other is not None:
                assert np.all(np.abs(got - other) <= 1e-5 * np.abs(other) + 1e-6), c
                worst_v = max(worst_v, float((np.abs(other.astype(np.float64) - ex) / (ex + 1e-3)).max()))
    assert worst < 2e-7 and worst < worst_v, (worst, worst_v)


def test_matrix_pipe_one_stream_per_channel_blocks(D, O, S, tune):
    """the same kernel end to end: 24 channels of ACARS traffic, eight callbacks at a time and one at a time -- identical blocks
    and dm for both chunkings, blocks exact given the GPU's dm, dm inside the bar."""
    tune("ACG_FIR_MM1", "1")
    rng = np.random.default_rng(4242)
    M, nch, nblk = 200, 24, 8
    nout = nblk * 1024
    offs = [25000.0 * int(k) for k in rng.integers(2, 40, size=nch) * rng.choice([-1, 1], size=nch)]
    rows, taps = [], []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, nframes=2, gap=(1500, 2500), text_len=(10, 60))
        rows.append(S.iq_u8_from_envelopes(np.array([0.5 * (1 + 0.5 * a)]), M, [offs[c]], phases=[float(rng.uniform(0, 6.28))], noise=0.01, rng=rng))
        taps.append(O.rtl_taps(int(131000000 + offs[c]), 131000000, M))
    iq = np.stack(rows)
    taps = np.stack(taps)

    def run(chunks):
        dec = D.Decoder(nch, decim=M, nstreams=nch, max_blocks=nblk)
        dec.set_taps(taps)
        per = nblk // chunks
        dms = [[] for _ in range(nch)]
        for k in range(chunks):
            dec.in_callback(np.ascontiguousarray(iq[:, k * per * 1024 * M * 2:(k + 1) * per * 1024 * M * 2]), nblocks=per)
            for c in range(nch):
                dms[c].append(dec.dm(c, per * 1024))
        got = {}
        for f in dec.drain_frames():
            got.setdefault(int(f.chn), []).append(D.frame_tuple(f))
        dec.close()
        return got, [np.concatenate(d) for d in dms]

    g1, d1 = run(1)
    g8, d8 = run(8)            # (one callback per call: launches of 1024 windows still take the matrix path)
    assert g1 == g8 and all(np.array_equal(a, b) for a, b in zip(d1, d8))
    total = 0
    for c in range(nch):
        want_dm = O.fir_u8(iq[c], M, taps[c])
        assert np.all(np.abs(d1[c] - want_dm) <= 1e-5 * np.abs(want_dm) + 1e-6)
        ch = O.Channel(c)
        ch.demod(d1[c])
        want = [O.frame_tuple(f) for f in ch.frames]
        assert g1.get(c, []) == want, c
        total += len(want)
    assert total >= nch
