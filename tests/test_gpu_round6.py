"""Round 6: the shared-stream down-converter on the matrix pipe (fir_mm.hip: rtl.c:344-354 with K channels per dongle as an
exact int8 contraction), the legacy view's context made at initMsk() time, and the configurations that were never timed before
(rtlMult 160 / 192, SDRplay planes) through the bench gate.  All through the C ABI; the oracle is the checker."""
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
