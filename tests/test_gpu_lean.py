"""Round 6: msk_lean.hip -- the demodulator with the framing state machine taken off the per-bit path (launches without a bit
log) -- against msk.hip's kernel (framing inline, after every bit): channel state, blocks and block text identical bit for bit
after every call, on input that visits every branch of decodeAcars() (acars.c:246-375), and both against the oracle."""
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


def zoo_tracks(S, nch, nsamp, seed):
    """12.5 kHz envelopes [nch, nsamp] that exercise the framing machine: plain traffic, every kind of corruption the repair
    tests use, blocks that are too long (acars.c:336), blocks with more parity errors than MAXPERR + 1 (acars.c:312), blocks whose
    ETX is lost so that they end at DEL (acars.c:323-333), noise only (false SYN / ~SYN all the time), silence, a constant."""
    rng = np.random.default_rng(seed)
    x = np.zeros((nch, nsamp), dtype=np.float32)
    kinds = []
    for c in range(nch):
        k = c % 10
        kinds.append(k)
        if k in (0, 1, 2):
            a, _ = S.channel_audio(rng, nsamp, gap=(300, 1500), text_len=(1, 220),
                                   corrupt=[None, "p1", "p2", "p3", "p4", "db", "crc", "p1crc"] if k else None)
            x[c] = S.envelope(a, noise=0.0 if k < 2 else 0.02, rng=rng)
        elif k in (3, 4, 5):
            frames = []
            for i in range(60):
                fr = bytearray(S.acars_frame(text=S.random_text(rng, 230, 250) if k == 3 else S.random_text(rng, 30, 120)))
                if k == 4:                         # six parity errors in the text
                    for j in rng.choice(np.arange(20, len(fr) - 6), size=6, replace=False):
                        fr[int(j)] ^= 1 << int(rng.integers(0, 7))
                if k == 5:                         # the terminator becomes a letter: the block runs on through the CRC to DEL
                    fr[len(fr) - 4] = S.odd_parity(0x41 + int(rng.integers(0, 26)))
                frames.append(bytes(fr))
            a = S.frames_audio(frames, rng, gap=(200, 900), lead=int(rng.integers(100, 900)))
            a = np.resize(a, nsamp) if len(a) >= nsamp else np.concatenate([a, np.zeros(nsamp - len(a))])
            x[c] = S.envelope(a[:nsamp])
        elif k == 6:
            x[c] = rng.normal(0.5, 0.2, size=nsamp).astype(np.float32)
        elif k == 7:
            x[c] = 0.0
        elif k == 8:
            x[c] = 0.37
        else:
            a, _ = S.channel_audio(rng, nsamp, gap=(100, 400), text_len=(1, 30))
            x[c] = S.envelope(a, noise=0.1, rng=rng)
    return x, kinds


def snapshot(dec, K):
    st = (K.ChanState * dec.nch)()
    dec._chk(dec.L.acg_get_state_n(dec.ctx, 0, dec.nch, st))
    raw = bytes(st)
    txt = []
    for ch in range(dec.nch):
        if st[ch].Acarsstate == 3:
            buf = (C.c_ubyte * 256)()
            dec._chk(dec.L.acg_get_block_text(dec.ctx, ch, buf))
            txt.append(bytes(buf[: st[ch].blk_len]))
        else:
            txt.append(b"")
    frames = sorted((f.chn, f.end_bit, f.len, f.err, f.lvl, bytes(f.crc), bytes(f.txt[: f.len]), f.end_sample, f.soh_sample)
                    for f in dec.drain_frames())
    return raw, txt, frames


def run(D, K, x, chunks, bitlog):
    nch = x.shape[0]
    dec = D.Decoder(nch, max_blocks=8, bitlog=bitlog)
    out = []
    a0 = 0
    for n in chunks:
        dec.demod_msk(x[:, a0:a0 + n])
        dec.sync()
        snap = snapshot(dec, K)
        if bitlog:
            cnt, vo, lvl = dec.bits_all()
            snap += (cnt.tobytes(), b"".join(vo[c, : cnt[c]].tobytes() + lvl[c, : cnt[c]].tobytes() for c in range(nch)))
        out.append(snap)
        a0 += n
    dec.close()
    return out


@pytest.mark.parametrize("lpc,cus", [(8, None), (4, None), (8, 0), (4, 0)])
def test_lean_demodulator_is_the_inline_demodulator(D, S, tune, lpc, cus):
    from acarsdec_amd import _capi as K
    nch = 70                                        # not a multiple of the channels per wave: the last wave replicates
    chunks = [8192, 1024, 32, 4096, 8192, 64, 2048, 8192, 8192, 96, 8192, 8192]
    x, kinds = zoo_tracks(S, nch, sum(chunks), 6)
    tune("ACG_MSK_LPC", str(lpc))
    if cus is not None:
        tune("ACG_MSK_CUS", str(cus))
    lean = run(D, K, x, chunks, bitlog=False)
    lean_log = run(D, K, x, chunks, bitlog=True)
    tune("ACG_MSK_NOLEAN", "1")
    inline = run(D, K, x, chunks, bitlog=False)
    logged = run(D, K, x, chunks, bitlog=True)
    for i, (a, b) in enumerate(zip(lean_log, logged)):
        assert a[:3] == lean[i][:3], "call %d: lean kernel with / without bit log" % i
        assert a[3] == b[3], "call %d: bits per channel differ" % i
        assert a[4] == b[4], "call %d: bit records {soft symbol, level} differ" % i
    nframes = 0
    for i, (a, b, c) in enumerate(zip(lean, inline, logged)):
        assert b[:3] == c[:3], "call %d: inline kernel with / without bit log" % i
        if a[0] != b[0]:
            sz = C.sizeof(K.ChanState)
            bad = [ch for ch in range(nch) if a[0][ch * sz:(ch + 1) * sz] != b[0][ch * sz:(ch + 1) * sz]]
            raise AssertionError("call %d: state differs on channels %s (kinds %s)" % (i, bad[:10], [kinds[ch] for ch in bad[:10]]))
        assert a[1] == b[1], "call %d: block text under assembly differs" % i
        assert a[2] == b[2], "call %d: blocks differ (%d / %d)" % (i, len(a[2]), len(b[2]))
        nframes += len(a[2])
    assert nframes > 200
    # every kind of track produced what it is there for
    per_kind = {}
    for snap in lean:
        for f in snap[2]:
            per_kind.setdefault(kinds[f[0]], []).append(f)
    assert per_kind.get(0) and per_kind.get(1) and per_kind.get(5), sorted(per_kind)
    assert all(f[2] > 20 for f in per_kind[5])       # blocks that ended at DEL: put_frame from the TXT state


def test_lean_demodulator_against_the_oracle(D, O, S):
    """blocks and the framing state of the lean kernel == the oracle's demodMSK + decodeAcars on the same dm (what the bench
    gate checks at scale), here on the zoo"""
    from acarsdec_amd import _capi as K
    nch = 20
    chunks = [8192] * 6
    x, kinds = zoo_tracks(S, nch, sum(chunks), 11)
    dec = D.Decoder(nch, max_blocks=8, bitlog=False)
    got = []
    a0 = 0
    for n in chunks:
        dec.demod_msk(x[:, a0:a0 + n])
        got += [(f.chn, f.len, f.err, bytes(f.crc), bytes(f.txt[: f.len]), f.end_bit, f.end_sample, f.soh_sample, f.lvl) for f in dec.drain_frames()]
        a0 += n
    ref = []
    for ch in range(nch):
        oc = O.Channel(ch, max_frames=1024)
        a0 = 0
        for n in chunks:
            oc.demod(x[ch, a0:a0 + n])
            a0 += n
        ref += [(f.chn, f.len, f.err, bytes(f.crc), bytes(f.txt[: f.len]), f.end_bit, f.end_sample, f.soh_sample, f.lvl) for f in oc.frames]
        g, o = dec.state(ch), oc.state()
        for k in ("MskS", "idx", "outbits", "nbits", "Acarsstate", "MskBitCount"):
            assert int(g[k]) == int(o[k]), (ch, kinds[ch], k, g[k], o[k])
        for k in ("MskPhi", "MskDf", "MskClk", "MskLvlSum"):
            assert np.isclose(g[k], o[k], rtol=1e-9, atol=1e-12), (ch, kinds[ch], k, g[k], o[k])
    assert len(ref) > 40
    assert sorted(got) == sorted(ref)
    dec.close()


def test_lean_demodulator_with_a_bit_count_only_a_host_could_have_written(D, S, tune):
    """nbits outside 1..8 never arises from decodeAcars(); a host can write it with acg_set_state.  Such a channel takes the inline
    path until its count is back in range: both kernels agree call by call."""
    from acarsdec_amd import _capi as K
    nch = 24
    chunks = [4096, 8192, 8192]
    x, kinds = zoo_tracks(S, nch, sum(chunks) + 4096, 21)

    def go():
        dec = D.Decoder(nch, max_blocks=8, bitlog=False)
        dec.demod_msk(x[:, :4096])
        dec.sync()
        st = (K.ChanState * nch)()
        dec._chk(dec.L.acg_get_state_n(dec.ctx, 0, nch, st))
        for ch in range(nch):
            st[ch].nbits = (0, -3, 9, 13, 40, 1000)[ch % 6]
        dec._chk(dec.L.acg_set_state_n(dec.ctx, 0, nch, st))
        out = []
        a0 = 4096
        for n in chunks:
            dec.demod_msk(x[:, a0:a0 + n])
            dec.sync()
            out.append(snapshot(dec, K))
            a0 += n
        dec.close()
        return out

    lean = go()
    tune("ACG_MSK_NOLEAN", "1")
    inline = go()
    for i, (a, b) in enumerate(zip(lean, inline)):
        assert a == b, "call %d" % i


def test_set_state_refuses_a_block_length_beyond_the_text_row(D):
    from acarsdec_amd import _capi as K
    dec = D.Decoder(4, max_blocks=1, bitlog=False)
    st = K.ChanState()
    dec._chk(dec.L.acg_get_state(dec.ctx, 1, C.byref(st)))
    for bad in (-1, 242, 300, 1 << 20):
        st.blk_len = bad
        assert dec.L.acg_set_state(dec.ctx, 1, C.byref(st)) == K.EINVAL
    st.blk_len = 241
    assert dec.L.acg_set_state(dec.ctx, 1, C.byref(st)) == K.OK
    dec.close()
