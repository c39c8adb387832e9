"""Round 5: the block's time stamp (acars.c:290), the block ring across the wrap of its 32-bit counters, a host that streams
more calls than the per-call ring holds without collecting, the batched state access of the legacy view, and the reference
program on a 16-channel dongle against its CPU twin.  All through the C ABI; the oracle is the checker."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

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


def traffic(S, nch, n, seed, corrupt=None, gap=(1200, 2500), text_len=(8, 60), noise=0.003):
    rng = np.random.default_rng(seed)
    x = np.zeros((nch, n), dtype=np.float32)
    for c in range(nch):
        a, _ = S.channel_audio(rng, n, gap=gap, text_len=text_len, corrupt=corrupt)
        x[c] = S.envelope(a, noise=noise, rng=rng)
    return x


@pytest.mark.parametrize("lpc", [8, 1])
def test_block_time_stamps_equal_the_oracle(D, O, S, lpc, tune):
    """acg_frame.soh_sample / acg_msg.soh_sample: the 12.5 kHz sample index at which the block's SOH byte completed -- where the
    reference takes blk->tv (acars.c:290) -- and end_sample, the closing bit's: both bit-exact against the oracle's own
    bookkeeping of the same loop, across calls of ragged length (the stamp lives in the channel state between launches), for
    raw blocks and for delivered messages, at 8 lanes and at 1 lane per channel."""
    from acarsdec_amd import _capi as K
    tune("ACG_MSK_LPC", lpc)
    nch, n = 24, 40 * 1024
    x = traffic(S, nch, n, 905)
    cuts = [0, 3000, 3001, 9000, 20480, 20481 + 7, 33333, n]
    want = {}
    for c in range(nch):
        ch = O.Channel(c, max_frames=512)
        for a, b in zip(cuts[:-1], cuts[1:]):
            ch.demod(x[c, a:b])
        want[c] = [(int(f.end_bit), int(f.end_sample), int(f.soh_sample)) for f in ch.frames]
        for eb, es, ss in want[c]:
            assert 0 <= ss < es                            # SOH comes before the closing bit
    assert sum(len(v) for v in want.values()) >= 3 * nch
    for repair in (False, True):
        dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=20, repair=repair, bitlog=False)
        got, gotm = {}, {}
        for a, b in zip(cuts[:-1], cuts[1:]):
            dec.demod_msk(x[:, a:b])
            if repair:
                for m in dec.drain_msgs():
                    gotm.setdefault(int(m.chn), []).append((int(m.end_bit), int(m.end_sample), int(m.soh_sample)))
                    assert m.reserved1 == 0 and m.reserved3 == 0
            else:
                for f in dec.drain_frames():
                    got.setdefault(int(f.chn), []).append((int(f.end_bit), int(f.end_sample), int(f.soh_sample)))
        dec.close()
        if not repair:
            assert got == {c: v for c, v in want.items() if v}
        else:
            # the delivered messages are a subset (blocks the block thread drops are omitted); every one carries its block's stamps
            for c, lst in gotm.items():
                assert set(lst) <= set(want[c]) and lst == sorted(lst)
            assert sum(len(v) for v in gotm.values()) >= 2 * nch
    assert K.load().acg_is_lab_build() == 0


@pytest.mark.parametrize("mode", ["frames", "msgs"])
def test_block_ring_across_the_wrap_of_its_counters(D, O, S, mode):
    """the ring's monotonic 32-bit counters (device queue length, per-call marks, repair mark, host consumer)
    wrap after 2^32 blocks -- hours at bench rates.  The ring's length is a power of two, so slot(count) has no jump there.
    The lab hook presets every counter to 2^32 - k right after a reset; calls that cross the wrap (collected one call behind,
    with a buffer small enough that ACG_EAGAIN and the two-piece copy happen as well) must deliver exactly the blocks /
    messages of a run started at 0."""
    from acarsdec_amd import _capi as K
    nch, ncalls, clen = 48, 6, 8 * 1024
    x = traffic(S, nch, ncalls * clen, 4242, corrupt=[None, None, "p1", None, "db", None, "crc", None])
    repair = mode == "msgs"
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=8, repair=repair, bitlog=False, max_lag=1)
    ring = dec.L.acg_lab_block_ring_size(dec.ctx)
    assert ring >= 2 and ring & (ring - 1) == 0

    def key(r):
        if repair:
            return (int(r.chn), int(r.end_bit), int(r.err), int(r.txt_len), bytes(r.txt[: r.txt_len]), bytes(r.addr), bytes(r.label), int(r.soh_sample))
        return D.frame_tuple(r) + (int(r.end_bit), int(r.soh_sample))

    def run(preset):
        dec.reset()
        if preset is not None:
            assert dec.L.acg_lab_set_block_counter(dec.ctx, preset) == K.OK
        out = []
        for k in range(ncalls):
            dec.demod_msk(x[:, k * clen:(k + 1) * clen])
            if repair:
                while True:
                    n, buf, more = dec.collect_msgs_raw(lag=1, max_msgs=7)
                    out += [key(K.Msg.from_buffer_copy(buf[i])) for i in range(n)]
                    if not more:
                        break
            else:
                while True:
                    n, buf, more = dec._frames_call(dec.L.acg_collect_frames, 7, 1)
                    out += [key(K.Frame.from_buffer_copy(buf[i])) for i in range(n)]
                    if not more:
                        break
        out += [key(r) for r in (dec.drain_msgs(5) if repair else dec.drain_frames(5))]
        return sorted(out)

    base = run(None)
    assert len(base) >= 3 * nch
    # the wrap inside the first call, between the calls, and right at the end of the stream
    for back in (5, len(base) // 2, len(base) - 3, 1):
        assert run((1 << 32) - back) == base, "counters preset to 2^32 - %d" % back
    # the hook refuses once calls have been issued
    dec.demod_msk(x[:, :clen])
    assert dec.L.acg_lab_set_block_counter(dec.ctx, 5) == K.ESTATE
    dec.close()
    # and the oracle agrees with the run from 0 (raw blocks)
    if not repair:
        want = []
        for c in range(nch):
            ch = O.Channel(c, max_frames=512)
            ch.demod(x[c])
            want += [O.frame_tuple(f) + (int(f.end_bit), int(f.soh_sample)) for f in ch.frames]
        assert sorted(want) == base


def test_streaming_past_the_per_call_ring_without_collecting(D, O, S):
    """the per-call marks live in an 8-slot ring.  A host that streams 20 calls and collects only at the end must
    get every message: no repair pass may read a mark that a later call has already re-used (it would re-process blocks whose
    parity is already stripped and drop them)."""
    nch, ncalls, clen = 32, 20, 2 * 1024
    x = traffic(S, nch, ncalls * clen, 777, corrupt=[None, "p1", None, "p2", None])
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=2, repair=True, bitlog=False)
    for k in range(ncalls):
        dec.demod_msk(x[:, k * clen:(k + 1) * clen])
    got = sorted((int(m.chn), int(m.end_bit), int(m.err), bytes(m.txt[: m.txt_len])) for m in dec.drain_msgs())
    dec.close()
    want = []
    for c in range(nch):
        ch = O.Channel(c, max_frames=512)
        ch.demod(x[c])
        for f in ch.frames:
            o = O.blk_process(f)
            if o is not None:
                m = O.msg_split(o)
                want.append((c, int(f.end_bit), int(o.err), bytes(m.txt[: m.txt_len])))
    assert got == sorted(want) and len(got) >= 2 * nch


def test_state_of_n_channels_in_one_transfer(D, S):
    """acg_get_state_n / acg_set_state_n / acg_read_dm_n (what the legacy view uses per callback) equal the one-channel calls,
    and a state set for all channels at once continues exactly like the original context."""
    from acarsdec_amd import _capi as K
    nch, n = 12, 6 * 1024
    x = traffic(S, nch, 2 * n, 99)
    a = D.Decoder(nch, decim=8, ntaps=8, max_blocks=6, bitlog=False)
    a.demod_msk(x[:, :n])
    a.drain_frames()
    st = (K.ChanState * nch)()
    assert a.L.acg_get_state_n(a.ctx, 0, nch, st) == K.OK
    for c in range(nch):
        one = K.ChanState()
        assert a.L.acg_get_state(a.ctx, c, C.byref(one)) == K.OK
        assert bytes(one) == bytes(st[c])
    part = (K.ChanState * 5)()
    assert a.L.acg_get_state_n(a.ctx, 4, 5, part) == K.OK and bytes(part) == bytes(st)[4 * C.sizeof(K.ChanState): 9 * C.sizeof(K.ChanState)]
    assert a.L.acg_get_state_n(a.ctx, 8, 5, part) == K.EINVAL and a.L.acg_set_state_n(a.ctx, -1, 2, part) == K.EINVAL
    dm = np.zeros((nch, 128), dtype=np.float32)
    assert a.L.acg_read_dm_n(a.ctx, 0, nch, dm.ctypes.data, 128, 100) == K.OK
    for c in range(nch):
        assert np.array_equal(dm[c, :100], a.dm(c, 100)) and np.array_equal(dm[c, :100], x[c, :100])
    # a fresh context given the whole state continues like the original.  (The text of a block that is being assembled at the
    # hand-over lives in the context's text buffer, which is not part of channel_t -- in the legacy view it is the caller's
    # blk->txt: per channel the first block after the hand-over may differ in its text, everything behind it must not.)
    b = D.Decoder(nch, decim=8, ntaps=8, max_blocks=6, bitlog=False)
    assert b.L.acg_set_state_n(b.ctx, 0, nch, st) == K.OK
    a.demod_msk(x[:, n:])
    b.demod_msk(x[:, n:])
    per = lambda frs: {c: [D.frame_tuple(f) for f in frs if int(f.chn) == c] for c in range(nch)}
    fa, fb = per(a.drain_frames()), per(b.drain_frames())
    later = 0
    for c in range(nch):
        assert [t[:2] for t in fa[c]] == [t[:2] for t in fb[c]]            # the same blocks with the same lengths ...
        assert fa[c][1:] == fb[c][1:]                                      # ... and, behind the first, the same bytes
        later += len(fa[c][1:])
    assert later >= 3
    sa, sb = (K.ChanState * nch)(), (K.ChanState * nch)()
    assert a.L.acg_get_state_n(a.ctx, 0, nch, sa) == K.OK and b.L.acg_get_state_n(b.ctx, 0, nch, sb) == K.OK and bytes(sa) == bytes(sb)
    a.close()
    b.close()


def test_compat_rtl_program_with_16_channels_equals_its_cpu_twin(S, tmp_path):
    """BASELINE configs[1]'s shape through the legacy view at the reference's limit of 16 channels per dongle (MAXNBCHANNELS,
    acarsdec.h:30): the reference's acarsdec.c + rtl.c (one-hunk binding) + acars.c + output.c on compat_msk.c must print what
    the CPU twin prints on the same 2.0 Msps I/Q file -- same messages, levels and error counts, per channel in order -- with the
    state of all 16 channels moved in one transfer per callback and dm_buffer left on the device."""
    gpu = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_rtl")
    cpu = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_rtl")
    if not (os.path.exists(gpu) and os.path.exists(cpu)):
        pytest.skip("demo binaries not built (they need the reference tree at build time)")
    rng = np.random.default_rng(1616)
    M, nch, nblk = 160, 16, 24
    freqs = ["%.3f" % (131.000 + 0.025 * (2 * k + 1)) for k in range(nch)]          # 50 kHz raster, all inside one dongle's 2 MHz
    from acarsdec_amd import decoder as D
    fr = [D.parse_freq_mhz(f) for f in freqs]
    fc, _ = D.choose_fc(fr, M)
    assert fc != 0
    env = np.zeros((nch, nblk * 1024))
    for c in range(nch):
        a, _ = S.channel_audio(rng, env.shape[1], gap=(2000, 5000), text_len=(10, 80))
        env[c] = 0.5 * (1 + 0.5 * a)
    iq = S.iq_u8_from_envelopes(env, M, [f - fc for f in fr], phases=list(rng.uniform(0, 6.28, nch)), scale=1.0 / nch, noise=0.004, rng=rng)
    path = tmp_path / "t16.iq"
    path.write_bytes(iq.tobytes())
    outs, stats = [], None
    for exe in (cpu, gpu):
        r = subprocess.run([exe, "-o", "1", "-r", "0"] + freqs, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path), ACARSDEC_AMD_STATS="1"),
                           capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode("latin-1")[-800:]
        lines = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1")).splitlines()
        per = {}
        for l in lines:
            if l.startswith("#"):
                per.setdefault(l.split()[0], []).append(l)
        outs.append(per)
        if exe == gpu:
            stats = re.search(r"acarsdec_amd compat: (\d+) calls, .* ([0-9.]+) ms per call", r.stderr.decode("latin-1"))
    assert outs[0] == outs[1] and sum(len(v) for v in outs[0].values()) >= nch
    assert stats and int(stats.group(1)) == nblk
    # a callback carries 81.92 ms of signal (rtl.c:49,213): the legacy view must stay far inside that budget
    assert float(stats.group(2)) < 20.0, stats.group(0)


def test_host_entry_points_do_not_share_a_carried_window(D, S):
    """acg_feed_samples_host keeps the samples of an incomplete window at the head of the two staging buffers that
    acg_process_iq_u8_host uses as well.  While a partial window is carried, the u8 entry point refuses (ACG_ESTATE) instead of
    overwriting it; after acg_reset (which drops the carry) it works again."""
    from acarsdec_amd import _capi as K
    M = 160
    dec = D.Decoder(2, decim=M, nstreams=2, max_blocks=1, bitlog=False)
    x = (np.arange(2 * (M + 37) * 2, dtype=np.int16) % 97).reshape(2, -1)          # one window and 37 samples per stream
    assert dec.L.acg_feed_samples_host(dec.ctx, K.FMT_CS16, x.ctypes.data, None, M + 37, M + 37) == K.OK
    iq = np.zeros((2, 1024 * M * 2), dtype=np.uint8)
    assert dec.L.acg_process_iq_u8_host(dec.ctx, iq.ctypes.data, iq.shape[1], 1) == K.ESTATE
    assert b"partial window" in dec.L.acg_last_error(dec.ctx)
    dec.reset()
    assert dec.L.acg_process_iq_u8_host(dec.ctx, iq.ctypes.data, iq.shape[1], 1) == K.OK
    assert dec.drain_frames() == []
    dec.close()
