"""Parity of the HIP path (through the C ABI) against the oracle and the committed golden
fixtures.  Runs on the GPU box only (-m gpu); /root/reference does not exist there.

Stated tolerances:
  * blocks (chn, len, err, crc, txt): bit-exact, per channel in order;
  * dm (|D| after the down-converter): |gpu - oracle| <= 1e-5*|oracle| + 1e-6  (f32 sums of up to
    320 products in a different association; the reference's own -Ofast build reassociates too);
  * soft symbols: hard decisions identical wherever |vo| > 0.05; |d vo| <= 1e-4 for >= 99 % of bits
    and <= 5e-2 for all (the reference's own -O2 vs -Ofast builds differ by up to 1.4e-2 on isolated
    bits because `o=(int)(12*(MskClk/s+0.5))`, msk.c:103, truncates);
  * level: within 0.05 dB.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden_blocks, soft_from_v, fhex, ROOT

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


def assert_soft_close(vo_g, vo_o, lvl_g=None, lvl_o=None):
    assert len(vo_g) == len(vo_o)
    d = np.abs(vo_g - vo_o)
    strong = np.abs(vo_o) > 0.05
    assert np.array_equal(vo_g[strong] > 0, vo_o[strong] > 0), "hard decision flipped on a strong bit"
    assert d.max() <= 5e-2, d.max()
    assert (d <= 1e-4).mean() >= 0.99, (d <= 1e-4).mean()
    if lvl_g is not None:
        assert np.allclose(lvl_g, lvl_o, rtol=1e-4, atol=1e-6)


def blocks_by_channel(frames, tup):
    out = {}
    for f in frames:
        out.setdefault(int(f.chn), []).append(tup(f))
    return out


def blocks_by_channel_tuples(tuples):
    out = {}
    for t in tuples:
        out.setdefault(t[0], []).append(t)
    return out


# ------------------------------------------------------------------------------------ FIR stage
@pytest.mark.parametrize("M,ntaps,nblk", [(160, 160, 2), (200, 200, 1), (192, 192, 1), (200, 192, 1),
                                          (320, 320, 1), (8, 8, 1), (164, 164, 1), (160, 37, 1)])
def test_fir_matches_oracle(D, O, M, ntaps, nblk):
    rng = np.random.default_rng(M * 1000 + ntaps)
    nch = 5
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(nch, nout * M * 2), dtype=np.uint8)
    # include the extremes of the u8 range
    iq[0, :64] = 0
    iq[1, :64] = 255
    taps = (rng.normal(size=(nch, ntaps, 2)) / M / 127.5).astype(np.float32)
    if ntaps == M and M % 8 == 0:
        taps[0] = O.rtl_taps(131525000, 131850000, M)[:ntaps]
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk)
    dec.set_taps(taps)
    dec.in_callback(iq)
    for c in range(nch):
        want = O.fir_u8(iq[c], M, taps[c], nout=nout, ntaps=ntaps)
        got = dec.dm(c, nout)
        assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6), (c, np.abs(got - want).max())
    dec.close()


def test_fir_shared_stream_and_stream_map(D, O):
    """rtl.c shape: several channels read ONE stream; and an explicit channel->stream map."""
    rng = np.random.default_rng(5)
    M, nch = 160, 6
    iq = rng.integers(0, 256, size=(2, 1024 * M * 2), dtype=np.uint8)
    taps = np.stack([O.rtl_taps(131525000 + 25000 * c, 131850000, M) for c in range(nch)])
    dec = D.Decoder(nch, decim=M, nstreams=2, max_blocks=1)
    dec.set_taps(taps)
    smap = [0, 1, 1, 0, 1, 0]
    dec.set_channel_streams(smap)
    dec.in_callback(iq)
    for c in range(nch):
        want = O.fir_u8(iq[smap[c]], M, taps[c])
        got = dec.dm(c, 1024)
        assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6)
    dec.close()


@pytest.mark.parametrize("M", [200, 160, 192])
def test_fir_shared_stream_kernel_ragged_groups(D, O, M, tune):
    """the shared-stream down-converter (one tile load + one u8->f32 conversion for all channels of a
    stream, groups of <= 8): streams feeding 1, 3, 8, 11 and 16 channels in scrambled channel order;
    dm within tolerance of the oracle AND bit-identical to the one-channel-per-unit kernel.  (Round 6: this vector-pipe kernel
    is the fallback for rtlMult values the matrix-pipe kernel does not take -- fir_mm.hip, tests/test_gpu_round6.py -- and is
    selected here with ACG_FIR_MM=0.)"""
    tune("ACG_FIR_MM", "0")
    rng = np.random.default_rng(77 + M)
    sizes = [1, 3, 8, 11, 16, 1]
    smap = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(smap)
    nch, nblk = int(smap.size), 2
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(len(sizes), nout * M * 2), dtype=np.uint8)
    taps = np.stack([O.rtl_taps(131000000 + 25000 * int(rng.integers(-40, 41)), 131000000, M) for c in range(nch)])

    def run():
        dec = D.Decoder(nch, decim=M, nstreams=len(sizes), max_blocks=nblk)
        dec.set_taps(taps)
        dec.set_channel_streams(smap)
        dec.in_callback(iq)
        out = np.stack([dec.dm(c, nout) for c in range(nch)])
        # taps changed after the first call: the regrouped tap table must follow
        dec.set_taps(taps[::-1].copy())
        dec.in_callback(iq)
        out2 = np.stack([dec.dm(c, nout) for c in range(nch)])
        dec.close()
        return out, out2

    shared, shared2 = run()
    tune("ACG_FIR_SHARED", "0")
    tune("ACG_FIR_VARIANT", "3")          # the workgroup-granular kernel shares the tap split and reduction order
    plain, plain2 = run()
    assert np.array_equal(shared, plain) and np.array_equal(shared2, plain2)
    tune("ACG_FIR_VARIANT", None)               # the default (wave-private) kernel: same sums in another order
    direct, direct2 = run()
    assert np.all(np.abs(direct - shared) <= 1e-5 * np.abs(shared) + 1e-6) and np.all(np.abs(direct2 - shared2) <= 1e-5 * np.abs(shared2) + 1e-6)
    for c in range(nch):
        want = O.fir_u8(iq[smap[c]], M, taps[c], nout=nout)
        assert np.all(np.abs(shared[c] - want) <= 1e-5 * np.abs(want) + 1e-6), c
        want2 = O.fir_u8(iq[smap[c]], M, taps[nch - 1 - c], nout=nout)
        assert np.all(np.abs(shared2[c] - want2) <= 1e-5 * np.abs(want2) + 1e-6), c


@pytest.mark.parametrize("M,ntaps", [(200, 200), (200, 192), (160, 160), (192, 192), (164, 164), (320, 320), (160, 37)])
def test_exact_order_mode_dm_is_bit_identical_to_oracle(D, O, M, ntaps):
    """ACG_F_EXACT_FIR: the down-converter in the reference's own order of operations (rtl.c:335-353 as an IEEE build runs it:
    127.37 per sample, products / difference / sum rounded separately, the terms added one after the other) -- dm equals the
    oracle's (== the -O2 reference's, tests/test_oracle_vs_ref.py) in every bit, extremes of the u8 range included."""
    rng = np.random.default_rng(31 * M + ntaps)
    nch, nblk = 5, 2
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(nch, nout * M * 2), dtype=np.uint8)
    iq[0, :4096] = 0
    iq[1, :4096] = 255
    win = np.ones(ntaps) if ntaps == M else np.hamming(ntaps) / np.hamming(ntaps).mean() * (M / ntaps)
    taps = np.stack([(O.rtl_taps(131000000 + 25000 * (c + 1), 131000000, M)[:ntaps] * win[:, None]).astype(np.float32) for c in range(nch)])
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, exact_fir=True)
    dec.set_taps(taps)
    dec.in_callback(iq)
    for c in range(nch):
        want = O.fir_u8(iq[c], M, taps[c], nout=nout, ntaps=ntaps)
        got = dec.dm(c, nout)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c, float(np.abs(got - want).max()))
    dec.close()
    # rtl.c's own shape: all channels on one dongle stream (scrambled channel -> stream map on two streams)
    smap = np.array([1, 0, 0, 1, 0])
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=2, max_blocks=nblk, exact_fir=True)
    dec.set_taps(taps)
    dec.set_channel_streams(smap)
    dec.in_callback(iq[:2])
    for c in range(nch):
        want = O.fir_u8(iq[smap[c]], M, taps[c], nout=nout, ntaps=ntaps)
        assert np.array_equal(dec.dm(c, nout).view(np.uint32), want.view(np.uint32)), c
    dec.close()


def test_exact_order_mode_is_bit_identical_end_to_end(D, O, S):
    """The parity argument in one test.  Noisy ACARS channels (20 dB SNR in the channel, long noise-only gaps: where razor-edge
    soft decisions live) through (a) the exact-order mode and (b) the streaming kernel, same GPU demodulator behind both:
      (a) blocks, every soft bit, and the final loop state identical to oracle down-converter -> oracle demodulator: with the
          same dm the GPU path is the reference, bit for bit, end to end;
      (b) dm within 1e-5 relative, blocks bit-exact against the oracle's demodulator fed with (b)'s own dm -- the streaming
          path's one and only deviation is the re-associated sum."""
    M, nblk, nch = 200, 24, 12
    nout = nblk * 1024
    sigma = 0.25 * 0.5 * (M / (2.0 * 10 ** 2.0)) ** 0.5
    rows, taps = [], []
    for c in range(nch):
        a, _ = S.channel_audio(np.random.default_rng(0xACA25 + c), nout, gap=(3125, 12500), text_len=(20, 220))
        off = 25000.0 * (2 + c) * (-1) ** c
        rows.append(S.iq_u8_from_envelopes((0.5 * (1 + 0.5 * a))[None, :], M, [off], phases=[0.37 * c], noise=sigma,
                                           rng=np.random.default_rng(500 + c)).reshape(-1))
        taps.append(O.rtl_taps(131000000 + int(off), 131000000, M))
    iq, taps = np.stack(rows), np.stack(taps)
    want, want_bits, want_state, dm_o = [], [], [], []
    for c in range(nch):
        ch = O.Channel(c, max_bits=nout // 4 + 8)
        dm_o.append(O.fir_u8(iq[c], M, taps[c]))
        ch.demod(dm_o[c])
        want.append([O.frame_tuple(f) for f in ch.frames])
        want_bits.append(ch.bits)
        want_state.append(ch.state())
    assert sum(len(w) for w in want) >= nch
    for exact in (True, False):
        dec = D.Decoder(nch, decim=M, max_blocks=nblk, exact_fir=exact)
        dec.set_taps(taps)
        dec.in_callback(iq)
        got = blocks_by_channel(dec.drain_frames(), D.frame_tuple)
        for c in range(nch):
            dm = dec.dm(c, nout)
            if exact:
                assert np.array_equal(dm.view(np.uint32), dm_o[c].view(np.uint32)), c
                assert got.get(c, []) == want[c], c
                vo, lvl = dec.bits(c)
                # (the mixer's sin/cos is the one operation that is not the reference's: <= 2.1 ulp in f64, and the float
                #  products the loop keeps equal glibc's except about once in 1e8 -- so: every soft bit identical, allowing
                #  for one such event in this test's 7e5 products)
                assert_soft_close(vo, want_bits[c][0], lvl, want_bits[c][1])
                assert (vo.view(np.uint32) == want_bits[c][0].view(np.uint32)).mean() >= 0.9999, c
                s, o = dec.state(c), want_state[c]
                for k in ("MskS", "idx", "nbits", "Acarsstate", "outbits", "MskBitCount"):
                    assert s[k] == o[k], (c, k)
                assert_state_close(s, o, "exact-order mode ch %d" % c)
            else:
                assert np.all(np.abs(dm - dm_o[c]) <= 1e-5 * np.abs(dm_o[c]) + 1e-6), c
                ch = O.Channel(c)
                ch.demod(dm)
                assert got.get(c, []) == [O.frame_tuple(f) for f in ch.frames], c
        dec.close()


FIR_VARIANT_CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from acarsdec_amd import decoder as D
from oracle import oracle as O
M = int(sys.argv[1])
rng = np.random.default_rng(99)
nch, nblk = 6, 2
nout = nblk * 1024
iq = rng.integers(0, 256, size=(nch, nout * M * 2), dtype=np.uint8)
taps = np.stack([O.rtl_taps(131000000 + 25000 * (c + 1), 131000000, M) for c in range(nch)])
dec = D.Decoder(nch, decim=M, max_blocks=nblk)
dec.set_taps(taps)
dec.in_callback(iq)
worst = 0.0
for c in range(nch):
    want = O.fir_u8(iq[c], M, taps[c], nout=nout)
    got = dec.dm(c, nout)
    assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-6), c
    worst = max(worst, float(np.abs(got - want).max()))
print("OK", worst)
'''


@pytest.mark.parametrize("variant,M", [("0", 200), ("1", 200), ("2", 160), ("4", 200), ("3", 192), ("3", 200), ("5", 200), ("5", 160), ("5", 192),
                                       ("54", 200), ("6", 200), ("7", 200), ("8", 200)])
def test_fir_kernel_variants_all_match_oracle(variant, M):
    """the down-converter's alternative kernels stay selectable (ACG_FIR_VARIANT: 0 one workgroup per
    segment, 1/2 static persistent partition without/with non-temporal loads, 3 the workgroup-granular dynamic
    dispenser, 4 LDS-DMA double buffering, 5 the default wave-private streaming kernel, 54 the same with 127.37
    subtracted per sample, 6 the matrix-pipe experiment, 7 the wave-private kernel with register-resident taps, 8 the same
    with the results parked in LDS and written in chip-wide bursts): each one against the oracle, in a fresh process.  Only 5
    and 3 exist in the product library; the others live in the lab build (libacarsdec_amd_lab.so), which the child loads
    instead (ACARSDEC_AMD_LIB) -- and in the product library a lab variant's number must fall back to the default kernel."""
    from acarsdec_amd import _capi as K
    env = dict(os.environ, ACG_FIR_VARIANT=variant, ACG_ALLOW_TUNING="1")
    if variant not in ("3", "5"):
        r = subprocess.run([sys.executable, "-c", FIR_VARIANT_CHILD % dict(root=ROOT), str(M)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), (r.stdout[-500:], r.stderr[-1500:])     # product: default kernel
        env["ACARSDEC_AMD_LIB"] = K.LAB_PATH
    r = subprocess.run([sys.executable, "-c", FIR_VARIANT_CHILD % dict(root=ROOT), str(M)], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("M,ntaps,variant", [(200, 200, None), (200, 192, None), (160, 160, None), (192, 192, None),
                                             (200, 200, "7"), (200, 192, "7"), (200, 200, "8"), (200, 192, "8")])
def test_fir_direct_kernel_many_runs_scrambled_streams(D, O, M, ntaps, variant, tune):
    """the wave-private streaming kernel where its run dispenser matters: far more runs than resident waves (every
    wave goes through many tickets, shards run dry and waves move on to the next shard), a channel -> stream map that
    is a permutation (row lookup instead of the identity shortcut), fewer taps than the window (zero columns), and
    three launches in a row on the same dispenser (it re-arms itself).  Every channel against the oracle."""
    if variant:
        tune("ACG_FIR_VARIANT", variant, lab=True)      # (looked up at every launch) 7: register-resident taps, 26 loads per tile; lab build
    rng = np.random.default_rng(4242 + M + ntaps)
    nch, nblk = 300, 8
    nout = nblk * 1024
    iq = torch_randint_u8((nch, nout * M * 2), seed=M)
    host = iq.cpu().numpy()
    win = np.ones(ntaps) if ntaps == M else np.hamming(ntaps) / np.hamming(ntaps).mean() * (M / ntaps)
    taps = np.stack([(O.rtl_taps(131000000 + 25000 * int(rng.integers(-40, 41)), 131000000, M)[:ntaps] * win[:, None]).astype(np.float32)
                     for c in range(nch)])
    perm = rng.permutation(nch)
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False, lab=bool(variant))
    dec.set_taps(taps)
    want = [None] * nch
    for rnd, smap in enumerate((np.arange(nch), perm, perm)):
        dec.set_channel_streams(smap)
        dec.fir_only(iq, nblk, iq.stride(0))
        for c in range(nch):
            if rnd < 2:
                want[c] = O.fir_u8(host[smap[c]], M, taps[c], nout=nout, ntaps=ntaps)
            got = dec.dm(c, nout)
            assert np.all(np.abs(got - want[c]) <= 1e-5 * np.abs(want[c]) + 1e-6), (rnd, c)
    dec.close()


def torch_randint_u8(shape, seed):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)


def test_placement_trial_leaves_the_decoder_reset(D, O):
    """acg_placement_trial / decoder.best_placed: a few contexts, the same call timed on each, the fastest kept -- and the
    kept one behaves like a fresh decoder afterwards (state and queues reset): same blocks as the oracle from a cold start."""
    import torch
    from acarsdec_amd import synth as S
    rng = np.random.default_rng(99)
    M, nch, nblk = 160, 6, 4
    nout = nblk * 1024
    freqs = [131525000 + 25000 * k for k in range(nch)]
    fc = 131700000
    env = []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, nframes=1, gap=(800, 1500), text_len=(10, 30))
        env.append(0.5 * (1 + 0.5 * a))
    iq = np.stack([S.iq_u8_from_envelopes(np.array(env[c])[None, :], M, [freqs[c] - fc], noise=0.01, rng=rng) for c in range(nch)])
    d = torch.from_numpy(iq).cuda()
    taps = np.stack([O.rtl_taps(freqs[c], fc, M) for c in range(nch)]).astype(np.float32)

    def factory():
        dec = D.Decoder(nch, decim=M, max_blocks=nblk, bitlog=False)
        dec.set_taps(taps)
        return dec
    dec, ms, best = D.best_placed(factory, 3, d, nblk, d.stride(0))
    assert len(ms) == 3 and all(x > 0 for x in ms) and ms[best] == min(ms)
    dec.in_callback(d, nblocks=nblk, pitch=d.stride(0))
    got = sorted(D.frame_tuple(f) for f in dec.drain_frames())
    want = []
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_u8(iq[c], M, taps[c]))
        want += [O.frame_tuple(f) for f in ch.frames]
    assert got == sorted(want) and len(got) >= nch
    from acarsdec_amd import _capi as K
    import ctypes as C
    ms1 = C.c_double(0)
    assert dec.L.acg_placement_trial(dec.ctx, d.data_ptr(), d.stride(0), nblk, 0, None, C.byref(ms1)) == K.EINVAL       # repeats < 1
    assert dec.L.acg_placement_trial(dec.ctx, None, d.stride(0), nblk, 1, None, C.byref(ms1)) == K.EINVAL
    dec.close()
    # the same for a sample-format input (acg_placement_trial_samples): CS16 here
    M2 = 200
    iq16 = np.stack([S.iq_s16_from_envelopes(np.array(env[c])[None, :], M2, [freqs[c] - fc], noise=0.01, rng=rng) for c in range(nch)])
    d16 = torch.from_numpy(iq16).cuda()
    taps16 = np.stack([O.soapy_taps(freqs[c], fc, M2) for c in range(nch)]).astype(np.float32)

    def factory16():
        dec = D.Decoder(nch, decim=M2, max_blocks=nblk, bitlog=False)
        dec.set_taps(taps16)
        return dec
    pitch16 = d16.stride(0) * 2                              # bytes
    dec, ms, best = D.best_placed(factory16, 2, d16, nblk, pitch16, fmt=K.FMT_CS16)
    assert len(ms) == 2 and all(x > 0 for x in ms) and ms[best] == min(ms)
    dec.process_samples(K.FMT_CS16, d16, nblk, pitch16)
    got = sorted(D.frame_tuple(f) for f in dec.drain_frames())
    want = []
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_cs16(iq16[c], M2, taps16[c]))
        want += [O.frame_tuple(f) for f in ch.frames]
    assert got == sorted(want) and len(got) >= nch
    assert dec.L.acg_placement_trial_samples(dec.ctx, K.FMT_CS16, d16.data_ptr(), pitch16, 0, nblk, 0, None, C.byref(ms1)) == K.EINVAL
    dec.close()


# ------------------------------------------------------------------------------------ error behaviour of the ABI
def test_abi_rejects_misuse_loudly(D):
    """bad arguments and call-sequence errors come back as codes with a message, never as silent no-ops:
    the reference's conventions are 0 = OK / non-zero = fail (acarsdec.c:456-459)."""
    import torch
    from acarsdec_amd import _capi as K
    L = K.load()
    M = 160
    dec = D.Decoder(2, decim=M, max_blocks=2, bitlog=False)
    iq = torch.zeros((2, 2 * 1024 * M * 2 + 64), dtype=torch.uint8, device="cuda")
    assert L.acg_process_iq_u8_dev(dec.ctx, iq.data_ptr(), iq.stride(0), 3, None) == K.EINVAL      # > max_blocks
    assert b"nblocks" in L.acg_last_error(dec.ctx)
    assert L.acg_process_iq_u8_dev(dec.ctx, iq.data_ptr() + 4, iq.stride(0), 1, None) == K.EINVAL  # misaligned base
    assert L.acg_process_iq_u8_dev(dec.ctx, iq.data_ptr(), 1024 * M * 2 - 16, 1, None) == K.EINVAL  # pitch < row
    assert L.acg_process_iq_u8_dev(dec.ctx, None, iq.stride(0), 1, None) == K.EINVAL
    assert L.acg_set_channel_streams(dec.ctx, (C.c_int * 2)(0, 2)) == K.EINVAL                    # stream out of range
    with pytest.raises(K.AcgError) as e:
        dec.bits(0)                                                        # context has no bit log
    assert e.value.code == K.ESTATE
    assert L.acg_process_samples_dev(dec.ctx, 9, iq.data_ptr(), iq.stride(0), 0, 1, None) == K.EINVAL   # unknown format
    # the context still works after the rejected calls
    dec.in_callback(iq[:, : 1024 * M * 2].contiguous())
    assert dec.drain_frames() == []
    dec.close()
    # u8 path above RTLMULTMAX, split planes above their limit
    dec = D.Decoder(1, decim=400, max_blocks=1)
    big = torch.zeros(1024 * 400 * 4, dtype=torch.uint8, device="cuda")
    assert L.acg_process_iq_u8_dev(dec.ctx, big.data_ptr(), big.numel(), 1, None) == K.EINVAL
    assert L.acg_process_samples_dev(dec.ctx, K.FMT_S16_SPLIT, big.data_ptr(), big.numel(), big.numel() // 2, 1, None) == K.EINVAL
    assert L.acg_process_samples_dev(dec.ctx, K.FMT_CS16, big.data_ptr(), big.numel(), 0, 1, None) == K.OK
    dec.close()
    # queue smaller than the result: EOVERFLOW, truncated, not corrupted
    fr = (K.Frame * 1)()
    n = C.c_int(0)
    assert L.acg_drain_frames(None, fr, 1, C.byref(n)) == K.EINVAL


def test_process_dm_dev_input_may_be_refilled_in_place(D, O):
    """acg_process_dm_dev is asynchronous like the iq entry points: the caller may overwrite dm_dev with the next
    chunk on the same stream right after the call -- the demodulator that reads it runs on another stream and the
    refill must be ordered behind it.  16 chunks through ONE device buffer, blocks vs the oracle."""
    import torch
    from acarsdec_amd import _capi as K, synth as S
    L = K.load()
    nch, chunk, nchunk = 64, 4096, 16
    rng = np.random.default_rng(31)
    audio = np.stack([S.envelope(S.channel_audio(np.random.default_rng(500 + c), chunk * nchunk, gap=(1500, 4000), text_len=(10, 60))[0],
                                 noise=0.01, rng=rng) for c in range(nch)]).astype(np.float32)
    src = torch.from_numpy(audio).cuda()
    buf = torch.empty((nch, chunk), dtype=torch.float32, device="cuda")
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=chunk // 1024, bitlog=False)
    st = torch.cuda.Stream()
    got = []
    with torch.cuda.stream(st):
        for k in range(nchunk):
            buf.copy_(src[:, k * chunk:(k + 1) * chunk])                  # refill in place, same stream
            assert L.acg_process_dm_dev(dec.ctx, buf.data_ptr(), chunk, chunk, st.cuda_stream) == 0
            buf.mul_(0.0)                                                 # and scribble over it right away
    got = sorted(D.frame_tuple(f) for f in dec.drain_frames(8192))
    want = []
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(audio[c])
        want += [O.frame_tuple(f) for f in ch.frames]
    assert got == sorted(want) and len(got) >= nch
    dec.close()


# ------------------------------------------------------------------------------------ MSK stage on test.wav
def run_wav(D, x, chunk):
    nch = x.shape[1]
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=(chunk + 1023) // 1024)
    frames, vo, lvl = [], [[] for _ in range(nch)], [[] for _ in range(nch)]
    for s in range(0, x.shape[0], chunk):
        dec.demod_msk(np.ascontiguousarray(x[s:s + chunk].T))
        frames += dec.drain_frames()
        for c in range(nch):
            a, b = dec.bits(c)
            vo[c].append(a)
            lvl[c].append(b)
    st = [dec.state(c) for c in range(nch)]
    dec.close()
    return frames, [np.concatenate(v) for v in vo], [np.concatenate(v) for v in lvl], st


@pytest.mark.parametrize("chunk", [4096, 1024, 1000, 53843])
def test_testwav_blocks_bits_state(D, testwav, golden, golden_bits, chunk):
    """config 1: the reference's only data fixture; 7 blocks, bit-exact, any chunking."""
    if chunk > 8192:
        chunk = ((testwav.shape[0] + 1023) // 1024) * 1024
    frames, vo, lvl, st = run_wav(D, testwav, chunk)
    gb = golden_blocks(golden["file"]["raw_blocks"])
    want = {}
    for t, l in gb:
        want.setdefault(t[0], []).append((t, l))
    got = blocks_by_channel(frames, D.frame_tuple)
    assert sorted(got) == sorted(want)
    for c in want:
        assert got[c] == [t for t, _ in want[c]], "blocks differ on channel %d" % c
    for f in frames:
        ref_lvl = [l for t, l in gb if t == D.frame_tuple(f)][0]
        assert abs(f.lvl - ref_lvl) < 0.05
    for c in range(4):
        m = golden_bits["chn"] == c
        vo_o, lvl_o = soft_from_v(golden_bits["vr"][m], golden_bits["vi"][m], golden_bits["MskS"][m])
        assert len(vo[c]) == golden["file"]["bits_per_channel"][c]
        assert_soft_close(vo[c], vo_o, lvl[c], lvl_o)
        g = golden["file"]["final_state"][c]
        assert st[c]["MskS"] == g["MskS"] and st[c]["idx"] == g["idx"]
        assert st[c]["nbits"] == g["nbits"] and st[c]["Acarsstate"] == g["Acarsstate"]
        assert st[c]["outbits"] == g["outbits"] and st[c]["MskBitCount"] == g["MskBitCount"]
        assert_state_close(st[c], dict(MskDf=fhex(g["MskDf"]), MskClk=fhex(g["MskClk"]), MskPhi=fhex(g["MskPhi"])), "golden test.wav ch %d" % c)


def test_shared_reciprocal_normalisation_is_ieee_division():
    """msk.c:111 divides Re v and Im v by the same double (lvl + 1e-8); the device computes one reciprocal and finishes
    each quotient with the remainder step of the IEEE algorithm.  Over the operand range the loop produces
    (d = |v| + 1e-8 with |v| from exact silence to a saturated filter, numerators up to |v|) the quotients must be
    bit-identical to the compiler's IEEE division, on 4 million random triples plus the edges."""
    from acarsdec_amd import _capi as K
    L = K.load()
    rng = np.random.default_rng(2024)
    n = 1 << 22
    lvl = np.concatenate([10.0 ** rng.uniform(-12, 2, n - 8), [0.0, 1e-45, 1e-38, 1.17549435e-38, 3.4e38 ** 0.25, 1.0, 127.5, 1e-8]]).astype(np.float32)
    d = lvl.astype(np.float64) + 1e-8
    ang = rng.uniform(0, 2 * np.pi, n)
    vr = (lvl * np.cos(ang)).astype(np.float32).astype(np.float64)
    vi = (lvl * np.sin(ang)).astype(np.float32).astype(np.float64)
    vr[:16] = 0.0
    vi[16:32] = 0.0
    vr[32:48] = lvl[32:48]
    vr[vr == 0] = 0.0                            # the filter output is a sum that starts from +0: it is never -0,
    vi[vi == 0] = 0.0                            # the one numerator whose quotient's sign the shared form would lose
    out = np.zeros((n, 8), dtype=np.float64)
    assert L.acg_selftest_div2(vr.ctypes.data, vi.ctypes.data, d.ctypes.data, out.ctypes.data, n) == K.OK
    assert np.array_equal(out[:, 0].view(np.uint64), out[:, 2].view(np.uint64))
    assert np.array_equal(out[:, 1].view(np.uint64), out[:, 3].view(np.uint64))
    assert np.array_equal(out[:, 2], vr / d) and np.array_equal(out[:, 3], vi / d)          # and the device's IEEE division is IEEE
    assert np.array_equal(out[:, 6].view(np.uint64), out[:, 2].view(np.uint64))             # the single-quotient form (tap phase)
    # |v| = sqrt(re^2 + im^2) without the exponent scaling: bit-identical to the IEEE square root, which is numpy's
    assert np.array_equal(out[:, 4].view(np.uint64), out[:, 5].view(np.uint64))
    assert np.array_equal(out[:, 5], np.sqrt(out[:, 7]))
    # the tap phase's operands: clock in (-1, 1) over s = 0.9 +- 0.05
    clk = rng.uniform(-1, 1, n).astype(np.float32).astype(np.float64)
    sden = 1800.0 / 12500 * 2 * np.pi + rng.uniform(-0.05, 0.05, n)
    assert L.acg_selftest_div2(clk.ctypes.data, clk.ctypes.data, sden.ctypes.data, out.ctypes.data, n) == K.OK
    assert np.array_equal(out[:, 6].view(np.uint64), out[:, 2].view(np.uint64)) and np.array_equal(out[:, 2], clk / sden)


def test_device_message_split_matches_reference_json_and_oracle(D, O, msgsplit_golden):
    """the batch sink.  The golden recording (transmissions covering every branch of outputmsg()'s
    field split) through demodulator + framing + block repair + field split, all on the device: the fixed binary
    records equal the JSON the unmodified reference program printed, field for field, and the oracle's split."""
    from conftest import msg_fields_from_json, msg_fields_from_record
    pcm, want = msgsplit_golden
    x = pcm.astype(np.float32) / 32768.0
    chunk = 4096
    pad = (-x.size) % chunk
    x = np.concatenate([x, np.zeros(pad, dtype=np.float32)])
    nch = 3                                           # the same recording on three channels
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False)
    msgs = []
    for s in range(0, x.size, chunk):
        dec.demod_msk(np.tile(x[s:s + chunk], (nch, 1)))
        msgs += dec.drain_msgs()
    assert len(msgs) == nch * len(want)
    for c in range(nch):
        got = [msg_fields_from_record(m) for m in msgs if m.chn == c]
        ref = [dict(msg_fields_from_json(j), chn=c) for j in want]
        assert got == ref, c
    ch = O.Channel(0, max_frames=512)
    ch.demod(x)
    orc = [O.msg_tuple(O.msg_split(b)) for b in (O.blk_process(f) for f in ch.frames) if b is not None]
    assert [O.msg_tuple(m) for m in msgs if m.chn == 0] == orc
    dec.close()
    # without ACG_F_REPAIR the sink refuses (outputmsg() only ever sees repaired blocks)
    from acarsdec_amd import _capi as K
    dec = D.Decoder(1, decim=8, ntaps=8, max_blocks=1)
    with pytest.raises(K.AcgError) as e:
        dec.drain_msgs()
    assert e.value.code == K.ESTATE
    dec.close()


def test_message_drain_keeps_what_does_not_fit_and_records_are_fully_defined(D, msgsplit_golden):
    """A drain into a buffer that is too small hands out the OLDEST messages that fit, reports ACG_EAGAIN ("call again")
    and keeps the rest queued -- nothing is lost; every byte of a record is defined (the device staging buffer comes from
    hipMalloc: the split clears the record before it fills it); the Python wrappers loop on ACG_EAGAIN instead of raising
    (ADVICE r03), and the block drain has the same keep-the-rest contract."""
    from acarsdec_amd import _capi as K
    pcm, want = msgsplit_golden
    x = pcm.astype(np.float32) / 32768.0
    chunk = 8192
    x = np.concatenate([x, np.zeros((-x.size) % chunk, dtype=np.float32)])
    dec = D.Decoder(2, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False)
    for s in range(0, x.size, chunk):
        dec.demod_msk(np.tile(x[s:s + chunk], (2, 1)))          # nothing drained in between: everything queues up
    total = 2 * len(want)
    got, rounds = [], 0
    while True:
        buf = (K.Msg * 7)()
        n = C.c_int(0)
        rc = dec.L.acg_drain_msgs(dec.ctx, buf, 7, C.byref(n))
        assert rc in (K.OK, K.EAGAIN), rc
        got += [K.Msg.from_buffer_copy(buf[i]) for i in range(n.value)]
        rounds += 1
        if rc == K.OK:
            break
        assert 0 < n.value <= 7 and b"call again" in dec.L.acg_last_error(dec.ctx)
    assert len(got) == total and rounds >= (total + 6) // 7
    ref = dec2 = None
    dec2 = D.Decoder(2, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False)
    for s in range(0, x.size, chunk):
        dec2.demod_msk(np.tile(x[s:s + chunk], (2, 1)))
    ref = dec2.drain_msgs(4096)
    key = lambda m: (int(m.chn), int(m.end_bit))
    assert sorted(bytes(m) for m in got) == sorted(bytes(m) for m in ref) and len({key(m) for m in got}) == total
    # the wrappers: a small buffer is looped over, never raised on; collect (lag 0) likewise
    for take in ("drain", "collect"):
        # (a host that collects after every call names its lag and gets the smallest queue; one that only drains at the end keeps the default)
        d3 = D.Decoder(2, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False, max_lag=1 if take == "collect" else 0)
        assert d3.max_lag == (1 if take == "collect" else 6)
        some = []
        for s0 in range(0, x.size, chunk):
            d3.demod_msk(np.tile(x[s0:s0 + chunk], (2, 1)))
            some += d3.collect_msgs(lag=0, max_msgs=5) if take == "collect" else []
        some += d3.drain_msgs(5)
        assert sorted(bytes(m) for m in some) == sorted(bytes(m) for m in ref), take
        d3.close()
    # blocks: the same keep-the-rest contract
    d4 = D.Decoder(2, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False)
    for s0 in range(0, x.size, chunk):
        d4.demod_msk(np.tile(x[s0:s0 + chunk], (2, 1)))
    fb = (K.Frame * 3)()
    nfr, codes = 0, set()
    while True:
        rc = d4.L.acg_drain_frames(d4.ctx, fb, 3, C.byref(n))
        codes.add(rc)
        nfr += n.value
        if rc != K.EAGAIN:
            break
    assert rc == K.OK and nfr == total and K.EAGAIN in codes
    assert len(d4.drain_frames(5)) == 0
    d4.close()
    for m in got:                                               # text beyond txt_len and the reserved fields are zero
        assert bytes(m.txt[m.txt_len:]) == bytes(242 - m.txt_len) and m.reserved1 == 0 and m.reserved3 == 0 and 0 <= m.soh_sample < m.end_sample
    dec.close()
    dec2.close()


def test_collect_behind_a_drain_finds_nothing_pending(D, msgsplit_golden):
    """A host that drains everything and then goes on collecting `lag` calls behind (bench.py: a pass ends with a drain, the next
    pass collects with lag 2) asks for calls whose blocks the drain has already handed out: their published queue length lies
    behind the consumer.  That is "nothing pending", not a lapped queue (round 4's first lag-2 run read the negative difference as
    4 294 9xx xxx lost blocks)."""
    from acarsdec_amd import _capi as K
    pcm, want = msgsplit_golden
    x = pcm.astype(np.float32) / 32768.0
    chunk = 8192
    x = np.concatenate([x, np.zeros((-x.size) % chunk, dtype=np.float32)])
    dec = D.Decoder(2, decim=8, ntaps=8, max_blocks=chunk // 1024, repair=True, bitlog=False, max_lag=2)
    got = []
    for rnd in range(2):
        for s in range(0, x.size, chunk):
            dec.demod_msk(np.tile(x[s:s + chunk], (2, 1)))
            got += dec.collect_msgs(lag=2, max_msgs=64)
        got += dec.drain_msgs(64)                    # ... and the next round's first collects look behind this drain
        n, buf = dec.collect_frames_raw(lag=2, max_frames=16)
        assert n == 0
    assert 2 * len(want) <= len(got) <= 2 * 2 * len(want) + 4          # (round 2 starts from round 1's loop state: nearly always the same messages)
    dec.close()


def test_collect_lag_is_bounded_by_what_the_block_queue_was_sized_for(D):
    """acg_max_lag(): the block queue holds the worst case of max_lag + 1 calls (it held two calls' worth while
    the API allowed a lag of 6), 6 where that costs <= 512 MiB, fewer for very wide contexts; a larger lag is refused."""
    from acarsdec_amd import _capi as K
    dec = D.Decoder(64, decim=8, ntaps=8, max_blocks=2, bitlog=False)
    assert dec.max_lag == 6
    n = C.c_int(0)
    buf = (K.Frame * 4)()
    assert dec.L.acg_collect_frames(dec.ctx, 6, buf, 4, C.byref(n)) == K.OK and dec.L.acg_collect_frames(dec.ctx, 7, buf, 4, C.byref(n)) == K.EINVAL
    dec.close()
    wide = D.Decoder(16384, decim=200, max_blocks=8, bitlog=False)
    assert 1 <= wide.max_lag < 6
    assert wide.L.acg_collect_frames(wide.ctx, wide.max_lag + 1, buf, 4, C.byref(n)) == K.EINVAL
    assert b"acg_max_lag" in wide.L.acg_last_error(wide.ctx)
    wide.close()


def assert_state_close(got, want, what):
    """The loop's continuous state against the reference's.  The device differs from glibc only in the last bit of
    the mixer's f64 sin/cos (< 1 ulp, and only the float-rounded product is kept, msk.c:90): a product moves by one
    f32 ulp about once in 2^29 samples, which the PLL (a contraction, msk.c:130) forgets within a few bits.  So the
    state agrees far below the soft-symbol tolerance: MskDf 1e-9, MskClk (f32, up to 3*pi/2: two ulps = 1e-6) and
    MskPhi (mod 2*pi) 1e-9.  Every deviation seen so far is exactly 0 (they are appended to
    gpurun_out/state_deviation.txt)."""
    ddf = abs(got["MskDf"] - want["MskDf"])
    dclk = abs(got["MskClk"] - want["MskClk"])
    dphi = abs(got["MskPhi"] - want["MskPhi"])
    dphi = min(dphi, abs(dphi - 2 * np.pi))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "state_deviation.txt"), "a") as f:
            f.write("%s dDf=%.3e dClk=%.3e dPhi=%.3e\n" % (what, ddf, dclk, dphi))
    except OSError:
        pass
    assert ddf < 1e-9 and dclk < 1e-6 and dphi < 1e-9, (what, ddf, dclk, dphi)


def test_msk_matches_oracle_noise_and_silence(D, O):
    """noise-only (the FSM reset path fires ~19x/s), exact silence (v == 0 -> vo == 0) and a
    saturated level, against the oracle run live."""
    rng = np.random.default_rng(11)
    n = 8192
    x = np.zeros((4, n), dtype=np.float32)
    x[0] = rng.normal(0.3, 0.1, n)
    x[1] = 0.0
    x[2] = 1.42
    x[3] = np.abs(rng.normal(0, 1e-3, n))
    dec = D.Decoder(4, decim=8, ntaps=8, max_blocks=8)
    dec.demod_msk(x)
    for c in range(4):
        ch = O.Channel(c, max_bits=4096)
        ch.demod(x[c])
        vo_o, lvl_o = ch.bits
        vo_g, lvl_g = dec.bits(c)
        if c == 1:
            assert np.all(vo_g == 0) and np.all(lvl_g == 0) and len(vo_g) == len(vo_o)
        else:
            assert_soft_close(vo_g, vo_o, lvl_g, lvl_o)
        s, o = dec.state(c), ch.state()
        for k in ("MskS", "idx", "nbits", "Acarsstate", "outbits", "MskBitCount"):
            assert s[k] == o[k], (c, k, s[k], o[k])
    dec.close()


def test_msk_ragged_and_empty_lengths(D, O):
    """lengths that are not multiples of anything, including 0 and 1; state carries across calls."""
    rng = np.random.default_rng(3)
    from acarsdec_amd import synth as S
    a, _ = S.channel_audio(rng, 30000, nframes=3, gap=(3000, 5000), text_len=(5, 60))
    e = S.envelope(a, noise=0.01, rng=rng)
    dec = D.Decoder(1, decim=8, ntaps=8, max_blocks=8)
    ch = O.Channel(0, max_bits=8000)
    pos, got = 0, []
    for ln in [0, 1, 7, 1023, 1025, 4096, 5, 0, 8192, 3333]:
        seg = e[pos:pos + ln]
        pos += ln
        dec.demod_msk(seg.reshape(1, -1))
        got += dec.drain_frames()
        ch.demod(seg)
    seg = e[pos:pos + 8000]
    dec.demod_msk(seg.reshape(1, -1))
    got += dec.drain_frames()
    ch.demod(seg)
    assert [D.frame_tuple(f) for f in got] == [O.frame_tuple(f) for f in ch.frames]
    assert len(got) >= 2
    s, o = dec.state(0), ch.state()
    for k in ("MskS", "idx", "nbits", "Acarsstate", "outbits", "MskBitCount"):
        assert s[k] == o[k]
    dec.close()


# ------------------------------------------------------------------------------------ config 2: 8 channels, 2.0 Msps
def _config2(S, testwav, fc, offsets, phases):
    env = S.pad_blocks(0.5 + 0.5 * testwav.T.astype(np.float64), 1024, 0.5)
    env8 = np.concatenate([env, env[::-1]])          # 4 wav channels used twice
    return env8


def test_config2_shared_stream_8ch(D, O, S, testwav):
    """8 channels on ONE 2.0 Msps stream (the rtl.c shape), blocks bit-exact vs the oracle."""
    M = 160
    freqs = ["131.525", "131.725", "131.825", "131.550", "131.450", "131.475", "131.650", "131.125"]
    dec = D.Decoder(8, decim=M, nstreams=1, max_blocks=53)
    fc = dec.init_rtl(freqs)
    fr = [D.parse_freq_mhz(f) for f in freqs]
    assert fc == O.choose_fc(fr, M)
    env8 = _config2(S, testwav, fc, None, None)
    iq = S.iq_u8_from_envelopes(env8, M, [f - fc for f in fr], phases=np.linspace(0.1, 5.0, 8), scale=0.12)
    dec.in_callback(iq.reshape(1, -1))
    got = blocks_by_channel(dec.drain_frames(), D.frame_tuple)
    total = 0
    for c in range(8):
        ch = O.Channel(c, max_bits=12000)
        dm_o = O.fir_u8(iq, M, O.rtl_taps(fr[c], fc, M))
        dm_g = dec.dm(c, dm_o.size)
        assert np.all(np.abs(dm_g - dm_o) <= 1e-5 * np.abs(dm_o) + 1e-6)
        ch.demod(dm_o)
        assert got.get(c, []) == [O.frame_tuple(f) for f in ch.frames], "channel %d" % c
        total += len(ch.frames)
        # soft symbols: the MSK stage in isolation, i.e. the oracle demodulator fed the GPU's own dm
        # (between blocks dm is u8 quantisation noise ~1e-3, where the 1e-5-relative dm differences of
        # the two summation orders legitimately decorrelate the free-running PLLs)
        ch2 = O.Channel(c, max_bits=12000)
        ch2.demod(dm_g)
        assert [O.frame_tuple(f) for f in ch2.frames] == [O.frame_tuple(f) for f in ch.frames]
        vo_g, lvl_g = dec.bits(c)
        vo_o, lvl_o = ch2.bits
        assert_soft_close(vo_g, vo_o, lvl_g, lvl_o)
    assert total == 14       # the 7 blocks of test.wav, twice


def test_config2_one_stream_per_channel(D, O, S, testwav):
    """8 independent 2.0 Msps streams, one channel each (the roofline-relevant shape)."""
    M = 160
    env8 = _config2(S, testwav, None, None, None)
    nblk = 6
    off = [-325000, -300000, 75000, 150000, -50000, 25000, 400000, -475000]
    iq = np.stack([S.iq_u8_from_envelopes(env8[c:c + 1, 40 * 1024:(40 + nblk) * 1024], M, [off[c]], phases=[0.3 * c])
                   for c in range(8)])
    taps = np.stack([D.rtl_taps(131850000 + off[c], 131850000, M) for c in range(8)])
    dec = D.Decoder(8, decim=M, max_blocks=nblk)
    dec.set_taps(taps)
    dec.in_callback(iq)
    got = blocks_by_channel(dec.drain_frames(), D.frame_tuple)
    for c in range(8):
        ch = O.Channel(c)
        ch.demod(O.fir_u8(iq[c], M, taps[c]))
        assert got.get(c, []) == [O.frame_tuple(f) for f in ch.frames]
    dec.close()


# ------------------------------------------------------------------------------------ many channels, synthetic MSK
def test_many_channels_synthetic_msk(D, O, S):
    """96 independent noisy MSK channels at 12.5 kHz: every block bit-exact vs the oracle."""
    rng = np.random.default_rng(2024)
    nch, n = 96, 8 * 1024
    x = np.zeros((nch, n), dtype=np.float32)
    for c in range(nch):
        a, _ = S.channel_audio(rng, n, gap=(1500, 3000), text_len=(5, 40))
        x[c] = S.envelope(a, depth=0.5, carrier=0.1 + 0.4 * rng.random(), noise=0.004, rng=rng)
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=8)
    dec.demod_msk(x[:, :5000])
    fr = dec.drain_frames()
    dec.demod_msk(x[:, 5000:])
    fr += dec.drain_frames()
    got = blocks_by_channel(fr, D.frame_tuple)
    nblocks = 0
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(x[c])
        assert got.get(c, []) == [O.frame_tuple(f) for f in ch.frames], "channel %d" % c
        nblocks += len(ch.frames)
        for f in ch.frames:
            assert O.lib().orc_frame_check(f) == 0
    assert nblocks >= nch
    dec.close()


def test_device_sincos_keeps_the_mixer_products_of_libm(D):
    """the mixer's sin/cos (msk.c:90 calls cexp) is a 128-entry table + a rotation by the remainder (msk.hip sincos_tab):
    device result vs libm on 2e5 phases in [0, 2*pi): <= 2.5 ulp (3e-17 absolute near the zeros), and the float-rounded
    products in * cos, in * -sin -- what the demodulator keeps -- identical to libm's."""
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 200000), [0.0, np.pi / 4, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi - 1e-15],
                        np.arange(8) * (np.pi / 4) + 1e-9, np.arange(1, 9) * (np.pi / 4) - 1e-9,
                        np.arange(129) * (2 * np.pi / 128), (np.arange(128) + 0.5) * (2 * np.pi / 128)])
    x = x[x < 2 * np.pi + 1e-12]
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    assert K.load().acg_selftest_sincos(x.ctypes.data, s.ctypes.data, c.ctypes.data, x.size) == 0
    for got, want in ((s, np.sin(x)), (c, np.cos(x))):
        ulp = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 2.0 ** -80)
        ok = (ulp <= 2.5) | (np.abs(got - want) < 3e-17)      # absolute bound near the zeros
        assert ok.all(), (ulp.max(), x[np.argmax(ulp)])
    assert (s == np.sin(x)).mean() > 0.7 and (c == np.cos(x)).mean() > 0.7      # most are bit-identical to libm
    amp = rng.uniform(1e-3, 1.0, x.size).astype(np.float32).astype(np.float64)
    for got, want in ((c, np.cos(x)), (-s, -np.sin(x))):
        # (at the zeros of sin / cos both are ~1e-16 and differ by ~1e-20: a product of that size is nothing in the filter's sum)
        far = np.abs(want) > 1e-9
        assert np.count_nonzero(((amp * got).astype(np.float32) != (amp * want).astype(np.float32)) & far) <= 1


@pytest.mark.parametrize("lpc", [1, 2, 4, 8])
def test_msk_lane_layouts_are_bit_identical(D, O, S, lpc, tune):
    """1/2/4/8 lanes per channel only change the SIMT schedule: bits, state and blocks identical."""
    rng = np.random.default_rng(77)
    nch, n = 19, 6000                      # not a multiple of the channels-per-wave of any layout
    x = np.zeros((nch, n), dtype=np.float32)
    for c in range(nch):
        a, _ = S.channel_audio(rng, n, gap=(800, 2000), text_len=(5, 30))
        x[c] = S.envelope(a, carrier=0.3, noise=0.01, rng=rng)
    x[3] = rng.normal(0.2, 0.1, n)         # noise only
    tune("ACG_MSK_LPC", str(lpc))
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=8)
    dec.demod_msk(x[:, :2999])
    fr = dec.drain_frames()
    dec.demod_msk(x[:, 2999:])
    fr += dec.drain_frames()
    got = blocks_by_channel(fr, D.frame_tuple)
    for c in range(nch):
        ch = O.Channel(c, max_bits=3000)
        ch.demod(x[c, :2999])
        ch.demod(x[c, 2999:])
        assert got.get(c, []) == [O.frame_tuple(f) for f in ch.frames], (lpc, c)
        s, o = dec.state(c), ch.state()
        for k in ("MskS", "idx", "nbits", "Acarsstate", "outbits", "MskBitCount"):
            assert s[k] == o[k], (lpc, c, k)
        assert_state_close(s, o, "lane layout %s ch %d" % (lpc, c))
    dec.close()


def test_msk_two_wave_kernel_is_bit_identical_to_the_one_wave_kernel(D, O, S, tune):
    """msk2.hip splits the demodulator's per-bit instruction stream over a wave pair (mixer chain / clock + framing) that
    meet at three barriers per bit; the framing state machine's write-back into the loop (MskDf = 0, acars.c:242) crosses as a
    precomputed verdict.  Same operations in the same order: every soft bit, the loop state (doubles included) and every
    block must equal the one-wave kernel's EXACTLY -- on ragged call lengths (periods straddling calls: the per-sample tail),
    empty and tiny calls, noise-only channels (resets fire ~19 times a second), and the in_callback pipeline (vector refills,
    two pairs per workgroup on the CU-masked stream).  Blocks also against the oracle."""
    rng = np.random.default_rng(2024)
    nch, n = 19, 12000                      # not a multiple of the channels per wave pair
    x = np.zeros((nch, n), dtype=np.float32)
    for c in range(nch):
        a, _ = S.channel_audio(rng, n, gap=(800, 2000), text_len=(5, 40))
        x[c] = S.envelope(a, carrier=0.3, noise=0.02, rng=rng)
    x[3] = rng.normal(0.2, 0.1, n)          # noise only
    x[7] = 0.0                              # exact silence
    cuts = [0, 2999, 2999, 3000, 3005, 3006, 3013, 3077, 7173, 7173 + 4096, n]        # calls of 2999, 0, 1, 5, 1, 7, 64, 4096, 4096, rest

    def run_dm(split):
        tune("ACG_MSK_SPLIT", split, lab=True)             # (the two-wave kernel exists in the lab build only)
        dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=8, lab=True)
        fr, bits = [], [[] for _ in range(nch)]
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            if a1 > a0:
                dec.demod_msk(x[:, a0:a1])
                fr += dec.drain_frames()
                cnt, vo, lvl = dec.bits_all()
                for c in range(nch):
                    bits[c].append((vo[c, :cnt[c]].copy(), lvl[c, :cnt[c]].copy()))
        st = [dec.state(c) for c in range(nch)]
        dec.close()
        return blocks_by_channel(fr, D.frame_tuple), bits, st

    one, two = run_dm("0"), run_dm("1")
    assert one[0] == two[0]
    for c in range(nch):
        for (v1, l1), (v2, l2) in zip(one[1][c], two[1][c]):
            assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(l1.view(np.uint32), l2.view(np.uint32)), c
        for k, v in one[2][c].items():
            assert np.array_equal(np.asarray(v), np.asarray(two[2][c][k])), (c, k, v, two[2][c][k])
        ch = O.Channel(c)
        ch.demod(x[c])
        assert two[0].get(c, []) == [O.frame_tuple(f) for f in ch.frames], c
    assert sum(len(v) for v in two[0].values()) >= nch - 4

    # the in_callback pipeline: 2.5 Msps u8 input, 20 channels on 3 streams, calls of 1..3 callbacks
    M, nblk = 200, 6
    env = []
    for c in range(20):
        a, _ = S.channel_audio(rng, nblk * 1024, gap=(1500, 3000), text_len=(5, 40))
        env.append(0.5 * (1 + 0.5 * a))
    smap = np.arange(20) % 3
    off = [25000.0 * (2 + c // 3) * (-1) ** c for c in range(20)]
    rows = [S.iq_u8_from_envelopes(np.array([env[c] for c in range(20) if smap[c] == s_]), M, [off[c] for c in range(20) if smap[c] == s_],
                                   scale=0.12, noise=0.02, rng=rng) for s_ in range(3)]
    iq = np.stack(rows)
    taps = np.stack([D.rtl_taps(131000000 + int(off[c]), 131000000, M) for c in range(20)])
    row1 = 1024 * M * 2

    def run_iq(split):
        tune("ACG_MSK_SPLIT", split, lab=True)
        dec = D.Decoder(20, decim=M, nstreams=3, max_blocks=3, lab=True)
        dec.set_taps(taps)
        dec.set_channel_streams(smap)
        fr, b0 = [], 0
        for nb_ in (1, 3, 2):
            dec.in_callback(iq[:, b0 * row1:(b0 + nb_) * row1])
            fr += dec.drain_frames()
            b0 += nb_
        st = [dec.state(c) for c in range(20)]
        dec.close()
        return blocks_by_channel(fr, D.frame_tuple), st

    one, two = run_iq("0"), run_iq("1")
    assert one[0] == two[0] and sum(len(v) for v in two[0].values()) >= 10
    for c in range(20):
        for k, v in one[1][c].items():
            assert np.array_equal(np.asarray(v), np.asarray(two[1][c][k])), (c, k)


SINCOS_AB_CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from acarsdec_amd import decoder as D, synth as S, _capi as K
pcm = np.load(%(wav)r)["pcm"]
x = (pcm.astype(np.float32) / np.float32(32768.0)).T.copy()            # [4, n] the golden recording's four channels
rng = np.random.default_rng(5)
noise = rng.normal(0.2, 0.1, (2, x.shape[1])).astype(np.float32)        # + two noise-only channels (resets, razor-edge decisions)
x = np.concatenate([x, noise])
n = (x.shape[1] // 4096) * 4096
out = {}
for split in ("0", "1"):
    K.tune("ACG_MSK_SPLIT", split)
    dec = D.Decoder(x.shape[0], decim=8, ntaps=8, max_blocks=4)
    vo, lv, fr = [[] for _ in range(x.shape[0])], [[] for _ in range(x.shape[0])], []
    for s in range(0, n, 4096):
        dec.demod_msk(x[:, s:s + 4096])
        fr += [D.frame_tuple(f) for f in dec.drain_frames()]
        cnt, v, l = dec.bits_all()
        for c in range(x.shape[0]):
            vo[c].append(v[c, :cnt[c]].copy()); lv[c].append(l[c, :cnt[c]].copy())
    for c in range(x.shape[0]):
        out["vo%%s_%%d" %% (split, c)] = np.concatenate(vo[c]); out["lv%%s_%%d" %% (split, c)] = np.concatenate(lv[c])
        st = dec.state(c)
        out["st%%s_%%d" %% (split, c)] = np.array([st["MskPhi"], st["MskDf"], st["MskClk"], st["MskLvlSum"], st["MskS"], st["idx"]], dtype=np.float64)
    out["fr%%s" %% split] = np.array([repr(sorted(fr))])
    dec.close()
np.savez(sys.argv[1], **out)
print("OK")
'''


def test_table_sincos_build_and_polynomial_sincos_build_agree_bit_for_bit(tmp_path):
    """the product's mixer evaluates sin/cos as a 128-entry table + rotation (<= 2.1 ulp); the checking build
    (-DACG_MSK_SINCOS_POLY, lib/libacarsdec_amd_poly.so) as a < 1 ulp polynomial.  What the loop keeps are the float-rounded
    products, and those are claimed identical (tests/sincos_model.c on the CPU).  Here on the GPU: the golden recording's four
    channels plus two noise-only channels through BOTH builds, one-wave and two-wave kernels: every soft bit, every level, the
    loop state and every block identical."""
    from acarsdec_amd import _build
    if not os.path.exists(_build.LIB_POLY):
        pytest.skip("checking build not present (build() makes it)")
    res = {}
    # (both are lab builds, which carry the two-wave kernel; msk.hip does not see ACG_LAB: the one-wave kernel of the lab build IS
    #  the product's object file)
    for name, lib in (("table", _build.LIB_LAB), ("poly", _build.LIB_POLY)):
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", SINCOS_AB_CHILD % dict(root=ROOT, wav=os.path.join(ROOT, "tests", "golden", "testwav_pcm16.npz")), out],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, ACARSDEC_AMD_LIB=lib))
        assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
        res[name] = np.load(out)
    a, b = res["table"], res["poly"]
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 2 * (3 * 6 + 1)
    nbits = 0
    for k in a.files:
        if k.startswith("fr"):
            assert a[k][0] == b[k][0] and len(a[k][0]) > 100, k
        else:
            assert a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
            nbits += a[k].size if k.startswith("vo") else 0
    assert a["fr0"][0] == a["fr1"][0]
    assert nbits > 2 * 6 * 9000


@pytest.mark.parametrize("lpc", [8, 1])
def test_precise_mixer_flag_changes_nothing_the_loop_keeps(D, testwav, lpc, tune):
    """ACG_F_PRECISE_MIXER (the < 1 ulp sin/cos was a separate compile, not a run-time verification switch like
    ACG_F_EXACT_FIR): the same product library, the demodulator launched with the polynomial sin/cos instead of table + rotation.
    The golden recording's four channels plus two noise-only channels (resets ~19 times a second, razor-edge decisions) in ragged
    calls: every soft bit, every level, the loop state doubles and every block are identical with and without the flag."""
    tune("ACG_MSK_LPC", str(lpc))
    x = testwav.T.copy()
    rng = np.random.default_rng(5)
    x = np.concatenate([x, rng.normal(0.2, 0.1, (2, x.shape[1])).astype(np.float32)])
    cuts = list(range(0, x.shape[1], 4096)) + [x.shape[1]]
    res = []
    for precise in (False, True):
        dec = D.Decoder(x.shape[0], decim=8, ntaps=8, max_blocks=4, precise_mixer=precise)
        vo, lv, fr = [[] for _ in range(x.shape[0])], [[] for _ in range(x.shape[0])], []
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            dec.demod_msk(x[:, a0:a1])
            fr += [D.frame_tuple(f) for f in dec.drain_frames()]
            cnt, v, l = dec.bits_all()
            for c in range(x.shape[0]):
                vo[c].append(v[c, :cnt[c]].copy())
                lv[c].append(l[c, :cnt[c]].copy())
        st = [dec.state(c) for c in range(x.shape[0])]
        dec.close()
        res.append((sorted(fr), [np.concatenate(v_).tobytes() for v_ in vo], [np.concatenate(l_).tobytes() for l_ in lv],
                    [(s_["MskPhi"], s_["MskDf"], s_["MskClk"], s_["MskLvlSum"], s_["MskS"], s_["idx"], s_["inb"].tobytes()) for s_ in st]))
    assert res[0] == res[1]
    assert len(res[0][0]) == 7 and sum(len(b) for b in res[0][1]) > 6 * 9000 * 4


def test_streaming_collect_equals_blocking_drain(D, O, S):
    """acg_collect_frames(lag=1) (one call in flight) delivers exactly the blocks of acg_drain_frames,
    in the same per-channel order, over many calls on the rtl path with the stream pipeline on."""
    rng = np.random.default_rng(5150)
    M, nch, nblk, ncalls = 160, 6, 2, 5
    env = np.zeros((nch, ncalls * nblk * 1024))
    for c in range(nch):
        a, _ = S.channel_audio(rng, env.shape[1], gap=(1200, 2500), text_len=(5, 30))
        env[c] = 0.5 * (1 + 0.5 * a)
    off = [-325000, -300000, 75000, 150000, -50000, 25000]
    iq = np.stack([S.iq_u8_from_envelopes(env[c:c + 1], M, [off[c]], phases=[0.4 * c], noise=0.01, rng=rng) for c in range(nch)])
    taps = np.stack([D.rtl_taps(131850000 + off[c], 131850000, M) for c in range(nch)])
    row = nblk * 1024 * M * 2
    results = []
    for mode in ("drain", "collect"):
        dec = D.Decoder(nch, decim=M, max_blocks=nblk)
        dec.set_taps(taps)
        got = []
        for k in range(ncalls):
            dec.in_callback(np.ascontiguousarray(iq[:, k * row:(k + 1) * row]))
            if mode == "drain":
                got += [D.frame_tuple(f) for f in dec.drain_frames()]
            else:
                n, buf = dec.collect_frames_raw(lag=1)
                got += [D.frame_tuple(buf[i]) for i in range(n)]      # tuples copy the bytes out
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
        results.append(blocks_by_channel_tuples(got))
        dec.close()
    assert results[0] == results[1]
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_u8(iq[c], M, taps[c]))
        assert results[0].get(c, []) == [O.frame_tuple(f) for f in ch.frames]
    assert sum(len(v) for v in results[0].values()) >= nch


@pytest.mark.parametrize("nch,M", [(6, 160), (1200, 200)])
def test_host_fed_calls_reuse_their_buffer_and_equal_device_fed_calls(D, O, S, nch, M):
    """acg_process_iq_u8_host: two device staging buffers and a copy stream, the copy of call i+1 beside the
    kernels of call i.  The contract is librtlsdr's (rtl.c:314-330): the buffer is the caller's again when the call returns --
    so this host OVERWRITES its one buffer with garbage right after every call, from pageable memory, from acg_host_alloc memory
    and from its own registered memory, with calls of varying size (1, 2, 2, 1, 2 callbacks: the staging slots alternate and a
    short call follows a long one).  Blocks, every soft bit and the channel state equal the device-fed run; at 1200 channels the
    down-converter runs on its CU-masked stream (two more events in the ordering)."""
    import torch
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(31337 + nch)
    sizes = [1, 2, 2, 1, 2]
    nsrc = min(nch, 6)
    total = sum(sizes) * 1024
    env = np.zeros((nsrc, total))
    for c in range(nsrc):
        a, _ = S.channel_audio(rng, total, gap=(600, 1500), text_len=(3, 20))
        env[c] = 0.5 * (1 + 0.5 * a)
    off = [-325000, -300000, 75000, 150000, -50000, 25000]
    src = np.stack([S.iq_u8_from_envelopes(env[c:c + 1], M, [off[c]], phases=[0.4 * c], noise=0.01, rng=rng).reshape(-1) for c in range(nsrc)])
    iq = src[np.arange(nch) % nsrc]                                  # [nch, total * M * 2]
    taps = np.stack([D.rtl_taps(131850000 + off[c % nsrc], 131850000, M) for c in range(nch)])
    L = K.load()

    def run(mode):
        dec = D.Decoder(nch, decim=M, max_blocks=2, max_lag=1)
        dec.set_taps(taps)
        cap = 2 * 1024 * M * 2
        if mode == "alloc":
            p = L.acg_host_alloc(nch * cap)
            assert p
            buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(nch, cap))
        else:
            buf = np.empty((nch, cap), dtype=np.uint8)
            p = buf.ctypes.data
            if mode == "registered":
                assert L.acg_host_register(p, buf.nbytes) == K.OK
        blocks, bits, pos = [], [[] for _ in range(nsrc)], 0
        dev_in = torch.from_numpy(iq).cuda() if mode == "dev" else None
        for nb in sizes:
            rowb = nb * 1024 * M * 2
            if mode == "dev":
                dec.in_callback(dev_in[:, pos:pos + rowb], nblocks=nb, pitch=dev_in.stride(0))
            else:
                buf[:, :rowb] = iq[:, pos:pos + rowb]
                assert L.acg_process_iq_u8_host(dec.ctx, p, cap, nb) == K.OK
                buf[:] = 0xA5                                   # the buffer is ours again: scribble over it at once
            pos += rowb
            n, fb = dec.collect_frames_raw(lag=1, max_frames=8 * nch + 64)
            blocks += [D.frame_tuple(fb[i]) for i in range(n)]
            if nch <= 64:                                       # (reading the bit log waits for the call: only where it is cheap)
                for c in range(nsrc):
                    bits[c].append(np.concatenate(dec.bits(c)))
        blocks += [D.frame_tuple(f) for f in dec.drain_frames(8 * nch + 64)]
        st = [dec.state(c) for c in range(0, nch, max(1, nch // 7))]
        key = [(s_["MskPhi"], s_["MskDf"], s_["MskClk"], s_["MskS"], s_["idx"], s_["nbits"], s_["Acarsstate"], s_["inb"].tobytes()) for s_ in st]
        dec.close()
        if mode == "alloc":
            L.acg_host_free(p)
        elif mode == "registered":
            assert L.acg_host_unregister(p) == K.OK
        return sorted(blocks), [np.concatenate(b).tobytes() if b else b"" for b in bits], key
    want = run("dev")
    assert len(want[0]) >= nch
    for mode in ("pageable", "alloc", "registered"):
        got = run(mode)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2], mode
    for c in range(nsrc):                                            # and the oracle, end to end, on the originals
        ch = O.Channel(c)
        ch.demod(O.fir_u8(iq[c], M, taps[c]))
        assert [b for b in want[0] if b[0] == c] == sorted(O.frame_tuple(f) for f in ch.frames), c


@pytest.mark.parametrize("pipe", ["", "1", "3"])
def test_soak_mixed_call_sizes_device_input(D, O, S, pipe, tune):
    """a long stream cut into calls of 1..4 callbacks in random order, device input refilled in place between
    calls (the stream contract), lagged collection, two dm buffers, shared streams (3 dongles x 4/5/3
    channels), CU partition, every pipeline chunking: blocks bit-exact per channel against the oracle run
    over the uncut stream, and state doubles identical to a one-call-per-callback run."""
    import torch
    if pipe:
        tune("ACG_PIPE_BLOCKS", pipe)
    rng = np.random.default_rng(90210)
    M, maxb = 160, 4
    sizes = [int(x) for x in rng.integers(1, maxb + 1, size=14)]
    total = sum(sizes)
    groups = [4, 5, 3]
    smap = np.repeat(np.arange(len(groups)), groups)
    nch = int(smap.size)
    offs = {}
    iq_rows = []
    for s_, g in enumerate(groups):
        env = []
        for k in range(g):
            a, _ = S.channel_audio(rng, total * 1024, gap=(1500, 3000), text_len=(5, 40))
            env.append(0.5 * (1 + 0.5 * a))
        off = [25000.0 * (k + 1) * (-1) ** k for k in range(g)]
        offs[s_] = off
        iq_rows.append(S.iq_u8_from_envelopes(np.array(env), M, off, phases=list(np.linspace(0.1, 2.0, g)), scale=0.5 / g,
                                              noise=0.01, rng=rng))
    iq = np.stack(iq_rows)
    fc = 131000000
    taps = np.stack([D.rtl_taps(fc + int(offs[smap[c]][c - int(np.flatnonzero(smap == smap[c])[0])]), fc, M) for c in range(nch)])
    row1 = 1024 * M * 2

    def run(call_sizes):
        dec = D.Decoder(nch, decim=M, nstreams=len(groups), max_blocks=maxb)
        dec.set_taps(taps)
        dec.set_channel_streams(smap)
        dbuf = torch.empty((len(groups), maxb * row1), dtype=torch.uint8, device="cuda")
        st = torch.cuda.Stream()
        got, pos = [], 0
        with torch.cuda.stream(st):
            for nb in call_sizes:
                chunk = torch.from_numpy(np.ascontiguousarray(iq[:, pos * row1:(pos + nb) * row1]))
                dbuf[:, : nb * row1].copy_(chunk, non_blocking=False)      # refill in place, ordered on the caller's stream
                dec.in_callback(dbuf, nblocks=nb, pitch=maxb * row1, stream=st.cuda_stream)
                n, buf = dec.collect_frames_raw(lag=1)
                got += [D.frame_tuple(buf[i]) for i in range(n)]
                pos += nb
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
        states = [dec.state(c) for c in range(nch)]
        dec.close()
        return blocks_by_channel_tuples(got), states

    mixed, st_mixed = run(sizes)
    single, st_single = run([1] * total)
    assert mixed == single
    for a, b in zip(st_mixed, st_single):
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    nblocks = 0
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_u8(iq[smap[c]], M, taps[c]))
        want = [O.frame_tuple(f) for f in ch.frames]
        nblocks += len(want)
        assert mixed.get(c, []) == want, c
    assert nblocks >= nch


# ------------------------------------------------------------------------------------ other front ends' formats
@pytest.mark.parametrize("M,feeds", [(160, [1000, 163841 - 1000, 70000, 999999]), (200, [5, 204800, 3 * 204800 + 17, 10 ** 7]), (192, [10 ** 7]),
                                      (400, [123457, 10 ** 7])])
def test_cs16_soapy_path_with_carry_matches_oracle(D, O, S, M, feeds):
    """soapy.c shape: CS16 samples fed in reads of arbitrary size (windows straddle reads).  dm within the
    stated tolerance, blocks bit-exact, and the result independent of how the stream was cut."""
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(M)
    nch, nblk = 5, 4
    nout = nblk * 1024
    freqs = [131525000, 131725000, 131825000, 131550000, 131450000]
    fc = D.choose_fc(freqs, M)[0]
    env = []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, gap=(400, 900), text_len=(3, 25))
        env.append(0.5 * (1 + 0.5 * a))
    iq = S.iq_s16_from_envelopes(np.array(env), M, [f - fc for f in freqs], phases=np.linspace(0, 3, nch), scale=0.15,
                                 noise=0.01, rng=rng)
    taps = np.stack([D.soapy_taps(float(f), fc, M) for f in freqs])
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    got, dms, pos, total = [], [[] for _ in range(nch)], 0, iq.size // 2
    for n in feeds:
        n = min(n, total - pos)
        if n <= 0:
            break
        dec.feed(K.FMT_CS16, iq[2 * pos:2 * (pos + n)].reshape(1, -1))
        pos += n
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
    assert pos == total
    by = blocks_by_channel_tuples(got)
    nfr = 0
    for c in range(nch):
        dm_o = O.fir_cs16(iq, M, O.soapy_taps(float(freqs[c]), fc, M))
        ch = O.Channel(c)
        ch.demod(dm_o)
        assert by.get(c, []) == [O.frame_tuple(f) for f in ch.frames], c
        nfr += len(ch.frames)
    assert nfr >= nch - 2
    dec.close()
    # window-aligned device path: same blocks, dm within tolerance
    import torch
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    d_iq = torch.from_numpy(iq.copy()).cuda()
    dec.process_samples(K.FMT_CS16, d_iq, nblk, pitch=iq.size * 2)
    got2 = blocks_by_channel(dec.drain_frames(), D.frame_tuple)
    for c in range(nch):
        dm_o = O.fir_cs16(iq, M, O.soapy_taps(float(freqs[c]), fc, M))
        dm_g = dec.dm(c, nout)
        assert np.all(np.abs(dm_g - dm_o) <= 1e-5 * np.abs(dm_o) + 1e-6), (c, np.abs(dm_g - dm_o).max())
        assert got2.get(c, []) == by.get(c, [])
    dec.close()


def test_split16_sdrplay_path_matches_oracle(D, O, S):
    """sdrplay.c (two int16 planes, |D|/4): arbitrary callback sizes
    with carry, then the window-aligned device path; dm within tolerance, blocks bit-exact."""
    import torch
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(2718)
    nch, nblk = 4, 4
    nout = nblk * 1024
    freqs = [131525000, 131725000, 131825000, 131550000]
    env = []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, gap=(400, 900), text_len=(3, 25))
        env.append(0.5 * (1 + 0.5 * a))
    env = np.array(env)
    # ---- sdrplay
    M = 160
    fc = D.choose_fc(freqs, M)[0]
    iq = S.iq_s16_from_envelopes(env, M, [f - fc for f in freqs], phases=np.linspace(0, 3, nch), scale=0.15, noise=0.01,
                                 rng=rng, full_scale=0.06)
    xi, xq = iq[0::2].copy(), iq[1::2].copy()
    taps = np.stack([D.sdrplay_taps(float(f), fc) for f in freqs])
    want = {}
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_split16(xi, xq, M, O.sdrplay_taps(float(freqs[c]), fc)))
        want[c] = [O.frame_tuple(f) for f in ch.frames]
    assert sum(len(v) for v in want.values()) >= nch - 1
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    got, pos = [], 0
    for n in (504, 100000, 3, 10 ** 7):
        n = min(n, xi.size - pos)
        dec.feed(K.FMT_S16_SPLIT, xi[pos:pos + n].reshape(1, -1), xq[pos:pos + n].reshape(1, -1))
        pos += n
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
    assert blocks_by_channel_tuples(got) == {c: v for c, v in want.items() if v}
    dec.close()
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    planes = torch.from_numpy(np.concatenate([xi, xq])).cuda()
    dec.process_samples(K.FMT_S16_SPLIT, planes, nblk, pitch=planes.numel() * 2, plane=xi.size * 2)
    assert blocks_by_channel(dec.drain_frames(), D.frame_tuple) == {c: v for c, v in want.items() if v}
    for c in range(nch):
        dm_o = O.fir_split16(xi, xq, M, O.sdrplay_taps(float(freqs[c]), fc))
        dm_g = dec.dm(c, nout)
        assert np.all(np.abs(dm_g - dm_o) <= 1e-5 * np.abs(dm_o) + 1e-6 * np.abs(dm_o).max())
    dec.close()


@pytest.mark.parametrize("rate", [2500000, 6000000, 10000000])
def test_f32_airspy_path_matches_oracle(D, O, S, rate):
    """air.c (real float32 around Fs/4) at the rates Airspy devices offer (R2: 10 and 2.5 Msps, Mini:
    6 Msps -> windows of 800 / 200 / 480 samples; the long ones pass through LDS in column slices):
    arbitrary callback sizes with carry, then the window-aligned device path."""
    import torch
    from acarsdec_amd import _capi as K
    rng = np.random.default_rng(2718 + rate // 100000)
    nch, nblk = 4, 4
    nout = nblk * 1024
    freqs = [131525000, 131725000, 131825000, 131550000]
    env = []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nout, gap=(400, 900), text_len=(3, 25))
        env.append(0.5 * (1 + 0.5 * a))
    env = np.array(env)
    M = rate // 12500
    fc = D.airspy_choose_fc(freqs)
    assert fc == O.air_choose_fc(freqs)
    x = S.real_f32_from_envelopes(env, M, [fc - f + rate / 4 for f in freqs], phases=np.linspace(0, 3, nch), scale=0.15,
                                  noise=0.01, rng=rng)
    taps = np.stack([D.airspy_taps(f, fc, rate) for f in freqs])
    want = {}
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_f32r(x, M, O.air_taps(freqs[c], fc, rate)))
        want[c] = [O.frame_tuple(f) for f in ch.frames]
    assert sum(len(v) for v in want.values()) >= nch - 1
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    got, pos = [], 0
    for n in (65536, 1000, 300001, 10 ** 7):
        n = min(n, x.size - pos)
        dec.feed(K.FMT_F32_REAL, x[pos:pos + n].reshape(1, -1))
        pos += n
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
    assert blocks_by_channel_tuples(got) == {c: v for c, v in want.items() if v}
    dec.close()
    dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=nblk)
    dec.set_taps(taps)
    dx = torch.from_numpy(x).cuda()
    dec.process_samples(K.FMT_F32_REAL, dx, nblk, pitch=x.size * 4)
    assert blocks_by_channel(dec.drain_frames(), D.frame_tuple) == {c: v for c, v in want.items() if v}
    for c in range(nch):
        dm_o = O.fir_f32r(x, M, O.air_taps(freqs[c], fc, rate))
        dm_g = dec.dm(c, nout)
        assert np.all(np.abs(dm_g - dm_o) <= 1e-5 * np.abs(dm_o) + 1e-6)
    dec.close()


# ------------------------------------------------------------------------------------ block repair on the device
def test_device_block_repair_matches_oracle_and_golden(D, O, S, testwav, golden):
    """ACG_F_REPAIR: what drain returns equals what the reference's blk_thread hands to outputmsg():
    repaired text, parity stripped, err = parity errors found, dropped blocks omitted."""
    # (1) the reference's own data: the 7 processed blocks of test.wav
    dec = D.Decoder(4, decim=8, ntaps=8, max_blocks=53, repair=True)
    x = np.zeros((4, 53 * 1024), dtype=np.float32)
    x[:, :testwav.shape[0]] = testwav.T
    dec.demod_msk(x)
    got = sorted(D.frame_tuple(f) for f in dec.drain_frames())
    want = sorted(t for t, _ in golden_blocks(golden["file"]["out_blocks"]))
    assert got == want and len(got) == 7
    dec.close()
    # (2) transmissions with injected bit errors: every repair path and every drop path
    rng = np.random.default_rng(31337)
    kinds = [None, "p1", "p2", "p3", "db", "crc", "p4", "p1crc"]
    nch, n = 16, 48 * 1024
    x = np.zeros((nch, n), dtype=np.float32)
    for c in range(nch):
        a, _ = S.channel_audio(rng, n, gap=(1200, 2500), text_len=(15, 50), corrupt=kinds[c % 8:] + kinds[:c % 8])
        x[c] = S.envelope(a, noise=0.002, rng=rng)
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=48, repair=True)
    got = []
    for k in range(0, n, 16 * 1024):                      # several calls: the repair runs per call
        dec.demod_msk(x[:, k:k + 16 * 1024])
        got += [D.frame_tuple(f) for f in dec.drain_frames()]
    dec.close()
    want, nraw, nfixed = [], 0, 0
    for c in range(nch):
        ch = O.Channel(c, max_frames=512)
        ch.demod(x[c])
        for f in ch.frames:
            nraw += 1
            o = O.blk_process(f)
            if o is not None:
                want.append(O.frame_tuple(o))
                nfixed += o.err > 0
    assert sorted(got) == sorted(want)
    assert nraw > len(want) and nfixed >= 10 and len(want) >= 40, (nraw, len(want), nfixed)


# ------------------------------------------------------------------------------------ the legacy call surface
def test_compat_program_output_is_golden(golden, tmp_path):
    """The reference's UNCHANGED acarsdec.c/acars.c/output.c linked against compat_msk.c
    (initMsk/demodMSK on the GPU): `acarsdec -o 1 -f test.wav` prints the golden text."""
    exe = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu")
    if not os.path.exists(exe):
        pytest.skip("demo binary not built (needs the reference tree at build time)")
    import hashlib
    import struct
    z = np.load(os.path.join(ROOT, "tests", "golden", "testwav_pcm16.npz"))
    pcm = z["pcm"].astype("<i2")
    wav = tmp_path / "t.wav"
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, pcm.shape[1], 12500, 12500 * 2 * pcm.shape[1], 2 * pcm.shape[1], 16) + b"data" + struct.pack("<I", len(data))
    wav.write_bytes(hdr + data)
    r = subprocess.run([exe, "-o", "1", "-f", str(wav)], capture_output=True, timeout=300)
    assert hashlib.md5(r.stdout).hexdigest() == golden["program"]["o1"]["md5"], r.stdout.decode("latin-1") + r.stderr.decode("latin-1")
    assert r.stdout.decode("latin-1") == golden["program"]["o1"]["stdout"]
    # JSON output (output.c:227-324) minus wall-clock time and host name
    import json
    r = subprocess.run([exe, "-o", "4", "-f", str(wav)], capture_output=True, timeout=300)
    recs = []
    for line in r.stdout.decode("latin-1").splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            d.pop("timestamp", None)
            d.pop("station_id", None)
            recs.append(d)
    assert recs == golden["program"]["o4"]["records"]


def test_compat_rtl_program_output_is_golden(golden, testwav, S, tmp_path):
    """The rtl.c path end to end: the reference's UNCHANGED acarsdec.c + rtl.c + acars.c + output.c, compat_msk.c
    instead of msk.c, and a file-playing dongle that hands every buffer to acarsdec_amd_in_callback()
    (= the one-line change at rtl.c:364): `acarsdec -o 1 -r 0 f1 f2 f3 f4` prints what the CPU reference prints."""
    import hashlib
    import re
    exe = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_rtl")
    if not os.path.exists(exe):
        pytest.skip("demo binary not built (needs the reference tree at build time)")
    g, pr = golden["rtl"], golden["program_rtl"]
    fr = [int(round(float(f) * 1e6)) for f in g["freqs"]]
    env = S.pad_blocks(0.5 + 0.5 * testwav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * pr["tail_blocks"]), 0.5)], axis=1)
    iq = S.iq_u8_from_envelopes(env, g["M"], [f - g["Fc"] for f in fr], phases=g["phases"])
    if hashlib.sha256(iq.tobytes()).hexdigest() != pr["iq_sha256"]:
        pytest.skip("numpy produced different synthetic IQ bytes than when the fixture was made")
    path = tmp_path / "t.iq"
    path.write_bytes(iq.tobytes())
    r = subprocess.run([exe] + pr["args"], env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)), capture_output=True, timeout=300)
    out = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1"))
    assert out == pr["stdout_no_timestamps"], out + r.stderr.decode("latin-1")[-800:]
    assert out.count("\n") == 7


def test_compat_soapy_program_output_equals_the_cpu_program(testwav, S, O, tmp_path):
    """The SoapySDR path end to end (VERDICT r03 "missing" 7): the reference's UNCHANGED acarsdec.c + acars.c + output.c ... and
    soapy.c with the one hunk of INTEGRATION.md applied at build time (its per-channel loop, soapy.c:228-254, replaced by a call of
    acarsdec_amd_soapy_samples() from compat_msk.c), fed by a file-playing SoapySDR stand-in that hands out reads of RAGGED size
    (the window carry of soapy.c:232-254 is exercised at every read).  `acarsdec -o 1 -m 160 -d file f1 f2 f3 f4` must print what
    the CPU twin (the same program with the reference's own soapy.c and msk.c, oracle/_ref/acarsdec_cpu_soapy) prints on the
    same CS16 file: the same messages with the same levels and error counts, per channel in the same order."""
    import re
    gpu = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_soapy")
    cpu = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_soapy")
    if not (os.path.exists(gpu) and os.path.exists(cpu)):
        pytest.skip("demo binaries not built (they need the reference tree at build time)")
    freqs, M = ["131.525", "131.725", "131.825", "131.550"], 160
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    fc = 131850000                                                   # soapy.c's chooseFc for these four (tests/test_oracle_vs_ref.py pins it against _ref)
    env = S.pad_blocks(0.5 + 0.5 * testwav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * 3), 0.5)], axis=1)    # (the CPU loop only demodulates whole 1024-output buffers, soapy.c:245)
    iq = S.iq_s16_from_envelopes(env, M, [f - fc for f in fr], phases=[0.3, 1.1, 2.2, 0.7])
    path = tmp_path / "t.cs16"
    path.write_bytes(iq.tobytes())
    outs = []
    for exe in (cpu, gpu):
        r = subprocess.run([exe, "-o", "1", "-m", str(M), "-d", "file"] + freqs, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)),
                           capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode("latin-1")[-800:]
        lines = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1")).splitlines()
        per = {}
        for l in lines:
            per.setdefault(l.split()[0], []).append(l)
        outs.append(per)
    assert outs[0] == outs[1] and sum(len(v) for v in outs[0].values()) == 7 and set(outs[0]) == {"#1", "#2", "#3", "#4"}


@pytest.mark.parametrize("fe", ["air", "sdrplay"])
def test_compat_callback_front_end_programs_equal_their_cpu_twins(fe, testwav, S, O, tmp_path):
    """The Airspy and SDRplay paths end to end: the reference's UNCHANGED acarsdec.c + air.c / sdrplay.c + acars.c + output.c ...,
    compat_msk.c instead of msk.c, and a file-playing vendor-library stand-in with ragged transfers that hands every transfer to
    acarsdec_amd_air_samples() / acarsdec_amd_sdrplay_samples() instead of the front end's own callback (= the one-line change
    inside rx_callback, air.c:291 / myStreamCallback, sdrplay.c:201).  The program must print what its CPU twin (the reference's own
    callback and msk.c behind the same stand-in, oracle/_ref/acarsdec_cpu_<fe>) prints on the same sample file: the same messages
    with the same levels and error counts, per channel in the same order."""
    import re
    gpu = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_" + fe)
    cpu = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_" + fe)
    if not (os.path.exists(gpu) and os.path.exists(cpu)):
        pytest.skip("demo binaries not built (they need the reference tree at build time)")
    freqs = ["131.525", "131.725", "131.825", "131.550"]
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    env = S.pad_blocks(0.5 + 0.5 * testwav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * 3), 0.5)], axis=1)
    if fe == "air":
        rate = 2500000
        fc = O.air_choose_fc(fr)                                              # air.c:62
        data = S.real_f32_from_envelopes(env, rate // 12500, [fc - f + rate / 4 for f in fr], phases=[0.3, 1.1, 2.2, 0.7], scale=0.15)
        args = ["-o", "1", "-s", "0"] + freqs
    else:
        fc = 131850000                                                        # sdrplay.c's choice for these four (pinned by the CPU test of the twin)
        data = S.iq_s16_from_envelopes(env, 160, [f - fc for f in fr], phases=[0.3, 1.1, 2.2, 0.7], full_scale=0.06)
        args = ["-o", "1", "-s"] + freqs
    path = tmp_path / ("t." + fe)
    path.write_bytes(data.tobytes())
    outs = []
    for exe in (cpu, gpu):
        r = subprocess.run([exe] + args, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)), capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode("latin-1")[-800:]
        lines = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1")).splitlines()
        per = {}
        for l in lines:
            if l.startswith("#"):
                per.setdefault(l.split()[0], []).append(l)
        outs.append(per)
    assert outs[0] == outs[1] and sum(len(v) for v in outs[0].values()) == 7 and set(outs[0]) == {"#1", "#2", "#3", "#4"}


def test_replay_sink_matches_device_blocks(D, O, testwav):
    """acg_replay_bits hands every bit to a putbit()-shaped sink; feeding an oracle FSM from it
    yields the same blocks the device assembled."""
    from acarsdec_amd import _capi as K
    x = np.ascontiguousarray(testwav[:16384].T)
    dec = D.Decoder(4, decim=8, ntaps=8, max_blocks=16)
    dec.demod_msk(x)
    seen = [[] for _ in range(4)]

    def sink(user, ch, vo, lvl):
        seen[ch].append((vo, lvl))
    cb = K.BIT_SINK(sink)
    assert dec.L.acg_replay_bits(dec.ctx, cb, None) == 0
    for c in range(4):
        vo, lvl = dec.bits(c)
        assert np.array_equal(np.array([v for v, _ in seen[c]], dtype=np.float32), vo)
        assert np.array_equal(np.array([l for _, l in seen[c]], dtype=np.float32), lvl)
    dec.close()
