"""Debug aid: which outputs of the wave-private down-converter differ from the oracle, by (channel, tile)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from acarsdec_amd import decoder as D
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once
from oracle import oracle as O
for M, nch, nblk in ((200, 5, 2), (160, 5, 2), (200, 40, 8)):
    rng = np.random.default_rng(M)
    nout = nblk * 1024
    iq = rng.integers(0, 256, size=(nch, nout * M * 2), dtype=np.uint8)
    taps = np.stack([O.rtl_taps(131000000 + 25000 * (c + 1), 131000000, M) for c in range(nch)])
    for variant in ("5", "3"):
        _K.tune("ACG_FIR_VARIANT", variant)
        dec = D.Decoder(nch, decim=M, max_blocks=nblk, bitlog=False)
        dec.set_taps(taps)
        dec.in_callback(iq)
        bad = np.zeros((nch, nout // 64), dtype=int)
        for c in range(nch):
            want = O.fir_u8(iq[c], M, taps[c], nout=nout)
            got = dec.dm(c, nout)
            e = np.abs(got - want) > 1e-5 * np.abs(want) + 1e-6
            bad[c] = e.reshape(-1, 64).sum(axis=1)
        print("M=%d nch=%d nblk=%d variant=%s: bad outputs %d of %d" % (M, nch, nblk, variant, bad.sum(), nch * nout))
        if bad.sum():
            for c in range(min(nch, 6)):
                print("  ch %d bad-per-tile: %s" % (c, " ".join("%d" % x for x in bad[c][:64])))
            w = np.argwhere(bad)[:1]
            c, t = w[0]
            want = O.fir_u8(iq[c], M, taps[c], nout=nout)
            got = dec.dm(c, nout)
            print("  first bad tile: ch %d tile %d\n   got  %s\n   want %s" % (c, t, got[t * 64:t * 64 + 8], want[t * 64:t * 64 + 8]))
        dec.close()
