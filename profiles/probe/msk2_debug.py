"""where does the two-wave demodulator (ACG_MSK_SPLIT=1) first differ from the one-wave kernel?  per call: soft bits, state"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from acarsdec_amd import decoder as D, synth as S, _capi as K
rng = np.random.default_rng(2024)
nch, n = 19, 12000
x = np.zeros((nch, n), dtype=np.float32)
for c in range(nch):
    a, _ = S.channel_audio(rng, n, gap=(800, 2000), text_len=(5, 40))
    x[c] = S.envelope(a, carrier=0.3, noise=0.02, rng=rng)
x[3] = rng.normal(0.2, 0.1, n)
x[7] = 0.0
cuts = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2999, 2999, 3000, 3005, 3006, 3013, 3077, 7173, 7173 + 4096, n]
def run(split):
    K.tune("ACG_MSK_SPLIT", split)
    dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=8)
    out = []
    for a0, a1 in zip(cuts[:-1], cuts[1:]):
        if a1 > a0:
            dec.demod_msk(x[:, a0:a1])
            fr = dec.drain_frames()
            cnt, vo, lvl = dec.bits_all()
            out.append((a0, a1, [(vo[c, :cnt[c]].copy(), lvl[c, :cnt[c]].copy()) for c in range(nch)], [dec.state(c) for c in range(nch)], len(fr)))
    dec.close()
    return out
one, two = run("0"), run("1")
for (a0, a1, b1, s1, f1), (_, _, b2, s2, f2) in zip(one, two):
    bad = []
    for c in range(nch):
        v1, l1 = b1[c]; v2, l2 = b2[c]
        why = None
        if len(v1) != len(v2):
            why = "nbits %d vs %d" % (len(v1), len(v2))
        else:
            d = np.flatnonzero((v1.view(np.uint32) != v2.view(np.uint32)) | (l1.view(np.uint32) != l2.view(np.uint32)))
            if d.size:
                k = d[0]
                why = "bit %d of %d: vo %r/%r lvl %r/%r" % (k, len(v1), v1[k], v2[k], l1[k], l2[k])
        sd = [k for k in s1[c] if not np.array_equal(np.asarray(s1[c][k]), np.asarray(s2[c][k]))]
        if why or sd:
            bad.append((c, why, [(k, s1[c][k], s2[c][k]) for k in sd if k != "inb"][:6]))
    print("call [%d, %d): frames %d/%d  differing channels %d" % (a0, a1, f1, f2, len(bad)))
    for b in bad[:4]:
        print("   ", b)
