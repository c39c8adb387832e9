"""dm of a down-converter variant against the oracle and against the default kernel, on one small launch:
    python profiles/probe/fir_variant_check.py VARIANT [channels] [blocks] [ntaps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, synth as S
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once
from oracle import oracle as O

variant = sys.argv[1] if len(sys.argv) > 1 else "6"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 96
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ntaps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
M = 200
rng = np.random.default_rng(5)
row = nblk * 1024 * M * 2
iq = rng.integers(0, 256, size=(nch, row), dtype=np.uint8)
iq[1, :] = 0
iq[2, :] = 255
# a carrier on some channels so that dm spans weak and strong levels
t = np.arange(row // 2)
for c in range(3, nch, 3):
    ph = 2 * np.pi * (25000.0 * (1 + c % 7)) / 2.5e6 * t
    a = 20 + 10 * (c % 9)
    iq[c, 0::2] = np.clip(127.4 + a * np.cos(ph) + rng.normal(0, 3, t.size), 0, 255).astype(np.uint8)
    iq[c, 1::2] = np.clip(127.4 + a * np.sin(ph) + rng.normal(0, 3, t.size), 0, 255).astype(np.uint8)
taps = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c % 7), 131000000, M)[:ntaps] for c in range(nch)]).astype(np.float32)
d = torch.from_numpy(iq).cuda()
out = {}
for v in ("5", variant):
    _K.tune("ACG_FIR_VARIANT", v)
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
    dec.set_taps(taps)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        dec.fir_only(d, nblk, row, stream=st.cuda_stream)
    torch.cuda.synchronize()
    out[v] = np.stack([dec.dm(c, nblk * 1024) for c in range(nch)])
    dec.close()
worst = 0.0
for c in list(range(0, min(nch, 24))) + [nch - 1]:
    want = O.fir_u8(iq[c], M, taps[c], ntaps=ntaps)
    for v in out:
        err = np.abs(out[v][c] - want)
        tol = 1e-5 * np.abs(want) + 1e-6
        r = float(np.max(err / tol))
        worst = max(worst, r) if v == variant else worst
        if c < 6 or r > 1:
            print("ch %3d variant %s: max |err| %.3e, max err/tol %.3f, dm range %.4g..%.4g" % (c, v, err.max(), r, want.min(), want.max()))
print("variant %s vs default: max |diff| %.3e; worst err/tol against the oracle %.3f -> %s" % (
    variant, np.abs(out[variant] - out["5"]).max(), worst, "OK" if worst <= 1 else "FAIL"))
