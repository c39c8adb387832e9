"""Same-process A/B of the two demodulator kernels on launches without a bit log: msk_lean.hip (framing off the per-bit path,
the default) against msk.hip (ACG_MSK_NOLEAN=1), nothing running beside them:
python profiles/probe/msk_lean_ab.py [channels] [blocks per call] [traffic: acars | noise | mixed] [bit log: 0 | 1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, synth as S, _capi as K

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
traffic = sys.argv[3] if len(sys.argv) > 3 else "acars"
bitlog = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
L = K.load()
rng = np.random.default_rng(7)
n = nblk * 1024
pool = []
for i in range(32):
    if traffic == "noise" or (traffic == "mixed" and i % 2):
        pool.append(rng.normal(0.5, 0.2, size=n).astype(np.float32))
    else:
        a, _ = S.channel_audio(rng, n, gap=(1500, 5000), text_len=(20, 160))
        pool.append(S.envelope(a, noise=0.02, rng=rng).astype(np.float32))
dm = np.stack([pool[c % 32] for c in range(nch)])
d = torch.from_numpy(dm).cuda()
dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=nblk, bitlog=bitlog, timing=True)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
bits = n / 5.2083
for rnd in range(2):
    for name, val in (("lean", None), ("inline", "1")):
        K.tune("ACG_MSK_NOLEAN", val)
        for _ in range(3):
            assert L.acg_process_dm_dev(dec.ctx, d.data_ptr(), n, n, st.cuda_stream) == 0
            dec.drain_frames_raw(65536)
        dec.timing()
        R = 10
        t0 = time.perf_counter()
        for _ in range(R):
            assert L.acg_process_dm_dev(dec.ctx, d.data_ptr(), n, n, st.cuda_stream) == 0
            nf = dec.drain_frames_raw(65536)[0]
        dt = (time.perf_counter() - t0) / R
        tim = dec.timing()
        print("%-6s nch=%d blk=%d %s log=%d: kernel %.4f ms per call (%.3f us/bit/wave), wall %.4f ms, %d blocks per call" % (
            name, nch, nblk, traffic, bitlog, tim["msk_ms"] / R, tim["msk_ms"] / R * 1e3 / bits, dt * 1e3, nf))
