"""demodulator alone, with and without the per-bit log compiled in (ACG_F_BITLOG): python profiles/probe/msk_bitlog_ab.py [channels] [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, synth as S, _capi as K
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = K.load()
rng = np.random.default_rng(7)
n = nblk * 1024
pool = [S.envelope(S.channel_audio(rng, n, gap=(1500, 5000), text_len=(20, 160))[0], noise=0.02, rng=rng).astype(np.float32) for i in range(32)]
d = torch.from_numpy(np.stack([pool[c % 32] for c in range(nch)])).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
decs = {b: D.Decoder(nch, decim=8, ntaps=8, max_blocks=nblk, bitlog=b, timing=True) for b in (True, False)}
for rnd in range(2):
    for b, dec in decs.items():
        for _ in range(3):
            assert L.acg_process_dm_dev(dec.ctx, d.data_ptr(), n, n, st.cuda_stream) == 0
            dec.drain_frames_raw(65536)
        dec.timing()
        R = 10
        for _ in range(R):
            assert L.acg_process_dm_dev(dec.ctx, d.data_ptr(), n, n, st.cuda_stream) == 0
            nf = dec.drain_frames_raw(65536)[0]
        tim = dec.timing()
        print("bitlog %-5s kernel %.4f ms per call (%.4f us/bit), %d blocks per call" % (b, tim["msk_ms"] / R, tim["msk_ms"] / R * 1e3 / (n / 5.2083), nf), flush=True)
