"""Why does the bench's 176-callback headline lose one block of channel 27 (oracle: a 66-character block ending at bit 31543)?
Rebuilds the bench's input for that channel, decodes it alone on the GPU with the bit log on, and compares soft symbols with the oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, synth as S, _capi as K
from oracle import oracle as O

CH, NCH, M, NBLK = 27, 1024, 200, 176
nout = NBLK * 1024
r0 = np.random.default_rng(0xACA25)
off = r0.integers(-48, 49, size=NCH) * 25000.0
off[np.abs(off) < 25000] = 50000.0
ph = r0.uniform(0, 2 * np.pi, NCH)
L = K.load()
SNR_DB, CARRIER, DEPTH, SCALE = 20.0, 0.5, 0.5, 0.25
sigma = SCALE * CARRIER * (M / (2.0 * 10 ** (SNR_DB / 10.0))) ** 0.5
row = nout * M * 2
# the bench synthesises all 1024 rows in one launch (noise is seeded per launch and row): do the same for rows 0..CH
n = CH + 1
iq = torch.empty((n, row), dtype=torch.uint8, device="cuda")
trk = torch.empty((n, nout), dtype=torch.float32, device="cuda")
for c in range(n):
    a, _ = S.channel_audio(np.random.default_rng(0xACA25 + c), nout, gap=(3125, 12500), text_len=(20, 220))
    trk[c] = torch.from_numpy((CARRIER * (1.0 + DEPTH * a)).astype(np.float32)).cuda()
d_idx = torch.arange(n, dtype=torch.int32, device="cuda")
d_off = torch.from_numpy(off[:n].astype(np.float32)).cuda()
d_ph = torch.from_numpy(ph[:n].astype(np.float32)).cuda()
assert L.acg_synth_iq_u8_dev(iq.data_ptr(), row, n, nout, M, trk.data_ptr(), nout, d_idx.data_ptr(), d_off.data_ptr(), d_ph.data_ptr(),
                             SCALE, sigma, 0xACA25, None) == 0
torch.cuda.synchronize()
host = iq[CH].cpu().numpy()
fc = 131000000
taps = D.rtl_taps(fc + int(off[CH]), fc, M).astype(np.float32)
dm = O.fir_u8(host, M, taps)
och = O.Channel(0, max_bits=40000)
och.demod(dm)
ovo, olvl = och.bits
print("oracle: %d blocks, end bits %s" % (len(och.frames), [int(f.end_bit) for f in och.frames]))
dec = D.Decoder(1, decim=M, max_blocks=8, bitlog=True)
dec.set_taps(taps[None])
gvo = []
frames = []
one = iq[CH:CH + 1]
for k in range(NBLK // 8):
    dec.in_callback(one[:, k * 8 * 1024 * M * 2:(k + 1) * 8 * 1024 * M * 2], nblocks=8, pitch=row)
    frames += dec.drain_frames()
    v, l = dec.bits(0)
    gvo.append(v)
gvo = np.concatenate(gvo)
print("gpu: %d blocks, end bits %s" % (len(frames), [int(f.end_bit) for f in frames]))
m = min(len(gvo), len(ovo))
hard = (gvo[:m] > 0) != (ovo[:m] > 0)
idx = np.nonzero(hard)[0]
print("bits compared %d; hard decisions that differ: %d; first ten: %s" % (m, len(idx), idx[:10]))
for i in idx[:5]:
    print("  bit %d: gpu vo %.6g  oracle vo %.6g   (neighbours gpu %s | oracle %s)" % (i, gvo[i], ovo[i], np.round(gvo[i - 2:i + 3], 4), np.round(ovo[i - 2:i + 3], 4)))
d = np.abs(gvo[:m] - ovo[:m])
print("max |dvo| over bits before the first difference: %.3g" % (d[:idx[0]].max() if len(idx) else d.max()))
dec.close()
