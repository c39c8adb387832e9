"""Stand-alone timing of the demodulator launch on ACARS traffic (nothing running beside it):
python profiles/probe/msk_only.py [channels] [blocks] [ACG_MSK_SPLIT values, e.g. 0,1,0,1]"""
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
pool = []
for i in range(32):
    a, _ = S.channel_audio(rng, n, gap=(1500, 5000), text_len=(20, 160))
    pool.append(S.envelope(a, noise=0.02, rng=rng).astype(np.float32))
dm = np.stack([pool[c % 32] for c in range(nch)])
d = torch.from_numpy(dm).cuda()
dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=nblk, bitlog=True, timing=True)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
for _ in range(3):
    assert L.acg_process_dm_dev(dec.ctx, d.data_ptr(), n, n, st.cuda_stream) == 0
    dec.drain_frames_raw(65536)
dec.timing()
for split in ([None] if len(sys.argv) <= 3 else sys.argv[3].split(",")):
  if split is not None:
    K.tune("ACG_MSK_SPLIT", split)              # 0: one wave per channel group (msk.hip), 1: wave pairs (msk2.hip)
    for _ in range(2):
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
  bits = n / 5.2083
  print("split %s  " % split, end="")
  print("msk_only nch=%d blk=%d lpc=%s: kernel %.4f ms per call (%.3f us/bit/wave), wall %.4f ms, %d blocks per call" % (
    nch, nblk, os.environ.get("ACG_MSK_LPC", "auto"), tim["msk_ms"] / R, tim["msk_ms"] / R * 1e3 / bits, dt * 1e3, nf))
