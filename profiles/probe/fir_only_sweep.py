"""Stand-alone timing of the down-converter launch (no demodulator running beside it):
python profiles/probe/fir_only_sweep.py [channels] [decim] [blocks]   -- env knobs: ACG_FIR_VARIANT, ACG_FIR_WG_PER_CU"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L = K.load()
row = nblk * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
dec = D.Decoder(nch, decim=M, max_blocks=nblk, bitlog=False)
dec.set_taps(np.stack([D.rtl_taps(131000000 + 25000 * (1 + c % 40), 131000000, M) for c in range(nch)]))
st = torch.cuda.Stream()            # a real stream: handle 0 means "the context's own stream" to the library
torch.cuda.set_stream(st)
s = st.cuda_stream
for _ in range(3):
    dec.fir_only(iq, nblk, row, stream=s)
torch.cuda.synchronize()
best = 1e9
tot = 0.0
R = 10
for _ in range(R):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec.fir_only(iq, nblk, row, stream=s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    best = min(best, ms)
    tot += ms
bytes_ = nch * nblk * 1024 * (2 * M + 4) + nch * M * 8
print("fir_only nch=%d M=%d blk=%d variant=%s wg/cu=%s: best %.4f ms = %.0f GB/s, mean %.4f ms = %.0f GB/s" % (
    nch, M, nblk, os.environ.get("ACG_FIR_VARIANT", "3"), os.environ.get("ACG_FIR_WG_PER_CU", "-"),
    best, bytes_ / best / 1e6, tot / R, bytes_ / (tot / R) / 1e6))
