"""Where a bit period of the two-wave demodulator (msk2.hip) goes: per wave role, cycles of work between the barriers and
cycles waited AT each barrier (measurement build, -DACG_MSK_STAMP):
    python -c "from acarsdec_amd import _build; _build.build_lib(stamp=True)";  python profiles/probe/msk2_stamps.py [channels] [blocks]"""
import ctypes as C
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from acarsdec_amd import _capi as K, _build
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K.LIB_PATH = _build.LIB_STAMP
L = K.load()
L.acg_msk_stamp_read.restype = C.c_int
L.acg_msk_stamp_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
from acarsdec_amd import decoder as D, synth as S
K.tune("ACG_MSK_SPLIT", 1)
nout = nblk * 1024
rng = np.random.default_rng(5)
pool = np.stack([S.envelope(S.channel_audio(np.random.default_rng(100 + i), nout, gap=(1500, 5000), text_len=(20, 160))[0],
                            noise=0.01, rng=rng) for i in range(64)])
dm = torch.from_numpy(pool[np.arange(nch) % 64].copy()).cuda()
dec = D.Decoder(nch, decim=8, ntaps=8, max_blocks=nblk, bitlog=True)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
for rep in range(3):
    assert L.acg_process_dm_dev(dec.ctx, dm.data_ptr(), nout, nout, st.cuda_stream) == 0
dec.sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert L.acg_process_dm_dev(dec.ctx, dm.data_ptr(), nout, nout, st.cuda_stream) == 0
dec.sync()
e1.record()
torch.cuda.synchronize()
nw = 2 * ((nch + 7) // 8)
buf = np.zeros((nw, 10), dtype=np.uint64)
assert L.acg_msk_stamp_read(dec.ctx, buf.ctypes.data, nw) == 0
names = ["B1->B2 work", "wait at B2", "B2->B3 work", "wait at B3", "B3->B1 work", "wait at B1"]
print("msk2 stamps: %d channels, %d waves, launch %.3f ms (instrumented), periods per wave %.0f" % (nch, nw, e0.elapsed_time(e1), buf[:, 8].mean()))
for role, nm in ((0, "wave M"), (1, "wave H")):
    b = buf[role::2].astype(np.float64)
    per = b[:, :6] / np.maximum(b[:, 8:9], 1)
    print("  %s: %s  | sum %.0f cycles per period" % (nm, "  ".join("%s %.0f" % (names[k], per[:, k].mean()) for k in range(6)), per.sum(axis=1).mean()))
