"""Cycle breakdown of the demodulator's per-bit serial chain: runs the MEASUREMENT build
(lib/libacarsdec_amd_stamp.so, -DACG_MSK_STAMP: s_memtime at the phase boundaries of msk_demod_kernel's loop)
on synthetic ACARS traffic and prints, per phase, the mean shader cycles per loop pass (= per bit period).
    python profiles/probe/msk_phase_stamps.py [channels] [blocks]
Build the library first:  python -c "from acarsdec_amd import _build; _build.build_lib(stamp=True)"
The stamps themselves cost ~10 % (MI355X_MICROARCH.md); compare with profiles/probe/msk_only.py for the
un-instrumented time per bit."""
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
assert os.path.exists(_build.LIB_STAMP), "build the measurement library first (see the docstring)"
K.LIB_PATH = _build.LIB_STAMP            # the binding loads whatever LIB_PATH names
L = K.load()
L.acg_msk_stamp_read.restype = C.c_int
L.acg_msk_stamp_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.acg_msk_lanes_per_channel.restype = C.c_int
L.acg_msk_lanes_per_channel.argtypes = [C.c_void_p]
from acarsdec_amd import decoder as D, synth as S

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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
assert L.acg_process_dm_dev(dec.ctx, dm.data_ptr(), nout, nout, st.cuda_stream) == 0
dec.sync()
e1.record()
torch.cuda.synchronize()
lpc = L.acg_msk_lanes_per_channel(dec.ctx)
nwaves = (nch * lpc + 63) // 64
buf = np.zeros((nwaves, 10), dtype=np.uint64)
assert L.acg_msk_stamp_read(dec.ctx, buf.ctypes.data, nwaves) == 0
names = ["upkeep+loop", "A vco/clock", "B sincos+mix+ring", "C1 tap phase+filter", "C2 |v|+normalise", "C3 decision+FSM", "C4 loop filter", "-"]
it = buf[:, 8].astype(np.float64)
bits = buf[:, 9].astype(np.float64)
print("msk phase stamps: %d channels, %d lanes per channel, %d waves, %d samples per channel; launch %.3f ms (instrumented)"
      % (nch, lpc, nwaves, nout, e0.elapsed_time(e1)))
print("loop passes per wave: mean %.0f; passes with a bit decision: mean %.0f" % (it.mean(), bits.mean()))
tot = 0.0
for k in range(7):
    per = buf[:, k].astype(np.float64) / np.maximum(it if k < 3 else bits, 1)
    tot += per.mean()
    print("  %-22s %8.1f cycles per pass  (min wave %.1f, max wave %.1f)" % (names[k], per.mean(), per.min(), per.max()))
print("  %-22s %8.1f cycles per bit period (stamps included)" % ("sum", tot))
