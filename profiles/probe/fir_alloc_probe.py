"""Does the down-converter's rate depend on WHERE its input buffer lies?  One process, one launch shape, the input
re-allocated several times behind paddings of different sizes (so that the virtual / physical placement changes);
also the same buffer timed again after each re-allocation of the decoder (dm, taps).
    python profiles/probe/fir_alloc_probe.py [channels] [blocks] [rounds]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ALT = sys.argv[4] if len(sys.argv) > 4 else "54"      # the kernel variant timed beside the default on every placement
M = ntaps = 200
L = K.load()
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
row = nblk * 1024 * M * 2
base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M)[:ntaps] for c in range(40)])
bytes_ = nch * nblk * 1024 * (2 * M + 4) + nch * ntaps * 8


def time_fir(dec, iq, reps=6):
    for _ in range(2):
        dec.fir_only(iq, nblk, row, stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.fir_only(iq, nblk, row, stream=s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return bytes_ / best / 8e9


pads = []
for r in range(rounds):
    pads.append(torch.empty(((3 + 5 * r) << 20) + 4096 * r, dtype=torch.uint8, device="cuda"))   # shifts the next allocation
    iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
    assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
    torch.cuda.synchronize()
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
    dec.set_taps(base[np.arange(nch) % 40])
    f1 = time_fir(dec, iq)
    _K.tune("ACG_FIR_VARIANT", ALT)
    f1n = time_fir(dec, iq)
    _K.tune("ACG_FIR_VARIANT", None)
    dec.close()
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
    dec.set_taps(base[np.arange(nch) % 40])
    f2 = time_fir(dec, iq)
    _K.tune("ACG_FIR_VARIANT", ALT)
    f2n = time_fir(dec, iq)
    _K.tune("ACG_FIR_VARIANT", None)
    g = C.c_double(0)
    L.acg_probe_read_dev(iq.data_ptr(), min(iq.numel(), 1 << 34), 3, C.byref(g))
    print("round %d: iq at 0x%x  fir %.3f of 8 TB/s (variant %s: %.3f); same iq, new decoder: %.3f (%.3f); pure reader %.0f GB/s" % (
        r, iq.data_ptr(), f1, ALT, f1n, f2, f2n, g.value), flush=True)
    dec.close()
    del iq
    torch.cuda.empty_cache()
