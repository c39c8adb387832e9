"""Does the down-converter's rate depend on WHERE the decoder's own buffers lie?  One input buffer, several decoders kept
alive (so every one has its own taps / dm / dispenser allocation; ACG_DEBUG_ADDR prints the addresses), each timed alone:
    ACG_DEBUG_ADDR=1 python profiles/probe/placement_probe.py [variant] [ndec]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once

variant = sys.argv[1] if len(sys.argv) > 1 else "5"
ndec = int(sys.argv[2]) if len(sys.argv) > 2 else 6
_K.tune("ACG_FIR_VARIANT", variant)
L = K.load()
nch, M, nblk, ntaps = 16384, 200, 4, 200
row = nblk * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
torch.cuda.synchronize()
print("iq at 0x%x (%.1f GB)" % (iq.data_ptr(), iq.numel() / 1e9), flush=True)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M)[:ntaps] for c in range(40)])
bytes_ = nch * nblk * 1024 * (2 * M + 4) + nch * ntaps * 8
decs = []
pads = []


def measure(dec):
    for _ in range(2):
        dec.fir_only(iq, nblk, row, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.fir_only(iq, nblk, row, stream=st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return bytes_ / (np.array(ts) * 1e-3) / 1e9


for k in range(ndec):
    sys.stderr.flush()
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
    dec.set_taps(base[np.arange(nch) % 40])
    decs.append(dec)
    g = measure(dec)
    print("decoder %d: %s GB/s  (%.3f of 8 TB/s)" % (k, " ".join("%.0f" % x for x in g), g.mean() / 8000), flush=True)
    # an odd-sized spacer so that the next decoder's buffers land elsewhere
    pads.append(torch.empty(((k + 1) * 37 * 4096 + 12345,), dtype=torch.uint8, device="cuda"))
print("second pass over the same decoders (same placements):")
for k, dec in enumerate(decs):
    g = measure(dec)
    _K.tune("ACG_FIR_DEBUG_DMPITCH0", "1")
    g0 = measure(dec)
    _K.tune("ACG_FIR_DEBUG_DMPITCH0", None)
    other = {}
    for v in ("5", "55", "7", "8"):
        _K.tune("ACG_FIR_VARIANT", v)
        other[v] = measure(dec).mean() / 8000
    _K.tune("ACG_FIR_VARIANT", variant)
    print("decoder %d: %s GB/s  (%.3f of 8 TB/s);  all dm rows folded into the first (no write stream): %.3f;  variants 5 / 55 / 7 / 8: %.3f %.3f %.3f %.3f" % (
        k, " ".join("%.0f" % x for x in g), g.mean() / 8000, g0.mean() / 8000, other["5"], other["55"], other["7"], other["8"]), flush=True)
for dec in decs:
    dec.close()
