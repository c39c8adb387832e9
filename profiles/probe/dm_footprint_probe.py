"""Does the down-converter's write stream still cost its reads when the dm it writes is SMALL enough to stay in the
Infinity Cache and is rewritten in place?  The same input, the same kernel, launches of `cb` callbacks on decoders whose dm
buffers hold exactly `cb` callbacks (two buffers, alternating per launch: footprint 2 x nch x cb x 4 KiB), several decoders
per size (placements differ), each also with the dm rows folded into one (ACG_FIR_DEBUG_DMPITCH0: no write stream at all):
    python profiles/probe/dm_footprint_probe.py [nch] [ndec] [cb ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ndec = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cbs = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
M, ntaps = 200, 200
L = K.load()
maxcb = max(cbs)
row = maxcb * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
torch.cuda.synchronize()
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M) for c in range(40)])


def measure(dec, cb, reps=12):
    """`reps` launches of cb callbacks back to back (walking through the input), one event pair around all of them"""
    per = cb * 1024 * M * 2
    nslice = maxcb // cb
    for k in range(3):
        dec.fir_only(iq[:, (k % nslice) * per:], cb, row, stream=st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        dec.fir_only(iq[:, (k % nslice) * per:], cb, row, stream=st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    bytes_ = nch * cb * 1024 * (2 * M + 4) + nch * ntaps * 8
    return bytes_ * reps / (e0.elapsed_time(e1) * 1e-3) / 8e12


pads = []
for cb in cbs:
    for k in range(ndec):
        dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=cb, bitlog=False)
        dec.set_taps(base[np.arange(nch) % 40])
        a = [measure(dec, cb) for _ in range(2)]
        K.tune("ACG_FIR_DEBUG_DMPITCH0", "1")
        b = measure(dec, cb)
        K.tune("ACG_FIR_DEBUG_DMPITCH0", None)
        print("nch %d  %d callback(s) per launch, dm footprint 2 x %.0f MB, decoder %d: %.3f %.3f of 8 TB/s;  rows folded (no write stream): %.3f"
              % (nch, cb, nch * cb * 4096 / 1e6, k, a[0], a[1], b), flush=True)
        pads.append(torch.empty(((k + 1) * 37 * 4096 + 12345,), dtype=torch.uint8, device="cuda"))
        dec.close()
