"""What seconds of load do to the two read ceilings (same process, same buffer, back to back):
  (a) the pure streaming reader (acg_probe_read_dev), called repeatedly for ~4 s: GB/s per call
  (b) the down-converter alone (16 384 channels x 4 callbacks per launch), launched back to back for ~4 s: GB/s per 10 launches
  (c) the pure reader again, right after (b)
    python profiles/probe/sustained_probe.py [variant]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once

variant = sys.argv[1] if len(sys.argv) > 1 else "5"
_K.tune("ACG_FIR_VARIANT", variant)
L = K.load()
nch, M, nblk, ntaps = 16384, 200, 4, 200
row = nblk * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
torch.cuda.synchronize()
time.sleep(2.0)                      # start from an idle device


def reader(seconds, label):
    g = C.c_double(0)
    out = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        assert L.acg_probe_read_dev(iq.data_ptr(), min(iq.numel(), 1 << 34), 10, C.byref(g)) == 0
        out.append(g.value)
    print("%s: %d calls of 10 x %.1f GB; GB/s first 5: %s ... last 5: %s; mean of the last half %.0f" % (
        label, len(out), min(iq.numel(), 1 << 34) / 1e9, " ".join("%.0f" % x for x in out[:5]), " ".join("%.0f" % x for x in out[-5:]),
        float(np.mean(out[len(out) // 2:]))), flush=True)


reader(4.0, "(a) pure reader from idle")
time.sleep(2.0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M)[:ntaps] for c in range(40)])
dec.set_taps(base[np.arange(nch) % 40])
bytes_ = nch * nblk * 1024 * (2 * M + 4) + nch * ntaps * 8
out = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 4.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dec.fir_only(iq, nblk, row, stream=st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    out.append(bytes_ * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
print("(b) down-converter (variant %s) alone from idle, per 10 launches: first 5: %s ... last 5: %s; mean of the last half %.0f GB/s = %.3f of 8 TB/s" % (
    variant, " ".join("%.0f" % x for x in out[:5]), " ".join("%.0f" % x for x in out[-5:]), float(np.mean(out[len(out) // 2:])),
    float(np.mean(out[len(out) // 2:])) / 8000.0), flush=True)
reader(2.0, "(c) pure reader right after (b)")
dec.close()
