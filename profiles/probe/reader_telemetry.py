"""clock / power telemetry while (a) a pure streaming reader, (b) the down-converter alone (variants 5, 7, 3) run for ~2 s each"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, ctypes as C
import bench
from acarsdec_amd import decoder as D, _capi as K
L = K.load()
nch, M, nblk = 16384, 200, 4
row = nblk * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
torch.cuda.synchronize()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
dec = D.Decoder(nch, decim=M, max_blocks=nblk, bitlog=False)
base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M) for c in range(40)])
dec.set_taps(base[np.arange(nch) % 40])
def sample(tag, fn, seconds=2.0):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; tele = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn(); n += 1
        tele.append(bench.gpu_telemetry(0))
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-34s %7.0f GB/s  telemetry first/last %s / %s" % (tag, n * nch * nblk * 1024 * (2 * M + 4) / ms / 1e6, tele[1] if len(tele) > 1 else tele[0], tele[-1]), flush=True)
g = C.c_double(0)
def reader():
    assert L.acg_probe_read_dev(iq.data_ptr(), iq.numel(), 1, C.byref(g)) == 0
sample("pure reader (acg_probe_read_dev)", reader)
for v in ("5", "7", "3", "6"):
    K.tune("ACG_FIR_VARIANT", v)
    sample("down-converter alone, variant " + v, lambda: dec.fir_only(iq, nblk, row, stream=st.cuda_stream))
