import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from acarsdec_amd import decoder as D, _capi as K
nch, M, nblk = 16384, 200, 8
L = K.load()
row = nblk * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
dec = D.Decoder(nch, decim=M, max_blocks=nblk, bitlog=True, timing=True)
dec.set_taps(np.stack([D.rtl_taps(131000000 + 25000 * (1 + c % 40), 131000000, M) for c in range(nch)]))
dec.set_timing(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
log = []
for i in range(8):
    a = time.perf_counter()
    dec.in_callback(iq, nblocks=nblk, pitch=row, stream=0)
    b = time.perf_counter()
    n, _ = dec.collect_frames_raw(1, 8 * nch)
    c = time.perf_counter()
    log.append((i, (a - t0) * 1e3, (b - a) * 1e3, (c - b) * 1e3, n))
torch.cuda.synchronize()
for r in log:
    print("step %d: enqueue at %.2f ms took %.3f ms; collect took %.3f ms (%d blocks)" % r)
print("total %.2f ms" % ((time.perf_counter() - t0) * 1e3))
