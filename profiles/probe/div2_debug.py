import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from acarsdec_amd import _capi as K
L = K.load()
rng = np.random.default_rng(2024)
n = 1 << 22
lvl = (10.0 ** rng.uniform(-12, 2, n)).astype(np.float32)
d = lvl.astype(np.float64) + 1e-8
ang = rng.uniform(0, 2 * np.pi, n)
vr = (lvl * np.cos(ang)).astype(np.float32).astype(np.float64)
vi = (lvl * np.sin(ang)).astype(np.float32).astype(np.float64)
out = np.zeros((n, 4), dtype=np.float64)
assert L.acg_selftest_div2(vr.ctypes.data, vi.ctypes.data, d.ctypes.data, out.ctypes.data, n) == K.OK
for a, b, num in ((0, 2, vr), (1, 3, vi)):
    bad = out[:, a].view(np.uint64) != out[:, b].view(np.uint64)
    print("mismatches:", int(bad.sum()), "of", n, " device-IEEE vs numpy:", int((out[:, b] != num / d).sum()))
    idx = np.flatnonzero(bad)[:8]
    for i in idx:
        print("  n=%r d=%r shared=%r ieee=%r (%d ulp)" % (num[i], d[i], out[i, a], out[i, b],
              int(out[i, a].view(np.int64)) - int(out[i, b].view(np.int64))))
