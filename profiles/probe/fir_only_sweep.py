"""Stand-alone timing of the down-converter launch (no demodulator running beside it), several kernels and
shapes in ONE process (ACG_FIR_VARIANT / ACG_FIR_WG_PER_CU are read at every launch):
    python profiles/probe/fir_only_sweep.py [spec ...]     spec = channels:decim:blocks:ntaps:variant[:wg_per_cu]
Default specs compare the workgroup-granular kernel (3) with the wave-private streaming kernel (5) on the bench
shapes.  Also prints what a pure streaming reader gets on the same buffer (acg_probe_read_dev)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K
from acarsdec_amd import _capi as _K   # switches go through acg_tune: the library reads the environment once

specs = sys.argv[1:] or ["1024:200:8:200:3", "1024:200:8:200:5", "1024:200:128:200:5", "16384:200:8:200:3", "16384:200:8:200:5",
                         "4096:200:32:192:3", "4096:200:32:192:5", "1024:160:8:160:3", "1024:160:8:160:5", "1024:192:8:192:5",
                         "1024:200:8:200:5:1"]
L = K.load()
cache = {}
st = torch.cuda.Stream()            # a real stream: handle 0 means "the context's own stream" to the library
torch.cuda.set_stream(st)
s = st.cuda_stream
for spec in specs:
    f = spec.split(":")
    nch, M, nblk, ntaps, variant = int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4]
    wg = f[5] if len(f) > 5 else None
    row = nblk * 1024 * M * 2
    key = (nch, row)
    if key not in cache:
        cache.clear()
        torch.cuda.empty_cache()
        iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 1234, None) == 0
        torch.cuda.synchronize()
        g = C.c_double(0)
        assert L.acg_probe_read_dev(iq.data_ptr(), min(iq.numel(), 1 << 34), 3, C.byref(g)) == 0
        print("pure reader on %.2f GB: %.0f GB/s" % (iq.numel() / 1e9, g.value), flush=True)
        cache[key] = iq
    iq = cache[key]
    _K.tune("ACG_FIR_VARIANT", variant)
    if wg:
        _K.tune("ACG_FIR_WG_PER_CU", wg)
    else:
        _K.tune("ACG_FIR_WG_PER_CU", None)
    dec = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=nblk, bitlog=False)
    base = np.stack([D.rtl_taps(131000000 + 25000 * (1 + c), 131000000, M)[:ntaps] for c in range(40)])
    dec.set_taps(base[np.arange(nch) % 40])
    for _ in range(3):
        dec.fir_only(iq, nblk, row, stream=s)
    torch.cuda.synchronize()
    best, tot, R = 1e9, 0.0, 10
    for _ in range(R):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.fir_only(iq, nblk, row, stream=s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = min(best, ms)
        tot += ms
    bytes_ = nch * nblk * 1024 * (2 * M + 4) + nch * ntaps * 8
    print("fir_only nch=%d M=%d blk=%d ntaps=%d variant=%s wg/cu=%s: best %.4f ms = %.0f GB/s (%.3f of 8 TB/s), mean %.4f ms = %.0f GB/s (%.3f)" % (
        nch, M, nblk, ntaps, variant, wg or "-", best, bytes_ / best / 1e6, bytes_ / best / 8e9, tot / R, bytes_ / (tot / R) / 1e6,
        bytes_ / (tot / R) / 8e9), flush=True)
    dec.close()
