"""Package power / clocks WHILE a kernel runs (a sampler thread reads sysfs every 20 ms; the main thread sits in the launch loop):
a pure streaming reader, and the down-converter alone -- default kernel, round 1's workgroup kernel with and without its arithmetic.
    python profiles/probe/power_probe.py"""
import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from benchlib.telemetry import gpu_telemetry
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
bytes_ = nch * nblk * 1024 * (2 * M + 4)

def run(tag, fn, seconds=2.5):
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            samples.append(gpu_telemetry(0)); time.sleep(0.02)
    fn(); torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            fn(); n += 1
        torch.cuda.current_stream().synchronize() if False else None
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    s = samples[len(samples) // 2:] or samples
    sclk = np.median([x["sclk"] for x in s if x["sclk"]]); pw = np.median([x["power_w"] for x in s if x["power_w"]])
    print("%-58s %6.0f GB/s = %.3f of 8 TB/s   shader %4.0f MHz  package %4.0f W  (fclk %s, mclk %s; %d samples)" % (
        tag, n * bytes_ / e0.elapsed_time(e1) / 1e6, n * bytes_ / e0.elapsed_time(e1) / 8e9, sclk, pw, s[-1]["fclk"], s[-1]["mclk"], len(s)), flush=True)

sink = torch.zeros(1, dtype=torch.int32, device="cuda")
# (the launcher acg_launch_read_probe is local to the library since round 5's version script: the exported lab entry point
#  acg_probe_read_dev runs the same kernel, one repeat per call -- ADVICE r05)
LI = K.load()
LI.acg_probe_read_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
_gbs = C.c_double(0)
run("pure streaming reader (read_probe_kernel)", lambda: LI.acg_probe_read_dev(iq.data_ptr(), iq.numel(), 1, C.byref(_gbs)))
for tag, var, extra in (("down-converter alone, default kernel (variant 5)", "5", None), ("..., register taps (variant 7)", "7", None),
                        ("..., default kernel without its arithmetic (variant 56: loads consumed, no cvt / FMA / tap reads)", "56", None),
                        ("..., round 1's workgroup kernel (variant 3)", "3", None), ("..., variant 3 without its arithmetic (loads + LDS staging only)", "3", "ACG_FIR_DEBUG_NOCOMPUTE")):
    K.tune("ACG_FIR_VARIANT", var)
    if extra: K.tune(extra, "1")
    run(tag, lambda: dec.fir_only(iq, nblk, row, stream=st.cuda_stream))
    if extra: K.tune(extra, None)
