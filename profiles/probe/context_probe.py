"""Several contexts alive in ONE process, the same pipelined calls timed on each in turn, a marker kernel between the
turns -- the question of VERDICT r03 item 3: why do contexts 0 / 3 of four run 7-16 % slower than 1 / 2 at 4096 channels?

  python profiles/probe/context_probe.py [channels] [ntaps] [contexts] [seconds per turn] [rounds]

Plain: prints ms per call and the event-timed down-converter / demodulator launches per context and round.
Under `rocprofv3 --kernel-trace`: profiles/probe/context_trace_summary.py splits the dispatch trace at the markers
(fill_random_kernel launches: turn k is preceded by k + 1 markers... see below) and reports, per turn, the HSA queue of
each kernel, kernel durations and how much of the demodulator's busy time the down-converter overlaps.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from acarsdec_amd import decoder as D, _capi as K

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ntaps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 4
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
M, cb = 200, 8
L = K.load()
row = cb * 1024 * M * 2
iq = torch.empty((nch, row), dtype=torch.uint8, device="cuda")
assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 0xACA25, None) == 0
mark = torch.empty(4096, dtype=torch.uint8, device="cuda")
taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
base = D.rtl_taps(131050000, 131000000, M)[:ntaps]
taps[:] = base
torch.cuda.synchronize()
decs = []
for k in range(nctx):
    d = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=cb, bitlog=True, timing=True, repair=True, max_lag=1)
    d.set_taps(taps)
    decs.append(d)
maxm = nch * 4 + 8192


def call(d):
    d.in_callback(iq, nblocks=cb, pitch=row, stream=None)
    while d.collect_msgs_raw(1, maxm)[2]:
        pass


def marker(n):
    for _ in range(n):
        assert L.acg_fill_random_u8_dev(mark.data_ptr(), 4096, 1, 4096, 1, None) == 0
    torch.cuda.synchronize()


for d in decs:                      # a round for nothing: clocks, first touch
    for _ in range(3):
        call(d)
    d.drain_msgs_raw(maxm)
    d.timing()
torch.cuda.synchronize()
print("context_probe: %d channels, %d taps, %d contexts, calls of %d callbacks" % (nch, ntaps, nctx, cb))
for r in range(rounds):
    res = []
    for k, d in enumerate(decs):
        marker(1)                   # one marker before every turn: the summariser counts turns
        d.set_timing(1)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < secs:
            call(d)
            n += 1
        while d.drain_msgs_raw(maxm)[2]:
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = d.timing()
        res.append(dt / n * 1e3)
        print("round %d context %d: %.4f ms per call (%d calls)  fir %.4f ms x %d launches, msk %.4f ms x %d launches per call" % (
            r, k, dt / n * 1e3, n, t["fir_ms"] / max(1, t["fir_launches"]), t["fir_launches"] // n,
            t["msk_ms"] / max(1, t["msk_launches"]), t["msk_launches"] // n))
    print("round %d: ms per call %s  spread %.1f %%" % (r, [round(x, 3) for x in res], (max(res) / min(res) - 1) * 100))
marker(2)
for d in decs:
    d.close()
