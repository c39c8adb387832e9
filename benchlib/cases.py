"""The workloads bench.py times (BASELINE.json configs and what SURVEY 8d / the verdicts added), the constants of the synthetic
signal, and the tap tables of a case."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
COPY_CEILING_GBS = 6290.0


# ------------------------------------------------------------------------------------------ workloads
# Every case is sized to 54-74 GB of input per GPU (one buffer serves them all); 20 steps of the headline case make >= 0.5 s
# (176 callbacks = 14.4 s of signal per channel per step).
CASES = {
    # BASELINE.json configs[2]: 1 GPU, 1024 channels, synthetic 2.5 Msps IQ, FIR decimate + MSK demod throughput
    "throughput": dict(tag="BASELINE configs[2]", channels=1024, decim=200, ntaps=200, blocks=176, content="acars"),
    # north-star regime: >= 10 000 concurrent channels at 2.5 Msps on one GPU
    "wide": dict(tag="north star (>= 10 000 channels per GPU)", channels=16384, decim=200, ntaps=200, blocks=8, content="acars", check_channels=256),
    # BASELINE.json configs[4]: 1 GPU stress, 192-tap LPF FIR, 2.5 Msps, 4096 channels
    "stress": dict(tag="BASELINE configs[4]", channels=4096, decim=200, ntaps=192, blocks=32, content="random+acars"),
    # the soapy.c front end's sample format (interleaved int16 I/Q) through the same pipeline
    "cs16": dict(tag="soapy.c CS16 front end (SURVEY 8f.2)", channels=4096, decim=200, ntaps=200, blocks=16, content="format+acars", format="cs16"),
    # ... and the air.c front end's (real float32 samples against complex taps)
    "f32": dict(tag="air.c real-f32 front end (SURVEY 8f.2)", channels=4096, decim=200, ntaps=200, blocks=16, content="format+acars", format="f32"),
    # BASELINE.json configs[3] per-GPU share: 16384 channels over 8 GPUs
    "shard2048": dict(tag="BASELINE configs[3], per-GPU share", channels=2048, decim=200, ntaps=200, blocks=64, content="acars"),
    # north_star: "from 2.0/2.5 Msps"; rtlMult 160 is the reference's default (acarsdec.c:57), 192 its other documented rate (rtl.c:35-37):
    # fir_u8_direct_kernel<20> / <24>, one stream per channel
    "m160": dict(tag="rtlMult 160 = 2.0 Msps, the reference's default (acarsdec.c:57)", channels=4096, decim=160, ntaps=160, blocks=16, content="random+acars"),
    "m192": dict(tag="rtlMult 192 = 2.4 Msps (rtl.c:35-37)", channels=4096, decim=192, ntaps=192, blocks=16, content="random+acars"),
    # the sdrplay.c front end's sample format (an int16 I plane and an int16 Q plane, sdrplay.c:215-236)
    "split16": dict(tag="sdrplay.c split int16 planes (SURVEY 8f.2)", channels=4096, decim=160, ntaps=160, blocks=16, content="format+acars", format="split16"),
    # rtl.c's OWN shape (rtl.c:344-354): one dongle stream feeds K channels -- 2048 dongles x 8 channels; the contraction runs on
    # the matrix pipe (fir_mm.hip).  Never the HBM-roofline figure of the one-stream-per-channel path.
    "share8": dict(tag="rtl.c's own shape: 2048 dongle streams x 8 channels each (rtl.c:344-354)", channels=16384, decim=200, ntaps=200, blocks=16,
                   content="shared+acars", share=8),
}
# The cases that saturate HBM hold the chip at its power cap and their rate sinks with the shader clock for tens of seconds
# (profiles/r05_soak.txt): their `value` is the >= 20 s figure (--sustain-hbm), the first 5 s are reported beside it (burst5s).
SUSTAIN_HBM = ("wide", "stress", "cs16", "f32")
# The reference's own interface: the input arrives in HOST memory that is only valid during the call (rtl.c:314-330,
# soapy.c:220-254).  10 000 channels x 2.5 Msps is 50 GB/s -- PCIe Gen5 territory: this case says whether the ">= 10 000
# concurrent channels" of the north star holds for inputs that come from a host (run_hostfed below; not a run_case shape).
HOSTFED = dict(tag="north star fed from pinned host memory (rtl.c:314-330 semantics)", channels=10000, decim=200, ntaps=200, call_blocks=2)
SNR_DB = 20.0               # AWGN at 20 dB, measured in the 12.5 kHz channel
CARRIER, DEPTH, SCALE = 0.5, 0.5, 0.25


class Job:
    """Process-wide state: device, library, collectives, the one input buffer."""
    pass


def make_taps(D, fmt_name, offs, M, ntaps):
    import numpy as np
    fc = 131000000
    nch = len(offs)
    taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
    # ntaps < M: a low-pass window over the NCO (Hamming, unit DC gain); the oracle for it is the same
    # sum(vb*wf) with these taps (SURVEY 8d config 5 -- the reference itself only has the boxcar)
    win = np.ones(ntaps) if ntaps == M else np.hamming(ntaps) / np.hamming(ntaps).mean() * (M / ntaps)
    cache = {}
    for c in range(nch):
        o = int(offs[c])
        if o not in cache:
            # the front end's own tap builder (rtl.c:283-286 / soapy.c:163-166 / air.c:278-285 / sdrplay.c:160-165)
            base = (D.rtl_taps(fc + o, fc, M) if fmt_name == "u8" else
                    D.airspy_taps(fc - o, fc, M * 12500) if fmt_name == "f32" else
                    D.sdrplay_taps(fc + o, fc) if (fmt_name == "split16" and M == 160) else D.soapy_taps(fc + o, fc, M))
            cache[o] = (base[:ntaps] * win[:, None]).astype(np.float32)
        taps[c] = cache[o]
    return taps
