"""bench.py's cpu_baseline leg: the unmodified reference (oracle/_ref) or the C restatement timed on the host cores."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline_child(variant, M, blocks_per_call, seconds):
    """Times the UNMODIFIED reference's in_callback (rtl.c:314-361 incl. demodMSK/decodeAcars),
    one channel per stream, on one host core.  Runs in a child process: the reference is all
    global state, and an -march=native build may not run on this host."""
    import numpy as np
    from oracle import oracle as O
    from acarsdec_amd import synth as S
    ref = O.Ref(variant)
    ref.init_rtl(["131.725"], M)
    rng = np.random.default_rng(1)
    a, _ = S.channel_audio(rng, blocks_per_call * 1024)
    iq = S.iq_u8_from_envelopes(0.5 * (1 + 0.5 * a)[None, :], M, [-25000.0], noise=0.01, rng=rng)
    blk = 1024 * M * 2
    bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]) for b in range(blocks_per_call)]
    for b in bufs:                      # warm-up
        ref.in_callback(b)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for b in bufs:
            ref.in_callback(b)
        n += len(bufs)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(value=n * 1024 * M / dt / 1e6, blocks=n, seconds=dt)))


def run_cpu_baseline(M, seconds=12.0):
    me = BENCH
    for variant, label in (("_fast", "-Ofast -march=native"), ("_v3", "-Ofast -march=x86-64-v3"), ("", "-O2")):
        so = os.path.join(ROOT, "oracle", "_ref", "libacarsref%s.so" % variant)
        if not os.path.exists(so):
            continue
        # 96 distinct callbacks = 39 MB per pass at rtlMult 200: the input streams from memory, as it does from a dongle (round 2
        # cycled through 4 cache-resident buffers, which flattered the CPU)
        cmd = [sys.executable, me, "--cpu-child", variant, str(M), "96", str(seconds)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip():
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out = dict(value=round(d["value"], 2), unit="channel*Msamples/s", cores=1, kind="reference",
                       sample="unmodified reference rtl.c in_callback + msk.c + acars.c (%s), 1 channel per stream, "
                              "rtlMult=%d, %d callbacks of 1024 outputs in %.1f s on one host core (the reference is "
                              "single-threaded; cycling through 96 distinct 410 KB callbacks = 39 MB, beyond the per-core caches)" % (label, M, d["blocks"], d["seconds"]))
            # the fair "all host cores" number: one independent reference process per core
            ncpu = min(os.cpu_count() or 1, 64)
            if ncpu > 1:
                cmd[-1] = str(max(4.0, seconds / 2))
                ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(ncpu)]
                tot = 0.0
                for p in ps:
                    o, _ = p.communicate()
                    if p.returncode == 0 and o.strip():
                        tot += json.loads(o.strip().splitlines()[-1])["value"]
                out["all_cores"] = dict(value=round(tot, 1), processes=ncpu)
            return out
    # no reference build travelled: time the C restatement instead
    import numpy as np
    from oracle import oracle as O
    iq = np.random.default_rng(0).integers(0, 256, size=1024 * M * 2, dtype=np.uint8)
    taps = O.rtl_taps(131725000, 131750000, M)
    ch = O.Channel(0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ch.demod(O.fir_u8(iq, M, taps))
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=round(n * 1024 * M / dt / 1e6, 2), unit="channel*Msamples/s", cores=1, kind="port",
                sample="oracle/acars_oracle.c (-O2 IEEE), 1 channel, %d callbacks in %.1f s" % (n, dt))
