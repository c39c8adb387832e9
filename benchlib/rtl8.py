"""BASELINE configs[1]: one 2.0 Msps dongle, 8 / 16 channels, a callback at a time -- legacy view, batched API, CPU reference."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter

from .cases import CARRIER, DEPTH
from .line import _short


# ------------------------------------------------------------------------------------------ BASELINE configs[1]: the rtl.c shape
RTL8 = dict(tag="BASELINE configs[1]: one dongle, 8 channels on ONE 2.0 Msps u8 stream (rtl.c's shape)", decim=160, callbacks=32)


def rtl8_cpu_child(variant, nfreq, path):
    """child process: the UNMODIFIED reference (oracle/_ref, its own flags) -- initRtl for the dongle's channels, then in_callback
    (rtl.c:314-361: mix + decimate for all channels, demodMSK per channel, decodeAcars) over the file's callbacks on one core"""
    import numpy as np
    from oracle import oracle as O
    M = RTL8["decim"]
    freqs = rtl8_freqs(int(nfreq))
    os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
    ref = O.Ref(variant)
    ref.init_rtl(freqs, M)
    iq = np.fromfile(path, dtype=np.uint8)
    blk = 1024 * M * 2
    bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]) for b in range(iq.size // blk)]
    for b in bufs:                      # warm-up: one pass over the file (page faults, caches, the core's clock)
        ref.in_callback(b)
    ref.init_rtl(freqs, M)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.5:   # whole passes over the file for >= 1.5 s (a single 32-callback pass is 10-80 ms: too short to time)
        for b in bufs:
            ref.in_callback(b)
        n += len(bufs)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(ms_per_callback=dt / n * 1e3, callbacks=n)))


def rtl8_freqs(nch):
    return ["%.3f" % (131.025 + 0.050 * k) for k in range(nch)]


def rtl8_oneline(chn, lvl, err, addr, fid, mode, label, no, txt):
    """printoneline() (output.c:327-346) without the date"""
    t = txt.split(b"\0")[0][:59].replace(b"\n", b" ").replace(b"\r", b" ")
    dec = lambda b: b.split(b"\0")[0].decode("latin-1")
    return "#%1d (L:%+5.1f E:%1d) %7s %6s %1s %2s %4s %s" % (chn + 1, lvl, err, dec(addr), dec(fid), dec(mode) or "\0", dec(label), dec(no), t.decode("latin-1"))


def run_rtl8(J, args):
    """BASELINE configs[1] and the path the north star names: nbch channels of ONE dongle on one 2.0 Msps u8 I/Q stream, handed
    over from host memory one reference callback (1024 outputs = 81.92 ms of signal, rtl.c:49,213) at a time.
      legacy   the reference's UNCHANGED acarsdec.c + acars.c + output.c + rtl.c with the one-hunk binding on
               compat_msk.c: acarsdec_amd_in_callback -> GPU -> every bit replayed through the unchanged decodeAcars() on the
               reference's own channel[] (lib/acarsdec_gpu_rtl, a file-playing librtlsdr stand-in); time inside the entry point
      batched  the same bytes through acg_process_iq_u8_host (nstreams = 1) + acg_collect_msgs one call behind
      cpu      the unmodified reference's in_callback on the same bytes, its own flags, one core (oracle/_ref)
    Parity: the legacy program's printed messages == the CPU twin program's (oracle/_ref/acarsdec_cpu_rtl), and the batched
    API's records, printed the same way, == both."""
    import re
    import tempfile
    import numpy as np
    from acarsdec_amd import decoder as D, synth as S, _capi as K
    M, ncb = RTL8["decim"], RTL8["callbacks"]
    gpu_exe = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_rtl")
    cpu_exe = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_rtl")
    out = {"workload": RTL8["tag"] + "; %d callbacks of 1024 outputs from host memory, rtlMult=%d" % (ncb, M), "budget_ms_per_callback": 81.92,
           "decim": M, "callbacks": ncb}
    strip = lambda txt: [l for l in re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", txt).splitlines() if l.startswith("#")]
    per_ch = lambda lines: {k: [l for l in lines if l.split()[0] == k] for k in sorted(set(l.split()[0] for l in lines))}
    with tempfile.TemporaryDirectory() as td:
        for nch in (8, 16):
            rng = np.random.default_rng(0x0881 + nch)
            freqs = rtl8_freqs(nch)
            fr = [D.parse_freq_mhz(f) for f in freqs]
            fc, _ = D.choose_fc(fr, M)
            env = np.zeros((nch, ncb * 1024))
            for c in range(nch):
                a_, _ = S.channel_audio(rng, env.shape[1], gap=(3125, 12500), text_len=(20, 120))
                env[c] = CARRIER * (1.0 + DEPTH * a_)
            iq = S.iq_u8_from_envelopes(env, M, [f - fc for f in fr], phases=list(rng.uniform(0, 2 * np.pi, nch)), scale=1.0 / nch, noise=0.004, rng=rng)
            path = os.path.join(td, "rtl%d.iq" % nch)
            iq.tofile(path)
            e = {"channels": nch}
            envp = dict(os.environ, ACARSDEC_IQ_FILE=path, ACARSDEC_AMD_STATS="1")
            lines = {}
            for name, exe in (("cpu", cpu_exe), ("legacy", gpu_exe)):
                if not os.path.exists(exe):
                    continue
                r = subprocess.run([exe, "-o", "1", "-r", "0"] + freqs, env=envp, capture_output=True, timeout=300)
                if r.returncode != 0:
                    e[name + "_error"] = _short(r.stderr.decode("latin-1"), 200)
                    continue
                lines[name] = per_ch(strip(r.stdout.decode("latin-1")))
                if name == "legacy":
                    m = re.search(r"first call ([0-9.]+) ms.*others ([0-9.]+) ms per call", r.stderr.decode("latin-1"))
                    if m:
                        e["legacy_first_call_ms"], e["legacy_ms_per_callback"] = float(m.group(1)), float(m.group(2))
            # the batched API, nstreams = 1: one host buffer per callback, messages collected one call behind
            dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=1, repair=True, bitlog=False, max_lag=1)
            dec.init_rtl(freqs)
            blk = 1024 * M * 2
            bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]).reshape(1, -1) for b in range(ncb)]

            def run_batched(sink):
                for b in bufs:
                    dec.in_callback(b)
                    while True:
                        n_, fb, more = dec.collect_msgs_raw(1, 256)
                        if sink is not None:
                            sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(n_)]
                        if not more:
                            break
                last = dec.drain_msgs(256)
                if sink is not None:
                    sink += last
            run_batched(None)                       # warm-up (first launches), then from reset
            dec.reset()
            msgs = []
            t0 = time.perf_counter()
            run_batched(msgs)
            e["batched_ms_per_callback"] = round((time.perf_counter() - t0) / ncb * 1e3, 4)
            dec.close()
            got = per_ch([rtl8_oneline(int(m.chn), m.lvl, int(m.err), m.addr, m.fid, m.mode, m.label, m.no, bytes(m.txt[: m.txt_len])) for m in msgs])
            lines["batched"] = got
            # the reference's in_callback on one host core (its own flags)
            for variant in ("_fast", "_v3", ""):
                if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libacarsref%s.so" % variant)):
                    continue
                r = subprocess.run([sys.executable, BENCH, "--rtl8-cpu-child", variant, str(nch), path], capture_output=True, text=True, timeout=300)
                if r.returncode == 0 and r.stdout.strip():
                    e["cpu_reference_ms_per_callback"] = round(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_callback"], 4)
                    break
            e["messages"] = sum(len(v) for v in lines.get("batched", {}).values())
            e["parity"] = {"legacy_program_equals_cpu_program": (lines["legacy"] == lines["cpu"]) if ("legacy" in lines and "cpu" in lines) else None,
                           "batched_equals_cpu_program": (lines["batched"] == lines["cpu"]) if "cpu" in lines else None,
                           "batched_equals_legacy_program": (lines["batched"] == lines["legacy"]) if "legacy" in lines else None}
            if any(v is False for v in e["parity"].values()):
                raise SystemExit("bench[rtl%d]: printed messages differ: %r" % (nch, {k: {c: len(v) for c, v in l.items()} for k, l in lines.items()}))
            out["ch%d" % nch] = e
    return out
