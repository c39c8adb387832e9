"""Generates the committed golden fixtures from the UNMODIFIED reference (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Outputs (all derived data, no reference source):
  testwav_pcm16.npz     the reference's only data fixture, test.wav, as int16 [frames, 4]
  testwav_golden.json   raw blocks queued by decodeAcars, final channel_t state, stdout of the
                        reference program (-o 1/2/4) and its md5s, rtl-path blocks
  testwav_bits.npz      per-bit matched-filter outputs (msk.c:110) of every channel
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from acarsdec_amd import synth as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = "/root/reference/test.wav"
RTL_FREQS = ["131.525", "131.725", "131.825", "131.550"]
RTL_PHASES = [0.1, 1.0, 2.0, 3.0]
RTL_M = 160


def fhex(x):
    return float(x).hex()


def frames_json(frames):
    return [dict(chn=int(f.chn), len=int(f.len), err=int(f.err), lvl=fhex(f.lvl),
                 crc=bytes(f.crc).hex(), txt=bytes(f.txt[:f.len]).hex()) for f in frames]


def state_json(s):
    return dict(MskPhi=fhex(s["MskPhi"]), MskDf=fhex(s["MskDf"]), MskClk=fhex(s["MskClk"]),
                MskLvlSum=fhex(s["MskLvlSum"]), MskBitCount=int(s["MskBitCount"]), MskS=int(s["MskS"]),
                idx=int(s["idx"]), inb=[fhex(v) for v in s["inb"]], outbits=int(s["outbits"]),
                nbits=int(s["nbits"]), Acarsstate=int(s["Acarsstate"]))


def child_file():
    """sound-file path: soundfile.c chunking (4096 frames, channels in order)."""
    rate, pcm = O.read_wav_pcm16(WAV)
    x = O.wav_to_float(pcm)
    nch = x.shape[1]
    ref = O.Ref()
    ref.init_file(nch)
    ref.bitlog_enable(1 << 20)
    for s in range(0, x.shape[0], 4096):
        for n in range(nch):
            ref.demod(n, np.ascontiguousarray(x[s:s + 4096, n]))
    ref.drain()
    vr, vi, S_, chn = ref.bitlog()
    np.savez_compressed(os.path.join(HERE, "testwav_bits.npz"), vr=vr, vi=vi, MskS=S_.astype(np.uint32),
                        chn=chn.astype(np.int8))
    out = dict(rate=rate, frames=int(pcm.shape[0]), channels=nch,
               raw_blocks=frames_json(ref.raw_frames()), out_blocks=frames_json(ref.out_frames()),
               final_state=[state_json(ref.state(n)) for n in range(nch)],
               bits_per_channel=[int((chn == n).sum()) for n in range(nch)])
    print(json.dumps(out))


def child_rtl():
    """rtl.c path: the same audio up-converted onto 4 carriers of one 2.0 Msps stream."""
    rate, pcm = O.read_wav_pcm16(WAV)
    x = O.wav_to_float(pcm)
    ref = O.Ref()
    Fc = ref.init_rtl(RTL_FREQS, RTL_M)
    Fr = [int(round(float(f) * 1e6)) for f in RTL_FREQS]
    env = S.pad_blocks(0.5 + 0.5 * x.T.astype(np.float64), 1024, 0.5)
    iq = S.iq_u8_from_envelopes(env, RTL_M, [f - Fc for f in Fr], phases=RTL_PHASES)
    blk = 1024 * RTL_M * 2
    dm0 = []
    for b in range(env.shape[1] // 1024):
        ref.in_callback(iq[b * blk:(b + 1) * blk])
        if b == 0:
            dm0 = [ref.dm(n).tolist() for n in range(4)]
    ref.drain()
    out = dict(Fc=int(Fc), M=RTL_M, freqs=RTL_FREQS, phases=RTL_PHASES, iq_sha256=hashlib.sha256(iq.tobytes()).hexdigest(),
               raw_blocks=frames_json(ref.raw_frames()),
               final_state=[state_json(ref.state(n)) for n in range(4)],
               dm_block0=[[fhex(v) for v in d[:64]] for d in dm0])
    print(json.dumps(out))


def main():
    O.build()
    rate, pcm = O.read_wav_pcm16(WAV)
    np.savez_compressed(os.path.join(HERE, "testwav_pcm16.npz"), pcm=pcm, rate=rate)
    # the reference is all global state: one process per scenario
    gold = {}
    for name in ("file", "rtl"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, check=True)
        gold[name] = json.loads(r.stdout.strip().splitlines()[-1])
    cpu = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu")
    prog = {}
    for o in ("1", "2", "4"):
        r = subprocess.run([cpu, "-o", o, "-f", WAV], capture_output=True)
        txt = r.stdout
        prog["o" + o] = dict(md5=hashlib.md5(txt).hexdigest())
        if o == "1":
            prog["o1"]["stdout"] = txt.decode("latin-1")
        if o == "4":       # JSON lines carry wall-clock time and the host name: keep the rest
            recs = []
            for line in txt.decode("latin-1").splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    d.pop("timestamp", None)
                    d.pop("station_id", None)
                    recs.append(d)
            prog["o4"]["records"] = recs
    gold["program"] = prog
    # the rtl.c path of the whole program: reference acarsdec.c + rtl.c + msk.c ... fed by the file-playing
    # librtlsdr stand-in (acarsdec_amd/csrc/demo) with the same synthetic I/Q plus 4 blocks of bare carrier
    import re
    import tempfile
    rate, pcm = O.read_wav_pcm16(WAV)
    x = O.wav_to_float(pcm)
    fr = [int(round(float(f) * 1e6)) for f in RTL_FREQS]
    fc = gold["rtl"]["Fc"]
    env = S.pad_blocks(0.5 + 0.5 * x.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 4096), 0.5)], axis=1)
    iq = S.iq_u8_from_envelopes(env, RTL_M, [f - fc for f in fr], phases=RTL_PHASES)
    with tempfile.NamedTemporaryFile(suffix=".iq", delete=False) as f:
        f.write(iq.tobytes())
        path = f.name
    args = ["-o", "1", "-r", "0"] + RTL_FREQS
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_rtl")] + args,
                       env=dict(os.environ, ACARSDEC_IQ_FILE=path), capture_output=True)
    os.unlink(path)
    gold["program_rtl"] = dict(args=args, tail_blocks=4, iq_sha256=hashlib.sha256(iq.tobytes()).hexdigest(),
                               stdout_no_timestamps=re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1")))
    with open(os.path.join(HERE, "testwav_golden.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "file":
        child_file()
    elif len(sys.argv) > 1 and sys.argv[1] == "rtl":
        child_rtl()
    else:
        main()
