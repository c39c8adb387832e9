"""Golden fixture for SURVEY 8f.4 (outputmsg()'s field split): a synthetic 12.5 kHz recording with 120 transmissions
that exercise every branch of output.c:486-560 is played through the UNMODIFIED reference program
(oracle/_ref/acarsdec_cpu -o 4 -f <wav>, built by oracle/Makefile from /root/reference) and its JSON lines are kept.
Run in the build container only:

    python tests/golden/make_msgsplit_golden.py

Outputs (derived data, no reference source):
  msgsplit_pcm16.npz     the recording as int16
  msgsplit_golden.json   the reference's JSON output, one object per message (timestamps dropped)
"""
import json
import os
import subprocess
import sys
import tempfile
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acarsdec_amd import synth as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make_recording(seed=20260925, n=120):
    rng = np.random.default_rng(seed)
    frames = S.message_zoo(rng, n)
    a = S.frames_audio(frames, rng)
    return np.rint(np.clip(0.5 * a, -1, 1) * 20000).astype(np.int16)


def reference_json(pcm):
    exe = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu")
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "zoo.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(12500)
            w.writeframes(pcm.tobytes())
        r = subprocess.run([exe, "-o", "4", "-f", p], capture_output=True, text=True)
    out = []
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            for k in ("timestamp", "station_id", "app", "freq"):
                j.pop(k, None)
            out.append(j)
    return out


if __name__ == "__main__":
    pcm = make_recording()
    js = reference_json(pcm)
    assert len(js) >= 110, len(js)          # (the reference drops a block now and then: acars.c:124-207)
    np.savez_compressed(os.path.join(HERE, "msgsplit_pcm16.npz"), pcm=pcm)
    with open(os.path.join(HERE, "msgsplit_golden.json"), "w") as f:
        json.dump(js, f, indent=0)
    print("wrote %d messages, %d samples" % (len(js), pcm.size))
