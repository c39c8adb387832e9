"""Pins the C restatement to the UNMODIFIED reference (oracle/_ref, built from /root/reference by
oracle/Makefile) run live on fresh inputs.  CPU only; skipped where no _ref build exists.
The reference is global state, so every scenario runs in a child process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no reference tree)")

CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from acarsdec_amd import synth as S
mode, seed = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(seed)
res = dict(ok=True, why=[])
def cmp_state(a, b, tag):
    for k in a:
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
            res["ok"] = False; res["why"].append("%%s %%s %%r %%r" %% (tag, k, a[k], b[k]))
def tup(f): return [int(f.chn), int(f.len), int(f.err), bytes(f.crc).hex(), bytes(f.txt[:f.len]).hex(), float(f.lvl).hex()]
ref = O.Ref()
if mode == "msk":
    nch, n = 3, 20000
    x = np.zeros((nch, n), dtype=np.float32)
    a, _ = S.channel_audio(rng, n, gap=(2000, 4000), text_len=(5, 80))
    x[0] = S.envelope(a, noise=0.02, rng=rng)
    x[1] = rng.normal(0.2, 0.2, n)                     # pure noise: exercises the reset path
    x[2, 5000:] = S.envelope(a[:n-5000], carrier=0.9, noise=0.0)
    ref.init_file(nch)
    chs = [O.Channel(c) for c in range(nch)]
    for s in range(0, n, 4096):
        for c in range(nch):
            ref.demod(c, x[c, s:s+4096]); chs[c].demod(x[c, s:s+4096])
    for c in range(nch): cmp_state(ref.state(c), chs[c].state(), "ch%%d" %% c)
    rf = sorted(tup(f) for f in ref.raw_frames()); of = sorted(tup(f) for c in chs for f in c.frames)
    if rf != of: res["ok"] = False; res["why"].append("frames %%d vs %%d" %% (len(rf), len(of)))
    res["nframes"] = len(rf)
else:
    M = int(mode)
    freqs = ["131.525", "131.725", "131.825"]
    fc = ref.init_rtl(freqs, M)
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    if O.choose_fc(fr, M) != fc: res["ok"] = False; res["why"].append("Fc")
    nblk = 3
    env = []
    for c in range(3):
        a, _ = S.channel_audio(rng, nblk * 1024, nframes=1, gap=(300, 600), text_len=(5, 20))
        env.append(0.5 * (1 + 0.5 * a))
    iq = S.iq_u8_from_envelopes(np.array(env), M, [f - fc for f in fr], phases=[0.2, 1.0, 4.0], noise=0.01, rng=rng)
    taps = [O.rtl_taps(fr[c], fc, M) for c in range(3)]
    for c in range(3):
        f, wf = ref.wf(c)
        if not np.array_equal(wf, taps[c]): res["ok"] = False; res["why"].append("taps%%d" %% c)
    chs = [O.Channel(c) for c in range(3)]
    blk = 1024 * M * 2
    for b in range(nblk):
        buf = iq[b*blk:(b+1)*blk]
        ref.in_callback(buf)
        for c in range(3):
            dm = O.fir_u8(buf, M, taps[c])
            if not np.array_equal(dm, ref.dm(c)): res["ok"] = False; res["why"].append("dm b%%d c%%d" %% (b, c))
            chs[c].demod(dm)
    for c in range(3): cmp_state(ref.state(c), chs[c].state(), "ch%%d" %% c)
    rf = sorted(tup(f) for f in ref.raw_frames()); of = sorted(tup(f) for c in chs for f in c.frames)
    if rf != of: res["ok"] = False; res["why"].append("frames")
    res["nframes"] = len(rf)
print(json.dumps(res))
'''


REPAIR_CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from acarsdec_amd import synth as S
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
kinds = [None, "p1", "p2", "p3", "db", "crc", "p4", "p1crc"]
nch, n = 4, 200000
ref = O.Ref(); ref.init_file(nch)
chs = [O.Channel(c, max_frames=512) for c in range(nch)]
for c in range(nch):
    a, sent = S.channel_audio(rng, n, gap=(1500, 3000), text_len=(15, 60), corrupt=kinds[c:] + kinds[:c])
    x = S.envelope(a, noise=0.002, rng=rng)
    for s in range(0, n, 4096):
        ref.demod(c, x[s:s+4096]); chs[c].demod(x[s:s+4096])
ref.drain()
def tup(f): return [int(f.chn), int(f.len), int(f.err), bytes(f.crc).hex(), bytes(f.txt[:f.len]).hex(), float(f.lvl).hex()]
raw = [f for c in chs for f in c.frames]
mine = sorted(tup(o) for o in (O.blk_process(f) for f in raw) if o is not None)
theirs = sorted(tup(f) for f in ref.out_frames())
nraw = len(raw)
print(json.dumps(dict(ok=mine == theirs, nraw=nraw, nout=len(theirs), nmine=len(mine),
                      repaired=sum(1 for t in theirs if t[2] > 0), raw_equal=sorted(tup(f) for f in raw) == sorted(tup(f) for f in ref.raw_frames()))))
'''


@pytest.mark.parametrize("seed", [5, 6])
def test_block_repair_identical_to_reference_blk_thread(seed):
    """acars.c:39-215 (parity/CRC check, fixprerr, fixdberr, parity strip) on transmissions with injected
    bit errors: what reaches outputmsg() is identical, block for block, drops included."""
    r = subprocess.run([sys.executable, "-c", REPAIR_CHILD % dict(root=ROOT), str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["raw_equal"] and res["ok"], res
    assert res["nraw"] >= 40 and res["nout"] < res["nraw"] and res["repaired"] >= 8, res


SOAPY_CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from acarsdec_amd import synth as S
M, seed, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
ref = O.Ref("_soapy")
freqs = ["131.525", "131.725", "131.825"]
fc = ref.init_soapy(freqs, M)
fr = [int(round(float(f) * 1e6)) for f in freqs]
res = dict(ok=True, why=[], fc=int(fc))
if O.choose_fc(fr, M) != fc: res["ok"] = False; res["why"].append("fc")
nblk = 3
env = []
for c in range(3):
    a, _ = S.channel_audio(rng, nblk * 1024, nframes=1, gap=(300, 600), text_len=(5, 20))
    env.append(0.5 * (1 + 0.5 * a))
iq = S.iq_s16_from_envelopes(np.array(env), M, [f - fc for f in fr], phases=[0.2, 1.0, 4.0], noise=0.01, rng=rng)
ref.dmlog_enable(nblk * 1024)
ref.soapy_feed(iq, chunk)          # reads of `chunk` samples: windows straddle read buffers (soapy.c:232-254)
ref.drain()
chs = [O.Channel(c) for c in range(3)]
for c in range(3):
    f, osc = ref.oscillator(c)
    mine = O.soapy_taps(float(fr[c]), fc, M)
    if not np.array_equal(osc, mine): res["ok"] = False; res["why"].append("osc%%d" %% c)
    dm = O.fir_cs16(iq, M, mine)
    rd = ref.dmlog(c)
    if not np.array_equal(dm[:rd.size], rd) or rd.size != nblk * 1024: res["ok"] = False; res["why"].append("dm%%d %%d" %% (c, rd.size))
    for b in range(0, dm.size, 1024): chs[c].demod(dm[b:b+1024])
    a, b2 = ref.state(c), chs[c].state()
    for k in a:
        if not np.array_equal(np.asarray(a[k]), np.asarray(b2[k])): res["ok"] = False; res["why"].append("state %%d %%s" %% (c, k))
def tup(f): return [int(f.chn), int(f.len), int(f.err), bytes(f.crc).hex(), bytes(f.txt[:f.len]).hex(), float(f.lvl).hex()]
rf = sorted(tup(f) for f in ref.raw_frames()); of = sorted(tup(f) for c in chs for f in c.frames)
if rf != of: res["ok"] = False; res["why"].append("frames")
res["nframes"] = len(rf)
print(json.dumps(res))
'''


@pytest.mark.parametrize("M,chunk", [(160, 0), (160, 1000), (192, 4097), (200, 777)])
def test_soapy_front_end_bit_identical_to_reference(M, chunk):
    """soapy.c (CS16, D carried across read buffers of any size): oscillator table, dm, state, blocks."""
    if not O.ref_available("_soapy"):
        pytest.skip("oracle/_ref/libacarsref_soapy.so not built")
    r = subprocess.run([sys.executable, "-c", SOAPY_CHILD % dict(root=ROOT), str(M), str(M + chunk), str(chunk)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"], res["why"]
    assert res["nframes"] >= 3


FE_CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from acarsdec_amd import synth as S
kind, seed, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
freqs = ["131.525", "131.725", "131.825"]
fr = [int(round(float(f) * 1e6)) for f in freqs]
res = dict(ok=True, why=[])
nblk = 3
env = []
for c in range(3):
    a, _ = S.channel_audio(rng, nblk * 1024, nframes=1, gap=(300, 600), text_len=(5, 20))
    env.append(0.5 * (1 + 0.5 * a))
if kind == "sdrplay":
    ref = O.Ref("_sdrplay"); M = 160
    fc = ref.init_sdrplay(freqs)
    if O.choose_fc(fr, M) != fc: res["ok"] = False; res["why"].append("fc %%d" %% fc)
    iq = S.iq_s16_from_envelopes(np.array(env), M, [f - fc for f in fr], phases=[0.2, 1.0, 4.0], noise=0.01, rng=rng, full_scale=0.05)
    xi, xq = iq[0::2].copy(), iq[1::2].copy()
    ref.dmlog_enable(nblk * 1024)
    ref.sdrplay_feed(xi, xq, chunk)
    taps = [O.sdrplay_taps(float(fr[c]), fc) for c in range(3)]
    dms = [O.fir_split16(xi, xq, M, taps[c]) for c in range(3)]
    gettaps = lambda c: ref.L.ref_get_oscillator
    step = 512
else:
    ref = O.Ref("_air"); rate = int(sys.argv[4]); M = rate // 12500
    fc = ref.init_air(freqs, rate)
    if O.air_choose_fc(fr) != fc: res["ok"] = False; res["why"].append("fc %%d" %% fc)
    # air.c mixes with Fc - Fr + Fs/4: a channel at Fr sits at that frequency of the real spectrum
    x = S.real_f32_from_envelopes(np.array(env), M, [fc - f + rate / 4 for f in fr], phases=[0.2, 1.0, 4.0], noise=0.01, rng=rng)
    ref.dmlog_enable(nblk * 1024 + 8)
    ref.air_feed(x, chunk)
    taps = [O.air_taps(fr[c], fc, rate) for c in range(3)]
    dms = [O.fir_f32r(x, M, taps[c]) for c in range(3)]
    step = None
ref.drain()
import ctypes as C
chs = [O.Channel(c) for c in range(3)]
for c in range(3):
    out = np.zeros((M, 2), dtype=np.float32)
    (ref.L.ref_get_oscillator if kind == "sdrplay" else ref.L.ref_get_wf)(c, out.ctypes.data, M)
    if not np.array_equal(out, taps[c]): res["ok"] = False; res["why"].append("taps%%d" %% c)
    rd = ref.dmlog(c)
    dm = dms[c]
    if rd.size == 0 or not np.array_equal(dm[:rd.size], rd): res["ok"] = False; res["why"].append("dm%%d %%d" %% (c, rd.size))
    chs[c].demod(dm[:rd.size])
    a, b2 = ref.state(c), chs[c].state()
    for k in a:
        if not np.array_equal(np.asarray(a[k]), np.asarray(b2[k])): res["ok"] = False; res["why"].append("state %%d %%s" %% (c, k))
def tup(f): return [int(f.chn), int(f.len), int(f.err), bytes(f.crc).hex(), bytes(f.txt[:f.len]).hex(), float(f.lvl).hex()]
rf = sorted(tup(f) for f in ref.raw_frames()); of = sorted(tup(f) for c in chs for f in c.frames)
if rf != of: res["ok"] = False; res["why"].append("frames %%d %%d" %% (len(rf), len(of)))
res["nframes"] = len(rf)
print(json.dumps(res))
'''


@pytest.mark.parametrize("kind,chunk,rate", [("sdrplay", 0, 0), ("sdrplay", 504, 0), ("sdrplay", 1001, 0),
                                             ("air", 32768, 2500000), ("air", 65536, 2500000), ("air", 1000, 2500000),
                                             ("air", 65536, 6000000), ("air", 40000, 10000000)])
def test_sdrplay_and_airspy_front_ends_bit_identical_to_reference(kind, chunk, rate):
    """sdrplay.c (split int16 planes, cabsf(D)/4) and air.c (real float32, complex taps): taps, centre
    frequency, dm, demodulator state and blocks, with the stream cut into callbacks of any size; Airspy
    at the rates its devices offer (air.c:211-214: R2 10 / 2.5 Msps, Mini 6 Msps -> 800 / 200 / 480 taps)."""
    if not O.ref_available("_" + kind):
        pytest.skip("oracle/_ref/libacarsref_%s.so not built" % kind)
    r = subprocess.run([sys.executable, "-c", FE_CHILD % dict(root=ROOT), kind, str(7 + chunk), str(chunk), str(rate)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"], res["why"]
    assert res["nframes"] >= 2


def test_syndrome_table_regenerated_equals_reference_header():
    import re
    path = "/root/reference/syndrom.h"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    txt = open(path).read()
    body = txt[txt.index("static const unsigned short syndrom[]"):]
    vals = np.array([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)], dtype=np.uint16)
    assert vals.size == 1936 and np.array_equal(vals, O.syndrome_table(1936))


def run_child(mode, seed):
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), mode, str(seed)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("seed", [1, 2])
def test_msk_and_framing_bit_identical_to_reference(seed):
    res = run_child("msk", seed)
    assert res["ok"], res["why"]
    assert res["nframes"] >= 2


@pytest.mark.parametrize("M", [160, 192, 200])
def test_rtl_front_end_bit_identical_to_reference(M):
    res = run_child(str(M), 10 + M)
    assert res["ok"], res["why"]
    assert res["nframes"] >= 3


def test_choose_fc_matches_reference_on_hard_sets():
    """chooseFc edge cases (mirror-image rejection makes Fc walk in 1 Hz steps, rtl.c:160)."""
    code = r'''
import sys, json; sys.path.insert(0, %r)
from oracle import oracle as O
ref = O.Ref(); sets = json.loads(sys.argv[1]); out = []
for s in sets:
    try: out.append(int(ref.init_rtl(s, 160)))
    except RuntimeError: out.append(0)
print(json.dumps(out))
''' % ROOT
    sets = [["131.550"], ["131.525", "131.550"], ["130.025", "131.825"], ["131.125", "131.450", "131.475", "131.525", "131.550", "131.650", "131.725", "131.825"],
            ["129.125", "131.125"]]
    r = subprocess.run([sys.executable, "-c", code, json.dumps(sets)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    ref_fc = json.loads(r.stdout.strip().splitlines()[-1])
    for s, fc in zip(sets, ref_fc):
        fr = [(int(1000000 * float(f) + 6250) // 12500) * 12500 for f in s]
        assert O.choose_fc(fr, 160) == fc, (s, fc)


def test_message_split_against_live_reference_program(tmp_path):
    """orc_msg_split (output.c:486-560) pinned to the reference program run live on a fresh random recording:
    300 transmissions, JSON output (-o 4), every field."""
    import json
    import subprocess
    import wave
    from conftest import msg_fields_from_json, msg_fields_from_record
    from acarsdec_amd import synth as S
    exe = os.path.join(os.path.dirname(O.ref_path()), "acarsdec_cpu")
    if not os.path.exists(exe):
        pytest.skip("reference program not built")
    rng = np.random.default_rng(777)
    a = S.frames_audio(S.message_zoo(rng, 300), rng)
    pcm = np.rint(np.clip(0.5 * a, -1, 1) * 20000).astype(np.int16)
    p = str(tmp_path / "zoo.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(12500)
        w.writeframes(pcm.tobytes())
    r = subprocess.run([exe, "-o", "4", "-f", p], capture_output=True, text=True)
    want = [msg_fields_from_json(json.loads(l)) for l in r.stdout.splitlines() if l.startswith("{")]
    ch = O.Channel(0, max_frames=1024)
    ch.demod(pcm.astype(np.float32) / 32768.0)
    got = [msg_fields_from_record(O.msg_split(b)) for b in (O.blk_process(f) for f in ch.frames) if b is not None]
    assert len(want) >= 280 and got == want


def test_reference_builds_against_each_other_and_the_oracle():
    """bench.py's gate measures how many blocks the reference's own -O2 and -Ofast builds differ by on its input
    (oracle.ref_blocks_forked: the unmodified in_callback -> demodMSK -> decodeAcars with a caller's tap table through
    ref_set_wf).  Here at a small size: the -O2 build equals the oracle block for block (that is the pin), the -Ofast build
    decodes the same blocks on this input (its dm differs at the 1e-7 level; a razor-edge soft decision could differ, none
    does here), and a build that is not there reports None instead of raising."""
    from acarsdec_amd import synth as S
    M, nblk, nch = 200, 20, 6
    nout = nblk * 1024
    sigma = 0.25 * 0.5 * (M / (2.0 * 10 ** 2.0)) ** 0.5           # 20 dB SNR in the 12.5 kHz channel (bench.py)
    rows, taps = [], []
    for c in range(nch):
        a, _ = S.channel_audio(np.random.default_rng(0xACA25 + c), nout, gap=(3125, 12500), text_len=(20, 220))
        off = 25000.0 * (2 + c) * (-1) ** c
        rows.append(S.iq_u8_from_envelopes((0.5 * (1 + 0.5 * a))[None, :], M, [off], phases=[0.37 * c], noise=sigma,
                                           rng=np.random.default_rng(500 + c)).reshape(-1))
        w = O.rtl_taps(131000000 + int(off), 131000000, M)
        taps.append(w if c % 2 == 0 else w[:192])                  # ref_set_wf zero-fills a shorter table
    o2 = O.ref_blocks_forked("", rows, M, taps)
    assert o2 is not None and sum(len(b) for b in o2) >= nch - 1
    for c in range(nch):
        ch = O.Channel(c)
        ch.demod(O.fir_u8(rows[c], M, taps[c], ntaps=taps[c].shape[0]))
        assert [O.frame_tuple(f)[1:] for f in ch.frames] == o2[c], c
    fast = O.ref_blocks_forked("_fast", rows, M, taps) or O.ref_blocks_forked("_v3", rows, M, taps)
    assert fast is not None
    assert sum(len(set(x) ^ set(y)) for x, y in zip(o2, fast)) == 0
    assert O.ref_blocks_forked("_no_such_build", rows, M, taps) is None


def test_reference_soapy_program_decodes_the_golden_recording(tmp_path):
    """The CPU twin of the SoapySDR demo: the reference's UNCHANGED soapy.c + msk.c + acars.c + output.c behind a file-playing
    SoapySDR stand-in with ragged reads (acarsdec_amd/csrc/demo/demo_soapy_file.c), fed the golden recording up-converted to CS16
    at the offsets soapy.c's own chooseFc gives: it prints the seven messages of SURVEY App. B.  This is what the GPU test
    test_compat_soapy_program_output_equals_the_cpu_program compares the bound program with; it also pins the centre frequency
    that test uses."""
    import re
    import subprocess
    from acarsdec_amd import synth as S
    exe = os.path.join(os.path.dirname(O.ref_path()), "acarsdec_cpu_soapy")
    if not (os.path.exists(exe) and O.ref_available("_soapy")):
        pytest.skip("oracle/_ref not built")
    freqs, M = ["131.525", "131.725", "131.825", "131.550"], 160
    code = "import sys; sys.path.insert(0, %r); from oracle import oracle as O; print(O.Ref('_soapy').init_soapy(%r, %d))" % (
        os.path.dirname(os.path.dirname(os.path.abspath(O.__file__))), freqs, M)
    fc = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
    assert fc == 131850000
    pcm = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testwav_pcm16.npz"))["pcm"]
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    env = S.pad_blocks(0.5 + 0.5 * wav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * 3), 0.5)], axis=1)
    iq = S.iq_s16_from_envelopes(env, M, [f - fc for f in fr], phases=[0.3, 1.1, 2.2, 0.7])
    path = tmp_path / "t.cs16"
    path.write_bytes(iq.tobytes())
    r = subprocess.run([exe, "-o", "1", "-m", str(M), "-d", "file"] + freqs, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)), capture_output=True, timeout=300)
    out = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1"))
    assert r.returncode == 0 and out.count("\n") == 7, out + r.stderr.decode("latin-1")[-500:]
    for tail in ("PH-BXR KL1681 E 5V S53A", "LN-DYY DY083J 2 Q0 S46A", "F-GTAE AF7728 G H1 D65C", "G-DBCK BA031T E Q0 S63A"):
        assert tail in out, tail


@pytest.mark.parametrize("fe", ["air", "sdrplay"])
def test_reference_callback_front_end_programs_decode_the_golden_recording(fe, tmp_path):
    """The CPU twins of the Airspy / SDRplay demos: the reference's UNCHANGED air.c / sdrplay.c + msk.c + acars.c + output.c behind
    file-playing vendor-library stand-ins with ragged transfers (acarsdec_amd/csrc/demo/demo_{airspy,sdrplay}_file.c), fed the
    golden recording as real float32 around Fs/4 / as int16 I/Q at the offsets the front end's own centre-frequency choice gives:
    each prints the seven messages of SURVEY App. B.  The GPU test compares the bound programs with these."""
    import re
    import subprocess
    from acarsdec_amd import synth as S
    exe = os.path.join(os.path.dirname(O.ref_path()), "acarsdec_cpu_" + fe)
    if not (os.path.exists(exe) and O.ref_available("_" + fe)):
        pytest.skip("oracle/_ref not built")
    freqs = ["131.525", "131.725", "131.825", "131.550"]
    fr = [int(round(float(f) * 1e6)) for f in freqs]
    pcm = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "testwav_pcm16.npz"))["pcm"]
    wav = pcm.astype(np.float32) / np.float32(32768.0)
    env = S.pad_blocks(0.5 + 0.5 * wav.T.astype(np.float64), 1024, 0.5)
    env = np.concatenate([env, np.full((4, 1024 * 3), 0.5)], axis=1)
    if fe == "air":
        rate = 2500000
        fc = O.air_choose_fc(fr)
        data = S.real_f32_from_envelopes(env, rate // 12500, [fc - f + rate / 4 for f in fr], phases=[0.3, 1.1, 2.2, 0.7], scale=0.15)
        args = ["-o", "1", "-s", "0"] + freqs
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(O.__file__)))
        code = "import sys; sys.path.insert(0, %r); from oracle import oracle as O; print(O.Ref('_sdrplay').init_sdrplay(%r))" % (root, freqs)
        fc = int(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip().splitlines()[-1])
        assert fc == 131850000
        data = S.iq_s16_from_envelopes(env, 160, [f - fc for f in fr], phases=[0.3, 1.1, 2.2, 0.7], full_scale=0.06)
        args = ["-o", "1", "-s"] + freqs
    path = tmp_path / ("t." + fe)
    path.write_bytes(data.tobytes())
    r = subprocess.run([exe] + args, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)), capture_output=True, timeout=300)
    out = re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1"))
    msgs = [l for l in out.splitlines() if l.startswith("#")]
    assert r.returncode == 0 and len(msgs) == 7, out + r.stderr.decode("latin-1")[-500:]
    for tail in ("PH-BXR KL1681 E 5V S53A", "LN-DYY DY083J 2 Q0 S46A", "F-GTAE AF7728 G H1 D65C", "G-DBCK BA031T E Q0 S63A"):
        assert tail in out, tail


def test_rtl8_one_line_format_equals_the_reference_program(tmp_path):
    """bench.py's rtl8 case (BASELINE configs[1]) compares three printed outputs: the CPU reference program's, the GPU legacy
    program's, and the batched API's records printed by bench.rtl8_oneline().  That formatter must be printoneline()
    (output.c:327-346) minus the date: here the oracle's messages for an 8-channel dongle, printed by it, against what the
    UNMODIFIED reference program (oracle/_ref/acarsdec_cpu_rtl: rtl.c + msk.c + acars.c + output.c) prints on the same I/Q file."""
    import re
    import subprocess
    import bench
    from acarsdec_amd import decoder as D, synth as S
    exe = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_rtl")
    if not os.path.exists(exe):
        pytest.skip("needs the reference program (oracle/_ref, built where /root/reference exists)")
    M, ncb, nch = 160, 10, 8
    rng = np.random.default_rng(0x0881 + nch)
    freqs = bench.rtl8_freqs(nch)
    fr = [D.parse_freq_mhz(f) for f in freqs]
    fc, _ = D.choose_fc(fr, M)
    env = np.zeros((nch, ncb * 1024))
    for c in range(nch):
        a_, _ = S.channel_audio(rng, env.shape[1], gap=(3125, 12500), text_len=(20, 120))
        env[c] = 0.5 * (1 + 0.5 * a_)
    iq = S.iq_u8_from_envelopes(env, M, [f - fc for f in fr], phases=list(rng.uniform(0, 6.28, nch)), scale=1.0 / nch, noise=0.004, rng=rng)
    path = tmp_path / "x.iq"
    iq.tofile(str(path))
    r = subprocess.run([exe, "-o", "1", "-r", "0"] + freqs, env=dict(os.environ, ACARSDEC_IQ_FILE=str(path)), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode("latin-1")[-500:]
    want = [l for l in re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", r.stdout.decode("latin-1")).splitlines() if l.startswith("#")]
    mine = []
    for c in range(nch):
        ch = O.Channel(c, max_frames=256)
        ch.demod(O.fir_u8(iq, M, D.rtl_taps(fr[c], fc, M)))
        for f in ch.frames:
            o = O.blk_process(f)
            if o is not None:
                m = O.msg_split(o)
                mine.append(bench.rtl8_oneline(c, m.lvl, int(m.err), m.addr, m.fid, m.mode, m.label, m.no, bytes(m.txt[: m.txt_len])))
    assert sorted(mine) == sorted(want) and len(want) >= 5
