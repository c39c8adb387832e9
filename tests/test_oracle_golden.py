"""The oracle (oracle/acars_oracle.c) against the committed golden fixtures, which were produced
by the UNMODIFIED reference sources (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np

from conftest import fhex, golden_blocks, soft_from_v
from oracle import oracle as O
from acarsdec_amd import synth as S


def run_file(x, chunk):
    chs = [O.Channel(n, max_bits=12000) for n in range(x.shape[1])]
    for s in range(0, x.shape[0], chunk):
        for n, ch in enumerate(chs):
            ch.demod(np.ascontiguousarray(x[s:s + chunk, n]))
    return chs


def test_oracle_reproduces_testwav_blocks_state_and_bits(testwav, golden, golden_bits):
    chs = run_file(testwav, 4096)                      # soundfile.c:27 chunking
    gb = golden_blocks(golden["file"]["raw_blocks"])
    got = sorted((O.frame_tuple(f), np.float32(f.lvl).tobytes()) for c in chs for f in c.frames)
    want = sorted((t, l.tobytes()) for t, l in gb)
    assert got == want and len(got) == 7               # blocks AND levels, to the bit
    for n, ch in enumerate(chs):
        g, s = golden["file"]["final_state"][n], ch.state()
        for k in ("MskPhi", "MskDf", "MskClk", "MskLvlSum"):
            assert s[k] == fhex(g[k]), (n, k)
        for k in ("MskBitCount", "MskS", "idx", "outbits", "nbits", "Acarsstate"):
            assert s[k] == g[k], (n, k)
        assert [float(v) for v in s["inb"]] == [fhex(v) for v in g["inb"]]
        m = golden_bits["chn"] == n
        vo_ref, lvl_ref = soft_from_v(golden_bits["vr"][m], golden_bits["vi"][m], golden_bits["MskS"][m])
        vo, lvl = ch.bits
        assert np.array_equal(vo, vo_ref) and np.array_equal(lvl, lvl_ref)
        assert len(vo) == golden["file"]["bits_per_channel"][n]


def test_oracle_chunking_invariance(testwav):
    a = run_file(testwav, 4096)
    b = run_file(testwav, 1000)
    for x, y in zip(a, b):
        assert [O.frame_tuple(f) for f in x.frames] == [O.frame_tuple(f) for f in y.frames]
        assert x.state()["MskPhi"] == y.state()["MskPhi"] and x.state()["MskS"] == y.state()["MskS"]


def test_oracle_rtl_path_golden(testwav, golden):
    """rtl.c path: taps, chooseFc, dm of the first callback and the decoded blocks."""
    g = golden["rtl"]
    M = g["M"]
    fr = [int(round(float(f) * 1e6)) for f in g["freqs"]]
    fc = O.choose_fc(fr, M)
    assert fc == g["Fc"]
    env = S.pad_blocks(0.5 + 0.5 * testwav.T.astype(np.float64), 1024, 0.5)
    iq = S.iq_u8_from_envelopes(env, M, [f - fc for f in fr], phases=g["phases"])
    if hashlib.sha256(iq.tobytes()).hexdigest() != g["iq_sha256"]:
        import pytest
        pytest.skip("numpy produced different synthetic IQ bytes than when the fixture was made")
    blocks = []
    for n in range(4):
        dm = O.fir_u8(iq, M, O.rtl_taps(fr[n], fc, M))
        assert [float(v) for v in dm[:64]] == [fhex(v) for v in g["dm_block0"][n]]
        ch = O.Channel(n)
        for b in range(0, dm.size, 1024):              # rtl.c:357-360: demodMSK per 1024-sample callback
            ch.demod(dm[b:b + 1024])
        blocks += [(O.frame_tuple(f), np.float32(f.lvl).tobytes()) for f in ch.frames]
        gs = g["final_state"][n]
        assert ch.state()["MskPhi"] == fhex(gs["MskPhi"]) and ch.state()["MskS"] == gs["MskS"]
    assert sorted(blocks) == sorted((t, l.tobytes()) for t, l in golden_blocks(g["raw_blocks"]))


def test_golden_program_output_lists_the_seven_messages(golden):
    out = golden["program"]["o1"]["stdout"]
    assert golden["program"]["o1"]["md5"] == "d2fbf112e9e07c970e52bbbb2f10b507"
    assert out.count("\n") == 7 and "F-GTAE AF7728" in out and "G-DBCK BA031T" in out


def test_frame_check_and_crc(golden):
    for t, _ in golden_blocks(golden["file"]["raw_blocks"]):
        f = O.OrcFrame()
        f.chn, f.len, f.err = t[0], t[1], t[2]
        f.crc[0], f.crc[1] = t[3][0], t[3][1]
        for i, b in enumerate(t[4]):
            f.txt[i] = b
        assert O.lib().orc_frame_check(f) == 0
        assert O.crc_ccitt(t[4] + t[3]) == 0
        assert O.crc_ccitt(t[4]) == S.crc_ccitt(t[4])
    f = O.OrcFrame()
    f.len = 5
    assert O.lib().orc_frame_check(f) == -1            # acars.c:124 too short


def test_oracle_message_split_matches_reference_json(msgsplit_golden):
    """120 transmissions covering every branch of outputmsg()'s field split (uplink / downlink, NAK, DEL
    label, dotted addresses, empty and short texts, ETB): oracle demod -> block repair -> orc_msg_split against the
    JSON the unmodified reference program printed for the same recording (tests/golden/make_msgsplit_golden.py)."""
    from conftest import msg_fields_from_json, msg_fields_from_record
    pcm, want = msgsplit_golden
    ch = O.Channel(0, max_frames=512)
    ch.demod(pcm.astype(np.float32) / 32768.0)
    got = []
    for f in ch.frames:
        b = O.blk_process(f)
        if b is not None:
            got.append(msg_fields_from_record(O.msg_split(b)))
    assert len(got) == len(want) >= 110
    assert got == [msg_fields_from_json(j) for j in want]
    kinds = {(g["flight"] is None, g["ack"] is False, g["text"] == "", g["end"], g["label"].endswith("d")) for g in got}
    assert len(kinds) >= 12            # the zoo really spans the branches
