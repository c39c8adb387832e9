import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        from acarsdec_amd import _capi as K
        return K.load().acg_device_count() > 0
    except Exception:
        return False


@pytest.fixture
def tune():
    """tune(name, value): a measurement switch of the library for the rest of this test (acg_tune; None removes it).  The
    library reads the environment only once per process, so setting os.environ inside a test would do nothing."""
    from acarsdec_amd import _capi as K
    touched = []

    def set_(name, value, lab=False):
        """lab=True: the switch of the LAB build of the library (Decoder(..., lab=True)), which has its own table"""
        K.tune(name, value, lab=lab)
        touched.append((name, lab))
    yield set_
    for n, lab in touched:
        K.tune(n, None, lab=lab)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "testwav_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def testwav():
    """test.wav as float32 [frames, 4] exactly as sf_read_float delivers it (x/32768)."""
    z = np.load(os.path.join(GOLDEN, "testwav_pcm16.npz"))
    pcm = z["pcm"]
    return (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


@pytest.fixture(scope="session")
def golden_bits():
    z = np.load(os.path.join(GOLDEN, "testwav_bits.npz"))
    return {k: z[k] for k in z.files}


def fhex(s):
    return float.fromhex(s)


def golden_blocks(lst):
    """[(chn,len,err,crc,txt)] + levels from a fixture list"""
    out = []
    for b in lst:
        out.append(((b["chn"], b["len"], b["err"], bytes.fromhex(b["crc"]), bytes.fromhex(b["txt"])),
                    np.float32(fhex(b["lvl"]))))
    return out


def soft_from_v(vr, vi, MskS):
    """msk.c:110-126 applied to logged matched-filter outputs: returns (signed vo, lvl) float32."""
    vr = vr.astype(np.float32)
    vi = vi.astype(np.float32)
    lvl = np.sqrt(vr.astype(np.float64) ** 2 + vi.astype(np.float64) ** 2).astype(np.float32)
    d = lvl.astype(np.float64) + 1e-8
    nr = (vr.astype(np.float64) / d).astype(np.float32)
    ni = (vi.astype(np.float64) / d).astype(np.float32)
    vo = np.where(MskS & 1, ni, nr)
    vo = np.where(MskS & 2, -vo, vo).astype(np.float32)
    return vo, lvl


def msg_fields_from_json(j):
    """the reference program's JSON line (output.c:227-324) reduced to the fields of outputmsg()'s split"""
    return dict(chn=j["channel"], err=j["error"], mode=j["mode"], label=j["label"], bid=j.get("block_id", ""), ack=j.get("ack"),
                tail=j.get("tail"), flight=j.get("flight"), msgno=j.get("msgno"), text=j.get("text", ""), end=j.get("end", False),
                level="%2.1f" % j["level"])


def msg_fields_from_record(m):
    """the same view of an oracle OrcMsg / library acg_msg record"""
    down = m.down not in (b"\x00", 0)
    return dict(chn=int(m.chn), err=int(m.err), mode=m.mode.decode("latin1"), label=m.label.decode("latin1"), bid=m.bid.decode("latin1"),
                ack=False if m.ack == b"!" else m.ack.decode("latin1"), tail=m.addr.decode("latin1"),
                flight=m.fid.decode("latin1") if down else None, msgno=m.no.decode("latin1") if down else None,
                text=bytes(m.txt[: m.txt_len]).decode("latin1"), end=(m.be == b"\x17"), level="%2.1f" % m.lvl)


@pytest.fixture(scope="session")
def msgsplit_golden():
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    pcm = np.load(os.path.join(here, "msgsplit_pcm16.npz"))["pcm"]
    with open(os.path.join(here, "msgsplit_golden.json")) as f:
        return pcm, json.load(f)
