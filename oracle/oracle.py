"""ctypes bindings for the CPU checkers in oracle/ -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  The product (acarsdec_amd/) never does.

Two libraries:
  * liboracle.so          -- the C restatement (acars_oracle.c)
  * _ref/libacarsref*.so  -- the unmodified reference sources + ref_glue.c (built only where
                             /root/reference exists; the built .so travels to the GPU box)
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INTRATE = 12500
FLEN = 11
TXTMAX = 250


def build(verbose=False):
    """(Re)build liboracle.so and, when the reference tree is present, _ref/."""
    r = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


# --------------------------------------------------------------------------- restatement
class OrcFrame(C.Structure):
    _fields_ = [("chn", C.c_int), ("len", C.c_int), ("err", C.c_int), ("lvl", C.c_float),
                ("crc", C.c_ubyte * 2), ("txt", C.c_ubyte * TXTMAX), ("end_bit", C.c_longlong),
                ("end_sample", C.c_longlong), ("soh_sample", C.c_longlong)]


class OrcBit(C.Structure):
    _fields_ = [("vo", C.c_float), ("lvl", C.c_float)]


class OrcChan(C.Structure):
    _fields_ = [("chn", C.c_int), ("MskPhi", C.c_double), ("MskDf", C.c_double),
                ("MskClk", C.c_float), ("MskLvlSum", C.c_double), ("MskBitCount", C.c_int),
                ("MskS", C.c_uint), ("idx", C.c_uint), ("inb", C.c_float * (2 * FLEN)),
                ("outbits", C.c_ubyte), ("nbits", C.c_int), ("Acarsstate", C.c_int),
                ("blk_len", C.c_int), ("blk_err", C.c_int),
                ("blk_txt", C.c_ubyte * (TXTMAX + 6)), ("blk_crc", C.c_ubyte * 2),
                ("nbit_total", C.c_longlong), ("nsamp_total", C.c_longlong), ("cur_sample", C.c_longlong), ("soh_sample", C.c_longlong),
                ("bitlog", C.POINTER(OrcBit)), ("bitlog_cap", C.c_size_t), ("bitlog_n", C.c_size_t),
                ("frames", C.POINTER(OrcFrame)), ("frames_cap", C.c_size_t), ("frames_n", C.c_size_t)]


_orc = None


def lib():
    global _orc
    if _orc is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_chan_init.argtypes = [C.POINTER(OrcChan), C.c_int]
        L.orc_msk_h.argtypes = [C.c_void_p]
        L.orc_demod_msk.argtypes = [C.POINTER(OrcChan), C.c_void_p, C.c_int]
        L.orc_rtl_taps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_choose_fc.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        L.orc_choose_fc.restype = C.c_int
        L.orc_fir_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_frame_check.argtypes = [C.POINTER(OrcFrame)]
        L.orc_frame_check.restype = C.c_int
        L.orc_crc_update.argtypes = [C.c_ushort, C.c_ubyte]
        L.orc_crc_update.restype = C.c_ushort
        L.orc_soapy_taps.argtypes = [C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orc_fir_cs16.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sdrplay_taps.argtypes = [C.c_float, C.c_uint, C.c_void_p]
        L.orc_fir_split16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_air_choose_fc.argtypes = [C.c_uint, C.c_uint]
        L.orc_air_choose_fc.restype = C.c_uint
        L.orc_air_taps.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p]
        L.orc_fir_f32r.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_syndrome_table.argtypes = [C.c_void_p, C.c_int]
        L.orc_blk_process.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcFrame)]
        L.orc_blk_process.restype = C.c_int
        _orc = L
    return _orc


def frame_tuple(f):
    """Hashable, comparable view of a frame: (chn, len, err, crc bytes, txt bytes)."""
    return (int(f.chn), int(f.len), int(f.err), bytes(f.crc), bytes(f.txt[: max(0, f.len)]))


class Channel:
    """One oracle channel with bit and frame sinks."""

    def __init__(self, chn=0, max_bits=0, max_frames=256):
        self.c = OrcChan()
        self._bits = (OrcBit * max_bits)() if max_bits else None
        self._frames = (OrcFrame * max_frames)()
        if max_bits:
            self.c.bitlog = C.cast(self._bits, C.POINTER(OrcBit))
            self.c.bitlog_cap = max_bits
        self.c.frames = C.cast(self._frames, C.POINTER(OrcFrame))
        self.c.frames_cap = max_frames
        lib().orc_chan_init(C.byref(self.c), chn)

    def demod(self, dm):
        dm = np.ascontiguousarray(dm, dtype=np.float32)
        lib().orc_demod_msk(C.byref(self.c), dm.ctypes.data, int(dm.size))

    @property
    def frames(self):
        return [self._frames[i] for i in range(min(self.c.frames_n, self.c.frames_cap))]

    @property
    def bits(self):
        n = min(self.c.bitlog_n, self.c.bitlog_cap)
        a = np.frombuffer(self._bits, dtype=np.float32, count=2 * n).reshape(n, 2)
        return a[:, 0].copy(), a[:, 1].copy()

    def state(self):
        c = self.c
        return dict(MskPhi=c.MskPhi, MskDf=c.MskDf, MskClk=c.MskClk, MskLvlSum=c.MskLvlSum,
                    MskBitCount=c.MskBitCount, MskS=c.MskS, idx=c.idx,
                    inb=np.array(c.inb[:], dtype=np.float32), outbits=c.outbits, nbits=c.nbits,
                    Acarsstate=c.Acarsstate)


def msk_h():
    h = np.zeros(133, dtype=np.float32)
    lib().orc_msk_h(h.ctypes.data)
    return h


def rtl_taps(Fr, Fc, M):
    wf = np.zeros((M, 2), dtype=np.float32)
    lib().orc_rtl_taps(int(Fr), int(Fc), int(M), wf.ctypes.data)
    return wf


def choose_fc(freqs_hz, M):
    fd = np.array(freqs_hz, dtype=np.uint32)
    return lib().orc_choose_fc(fd.ctypes.data, len(fd), INTRATE * M)


def fir_u8(iq, M, wf, nout=None, ntaps=None):
    iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    if ntaps is None:
        ntaps = wf.shape[0]
    if nout is None:
        nout = iq.size // (2 * M)
    dm = np.zeros(nout, dtype=np.float32)
    lib().orc_fir_u8(iq.ctypes.data, nout, M, ntaps, wf.ctypes.data, dm.ctypes.data)
    return dm


def soapy_taps(Fr, freq, M):
    osc = np.zeros((M, 2), dtype=np.float32)
    lib().orc_soapy_taps(float(Fr), int(freq), int(M), osc.ctypes.data)
    return osc


def fir_cs16(iq, M, osc, nout=None):
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
    osc = np.ascontiguousarray(osc, dtype=np.float32)
    if nout is None:
        nout = iq.size // (2 * M)
    dm = np.zeros(nout, dtype=np.float32)
    lib().orc_fir_cs16(iq.ctypes.data, nout, M, osc.ctypes.data, dm.ctypes.data)
    return dm


def sdrplay_taps(Fr, Fc):
    osc = np.zeros((160, 2), dtype=np.float32)
    lib().orc_sdrplay_taps(float(Fr), int(Fc), osc.ctypes.data)
    return osc


def fir_split16(xi, xq, M, osc, nout=None):
    xi = np.ascontiguousarray(xi, dtype=np.int16).reshape(-1)
    xq = np.ascontiguousarray(xq, dtype=np.int16).reshape(-1)
    osc = np.ascontiguousarray(osc, dtype=np.float32)
    if nout is None:
        nout = xi.size // M
    dm = np.zeros(nout, dtype=np.float32)
    lib().orc_fir_split16(xi.ctypes.data, xq.ctypes.data, nout, M, osc.ctypes.data, dm.ctypes.data)
    return dm


def air_choose_fc(freqs_hz):
    return lib().orc_air_choose_fc(int(min(freqs_hz)), int(max(freqs_hz)))


def air_taps(Fr, Fc, inrate):
    wf = np.zeros((inrate // INTRATE, 2), dtype=np.float32)
    lib().orc_air_taps(int(Fr), int(Fc), int(inrate), wf.ctypes.data)
    return wf


def fir_f32r(x, M, wf, nout=None):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    wf = np.ascontiguousarray(wf, dtype=np.float32)
    if nout is None:
        nout = x.size // M
    dm = np.zeros(nout, dtype=np.float32)
    lib().orc_fir_f32r(x.ctypes.data, nout, M, wf.ctypes.data, dm.ctypes.data)
    return dm


def syndrome_table(n=1936):
    t = np.zeros(n, dtype=np.uint16)
    lib().orc_syndrome_table(t.ctypes.data, n)
    return t


def blk_process(frame):
    """acars.c:123-207 on one raw block: returns the processed OrcFrame or None if dropped."""
    out = OrcFrame()
    return out if lib().orc_blk_process(C.byref(frame), C.byref(out)) else None


class OrcMsg(C.Structure):
    _fields_ = [("chn", C.c_int), ("err", C.c_int), ("lvl", C.c_float), ("txt_len", C.c_int), ("mode", C.c_char),
                ("addr", C.c_char * 8), ("ack", C.c_char), ("label", C.c_char * 3), ("bid", C.c_char), ("no", C.c_char * 5),
                ("fid", C.c_char * 7), ("bs", C.c_char), ("be", C.c_char), ("down", C.c_char), ("txt", C.c_ubyte * 256)]


def msg_split(frame):
    """output.c:486-560 on a processed block (blk_process output): the fields every sink formats."""
    out = OrcMsg()
    lib().orc_msg_split(C.byref(frame), C.byref(out))
    return out


def msg_tuple(m):
    """comparable view of a split message (OrcMsg or the library's acg_msg)"""
    return (int(m.chn), int(m.err), bytes(m.mode), bytes(m.addr), bytes(m.ack), bytes(m.label), bytes(m.bid), bytes(m.no),
            bytes(m.fid), bytes(m.bs), bytes(m.be), m.down not in (b"\x00", 0), bytes(m.txt[: m.txt_len]))


def crc_ccitt(data, crc=0):
    for b in bytes(data):
        crc = lib().orc_crc_update(crc, b)
    return crc


# --------------------------------------------------------------------------- real reference
class RefFrame(C.Structure):
    _fields_ = [("chn", C.c_int), ("len", C.c_int), ("err", C.c_int), ("lvl", C.c_float),
                ("crc", C.c_ubyte * 2), ("txt", C.c_ubyte * TXTMAX)]


class RefBit(C.Structure):
    _fields_ = [("vr", C.c_float), ("vi", C.c_float), ("MskS", C.c_uint), ("chn", C.c_int)]


class RefState(C.Structure):
    _fields_ = [("MskPhi", C.c_double), ("MskDf", C.c_double), ("MskLvlSum", C.c_double),
                ("MskClk", C.c_float), ("MskBitCount", C.c_int), ("MskS", C.c_uint),
                ("idx", C.c_uint), ("inb", C.c_float * 22), ("outbits", C.c_int),
                ("nbits", C.c_int), ("Acarsstate", C.c_int)]


def ref_path(variant=""):
    return os.path.join(HERE, "_ref", "libacarsref%s.so" % variant)


def ref_available(variant=""):
    return os.path.exists(ref_path(variant))


class Ref:
    """The unmodified reference (one instance per process: it is all global state)."""

    def __init__(self, variant=""):
        L = C.CDLL(ref_path(variant))
        if hasattr(L, 'ref_init_rtl'):
            L.ref_init_rtl.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int]
            L.ref_init_rtl.restype = C.c_long
            L.ref_in_callback.argtypes = [C.c_void_p, C.c_uint]
            L.ref_get_wf.argtypes = [C.c_int, C.c_void_p, C.c_int]
            L.ref_get_wf.restype = C.c_int
            if hasattr(L, 'ref_set_wf'):
                L.ref_set_wf.argtypes = [C.c_int, C.c_void_p, C.c_int]
                L.ref_set_wf.restype = C.c_int
        L.ref_init_file.argtypes = [C.c_int]
        L.ref_demod.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.ref_get_dm.argtypes = [C.c_int]
        L.ref_get_dm.restype = C.POINTER(C.c_float)
        L.ref_get_state.argtypes = [C.c_int, C.POINTER(RefState)]
        L.ref_raw.restype = C.POINTER(RefFrame)
        L.ref_out.restype = C.POINTER(RefFrame)
        if hasattr(L, 'ref_init_soapy'):
            L.ref_init_soapy.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int]
            L.ref_init_soapy.restype = C.c_long
            L.ref_soapy_feed.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
            L.ref_get_oscillator.argtypes = [C.c_int, C.c_void_p, C.c_int]
        if hasattr(L, 'ref_init_sdrplay'):
            L.ref_init_sdrplay.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
            L.ref_init_sdrplay.restype = C.c_long
            L.ref_sdrplay_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
            L.ref_get_oscillator.argtypes = [C.c_int, C.c_void_p, C.c_int]
        if hasattr(L, 'ref_init_air'):
            L.ref_init_air.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_uint]
            L.ref_init_air.restype = C.c_long
            L.ref_air_feed.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
            L.ref_get_wf.argtypes = [C.c_int, C.c_void_p, C.c_int]
            L.ref_get_wf.restype = C.c_int
        L.ref_dmlog_enable.argtypes = [C.c_size_t]
        L.ref_dmlog_count.argtypes = [C.c_int]
        L.ref_dmlog_count.restype = C.c_size_t
        L.ref_dmlog.argtypes = [C.c_int]
        L.ref_dmlog.restype = C.POINTER(C.c_float)
        L.ref_bitlog_enable.argtypes = [C.c_size_t]
        L.ref_bitlog_count.restype = C.c_size_t
        L.ref_bitlog.restype = C.POINTER(RefBit)
        self.L = L
        self.M = None

    def init_rtl(self, freqs_mhz, mult):
        """freqs_mhz: strings as on the acarsdec command line. Returns Fc (Hz)."""
        arr = (C.c_char_p * len(freqs_mhz))(*[f.encode() for f in freqs_mhz])
        fc = self.L.ref_init_rtl(len(freqs_mhz), arr, mult)
        if fc <= 0:
            raise RuntimeError("initRtl failed (%d)" % fc)
        self.M = mult
        return fc

    def init_soapy(self, freqs_mhz, mult):
        arr = (C.c_char_p * len(freqs_mhz))(*[f.encode() for f in freqs_mhz])
        fc = self.L.ref_init_soapy(len(freqs_mhz), arr, mult)
        if fc <= 0:
            raise RuntimeError("initSoapy failed (%d)" % fc)
        self.M = mult
        return fc

    def soapy_feed(self, iq, chunk=0):
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        self.L.ref_soapy_feed(iq.ctypes.data, iq.size // 2, chunk)

    def oscillator(self, n):
        out = np.zeros((self.M, 2), dtype=np.float32)
        fr = self.L.ref_get_oscillator(n, out.ctypes.data, self.M)
        return fr, out

    def init_sdrplay(self, freqs_mhz):
        arr = (C.c_char_p * len(freqs_mhz))(*[f.encode() for f in freqs_mhz])
        fc = self.L.ref_init_sdrplay(len(freqs_mhz), arr)
        if fc <= 0:
            raise RuntimeError("initSdrplay failed (%d)" % fc)
        self.M = 160
        return fc

    def sdrplay_feed(self, xi, xq, chunk=0):
        xi = np.ascontiguousarray(xi, dtype=np.int16).reshape(-1)
        xq = np.ascontiguousarray(xq, dtype=np.int16).reshape(-1)
        self.L.ref_sdrplay_feed(xi.ctypes.data, xq.ctypes.data, xi.size, chunk)

    def init_air(self, freqs_mhz, rate):
        arr = (C.c_char_p * len(freqs_mhz))(*[f.encode() for f in freqs_mhz])
        fc = self.L.ref_init_air(len(freqs_mhz), arr, rate)
        if fc <= 0:
            raise RuntimeError("initAirspy failed (%d)" % fc)
        self.M = rate // INTRATE
        return fc

    def air_feed(self, x, chunk=0):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        self.L.ref_air_feed(x.ctypes.data, x.size, chunk)

    def dmlog_enable(self, cap):
        self.L.ref_dmlog_enable(cap)

    def dmlog(self, n):
        k = self.L.ref_dmlog_count(n)
        return np.ctypeslib.as_array(self.L.ref_dmlog(n), shape=(k,)).copy() if k else np.zeros(0, np.float32)

    def init_file(self, nch):
        if self.L.ref_init_file(nch):
            raise RuntimeError("ref_init_file failed")

    def in_callback(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1)
        self.L.ref_in_callback(buf.ctypes.data, buf.size)

    def demod(self, n, dm):
        dm = np.ascontiguousarray(dm, dtype=np.float32)
        assert dm.size <= 4096
        self.L.ref_demod(n, dm.ctypes.data, dm.size)

    def wf(self, n):
        out = np.zeros((self.M, 2), dtype=np.float32)
        fr = self.L.ref_get_wf(n, out.ctypes.data, self.M)
        return fr, out

    def set_wf(self, n, taps):
        """channel n's taps := taps ([ntaps, 2] float32, zero beyond): the reference's in_callback on a caller's tap table"""
        taps = np.ascontiguousarray(taps, dtype=np.float32).reshape(-1, 2)
        if self.L.ref_set_wf(n, taps.ctypes.data, taps.shape[0]) != 0:
            raise RuntimeError("ref_set_wf failed")

    def dm(self, n, count=1024):
        return np.ctypeslib.as_array(self.L.ref_get_dm(n), shape=(count,)).copy()

    def state(self, n):
        s = RefState()
        self.L.ref_get_state(n, C.byref(s))
        return dict(MskPhi=s.MskPhi, MskDf=s.MskDf, MskClk=s.MskClk, MskLvlSum=s.MskLvlSum,
                    MskBitCount=s.MskBitCount, MskS=s.MskS, idx=s.idx,
                    inb=np.array(s.inb[:], dtype=np.float32), outbits=s.outbits, nbits=s.nbits,
                    Acarsstate=s.Acarsstate)

    def drain(self):
        self.L.ref_drain()

    def raw_frames(self):
        p = self.L.ref_raw()
        return [p[i] for i in range(self.L.ref_nraw())]

    def out_frames(self):
        p = self.L.ref_out()
        return [p[i] for i in range(self.L.ref_nout())]

    def bitlog_enable(self, cap):
        self.L.ref_bitlog_enable(cap)

    def bitlog(self):
        n = self.L.ref_bitlog_count()
        p = self.L.ref_bitlog()
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n, 4)).copy()
        v = a[:, :2].copy().view(np.float32)
        return v[:, 0], v[:, 1], a[:, 2].copy(), a[:, 3].view(np.int32).copy()


# --------------------------------------------------------------------------- WAV (PCM16, incl. WAVE_FORMAT_EXTENSIBLE)
def read_wav_pcm16(path):
    """Returns (rate, int16 array [frames, channels]).  Stand-in for libsndfile (soundfile.c:36)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid, sz = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + sz]
        if cid == b"fmt ":
            tag, nch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE:
                tag = struct.unpack("<H", body[24:26])[0]
            assert tag == 1 and bits == 16, (tag, bits)
            fmt = (nch, rate)
        elif cid == b"data":
            pcm = np.frombuffer(body, dtype="<i2")
        pos += 8 + sz + (sz & 1)
    nch, rate = fmt
    return rate, pcm[: (pcm.size // nch) * nch].reshape(-1, nch).copy()


def wav_to_float(pcm16):
    """libsndfile's sf_read_float normalisation for PCM16: x / 32768 (soundfile.c:65)."""
    return (pcm16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


# --------------------------------------------------------------------------- reference builds against each other
def _ref_blocks_child(variant, path_in, path_out):
    """child side of ref_blocks: a FRESH interpreter (the reference is all global state, an -march=native build may hit an
    illegal instruction on this host, and forking a process that holds a GPU runtime can deadlock on a lock whose owner
    thread does not exist in the child -- ADVICE r03)."""
    import pickle
    z = np.load(path_in)
    rows, taps, M, front = z["rows"], z["taps"], int(z["M"]), str(z["front"])
    row_of = z["row_of"]
    os.dup2(os.open(os.devnull, os.O_WRONLY), 2)          # init* narrates on stderr
    blk = 1024 * M * 2
    ref = Ref(variant)
    if front != "rtl":
        getattr(ref.L, "ref_set_oscillator" if front in ("soapy", "sdrplay") else "ref_set_wf").argtypes = [C.c_int, C.c_void_p, C.c_int]
    grab = lambda fr: [(int(f.len), int(f.err), bytes(f.crc), bytes(f.txt[: max(0, f.len)])) for f in fr]
    raw, out = [], []
    for c in range(taps.shape[0]):
        t = np.ascontiguousarray(taps[c], dtype=np.float32)
        r = np.ascontiguousarray(rows[int(row_of[c])]).reshape(-1)
        if front == "rtl":
            ref.init_rtl(["131.725"], M)                  # one channel; its state is re-initialised by every init
            ref.set_wf(0, t)
            for b in range(r.size // blk):
                ref.in_callback(r[b * blk:(b + 1) * blk])
        elif front == "soapy":                            # CS16 through the reader loop of soapy.c:220-254, 1000-sample reads
            ref.init_soapy(["131.725"], M)
            assert ref.L.ref_set_oscillator(0, t.ctypes.data, t.shape[0]) == 0
            ref.soapy_feed(r.view(np.int16), 1000)
        elif front == "sdrplay":                          # int16 I and Q planes through myStreamCallback (sdrplay.c:202-237), ragged callbacks
            ref.init_sdrplay(["131.725"])
            assert ref.L.ref_set_oscillator(0, t.ctypes.data, t.shape[0]) == 0
            h = r.view(np.int16)
            ref.sdrplay_feed(h[: h.size // 2], h[h.size // 2:], 1008)
        else:                                             # real f32 through rx_callback (air.c:291-341), ragged transfers
            ref.init_air(["131.725"], INTRATE * M)
            assert ref.L.ref_set_wf(0, t.ctypes.data, t.shape[0]) == 0
            ref.air_feed(r.view(np.float32), 50000)
        ref.drain()                                       # blk_thread's pass over what decodeAcars queued (acars.c:93-215)
        raw.append(grab(ref.raw_frames()))
        out.append(grab(ref.out_frames()))
    with open(path_out, "wb") as w:
        pickle.dump(dict(raw=raw, out=out), w)


def ref_blocks(variant, rows, M, taps, timeout_s=600, front="rtl", row_of=None):
    """The UNMODIFIED reference build `variant` ("" = -O2 IEEE, "_fast" = the reference's own -Ofast -march=native,
    "_v3"; front="soapy" / "air" / "sdrplay": "_soapy" / "_air" / "_sdrplay" and their "_fast" twins) run over rows[c] (whole
    callbacks of u8 I/Q; CS16 samples; real f32 samples; an int16 I plane followed by the Q plane) with channel c's tap table taps[c], one channel per pass, through its own front end
    (rtl.c in_callback / soapy.c's reader loop / air.c rx_callback) -> demodMSK -> decodeAcars -> blk_thread.  Returns
    {"raw": [...], "out": [...]}: per channel the blocks as decodeAcars queued them and as outputmsg() received them, each
    (len, err, crc, txt) -- or None when the build is missing or cannot run on this host.  Runs in a child interpreter; rows
    and taps travel through a temporary file.  row_of[c] = the row channel c reads (default c: one row per channel; several
    channels of one dongle name the same row)."""
    import pickle
    import sys
    import tempfile
    if not ref_available(variant):
        return None
    with tempfile.TemporaryDirectory() as td:
        pin, pout = os.path.join(td, "in.npz"), os.path.join(td, "out.pkl")
        tp = np.zeros((len(taps), M, 2), dtype=np.float32)     # (a shorter table is zero-filled, as ref_set_wf does anyway)
        for c, t in enumerate(taps):
            t = np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 2)
            tp[c, : t.shape[0]] = t
        np.savez(pin, rows=np.stack([np.ascontiguousarray(r).reshape(-1).view(np.uint8) for r in rows]), taps=tp, M=M, front=front,
                 row_of=np.arange(len(taps)) if row_of is None else np.asarray(row_of, dtype=np.int64))
        code = "import sys; sys.path.insert(0, %r); from oracle import oracle as O; O._ref_blocks_child(%r, %r, %r)" % (
            os.path.dirname(HERE), variant, pin, pout)
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None
        if r.returncode != 0 or not os.path.exists(pout):
            return None
        with open(pout, "rb") as f:
            return pickle.load(f)


def ref_blocks_forked(variant, rows, M, taps, timeout_s=600):
    """raw blocks only (the name is historical: the child is a fresh interpreter now, see ref_blocks)"""
    d = ref_blocks(variant, rows, M, taps, timeout_s)
    return None if d is None else d["raw"]
