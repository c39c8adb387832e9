"""ctypes binding of libacarsdec_amd.so (include/acarsdec_amd.h) -- the same stub a cffi/ctypes
host would write against the C ABI.  No compute happens in Python."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ACARSDEC_AMD_LIB: measurement aid (A/B timing of two builds on one box, the stamp build); the product is the in-tree path
LIB_PATH = os.environ.get("ACARSDEC_AMD_LIB") or os.path.join(HERE, "lib", "libacarsdec_amd.so")
# the lab build: every measurement kernel variant and debug shape; tests and probes only (Decoder(lab=True), tune(..., lab=True))
LAB_PATH = os.path.join(HERE, "lib", "libacarsdec_amd_lab.so")

OK, EINVAL, ENOMEM, EHIP, ENODEV, EOVERFLOW, ESTATE, EAGAIN = 0, -1, -2, -3, -4, -5, -6, -7
F_BITLOG, F_TIMING, F_REPAIR, F_EXACT_FIR, F_PRECISE_MIXER = 1, 2, 4, 8, 16
INTRATE, BLOCK, MAXDECIM, FLEN, TXTMAX = 12500, 1024, 320, 11, 250
MAXDECIM_SAMPLES = 1024
FMT_CS16, FMT_S16_SPLIT, FMT_F32_REAL = 1, 2, 3


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("nch", C.c_int), ("nstreams", C.c_int), ("decim", C.c_int),
                ("ntaps", C.c_int), ("max_blocks", C.c_int), ("flags", C.c_uint32), ("max_lag", C.c_int)]


class ChanState(C.Structure):
    _fields_ = [("MskPhi", C.c_double), ("MskDf", C.c_double), ("MskLvlSum", C.c_double),
                ("MskClk", C.c_float), ("MskBitCount", C.c_int), ("MskS", C.c_uint), ("idx", C.c_uint),
                ("inb", C.c_float * (2 * FLEN)), ("outbits", C.c_int), ("nbits", C.c_int),
                ("Acarsstate", C.c_int), ("blk_len", C.c_int), ("blk_err", C.c_int), ("soh_back", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("chn", C.c_int), ("len", C.c_int), ("err", C.c_int), ("lvl", C.c_float),
                ("crc", C.c_ubyte * 2), ("txt", C.c_ubyte * TXTMAX),
                ("end_bit", C.c_longlong), ("end_sample", C.c_longlong), ("soh_sample", C.c_longlong)]


class Msg(C.Structure):
    """acg_msg: outputmsg()'s field split as a fixed binary record"""
    _fields_ = [("chn", C.c_int), ("err", C.c_int), ("lvl", C.c_float), ("txt_len", C.c_int),
                ("end_bit", C.c_longlong), ("end_sample", C.c_longlong), ("soh_sample", C.c_longlong), ("reserved1", C.c_int),
                ("reserved2", C.c_char), ("mode", C.c_char), ("addr", C.c_char * 8), ("ack", C.c_char), ("label", C.c_char * 3),
                ("bid", C.c_char), ("no", C.c_char * 5), ("fid", C.c_char * 7), ("bs", C.c_char), ("be", C.c_char),
                ("down", C.c_char), ("txt", C.c_ubyte * 242), ("reserved3", C.c_int)]


BIT_SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float, C.c_float)

# name -> (restype, argtypes): every symbol include/acarsdec_amd.h (the product API) and include/acarsdec_amd_lab.h (measurement
# and diagnostics: LAB_SYMBOLS below) declare; the library exports exactly these (tests/test_host_logic.py)
SYMBOLS = {
    "acg_device_count": (C.c_int, []),
    "acg_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config)]),
    "acg_destroy": (None, [C.c_void_p]),
    "acg_strerror": (C.c_char_p, [C.c_int]),
    "acg_last_error": (C.c_char_p, [C.c_void_p]),
    "acg_version": (C.c_char_p, []),
    "acg_rtl_choose_fc": (C.c_uint, [C.c_void_p, C.c_uint, C.c_int]),
    "acg_rtl_taps": (C.c_int, [C.c_int, C.c_uint, C.c_int, C.c_void_p]),
    "acg_set_taps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "acg_set_channel_streams": (C.c_int, [C.c_void_p, C.c_void_p]),
    "acg_reset": (C.c_int, [C.c_void_p]),
    "acg_process_iq_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "acg_process_iq_u8_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "acg_process_dm_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "acg_process_dm_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "acg_fir_only_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "acg_sync": (C.c_int, [C.c_void_p]),
    "acg_soapy_taps": (C.c_int, [C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "acg_sdrplay_taps": (C.c_int, [C.c_float, C.c_uint, C.c_void_p]),
    "acg_airspy_choose_fc": (C.c_uint, [C.c_uint, C.c_uint]),
    "acg_airspy_taps": (C.c_int, [C.c_int, C.c_int, C.c_uint, C.c_void_p]),
    "acg_process_samples_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    "acg_feed_samples_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "acg_host_alloc": (C.c_void_p, [C.c_size_t]),
    "acg_host_free": (None, [C.c_void_p]),
    "acg_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "acg_host_unregister": (C.c_int, [C.c_void_p]),
    "acg_drain_frames": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_int, C.POINTER(C.c_int)]),
    "acg_collect_frames": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Frame), C.c_int, C.POINTER(C.c_int)]),
    "acg_drain_msgs": (C.c_int, [C.c_void_p, C.POINTER(Msg), C.c_int, C.POINTER(C.c_int)]),
    "acg_collect_msgs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Msg), C.c_int, C.POINTER(C.c_int)]),
    "acg_read_bits": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "acg_read_bits_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acg_bit_capacity": (C.c_int, [C.c_void_p]),
    "acg_max_lag": (C.c_int, [C.c_void_p]),
    "acg_read_dm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "acg_get_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ChanState)]),
    "acg_set_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ChanState)]),
    "acg_get_state_n": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(ChanState)]),
    "acg_set_state_n": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(ChanState)]),
    "acg_get_block_text": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "acg_set_block_text": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "acg_read_dm_n": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]),
    "acg_replay_bits": (C.c_int, [C.c_void_p, BIT_SINK, C.c_void_p]),
    "acg_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                 C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "acg_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
}

# include/acarsdec_amd_lab.h: measurement / diagnostics (exported by the product library too: bench.py times the product)
LAB_SYMBOLS = {
    "acg_placement_trial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]),
    "acg_placement_trial_samples": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                              C.POINTER(C.c_double)]),
    "acg_tune": (C.c_int, [C.c_char_p, C.c_char_p]),
    "acg_is_lab_build": (C.c_int, []),
    "acg_fill_random_u8_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_uint64, C.c_void_p]),
    "acg_synth_iq_u8_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_void_p]),
    "acg_probe_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "acg_probe_read_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "acg_selftest_div2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "acg_selftest_sincos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "acg_lab_set_block_counter": (C.c_int, [C.c_void_p, C.c_uint]),
    "acg_lab_block_ring_size": (C.c_uint, [C.c_void_p]),
}
# declared in acarsdec_amd_lab.h, present in the stamp build only (-DACG_MSK_STAMP)
STAMP_SYMBOLS = ("acg_msk_stamp_read", "acg_msk_lanes_per_channel")

_lib = None
_lab = None


def _open(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    try:
        # PyTorch bundles its own libamdhip64.so.7; importing it first makes this library bind
        # to the SAME HIP runtime instance so torch device pointers / streams are usable here.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not required by the library
        pass
    L = C.CDLL(path)
    for name, (res, args) in list(SYMBOLS.items()) + list(LAB_SYMBOLS.items()):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def load(lab=False):
    """Load the native library.  Raises (never falls back) if it is missing.  lab=True: the lab build (a second, independent
    library with its own tuning table), for tests and probes that need a measurement variant."""
    global _lib, _lab
    if lab:
        if _lab is None:
            _lab = _open(LAB_PATH)
            assert _lab.acg_is_lab_build() == 1
        return _lab
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def tune(name, value, lab=False):
    """Measurement switch NAME (ACG_...) := value for the following launches; None removes the override.  The library
    reads the environment only once, at its first look-up, so changing os.environ later has no effect: use this."""
    rc = load(lab).acg_tune(name.encode(), None if value is None else str(value).encode())
    if rc != OK:
        raise AcgError(rc, "acg_tune(%s)" % name)


class AcgError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("acarsdec_amd error %d: %s %s" % (code, load().acg_strerror(code).decode(), msg))
