"""Host-side mirror of the reference's per-channel interface on top of the C ABI.

Names follow the reference: a Decoder owns `nch` channels (channel_t, acarsdec.h:59-92);
`init_rtl` is initRtl's tap set-up (rtl.c:243-287), `in_callback` is rtl.c:314-361 for all
channels, `demod_msk` is demodMSK (msk.c:67-137) fed from 12.5 kHz samples, frames are the
msgblk_t blocks decodeAcars queues (acars.c:350-364).  Everything numerical happens in
libacarsdec_amd.so on the GPU; this file only marshals pointers.
"""
import ctypes as C

import numpy as np

from . import _capi as K


def _chk(ctx, rc, allow=(), lib=None):
    """lib = the library that owns ctx (a lab-build context must not be handed to the product library's acg_last_error)"""
    if rc != K.OK and rc not in allow:
        msg = (lib or K.load()).acg_last_error(ctx).decode() if ctx else ""
        raise K.AcgError(rc, msg)
    return rc


def rtl_taps(Fr_hz, Fc_hz, decim):
    """wf[] of rtl.c:283-286 as float32 [decim, 2]."""
    out = np.zeros((decim, 2), dtype=np.float32)
    _chk(None, K.load().acg_rtl_taps(int(Fr_hz), int(Fc_hz), int(decim), out.ctypes.data))
    return out


def choose_fc(freqs_hz, decim):
    """chooseFc of rtl.c:131-168. Returns (Fc, sorted freqs)."""
    fd = np.array(freqs_hz, dtype=np.uint32)
    fc = K.load().acg_rtl_choose_fc(fd.ctypes.data, len(fd), int(decim))
    return int(fc), fd


def soapy_taps(Fr_hz, freq_hz, decim):
    """oscillator[] of soapy.c:163-166 as float32 [decim, 2]."""
    out = np.zeros((decim, 2), dtype=np.float32)
    _chk(None, K.load().acg_soapy_taps(float(Fr_hz), int(freq_hz), int(decim), out.ctypes.data))
    return out


def sdrplay_taps(Fr_hz, Fc_hz):
    out = np.zeros((160, 2), dtype=np.float32)
    _chk(None, K.load().acg_sdrplay_taps(float(Fr_hz), int(Fc_hz), out.ctypes.data))
    return out


def airspy_taps(Fr_hz, Fc_hz, inrate):
    out = np.zeros((inrate // K.INTRATE, 2), dtype=np.float32)
    _chk(None, K.load().acg_airspy_taps(int(Fr_hz), int(Fc_hz), int(inrate), out.ctypes.data))
    return out


def airspy_choose_fc(freqs_hz):
    return int(K.load().acg_airspy_choose_fc(int(min(freqs_hz)), int(max(freqs_hz))))


def parse_freq_mhz(s):
    """rtl.c:245-247: command-line MHz string -> Hz rounded to the 12.5 kHz raster."""
    return (int(1000000 * float(s) + K.INTRATE / 2) // K.INTRATE) * K.INTRATE


class Decoder:
    def __init__(self, nch, decim=160, ntaps=None, nstreams=None, max_blocks=1, device=0,
                 bitlog=True, timing=False, repair=False, exact_fir=False, max_lag=0, lab=False, precise_mixer=False):
        self.L = K.load(lab)          # lab=True: the lab build of the library (measurement variants; tests and probes only)
        self.nch, self.decim = int(nch), int(decim)
        self.ntaps = int(ntaps if ntaps is not None else decim)
        self.nstreams = int(nstreams if nstreams is not None else nch)
        self.max_blocks = int(max_blocks)
        cfg = K.Config(device, self.nch, self.nstreams, self.decim, self.ntaps, self.max_blocks,
                       (K.F_BITLOG if bitlog else 0) | (K.F_TIMING if timing else 0) | (K.F_REPAIR if repair else 0) |
                       (K.F_EXACT_FIR if exact_fir else 0) | (K.F_PRECISE_MIXER if precise_mixer else 0), int(max_lag))
        self.ctx = C.c_void_p()
        rc = self.L.acg_create(C.byref(self.ctx), C.byref(cfg))
        if rc != K.OK:
            raise K.AcgError(rc, "acg_create")
        self.timing_flag = bool(timing)
        self.bit_cap = self.L.acg_bit_capacity(self.ctx)
        self.max_lag = self.L.acg_max_lag(self.ctx)
        self.Fc = None

    def _chk(self, rc, allow=()):
        return _chk(self.ctx, rc, allow, self.L)

    def close(self):
        if self.ctx:
            self.L.acg_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up -------------------------------------------------------------------------
    def reset(self):
        self._chk(self.L.acg_reset(self.ctx))

    def set_taps(self, taps, ch0=0):
        taps = np.ascontiguousarray(taps, dtype=np.float32).reshape(-1, self.ntaps, 2)
        self._chk(self.L.acg_set_taps(self.ctx, ch0, taps.shape[0], taps.ctypes.data))

    def set_channel_streams(self, stream_of):
        a = np.ascontiguousarray(stream_of, dtype=np.int32)
        assert a.size == self.nch
        self._chk(self.L.acg_set_channel_streams(self.ctx, a.ctypes.data))

    def init_rtl(self, freqs_mhz):
        """One dongle, nch channels sharing its stream: rtl.c:243-287.  Returns Fc."""
        assert len(freqs_mhz) == self.nch
        fr = [parse_freq_mhz(f) for f in freqs_mhz]
        fc, _ = choose_fc(fr, self.decim)
        if fc == 0:
            raise ValueError("Frequencies too far apart")          # rtl.c:150
        self.set_taps(np.stack([rtl_taps(f, fc, self.decim) for f in fr]))
        self.Fc = fc
        return fc

    # ---- hot path -----------------------------------------------------------------------
    @staticmethod
    def _ptr(x):
        """(pointer, is_device) of a numpy array or a torch tensor."""
        if hasattr(x, "data_ptr"):
            return x.data_ptr(), bool(x.is_cuda)
        return x.ctypes.data, False

    def in_callback(self, iq, nblocks=None, pitch=None, stream=None):
        """in_callback() for all channels.  iq: uint8, [nstreams, nblocks*1024*decim*2] (numpy = host,
        torch cuda tensor = device, asynchronous on `stream`)."""
        row = K.BLOCK * self.decim * 2
        if hasattr(iq, "data_ptr"):
            nbytes_row = iq.shape[-1] if iq.dim() > 1 else iq.numel() // self.nstreams
            pitch = pitch or (iq.stride(0) if iq.dim() > 1 else nbytes_row)
        else:
            iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(self.nstreams, -1)
            nbytes_row = iq.shape[1]
            pitch = pitch or nbytes_row
        nblocks = nblocks or nbytes_row // row
        p, dev = self._ptr(iq)
        if dev:
            self._chk(self.L.acg_process_iq_u8_dev(self.ctx, p, pitch, nblocks, stream))
        else:
            self._chk(self.L.acg_process_iq_u8_host(self.ctx, p, pitch, nblocks))

    def feed(self, fmt, samples, q_plane=None):
        """Host samples of any length (the SDR drivers' shape): [nstreams, n] int16 pairs / planes / float32."""
        if fmt == K.FMT_CS16:
            a = np.ascontiguousarray(samples, dtype=np.int16).reshape(self.nstreams, -1)
            n, pitch = a.shape[1] // 2, a.shape[1] // 2
        elif fmt == K.FMT_S16_SPLIT:
            a = np.ascontiguousarray(samples, dtype=np.int16).reshape(self.nstreams, -1)
            q_plane = np.ascontiguousarray(q_plane, dtype=np.int16).reshape(self.nstreams, -1)
            n, pitch = a.shape[1], a.shape[1]
        else:
            a = np.ascontiguousarray(samples, dtype=np.float32).reshape(self.nstreams, -1)
            n, pitch = a.shape[1], a.shape[1]
        self._chk(self.L.acg_feed_samples_host(self.ctx, fmt, a.ctypes.data,
                                                     q_plane.ctypes.data if q_plane is not None else None, pitch, n))

    def process_samples(self, fmt, dev_tensor, nblocks, pitch, plane=0, stream=None):
        self._chk(self.L.acg_process_samples_dev(self.ctx, fmt, dev_tensor.data_ptr(), pitch, plane, nblocks, stream))

    def fir_only(self, iq_dev, nblocks, pitch, stream=None):
        self._chk(self.L.acg_fir_only_dev(self.ctx, iq_dev.data_ptr(), pitch, nblocks, stream))

    def placement_trial(self, iq_dev, nblocks, pitch, repeats=2, stream=None):
        """ms per in_callback-sized call on this decoder's placement (acg_placement_trial); the decoder comes back reset."""
        ms = C.c_double(0)
        self._chk(self.L.acg_placement_trial(self.ctx, iq_dev.data_ptr(), pitch, nblocks, repeats, stream, C.byref(ms)))
        return ms.value

    def placement_trial_samples(self, fmt, dev_tensor, nblocks, pitch, plane=0, repeats=2, stream=None):
        """the same for a sample-format input (acg_placement_trial_samples)"""
        ms = C.c_double(0)
        self._chk(self.L.acg_placement_trial_samples(self.ctx, fmt, dev_tensor.data_ptr(), pitch, plane, nblocks, repeats, stream, C.byref(ms)))
        return ms.value

    def demod_msk(self, dm):
        """demodMSK() for all channels from 12.5 kHz samples: dm float32 [nch, len]."""
        dm = np.ascontiguousarray(dm, dtype=np.float32)
        dm = dm.reshape(self.nch, dm.size // self.nch)
        self._chk(self.L.acg_process_dm_host(self.ctx, dm.ctypes.data, dm.shape[1], dm.shape[1]))

    def sync(self):
        self._chk(self.L.acg_sync(self.ctx))

    # ---- results ------------------------------------------------------------------------
    def drain_frames(self, max_frames=4096):
        """Blocks completed since the last drain as a list of ctypes Frame objects (loops while the C side says ACG_EAGAIN)."""
        out = []
        while True:
            n, buf, more = self._frames_call(self.L.acg_drain_frames, max_frames)
            out += [K.Frame.from_buffer_copy(buf[i]) for i in range(n)]      # copies: the array is reused
            if not more:
                return out

    def _frames_call(self, fn, max_frames, *lead):
        max_frames = max(1, int(max_frames))          # (a zero-length buffer would make every EAGAIN loop spin without progress)
        if getattr(self, "_fbuf_cap", 0) < max_frames:
            self._fbuf = (K.Frame * max_frames)()
            self._fbuf_cap = max_frames
        n = C.c_int(0)
        rc = self._chk(fn(self.ctx, *lead, self._fbuf, self._fbuf_cap, C.byref(n)), allow=(K.EAGAIN,))
        return n.value, self._fbuf, rc == K.EAGAIN

    def drain_frames_raw(self, max_frames=4096):
        """(count, ctypes array): no per-block Python objects; the array is reused by the next call.  Blocks beyond
        max_frames stay queued (the next drain / collect hands them out)."""
        n, buf, _ = self._frames_call(self.L.acg_drain_frames, max_frames)
        return n, buf

    def collect_frames_raw(self, lag=1, max_frames=4096):
        """Streaming drain: blocks of all calls but the `lag` newest; waits only for those calls."""
        n, buf, _ = self._frames_call(self.L.acg_collect_frames, max_frames, lag)
        return n, buf

    def _msg_buf(self, max_msgs):
        if getattr(self, "_mbuf_cap", 0) < max_msgs:
            self._mbuf = (K.Msg * max_msgs)()
            self._mbuf_cap = max_msgs
        return self._mbuf

    def drain_msgs(self, max_msgs=4096):
        """outputmsg()'s field split of every block completed since the last drain (needs repair=True): K.Msg records.
        The C side hands out the oldest messages that fit and says ACG_EAGAIN ("call again, nothing lost") while more are
        queued: this wrapper calls again until the queue is empty, so the list is complete whatever max_msgs is (per call the
        records are ordered by (chn, end_bit); across calls the order is completion order)."""
        buf = self._msg_buf(max(1, max_msgs))
        out = []
        while True:
            n = C.c_int(0)
            rc = self._chk(self.L.acg_drain_msgs(self.ctx, buf, self._mbuf_cap, C.byref(n)), allow=(K.EAGAIN,))
            out += [K.Msg.from_buffer_copy(buf[i]) for i in range(n.value)]
            if rc == K.OK:
                return out

    def drain_msgs_raw(self, max_msgs=4096):
        """(count, ctypes array, more): one acg_drain_msgs call, no per-message Python objects"""
        buf = self._msg_buf(max(1, max_msgs))
        n = C.c_int(0)
        rc = self._chk(self.L.acg_drain_msgs(self.ctx, buf, self._mbuf_cap, C.byref(n)), allow=(K.EAGAIN,))
        return n.value, buf, rc == K.EAGAIN

    def collect_msgs_raw(self, lag=1, max_msgs=4096):
        """Streaming variant (acg_collect_msgs): (count, ctypes array, more) -- messages of all calls but the `lag` newest; `more`
        is True when the buffer was too small and the rest stays queued for the next collect (nothing is lost)."""
        buf = self._msg_buf(max(1, max_msgs))
        n = C.c_int(0)
        rc = self._chk(self.L.acg_collect_msgs(self.ctx, lag, buf, self._mbuf_cap, C.byref(n)), allow=(K.EAGAIN,))
        return n.value, buf, rc == K.EAGAIN

    def collect_msgs(self, lag=1, max_msgs=4096):
        """list of K.Msg of all calls but the `lag` newest (loops while the C side says "call again")"""
        out = []
        while True:
            n, buf, more = self.collect_msgs_raw(lag, max_msgs)
            out += [K.Msg.from_buffer_copy(buf[i]) for i in range(n)]
            if not more:
                return out

    def bits(self, ch):
        vo = np.zeros(self.bit_cap, dtype=np.float32)
        lvl = np.zeros(self.bit_cap, dtype=np.float32)
        n = C.c_int(0)
        self._chk(self.L.acg_read_bits(self.ctx, ch, vo.ctypes.data, lvl.ctypes.data, self.bit_cap, C.byref(n)))
        return vo[: n.value].copy(), lvl[: n.value].copy()

    def bits_all(self):
        counts = np.zeros(self.nch, dtype=np.int32)
        vo = np.zeros((self.nch, self.bit_cap), dtype=np.float32)
        lvl = np.zeros((self.nch, self.bit_cap), dtype=np.float32)
        self._chk(self.L.acg_read_bits_all(self.ctx, counts.ctypes.data, vo.ctypes.data, lvl.ctypes.data))
        return counts, vo, lvl

    def dm(self, ch, n):
        out = np.zeros(n, dtype=np.float32)
        self._chk(self.L.acg_read_dm(self.ctx, ch, out.ctypes.data, n))
        return out

    def state(self, ch):
        s = K.ChanState()
        self._chk(self.L.acg_get_state(self.ctx, ch, C.byref(s)))
        return dict(MskPhi=s.MskPhi, MskDf=s.MskDf, MskClk=s.MskClk, MskLvlSum=s.MskLvlSum,
                    MskBitCount=s.MskBitCount, MskS=s.MskS, idx=s.idx,
                    inb=np.array(s.inb[:], dtype=np.float32), outbits=s.outbits, nbits=s.nbits,
                    Acarsstate=s.Acarsstate, blk_len=s.blk_len, blk_err=s.blk_err)

    def set_timing(self, mode):
        """0 = off, 1 = both stages, 2 = down-converter launches only."""
        self._chk(self.L.acg_set_timing(self.ctx, int(mode)))

    def timing(self):
        f, m = C.c_double(0), C.c_double(0)
        nf, nm = C.c_int(0), C.c_int(0)
        self._chk(self.L.acg_get_timing(self.ctx, C.byref(f), C.byref(nf), C.byref(m), C.byref(nm)))
        return dict(fir_ms=f.value, fir_launches=nf.value, msk_ms=m.value, msk_launches=nm.value)


def frame_tuple(f):
    """(chn, len, err, crc, txt) -- the bit-exact part of a block."""
    return (int(f.chn), int(f.len), int(f.err), bytes(f.crc), bytes(f.txt[: max(0, f.len)]))


def best_placed(factory, n, iq_dev, nblocks, pitch, repeats=2, stream=None, fmt=0, plane=0, keep="best"):
    """Creates `n` decoders with factory() -- all alive during the trials, so that their buffers lie in different places --
    runs the same call on each once for nothing, then times it (Decoder.placement_trial) and keeps the fastest (keep="best")
    or the first (keep="first": the trial is then a diagnostic of how much the contexts of a process differ).  Returns
    (decoder, [ms per call], index); best_placed.last_fir_ms holds the down-converter tie-break figures where it was used."""
    decs = [factory() for _ in range(max(1, n))]
    if len(decs) == 1:
        return decs[0], [], 0
    def trial(d):
        return (d.placement_trial(iq_dev, nblocks, pitch, repeats, stream) if fmt == 0 else
                d.placement_trial_samples(fmt, iq_dev, nblocks, pitch, plane, repeats, stream))
    for d in decs:                      # a round for nothing: the first context timed is otherwise timed on a cold device (clocks,
        trial(d)                        # first touch of its buffers) and loses by 10-15 % to that alone (round 3, queue_probe.sh)
    ms = [trial(d) for d in decs]
    best = min(range(len(decs)), key=lambda i: ms[i])
    # Up to 2048 channels (CU partition) the demodulator sets the call, so the calls of all contexts are nearly alike while their
    # down-converters differ by a few per cent: among the contexts within 1 % of the fastest call keep the one whose
    # down-converter launches (event-timed) are the shortest.
    best_placed.last_fir_ms = None
    if fmt == 0 and all(getattr(d, "timing_flag", False) and d.nch <= 2048 for d in decs):
        fir = []
        for d in decs:
            d.set_timing(2)
            d.timing()
            for _ in range(3):
                d.in_callback(iq_dev, nblocks=nblocks, pitch=pitch, stream=stream)
            t = d.timing()
            fir.append(t["fir_ms"] / max(1, t["fir_launches"]))
            d.reset()
            d.set_timing(1)
        near = [i for i in range(len(decs)) if ms[i] <= 1.01 * min(ms)]
        best = min(near, key=lambda i: fir[i])
        best_placed.last_fir_ms = fir
    if keep == "first":
        best = 0
    for i, d in enumerate(decs):
        if i != best:
            d.close()
    return decs[best], ms, best
