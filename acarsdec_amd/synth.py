"""Host-side synthetic signal sources (numpy) for tests and the bench.

The reference has no signal generator; these produce inputs shaped like what its front ends
consume: 12.5 kHz real envelopes (soundfile.c:58-81) and interleaved u8 I/Q at INTRATE*rtlMult
(rtl.c:314-342).  The ACARS/MSK modulator is derived from the demodulator's conventions
(msk.c:115-127, acars.c:22-27,138,159-165,303-341): bytes LSB first, odd parity in bit 7,
2400 Hz when a bit equals the previous one and 1200 Hz when it differs, phase continuous,
CRC-CCITT (reflected 0x8408, init 0) over mode..ETX including parity bits.
"""
import numpy as np

INTRATE = 12500
BITRATE = 2400
SYN, SOH, STX, ETX, ETB, DEL = 0x16, 0x01, 0x02, 0x03, 0x17, 0x7F


# ------------------------------------------------------------------ ACARS frame bytes
def _crc_table():
    t = np.zeros(256, dtype=np.uint16)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
        t[i] = c
    return t


_CRC = _crc_table()


def crc_ccitt(data, crc=0):
    for b in data:
        crc = (crc >> 8) ^ int(_CRC[(crc ^ b) & 0xFF])
    return crc


def odd_parity(b):
    b &= 0x7F
    return b | (0x80 if bin(b).count("1") % 2 == 0 else 0)


def acars_frame(text=b"", mode=b"2", addr=b".N12345", ack=b"\x15", label=b"H1", bid=b"3",
                etb=False):
    """Bytes of one ACARS transmission (with parity applied), pre-key excluded."""
    body = mode + addr + ack + label + bid
    if text:
        body += bytes([STX]) + text
    body += bytes([ETB if etb else ETX])
    body_p = bytes(odd_parity(b) for b in body)
    crc = crc_ccitt(body_p)
    head = bytes(odd_parity(b) for b in (ord("+"), ord("*"), SYN, SYN, SOH))
    return head + body_p + bytes([crc & 0xFF, crc >> 8]) + bytes([DEL])


def corrupt_frame(frame, rng, kind):
    """Flip bits of a transmission (bytes from acars_frame) the way a noisy channel would, to exercise
    the block thread's repair (acars.c:39-215).  Only text/CRC bytes are touched (index >= 5, before DEL):
      'p1','p2','p3','p4' : that many single-bit errors in distinct text bytes (parity errors)
      'db'                : two bits in one text byte (parity stays odd, CRC fails -> fixdberr)
      'crc'               : one bit in a CRC byte
      'p1crc'             : one parity error + one CRC-byte bit (unrepairable by construction)"""
    b = bytearray(frame)
    text = list(range(6, len(b) - 4))        # after SOH+mode, before ETX/CRC/DEL; keeps framing bytes intact
    if kind in ("p1", "p2", "p3", "p4", "p1crc"):
        n = int(kind[1])
        for i in rng.choice(text, size=n, replace=False):
            b[int(i)] ^= 1 << int(rng.integers(0, 8))
        if kind == "p1crc":
            b[len(b) - 3] ^= 1 << int(rng.integers(0, 8))
            b[len(b) - 2] ^= 1 << int(rng.integers(0, 8))
    elif kind == "db":
        i = int(rng.choice(text))
        x, y = rng.choice(8, size=2, replace=False)
        b[i] ^= (1 << int(x)) | (1 << int(y))
    elif kind == "crc":
        b[len(b) - 2 - int(rng.integers(0, 2))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def random_text(rng, nmin=20, nmax=220):
    n = int(rng.integers(nmin, nmax + 1))
    return bytes(rng.integers(0x20, 0x7F, size=n).astype(np.uint8).tolist())


def frame_bits(frame, prekey=32, tail=16):
    """LSB-first bits with `prekey` one-bits before and `tail` one-bits after."""
    b = np.unpackbits(np.frombuffer(frame, dtype=np.uint8), bitorder="little")
    return np.concatenate([np.ones(prekey, np.uint8), b, np.ones(tail, np.uint8)])


# ------------------------------------------------------------------ MSK audio at 12.5 kHz
def msk_audio(bits, lead=0, trail=0, rate=INTRATE, phase0=0.0):
    """Phase-continuous 1200/2400 Hz MSK of `bits` sampled at `rate`.
    Returns float64 samples in [-1,1]; `lead`/`trail` silent samples around it."""
    bits = np.asarray(bits, dtype=np.uint8)
    prev = np.concatenate([[1], bits[:-1]])
    f = np.where(bits == prev, 2400.0, 1200.0)           # tone of each bit period
    n = int(np.floor(len(bits) * rate / BITRATE))
    t = np.arange(n) / rate                               # sample instants
    k = np.minimum((t * BITRATE).astype(np.int64), len(bits) - 1)
    # phase at start of bit k = 2*pi*sum_{i<k} f_i / BITRATE
    cum = np.concatenate([[0.0], np.cumsum(f)]) / BITRATE
    theta = 2 * np.pi * (cum[k] + f[k] * (t - k / BITRATE)) + phase0
    a = np.sin(theta)
    return np.concatenate([np.zeros(lead), a, np.zeros(trail)])


def channel_audio(rng, nsamp, nframes=None, gap=(3000, 12500), text_len=(20, 220), corrupt=None):
    """A 12.5 kHz audio track of length nsamp with random ACARS frames separated by
    silence (un-modulated carrier).  Returns (audio float64 [nsamp], list of frame bytes)."""
    out = np.zeros(nsamp)
    frames = []
    pos = int(rng.integers(gap[0] // 4, gap[0]))
    while True:
        if nframes is not None and len(frames) >= nframes:
            break
        fr = acars_frame(text=random_text(rng, *text_len),
                         mode=bytes([int(rng.choice(list(b"2EGx")))]),
                         addr=b"." + bytes(rng.integers(0x41, 0x5B, size=6).astype(np.uint8).tolist()),
                         label=bytes(rng.integers(0x30, 0x3A, size=2).astype(np.uint8).tolist()),
                         bid=bytes([int(rng.integers(0x30, 0x3A))]))
        if corrupt is not None:
            kinds = corrupt if isinstance(corrupt, (list, tuple)) else [corrupt]
            k = kinds[len(frames) % len(kinds)]
            if k:
                fr = corrupt_frame(fr, rng, k)
        a = msk_audio(frame_bits(fr), phase0=float(rng.uniform(0, 2 * np.pi)))
        if pos + len(a) + 64 > nsamp:
            break
        out[pos:pos + len(a)] = a
        frames.append(fr)
        pos += len(a) + int(rng.integers(gap[0], gap[1]))
    return out, frames


def envelope(audio, depth=0.5, carrier=0.5, noise=0.0, rng=None):
    """AM envelope carrier*(1+depth*a) (+ gaussian noise), float32 -- what a 12.5 kHz
    front end (soundfile.c) hands to demodMSK."""
    e = carrier * (1.0 + depth * np.asarray(audio, dtype=np.float64))
    if noise > 0:
        e = e + (rng or np.random.default_rng(0)).normal(0.0, noise, size=e.shape)
    return e.astype(np.float32)


# ------------------------------------------------------------------ u8 I/Q at INTRATE*M
def iq_u8_from_envelopes(envs, M, offsets_hz, phases=None, scale=0.25, noise=0.0, rng=None,
                         chunk=4096):
    """Up-convert 12.5 kHz envelopes onto carriers of one wide-band stream.

    envs: [ncarrier, nout] float (nout 12.5 kHz samples, zero-order-held M times),
    offsets_hz: carrier offsets from the tuner centre (Fr - Fc), phases: radians.
    x[n] = scale * sum_c envs[c, n//M] * exp(j(2*pi*f_c*n/(INTRATE*M) + phi_c)) (+ noise)
    u8 = clip(rint(127.37 + 127.5*x)) interleaved I,Q  (the rtl.c:338-339 convention).
    Returns uint8 [nout*M*2]."""
    envs = np.atleast_2d(np.asarray(envs, dtype=np.float64))
    nc, nout = envs.shape
    offsets_hz = np.asarray(offsets_hz, dtype=np.float64).reshape(nc)
    phases = np.zeros(nc) if phases is None else np.asarray(phases, dtype=np.float64).reshape(nc)
    out = np.empty(nout * M * 2, dtype=np.uint8)
    rate = float(INTRATE * M)
    rng = rng or np.random.default_rng(0)
    for s in range(0, nout, chunk):
        e = envs[:, s:s + chunk]
        w = e.shape[1]
        n = np.arange(s * M, (s + w) * M, dtype=np.float64)
        x = np.zeros(w * M, dtype=np.complex128)
        for c in range(nc):
            # reduce the phase exactly: f_c*n/rate is rational, keep the fractional turn
            turns = np.mod(offsets_hz[c] * n / rate, 1.0)
            x += np.repeat(e[c], M) * np.exp(1j * (2 * np.pi * turns + phases[c]))
        x *= scale
        if noise > 0:
            x += rng.normal(0, noise, size=x.shape) + 1j * rng.normal(0, noise, size=x.shape)
        iq = np.empty(w * M * 2, dtype=np.float64)
        iq[0::2] = x.real
        iq[1::2] = x.imag
        out[s * M * 2:(s + w) * M * 2] = np.clip(np.rint(127.37 + 127.5 * iq), 0, 255).astype(np.uint8)
    return out


def iq_complex_from_envelopes(envs, M, offsets_hz, phases=None, scale=0.25, noise=0.0, rng=None):
    """The complex baseband of iq_u8_from_envelopes before quantisation (complex128 [nout*M])."""
    envs = np.atleast_2d(np.asarray(envs, dtype=np.float64))
    nc, nout = envs.shape
    offsets_hz = np.asarray(offsets_hz, dtype=np.float64).reshape(nc)
    phases = np.zeros(nc) if phases is None else np.asarray(phases, dtype=np.float64).reshape(nc)
    rate = float(INTRATE * M)
    rng = rng or np.random.default_rng(0)
    n = np.arange(nout * M, dtype=np.float64)
    x = np.zeros(nout * M, dtype=np.complex128)
    for c in range(nc):
        turns = np.mod(offsets_hz[c] * n / rate, 1.0)
        x += np.repeat(envs[c], M) * np.exp(1j * (2 * np.pi * turns + phases[c]))
    x *= scale
    if noise > 0:
        x += rng.normal(0, noise, size=x.shape) + 1j * rng.normal(0, noise, size=x.shape)
    return x


def iq_s16_from_envelopes(envs, M, offsets_hz, phases=None, scale=0.25, noise=0.0, rng=None, full_scale=0.9):
    """CS16 interleaved I,Q (what SoapySDR delivers, soapy.c:181,238-239): int16 = rint(32767*full_scale*x)."""
    x = iq_complex_from_envelopes(envs, M, offsets_hz, phases, scale, noise, rng)
    out = np.empty(x.size * 2, dtype=np.int16)
    out[0::2] = np.clip(np.rint(32767.0 * full_scale * x.real), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(32767.0 * full_scale * x.imag), -32768, 32767).astype(np.int16)
    return out


def real_f32_from_envelopes(envs, M, offsets_hz, phases=None, scale=0.25, noise=0.0, rng=None):
    """Real float32 samples at INTRATE*M (what an Airspy in FLOAT32_REAL mode delivers, air.c:195,314):
    carriers at offsets_hz from 0 Hz of the REAL spectrum (air.c tunes so that channels sit around Fs/4)."""
    envs = np.atleast_2d(np.asarray(envs, dtype=np.float64))
    nc, nout = envs.shape
    offsets_hz = np.asarray(offsets_hz, dtype=np.float64).reshape(nc)
    phases = np.zeros(nc) if phases is None else np.asarray(phases, dtype=np.float64).reshape(nc)
    rate = float(INTRATE * M)
    rng = rng or np.random.default_rng(0)
    n = np.arange(nout * M, dtype=np.float64)
    x = np.zeros(nout * M)
    for c in range(nc):
        turns = np.mod(offsets_hz[c] * n / rate, 1.0)
        x += np.repeat(envs[c], M) * np.cos(2 * np.pi * turns + phases[c])
    x *= 2 * scale
    if noise > 0:
        x += rng.normal(0, noise, size=x.shape)
    return x.astype(np.float32)


def pad_blocks(x, block=1024, fill=0.0):
    """Pad the last axis to a multiple of `block` (rtl.c:49 RTLOUTBUFSZ) with `fill`."""
    x = np.asarray(x)
    n = x.shape[-1]
    pad = (-n) % block
    if pad == 0:
        return x
    shape = x.shape[:-1] + (pad,)
    return np.concatenate([x, np.full(shape, fill, dtype=x.dtype)], axis=-1)


def xorshift_bytes(seed, n):
    """Deterministic pseudo-random bytes (numpy PCG; the name records the role in SURVEY 8d)."""
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


def message_zoo(rng, n):
    """n transmissions that exercise every branch of outputmsg()'s field split (output.c:486-560): uplinks and
    downlinks, NAK and letter acknowledgements, labels ending in DEL, addresses with leading dots, empty texts (ETX
    right behind the block id), texts shorter than message number + flight id, ETB terminated blocks."""
    out = []
    for i in range(n):
        down = bool(rng.integers(0, 2))
        bid = bytes([int(rng.integers(0x30, 0x3A))]) if down else bytes([int(rng.integers(0x41, 0x5B))])
        ndots = int(rng.integers(1, 4))
        addr = b"." * ndots + bytes(rng.integers(0x41, 0x5B, size=7 - ndots).astype(np.uint8).tolist())
        ack = b"\x15" if rng.integers(0, 2) else bytes([int(rng.integers(0x41, 0x5B))])
        label = bytes([int(rng.integers(0x30, 0x5B))]) + (b"\x7f" if rng.integers(0, 4) == 0 else bytes([int(rng.integers(0x30, 0x5B))]))
        kind = i % 4
        text = b"" if kind == 0 else random_text(rng, 1, 9) if kind == 1 else random_text(rng, 10, 40) if kind == 2 else random_text(rng, 40, 200)
        out.append(acars_frame(text=text, mode=bytes([int(rng.choice(list(b"2EGx")))]), addr=addr, ack=ack, label=label, bid=bid,
                               etb=bool(rng.integers(0, 3) == 0)))
    return out


def frames_audio(frames, rng, gap=(3000, 6000), lead=4000):
    """12.5 kHz audio carrying the given transmissions one after the other (silence between them)."""
    parts = [np.zeros(lead)]
    for fr in frames:
        parts.append(msk_audio(frame_bits(fr), phase0=float(rng.uniform(0, 2 * np.pi))))
        parts.append(np.zeros(int(rng.integers(gap[0], gap[1]))))
    return np.concatenate(parts)
