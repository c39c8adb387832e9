"""msk_lean.hip frames the bits of a SEGMENT (up to 8 bit periods) at once instead of calling putbit() / decodeAcars() after every
bit.  This file restates both sides in Python integers and runs them against each other on random bit streams from random
starting states (CPU suite; the device code itself is compared with msk.hip's kernel and the oracle in tests/test_gpu_lean.py):

  * `ref_bit`      putbit() (msk.c:53-63) + decodeAcars() (acars.c:246-375) + `MskS++` (msk.c:127), bit by bit -- the reference;
  * `lean_segment` the deferred framing of msk_lean.hip, statement for statement: the two 1-bit shift registers (vo > 0, vo < 0),
                   the polarity mask from MskS, the bit-parallel SYN / ~SYN search, the one text byte a segment can close, the rare
                   branch through the full machine -- and the rule that ends a segment before any bit at which the reference
                   could reset the loop (`MskDf = 0`, acars.c:242), which is what makes deferring exact.

What must hold after every segment + closing bit: the framing state (Acarsstate, nbits, outbits, MskS, blk len / err / text,
crc[0]), the blocks put on the queue with the bit at which they were put, and the bits at which the loop was reset."""
import numpy as np
import pytest

SYN, SOH, ETX, ETB, DLE, MAXPERR = 0x16, 0x01, 0x83, 0x97, 0x7F, 3
WSYN, SYN2, SOH1, TXT, CRC1, CRC2, END = range(7)
SEG = 8


class St:
    def __init__(self):
        self.astate, self.nbits, self.outbits, self.S = WSYN, 8, 0, 0          # initAcars (acars.c:230-232)
        self.blen, self.berr, self.txt, self.crc0 = 0, 0, bytearray(256), 0
        self.resets, self.blocks, self.soh = [], [], []                      # (bit index) of MskDf = 0; blocks put; SOH stamps

    def key(self):
        return (self.astate, self.nbits, self.outbits, self.S & 0xFFFFFFFF, self.blen, self.berr,
                bytes(self.txt[: self.blen]) if self.astate == TXT else b"", self.crc0 if self.astate == CRC2 else 0,
                tuple(self.resets), tuple(self.blocks), tuple(self.soh))

    def copy(self):
        o = St()
        o.__dict__.update({k: (v.copy() if isinstance(v, (bytearray, list)) else v) for k, v in self.__dict__.items()})
        return o


def reset_acars(st, i):                                    # acars.c:239-244
    st.astate, st.nbits = WSYN, 1
    st.resets.append(i)


def put_frame(st, crc1, i, via="CRC2"):                    # acars.c:350-369
    st.blocks.append((i, st.blen, st.berr, st.crc0, crc1, bytes(st.txt[: st.blen]), via))
    st.astate, st.nbits = END, 8


def decode_acars(st, i):                                   # acars.c:246-375 (msk_common.h decode_acars)
    r = st.outbits & 0xFF
    a = st.astate
    if a == WSYN:
        if r == SYN:
            st.astate, st.nbits = SYN2, 8
        elif r == (~SYN & 0xFF):
            st.S ^= 2
            st.astate, st.nbits = SYN2, 8
        else:
            st.nbits = 1
    elif a == SYN2:
        if r == SYN:
            st.astate, st.nbits = SOH1, 8
        elif r == (~SYN & 0xFF):
            st.S ^= 2
            st.nbits = 8
        else:
            reset_acars(st, i)
    elif a == SOH1:
        if r == SOH:
            st.astate, st.blen, st.berr, st.nbits = TXT, 0, 0, 8
            st.soh.append(i)
        else:
            reset_acars(st, i)
    elif a == TXT:
        st.txt[st.blen] = r
        st.blen += 1
        if bin(r).count("1") % 2 == 0:
            st.berr += 1
            if st.berr > MAXPERR + 1:
                return reset_acars(st, i)
        if r in (ETX, ETB):
            st.astate, st.nbits = CRC1, 8
            return
        if st.blen > 20 and r == DLE:
            st.blen -= 3
            st.crc0 = st.txt[st.blen]
            return put_frame(st, st.txt[st.blen + 1], i, via="DLE")
        if st.blen > 240:
            return reset_acars(st, i)
        st.nbits = 8
    elif a == CRC1:
        st.crc0, st.astate, st.nbits = r, CRC2, 8
    elif a == CRC2:
        put_frame(st, r, i)
    else:
        reset_acars(st, i)
        st.nbits = 8


def ref_bit(st, P, N, i):
    """one bit the reference's way: msk.c:122-127 (polarity from MskS & 2), putbit, decodeAcars, MskS++"""
    bit = N if (st.S & 2) else P
    st.outbits = ((st.outbits >> 1) & 0x7F) | (0x80 if bit else 0)
    st.nbits -= 1
    if st.nbits <= 0:
        decode_acars(st, i)
    st.S = (st.S + 1) & 0xFFFFFFFF


def lim_of(st):
    """how many bits of this channel may wait (msk_lean.hip, segment setup)"""
    safe = (st.astate == WSYN or (st.astate == TXT and st.berr <= MAXPERR and st.blen <= 239) or st.astate == CRC1) and 1 <= st.nbits <= 8
    return SEG if safe else st.nbits - 1


def bitrev32(x):
    return int("{:032b}".format(x & 0xFFFFFFFF)[::-1], 2)


def lean_segment(st, Ps, Ns, i0):
    """msk_lean.hip's deferred framing of c = len(Ps) bits (bit k of the segment is bit i0 + k of the stream)"""
    c = len(Ps)
    assert 1 <= c <= lim_of(st)
    M32 = 0xFFFFFFFF
    P = N = 0
    for k in range(c):                                     # shift_in: oldest bit ends up highest
        P = (2 * P + Ps[k]) & M32
        N = (2 * N + Ns[k]) & M32
    S0 = st.S
    isW, isT = st.astate == WSYN, st.astate == TXT
    Pr, Nr = bitrev32(P) >> (32 - c), bitrev32(N) >> (32 - c)
    Mpol = (0x993366CC >> ((S0 & 3) * 8)) & M32
    B0 = (Mpol & Nr) | (~Mpol & Pr & M32)
    old = st.outbits
    W = ((B0 << 8) | old) & M32
    m = st.nbits
    out_end = (W >> c) & 0xFF
    reach = m <= c
    V = W >> 1
    X = (((~V) << 16) | V) & M32
    Y = ~X & M32
    dif = (X << 7) | (Y << 6) | (Y << 5) | (X << 4) | (Y << 3) | (X << 2) | (X << 1) | X
    valid = (((((1 << c) - 1) >> (m - 1)) << (m - 1)) & M32) if reach else 0
    hit = ~dif & M32
    mlo, mhi = (hit >> 7) & valid, (hit >> 23) & valid
    mm = mlo | mhi
    rb = (W >> m) & 0xFF
    plain = isT and reach and bin(rb).count("1") % 2 == 1 and ((rb + 1) & 0x60) != 0
    if plain:
        st.txt[st.blen] = rb
        st.blen += 1
    nbits_n = m - c if m > c else (1 if isW else m + 8 - c)
    S_n = (S0 + c) & M32
    if (isW and mm) or (reach and not isW and not plain):
        if isW:
            k = (mm & -mm).bit_length() - 1                # ctz
            st.astate = SYN2
            nbits_n = 8 - (c - 1 - k)
            if (mhi >> k) & 1:
                S_n = ((((S0 + k) & M32) ^ 2) + (c - k)) & M32
                B1 = (Mpol & Pr) | (~Mpol & Nr & M32)
                lowm = (2 << k) - 1
                Bx = (B0 & lowm) | (B1 & ~lowm & M32)
                out_end = ((((Bx << 8) | old) & M32) >> c) & 0xFF
        else:
            k = m - 1
            st.outbits = rb
            st.S = (S0 + k) & M32
            decode_acars(st, i0 + k)
            assert st.S == (S0 + k) & M32                  # (the states that touch MskS are never framed late)
            nbits_n = st.nbits - (c - 1 - k)
    st.nbits, st.outbits, st.S = nbits_n, out_end, S_n


def random_bits(rng, n, kind, S0=0):
    """(P, N) per bit: P = soft symbol > 0, N = soft symbol < 0 (both 0: a soft symbol of exactly 0).  The transmitted bits are
    folded with bit 1 of a counter that starts at S0, the way the demodulator's alternating decision (msk.c:115-126) sees them."""
    if kind == "noise":
        b = rng.integers(0, 2, n)
    elif kind == "frames":
        from acarsdec_amd import synth as S
        parts = []
        while sum(len(p) for p in parts) < n:
            fr = bytearray(S.acars_frame(text=S.random_text(rng, 1, 250)))
            r = rng.integers(0, 6)
            if r == 1:
                for j in rng.choice(np.arange(6, len(fr) - 4), size=min(6, len(fr) - 10), replace=False):
                    fr[int(j)] ^= 1 << int(rng.integers(0, 8))
            if r == 2:
                fr[len(fr) - 4] = S.odd_parity(0x41 + int(rng.integers(0, 26)))      # no terminator: the block ends at DEL
            if r == 3:
                fr[int(rng.integers(0, 5))] ^= 1 << int(rng.integers(0, 8))          # damaged head: SYN2 / SOH1 fail
            bits = S.frame_bits(bytes(fr), prekey=int(rng.integers(4, 40)), tail=int(rng.integers(2, 30)))
            if rng.integers(0, 2):
                bits = 1 - bits                                                      # the other polarity: ~SYN
            parts.append(bits)
            parts.append(rng.integers(0, 2, int(rng.integers(0, 40))))
        b = np.concatenate(parts)[:n]
    else:
        b = np.zeros(n, dtype=np.int64) + (1 if kind == "ones" else 0)
    b = np.asarray(b, dtype=np.int64) ^ (((S0 + np.arange(n)) >> 1) & 1)
    zero = rng.random(n) < {"silence": 1.0, "frames": 0.0003}.get(kind, 0.02)                      # soft symbols of exactly 0
    return np.where(zero, 0, b), np.where(zero, 0, 1 - b)


@pytest.mark.parametrize("kind", ["frames", "noise", "ones", "silence"])
def test_deferred_framing_is_the_bit_by_bit_framing(kind):
    rng = np.random.default_rng({"frames": 1, "noise": 2, "ones": 3, "silence": 4}[kind])
    n = 120000 if kind == "frames" else 40000
    S0 = int(rng.integers(0, 2 ** 32))
    P, N = random_bits(rng, n, kind, S0)
    ref, lean = St(), St()
    ref.S = lean.S = S0
    i = 0
    segs = late = 0
    while i < n - 9:
        # a segment ends where the channel's own lim says, or earlier (another channel of the wave, the end of the buffer)
        c = min(lim_of(lean), int(rng.integers(0, SEG + 1)) if rng.integers(0, 3) == 0 else SEG)
        if c > 0:
            resets_before = len(lean.resets)
            lean_segment(lean, [int(x) for x in P[i:i + c]], [int(x) for x in N[i:i + c]], i)
            assert len(lean.resets) == resets_before, "a bit that resets the loop was framed late"
            for k in range(c):
                ref_bit(ref, int(P[i + k]), int(N[i + k]), i + k)
            i += c
            segs += 1
            late += c
            assert lean.key() == ref.key(), (i, c)
        if c < SEG:                                       # the closing period: framing inline, both sides the reference's way
            ref_bit(ref, int(P[i]), int(N[i]), i)
            ref_bit(lean, int(P[i]), int(N[i]), i)
            i += 1
            assert lean.key() == ref.key(), i
    assert late > 0.6 * n
    if kind == "frames":
        assert len(ref.blocks) > 30 and len(ref.resets) > 60
        assert any(b[6] == "DLE" for b in ref.blocks) and any(b[6] == "CRC2" for b in ref.blocks)      # both ways a block ends
    if kind == "noise":
        assert len(ref.resets) > 100                      # false SYN / ~SYN all the time
