// blk.hip -- the block thread's work (acars.c:93-215) batched on the device: parity check, CRC
// check, recursive parity-error repair (fixprerr, acars.c:39-64), two-bits-in-a-byte repair
// (fixdberr, acars.c:66-90), parity strip -- results written back in place into the block queue -- and
// outputmsg()'s field split (output.c:486-560).  The arithmetic is nothing; what matters is LATENCY: both
// run beside a down-converter that saturates HBM, where anything that walks a block's text byte by byte
// is a chain of microsecond round trips.  The repair gives every block a wave (the CRC is linear: see
// below), the split stages the text through LDS and builds its record there.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "acg_internal.h"

#define MAXPERR 3
#define ETX 0x83
#define STX 0x02

__device__ __forceinline__ bool crc_acceptable(const unsigned short* synd, unsigned short crc)
{
    // acars.c:54-62 / 70-74: clean, or a single wrong bit inside the two CRC bytes themselves
    if (crc == 0) return true;
    for (int i = 0; i < 16; ++i)
        if (synd[i] == crc) return true;
    return false;
}

// ---- block repair: one WAVE per block -----------------------------------------------------------------------------------
// Rounds 3-4 gave every block one thread that walked its text byte by byte (parity, table-driven CRC).  Beside a down-converter
// that saturates HBM and LDS that is a chain of a few hundred dependent memory round trips per block: the pass measured 0.25 -
// 4.4 ms per call (up to a third of the GPU time), sat between the host and its next call, and cost the whole job ~12 % at
// >= 4096 channels.  The CRC is linear over GF(2) (CRC-CCITT, initial value 0, no final xor), and the reference's own syndrome
// table IS its basis: synd[bit + 8 k] is the CRC of a message whose only set bit is `bit` of the byte that has k bytes behind it
// (syndrom.h:52-295; fixprerr / fixdberr index it as 8 (len - i + 1) + bit, acars.c:45-88).  So
//     crc(text, crc0, crc1) = XOR over every set bit of synd[bit + 8 (bytes behind it)]
// -- identical to acars.c:159-165's table walk (tests/test_host_logic.py checks the identity against the oracle's update_crc),
// and embarrassingly parallel: lane L takes bytes L, L + 64, L + 128, L + 192 (one coalesced 256-byte read per block), xors the
// syndromes of their set bits out of an LDS copy of the table, and six butterfly steps fold the wave.  Parity errors are ballots.
// The rare repairs (fixprerr, fixdberr) run replicated on every lane from wave-uniform values, in the reference's search order.
#define BLK_T 32        // threads per workgroup of the field split below
#define BLK_ROW 336     // its LDS row: a block's 256 text bytes, then the 320-byte record built in place (84 dwords: <= 8 lanes per bank)
#define BLK_WAVES 4
#define NSYND (8 * 243)

// pulls block f's text (256 bytes, 16-byte aligned in the ring) into this thread's LDS row: sixteen independent loads
__device__ __forceinline__ void stage_text(const AcgFrameRec* f, unsigned char* row)
{
    const uint4* src = (const uint4*)f->txt;
    uint4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) ((uint4*)row)[j] = v[j];
}

__device__ __forceinline__ unsigned int synd_of_bits(const unsigned short* synd, unsigned int byte, int k8)
{
    unsigned int x = 0;
    while (byte) {
        const int b = __ffs((int)byte) - 1;
        x ^= synd[b + k8];
        byte &= byte - 1;
    }
    return x;
}

__global__ __launch_bounds__(64 * BLK_WAVES) void blk_repair_kernel(AcgFrameRec* frames, unsigned int cap, const unsigned int* upto,
                                                                    unsigned int* done_upto, unsigned int* done_ctr,
                                                                    const unsigned short* synd_g, const unsigned short* crctab_g)
{
    // The pass covers blocks [*done_upto, *upto): `upto` is the queue length the call's last demodulator launch published
    // (a host-mapped word), NOT the live counter -- the pass runs on a stream of its own beside the demodulator of the NEXT
    // call, which is appending records behind that mark.  At most one lap of the ring: if more than `cap` blocks were queued
    // since the last pass (the host reports that as ACG_EOVERFLOW), the surviving newest `cap` are each processed once.
    // (the table in LDS: read where it lies -- 3.9 KB that stay in L2 -- every look-up is a ~1.5 us round trip beside the
    //  streaming down-converter and a block took ~50 us per wave: the pass fell behind the calls at 2048 channels, call 9)
    (void)crctab_g;
    __shared__ unsigned short synd[NSYND];
    for (int i = threadIdx.x; i < NSYND; i += 64 * BLK_WAVES) synd[i] = synd_g[i];
    __syncthreads();
    const unsigned int hi = __hip_atomic_load(upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int from = *done_upto;
    const unsigned int lo = (hi - from > cap) ? hi - cap : from;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * BLK_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * BLK_WAVES;
    for (unsigned int q = lo + wave; q - lo < hi - lo; q += nwaves) {      // (q is wave-uniform: no divergence around the ballots)
        AcgFrameRec* f = frames + (q % cap);
        const int len = f->len;
        if (len < 13) {                                                    // acars.c:124
            if (lane == 0) f->status = 2;
            continue;
        }
        unsigned char* txt = f->txt;
        unsigned int c[4];
        bool have[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int i = lane + 64 * s_;
            have[s_] = i < len;                                            // (len <= 250)
            c[s_] = have[s_] ? txt[i] : 0u;
        }
        if (lane == 12) c[0] = (c[0] & (ETX | STX)) | (ETX & STX);         // acars.c:132-133
        // parity errors (acars.c:136-144) and the CRC over text + the two CRC bytes (acars.c:159-165)
        unsigned long long bad[4];
        unsigned int x = 0;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            bad[s_] = __ballot(have[s_] && (__popc(c[s_]) & 1) == 0);
            if (have[s_]) x ^= synd_of_bits(synd, c[s_], 8 * (len - (lane + 64 * s_) + 1));
        }
        if (lane < 2) x ^= synd_of_bits(synd, f->crc[lane], 8 * (1 - lane));
#pragma unroll
        for (int off = 32; off; off >>= 1) x ^= (unsigned int)__shfl_xor((int)x, off);
        const unsigned short crc = (unsigned short)x;
        const int pn = __popcll(bad[0]) + __popcll(bad[1]) + __popcll(bad[2]) + __popcll(bad[3]);
        if (pn > MAXPERR) {                                                // acars.c:145
            if (lane == 0) f->status = 2;
            continue;
        }
        int pr[MAXPERR] = {0, 0, 0};                                       // positions of the flagged bytes, ascending
        {
            int k = 0;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                unsigned long long m = bad[s_];
                while (m && k < MAXPERR) {
                    pr[k++] = 64 * s_ + __ffsll((long long)m) - 1;
                    m &= m - 1;
                }
            }
        }
        bool ok = true;
        if (pn) {
            // fixprerr: depth-first over bit positions of the pn flagged bytes, first byte outermost,
            // i.e. lexicographic order of (i0, i1, i2) -- the recursion's visiting order (acars.c:45-50)
            ok = false;
            const int ncomb = 1 << (3 * pn);
            for (int comb = 0; comb < ncomb && !ok; ++comb) {
                unsigned short c2 = crc;
                for (int d = 0; d < pn; ++d) {
                    const int bit = (comb >> (3 * (pn - 1 - d))) & 7;
                    c2 ^= synd[bit + 8 * (len - pr[d] + 1)];
                }
                if (crc_acceptable(synd, c2)) {
                    for (int d = 0; d < pn; ++d)
                        if (lane == (pr[d] & 63)) {
                            const unsigned int flip = 1u << ((comb >> (3 * (pn - 1 - d))) & 7);
                            if ((pr[d] >> 6) == 0) c[0] ^= flip;
                            else if ((pr[d] >> 6) == 1) c[1] ^= flip;
                            else if ((pr[d] >> 6) == 2) c[2] ^= flip;
                            else c[3] ^= flip;
                        }
                    ok = true;
                }
            }
        } else if (crc) {
            // fixdberr (acars.c:66-90)
            ok = false;
            for (int i = 0; i < 16 && !ok; ++i) ok = synd[i] == crc;
            for (int k = 0; k < len && !ok; ++k) {
                const int bo = 8 * (len - k + 1);
                for (int i = 0; i < 8 && !ok; ++i)
                    for (int j = 0; j < 8 && !ok; ++j) {
                        if (i == j) continue;
                        if ((crc ^ synd[i + bo] ^ synd[j + bo]) == 0) {
                            if (lane == (k & 63)) {
                                const unsigned int flip = (1u << i) | (1u << j);
                                if ((k >> 6) == 0) c[0] ^= flip;
                                else if ((k >> 6) == 1) c[1] ^= flip;
                                else if ((k >> 6) == 2) c[2] ^= flip;
                                else c[3] ^= flip;
                            }
                            ok = true;
                        }
                    }
            }
        }
        if (!ok) {
            if (lane == 0) f->status = 2;
            continue;
        }
        int pn2 = 0;                                                        // acars.c:195-207: parity once more, then strip it
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            pn2 += __popcll(__ballot(have[s_] && (__popc(c[s_]) & 1) == 0));
            if (have[s_]) txt[lane + 64 * s_] = (unsigned char)(c[s_] & 0x7f);
        }
        if (lane == 0) {
            f->err = pn;                                                    // acars.c:156
            f->status = pn2 ? 2 : 1;
        }
    }
    // the last workgroup out moves the mark (every workgroup has read it by then) and re-arms the counter
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int d = atomicAdd(done_ctr, 1u);
        if (d == gridDim.x - 1) {
            __hip_atomic_store(done_upto, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// SURVEY 8f.4 -- the field split of outputmsg() (output.c:486-560,566-568,623-631, the build without libacars) on the
// device: processed blocks [first, first + n) of the queue -> fixed binary records (AcgMsgRec == acg_msg, see
// include/acarsdec_amd.h), one thread per block.  valid = 0 marks blocks the repair dropped (acars.c:124-207) and
// blocks the repair has not seen.  The level (a log10) is filled in on the host, like for acg_frame.
__global__ __launch_bounds__(BLK_T) void msg_split_kernel(const AcgFrameRec* frames, unsigned int cap, unsigned int first, unsigned int n, AcgMsgRec* out)
{
    // The record is BUILT IN the thread's LDS row and leaves as twenty 16-byte stores (byte-wise stores of up to 240 text bytes
    // per message into global memory took 0.9 ms per collect beside the streaming down-converter): the block's text is staged
    // into row[0, 256), moved to its place in the record (row + 74: behind its source, so the move runs from the end), then the
    // header goes over the start of the row -- whose bytes have been read into registers first.
    static_assert(sizeof(AcgMsgRec) == 320 && offsetof(AcgMsgRec, txt) == 74 && offsetof(AcgMsgRec, valid) == 44, "record layout");
    __shared__ __attribute__((aligned(16))) unsigned char rows[BLK_T * BLK_ROW];
    const unsigned int q = blockIdx.x * BLK_T + threadIdx.x;
    if (q >= n) return;
    const AcgFrameRec* f = frames + ((first + q) % cap);
    unsigned char* t = rows + threadIdx.x * BLK_ROW;
    const int len = f->len;
    const int chn = f->chn, err = f->err, bitcount = f->bitcount;
    const long long end_bit = f->end_bit, end_sample = f->end_sample;
    const double lvlsum = f->lvlsum;
    const bool valid = f->status == 1 && len >= 13;
    stage_text(f, t);
    AcgMsgRec* m = (AcgMsgRec*)t;                                           // the record, in place (LDS)
    // ---- the fields, read out of the text before anything is overwritten
    char mode = 0, addr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ack = 0, label[3] = {0, 0, 0}, bid = 0, no[5] = {0, 0, 0, 0, 0}, fid[7] = {0, 0, 0, 0, 0, 0, 0};
    char bs = 0, be = 0, down = 0;
    int k = 0, tl = 0;
    if (valid) {
        mode = (char)t[k++];
        int j = 0;
        for (int i = 0; i < 7; ++i, ++k)
            if (t[k] != '.') addr[j++] = (char)t[k];                       // output.c:502-508
        ack = t[k] == 0x15 ? '!' : (char)t[k];                             // NAK is not printable, output.c:511-514
        ++k;
        label[0] = (char)t[k++];
        label[1] = t[k] == 0x7f ? 'd' : (char)t[k];                        // output.c:518-520
        ++k;
        bid = (char)t[k++];
        down = (bid >= '0' && bid <= '9') ? 1 : 0;                         // IS_DOWNLINK_BLK, output.c:31
        bs = (char)t[k++];
        be = (char)t[len - 1];
        if (bs != 0x03) {
            if (down) {
                int i;
                for (i = 0; i < 4 && k < len - 1; ++i, ++k) no[i] = (char)t[k];       // output.c:547-550
                for (i = 0; i < 6 && k < len - 1; ++i, ++k) fid[i] = (char)t[k];      // output.c:560-563
            }
            tl = len - k - 1;                                              // output.c:567
            if (tl < 0) tl = 0;
        }
    }
    // ---- the text to its place (destination 74 + i lies behind source k + i: last byte first), the rest of the text area zeroed
    for (int i = tl - 1; i >= 0; --i) t[74 + i] = t[k + i];
    for (int i = 74 + tl; i < 320; ++i) t[i] = 0;
    // ---- the header (every byte of the record is defined: nothing stale crosses the ABI)
    m->chn = chn;
    m->err = err;
    m->lvl = 0.f;
    m->txt_len = tl;
    m->end_bit = end_bit;
    m->end_sample = end_sample;
    m->lvlsum = lvlsum;
    m->bitcount = bitcount;
    m->valid = valid ? 1 : 0;
    m->mode = mode;
    for (int i = 0; i < 8; ++i) m->addr[i] = addr[i];
    m->ack = ack;
    for (int i = 0; i < 3; ++i) m->label[i] = label[i];
    m->bid = bid;
    for (int i = 0; i < 5; ++i) m->no[i] = no[i];
    for (int i = 0; i < 7; ++i) m->fid[i] = fid[i];
    m->bs = bs;
    m->be = be;
    m->down = down;
    // ---- out
    uint4* dst = (uint4*)(out + q);
#pragma unroll
    for (int j = 0; j < 20; ++j) dst[j] = ((const uint4*)t)[j];
}

extern "C" int acg_launch_msg_split(const AcgFrameRec* frames, unsigned int cap, unsigned int first, unsigned int n,
                                    AcgMsgRec* out, void* stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(msg_split_kernel, dim3((n + BLK_T - 1) / BLK_T), dim3(BLK_T), 0, (hipStream_t)stream, frames, cap, first, n, out);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_blk_repair(AcgFrameRec* frames, unsigned int cap, const unsigned int* upto,
                                     unsigned int* done_upto, unsigned int* done_ctr, const unsigned short* synd,
                                     const unsigned short* crctab, int nch, void* stream)
{
    // a wave per block, the waves looping over the call's blocks (about one per channel per call of 8 callbacks): one wave per 8
    // channels, 128 ... 1024 waves -- enough that the pass keeps up with the calls at every width (64 waves did not at 2048
    // channels), small enough not to crowd the demodulator's CUs at 1024 (1024 waves, most of them finding nothing, did)
    int wgs = nch / (8 * BLK_WAVES);
    wgs = wgs < 32 ? 32 : wgs > 256 ? 256 : wgs;
    hipLaunchKernelGGL(blk_repair_kernel, dim3(wgs), dim3(64 * BLK_WAVES), 0, (hipStream_t)stream, frames, cap, upto, done_upto, done_ctr, synd, crctab);
    return (int)hipGetLastError();
}
