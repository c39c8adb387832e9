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
// The repairs (fixprerr, fixdberr) are searched 64 candidates at a time in the reference's search order (see below).
#define BLK_ROW 336     // LDS row of the field split: a block's 256 text bytes, then the 320-byte record built in place
#define BLK_WAVES 4
#define NSYND (8 * 243)

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
#ifdef ACG_BLK_AB_PRIO
    __builtin_amdgcn_s_setprio(3);          // A/B build only: see profiles/LEDGER.md round 5
#endif
    __shared__ unsigned short synd[NSYND];
    // The pass's fixed cost is a chain of memory round trips beside a down-converter that saturates HBM (a few microseconds
    // each): the marks, the table, the first block.  They are requested in that order WITHOUT waiting in between: the two marks
    // first, the table's rows while those are in flight, the wave's first block as soon as the marks say which one it is, and
    // only then the barrier behind which the table is needed.
    const unsigned int mark = __hip_atomic_load(upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int from = *done_upto;
    unsigned short trow[(NSYND + 64 * BLK_WAVES - 1) / (64 * BLK_WAVES)];
#pragma unroll
    for (int k = 0; k < (NSYND + 64 * BLK_WAVES - 1) / (64 * BLK_WAVES); ++k) {
        const int i = threadIdx.x + k * 64 * BLK_WAVES;
        trow[k] = i < NSYND ? synd_g[i] : (unsigned short)0;
    }
    // A mark at or BEHIND what has been processed means nothing to do (the counters are monotonic and wrap: a signed
    // difference).  It cannot move the pass backwards: re-processing blocks whose parity has already been stripped would
    // fail their parity check and drop valid messages (a pass that reads a mark word which a later call
    // has already re-used; the host also keeps the words from being re-used early, acg_api.cpp begin_call).
    const unsigned int hi = (int)(mark - from) > 0 ? mark : from;
    const unsigned int lo = (hi - from > cap) ? hi - cap : from;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = blockIdx.x * BLK_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * BLK_WAVES;
    // Everything a block needs from memory is requested at once -- its length, all 256 bytes of its text row (what lies beyond
    // len is masked below, the row is always there) and the two CRC bytes: ONE round trip per block instead of three dependent
    // ones (length -> text -> CRC bytes) -- and the request for the wave's NEXT block is in flight while it works on this one.
    unsigned int nraw[4] = {0, 0, 0, 0}, ncrcb = 0;
    int nlen = 0;
    auto request = [&](unsigned int q_) {
        const AcgFrameRec* g_ = frames + (q_ & (cap - 1));                  // (cap is a power of two)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) nraw[s_] = g_->txt[lane + 64 * s_];
        ncrcb = g_->crc[lane & 1];
        nlen = g_->len;
    };
    if (wave < hi - lo) request(lo + wave);
#pragma unroll
    for (int k = 0; k < (NSYND + 64 * BLK_WAVES - 1) / (64 * BLK_WAVES); ++k) {
        const int i = threadIdx.x + k * 64 * BLK_WAVES;
        if (i < NSYND) synd[i] = trow[k];
    }
    __syncthreads();
    for (unsigned int q = lo + wave; q - lo < hi - lo; q += nwaves) {      // (q is wave-uniform: no divergence around the ballots)
        AcgFrameRec* f = frames + (q & (cap - 1));
        unsigned char* txt = f->txt;
        unsigned int raw[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) raw[s_] = nraw[s_];
        const unsigned int crcb = ncrcb;
        const int len = nlen;
        if (q + nwaves - lo < hi - lo) request(q + nwaves);
        if (len < 13) {                                                    // acars.c:124
            if (lane == 0) f->status = 2;
            continue;
        }
        unsigned int c[4];
        bool have[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int i = lane + 64 * s_;
            have[s_] = i < len;                                            // (len <= 250)
            c[s_] = have[s_] ? raw[s_] : 0u;
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
        if (lane < 2) x ^= synd_of_bits(synd, crcb, 8 * (1 - lane));
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
        // The two searches are the reference's, candidate for candidate, but the WAVE walks them 64 candidates at a time in the
        // reference's visiting order and takes the first hit of the first 64-group that has one (ballot + find-first): a search
        // that fails -- every block cut by noise whose parity happens to hold, 512 candidates or 250 x 56 -- is then a few
        // microseconds instead of the 0.5 - 1.7 ms one lane took for it (rounds 3-4 ran them replicated on every lane; those
        // passes were the 12 - 14 % of GPU time of profiles/r04_*_stats.txt).
        bool ok = true;
        if (pn) {
            // fixprerr: depth-first over bit positions of the pn flagged bytes, first byte outermost,
            // i.e. lexicographic order of (i0, i1, i2) -- the recursion's visiting order (acars.c:45-50)
            ok = false;
            const int ncomb = 1 << (3 * pn);
            for (int c0 = 0; c0 < ncomb && !ok; c0 += 64) {
                const int comb = c0 + lane;
                unsigned short c2 = crc;
                for (int d = 0; d < pn; ++d) {
                    const int bit = (comb >> (3 * (pn - 1 - d))) & 7;
                    c2 ^= synd[bit + 8 * (len - pr[d] + 1)];
                }
                const unsigned long long hit = __ballot(comb < ncomb && crc_acceptable(synd, c2));
                if (hit) {
                    const int first = c0 + __ffsll((long long)hit) - 1;
                    for (int d = 0; d < pn; ++d)
                        if (lane == (pr[d] & 63)) {
                            const unsigned int flip = 1u << ((first >> (3 * (pn - 1 - d))) & 7);
                            if ((pr[d] >> 6) == 0) c[0] ^= flip;
                            else if ((pr[d] >> 6) == 1) c[1] ^= flip;
                            else if ((pr[d] >> 6) == 2) c[2] ^= flip;
                            else c[3] ^= flip;
                        }
                    ok = true;
                }
            }
        } else if (crc) {
            // fixdberr (acars.c:66-90): the CRC bytes first, then text bytes in ascending order, bit pairs (i, j) in the loops' order
            ok = __ballot(lane < 16 && synd[lane & 15] == crc) != 0;
            for (int k0 = 0; k0 < len && !ok; k0 += 64) {
                const int k = k0 + lane;
                unsigned int flip = 0;
                if (k < len) {
                    const int bo = 8 * (len - k + 1);
                    unsigned short sy[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) sy[i] = synd[i + bo];
#pragma unroll
                    for (int i = 7; i >= 0; --i)                           // (descending: the last assignment is the first pair of the loops)
#pragma unroll
                        for (int j = 7; j >= 0; --j)
                            if (i != j && (unsigned short)(crc ^ sy[i] ^ sy[j]) == 0) flip = (1u << i) | (1u << j);
                }
                const unsigned long long hit = __ballot(flip != 0);
                if (hit) {
                    const int first = __ffsll((long long)hit) - 1;      // lowest lane = lowest k of this group
                    if (lane == first) {
                        if ((k >> 6) == 0) c[0] ^= flip;
                        else if ((k >> 6) == 1) c[1] ^= flip;
                        else if ((k >> 6) == 2) c[2] ^= flip;
                        else c[3] ^= flip;
                    }
                    ok = true;
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
// include/acarsdec_amd.h), one WAVE per block.  valid = 0 marks blocks the repair dropped (acars.c:124-207) and
// blocks the repair has not seen.  The level (a log10) is filled in on the host, like for acg_frame.
//
// Round 4 gave every block one THREAD with a 336-byte LDS row (10.5 KiB per 32-thread workgroup) that moved the text byte by
// byte.  Beside the down-converter (8 x 18 KiB of a CU's 160 KiB) and a demodulator workgroup (9 KiB) such a workgroup does not
// FIT on most CUs: the trace of BASELINE configs[4] (profiles/r05_tails.md) has one launch in ten waiting ~1 ms for a
// down-converter workgroup to retire -- and acg_collect_msgs waits for it.  Now: 336 bytes of LDS per block (1.3 KiB per
// workgroup of four waves), the text moved by all 64 lanes at once, the record leaving as twenty 16-byte stores.
#define SPLIT_WAVES 4
__global__ __launch_bounds__(64 * SPLIT_WAVES) void msg_split_kernel(const AcgFrameRec* frames, unsigned int cap, unsigned int first, unsigned int n, AcgMsgRec* out)
{
    static_assert(sizeof(AcgMsgRec) == 320 && offsetof(AcgMsgRec, txt) == 74 && offsetof(AcgMsgRec, valid) == 44 &&
                  offsetof(AcgMsgRec, soh_back) == 316, "record layout");
    __shared__ __attribute__((aligned(16))) unsigned char rows[SPLIT_WAVES][BLK_ROW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned int q = blockIdx.x * SPLIT_WAVES + wv;                   // wave-uniform
    if (q >= n) return;
    const AcgFrameRec* f = frames + ((first + q) & (cap - 1));              // (cap is a power of two)
    unsigned char* t = rows[wv];
    const int len = f->len;
    const int chn = f->chn, err = f->err, bitcount = f->bitcount, soh_back = f->soh_back;
    const long long end_bit = f->end_bit, end_sample = f->end_sample;
    const double lvlsum = f->lvlsum;
    const bool valid = f->status == 1 && len >= 13;
    // the block's text (256 bytes, 16-byte aligned in the ring) into the row: one 16-byte load per lane
    if (lane < 16) ((uint4*)t)[lane] = ((const uint4*)f->txt)[lane];
    // (LDS operations of one wave execute in order; the fences keep the compiler from moving them across each other)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the fields, read out of the text before anything is overwritten (every lane the same: broadcast reads)
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
    // ---- the text to its place: all of it is read (lane L: bytes L, L + 64, ...) before any of it is written; the rest of
    // the text area is zeroed by the same stores (every byte of the record is defined: nothing stale crosses the ABI)
    unsigned char mv[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const int i = lane + 64 * s_;
        mv[s_] = i < tl ? t[k + i] : (unsigned char)0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const int i = lane + 64 * s_;
        if (i < ACG_MSG_TXT) t[74 + i] = mv[s_];
    }
    // ---- the header
    if (lane == 0) {
        AcgMsgRec* m = (AcgMsgRec*)t;                                       // the record, in place (LDS)
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
        m->soh_back = soh_back;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- out: twenty 16-byte stores
    if (lane < 20) ((uint4*)(out + q))[lane] = ((const uint4*)t)[lane];
}

extern "C" int acg_launch_msg_split(const AcgFrameRec* frames, unsigned int cap, unsigned int first, unsigned int n,
                                    AcgMsgRec* out, void* stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(msg_split_kernel, dim3((n + SPLIT_WAVES - 1) / SPLIT_WAVES), dim3(64 * SPLIT_WAVES), 0, (hipStream_t)stream, frames, cap, first, n, out);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_blk_repair(AcgFrameRec* frames, unsigned int cap, const unsigned int* upto,
                                     unsigned int* done_upto, unsigned int* done_ctr, const unsigned short* synd,
                                     const unsigned short* crctab, int nch, void* stream)
{
    // a wave per block, the waves looping over the call's blocks (0.6 - 1 per channel per call of 8 callbacks): one wave per 8
    // channels, 128 ... 1024 waves -- enough that the pass keeps up with the calls at every width, small enough not to crowd
    // the CUs: round 5 tried one wave per 2 channels and the pass got SLOWER (88 -> 97 us at 1024 channels, 105 -> 141 us at
    // 2048, 111 -> 201 us at 4096: profiles/r05_*_stats_call9_wave_per_2_channels.txt) -- what a pass costs is workgroups finding
    // a place beside the down-converter and filling their table, not the handful of blocks a wave takes one after the other.
    int wgs = nch / (8 * BLK_WAVES);
    wgs = wgs < 32 ? 32 : wgs > 256 ? 256 : wgs;
    hipLaunchKernelGGL(blk_repair_kernel, dim3(wgs), dim3(64 * BLK_WAVES), 0, (hipStream_t)stream, frames, cap, upto, done_upto, done_ctr, synd, crctab);
    return (int)hipGetLastError();
}
