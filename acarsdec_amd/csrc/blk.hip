// blk.hip -- the block thread's work (acars.c:93-215) batched on the device: parity check, CRC
// check, recursive parity-error repair (fixprerr, acars.c:39-64), two-bits-in-a-byte repair
// (fixdberr, acars.c:66-90), parity strip.  One thread per queued block, results written back in place
// into the block queue.  The arithmetic is nothing; what matters is LATENCY: the pass runs beside the
// streaming down-converter, where a byte-wise walk over a block's text in global memory is a chain of
// ~2 us HBM round trips (round 4 measured 0.5 ms per pass at 1024 channels, 4 ms at 16 384: a fifth of
// the GPU time).  So a thread first pulls its block's text into its own LDS row with sixteen
// independent 16-byte loads, works there (tables in LDS too), and writes the row back as vectors.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "acg_internal.h"

#define MAXPERR 3
#define ETX 0x83
#define STX 0x02

__device__ __forceinline__ unsigned short crc_upd(const unsigned short* tab, unsigned short crc, unsigned int c)
{
    return (unsigned short)((crc >> 8) ^ tab[(crc ^ c) & 0xff]);          // syndrom.h:49
}

__device__ __forceinline__ bool crc_acceptable(const unsigned short* synd, unsigned short crc)
{
    // acars.c:54-62 / 70-74: clean, or a single wrong bit inside the two CRC bytes themselves
    if (crc == 0) return true;
    for (int i = 0; i < 16; ++i)
        if (synd[i] == crc) return true;
    return false;
}

// Workgroups of 32 threads with one LDS row of 336 bytes each (a block's 256 text bytes; the split builds its 320-byte record
// in the same row; 84 dwords: at most 8 lanes share a bank): 10.5 KiB per workgroup, small enough to sit in what the
// down-converter's persistent workgroups leave free on a CU (16 KiB) -- with 64 rows + the syndrome table in LDS (22 KiB) the
// pass only fitted beside the demodulator's waves, was starved by them (0.25-0.9 ms per pass) and took issue slots from the
// one chain that sets the step at <= 2048 channels.  The syndrome table (only touched when a block needs repair) stays global.
#define BLK_T 32
#define BLK_ROW 336

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

__global__ __launch_bounds__(BLK_T) void blk_repair_kernel(AcgFrameRec* frames, unsigned int cap, const unsigned int* upto,
                                                           unsigned int* done_upto, unsigned int* done_ctr,
                                                           const unsigned short* synd, const unsigned short* crctab_g)
{
    // The pass covers blocks [*done_upto, *upto): `upto` is the queue length the call's last demodulator launch published
    // (a host-mapped word), NOT the live counter -- the pass runs on a stream of its own beside the demodulator of the NEXT
    // call, which is appending records behind that mark.  At most one lap of the ring: if more than `cap` blocks were queued
    // since the last pass (the host reports that as ACG_EOVERFLOW), the surviving newest `cap` are each processed once.
    __shared__ unsigned short crctab[256];
    __shared__ __attribute__((aligned(16))) unsigned char rows[BLK_T * BLK_ROW];
    for (int i = threadIdx.x; i < 256; i += BLK_T) crctab[i] = crctab_g[i];
    __syncthreads();
    unsigned char* txt = rows + threadIdx.x * BLK_ROW;                       // this thread's row; nobody else touches it
    const unsigned int hi = __hip_atomic_load(upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int from = *done_upto;
    const unsigned int lo = (hi - from > cap) ? hi - cap : from;
    for (unsigned int q = lo + blockIdx.x * BLK_T + threadIdx.x; q - lo < hi - lo; q += gridDim.x * BLK_T) {
        AcgFrameRec* f = frames + (q % cap);
        const int len = f->len;
        const unsigned int c0 = f->crc[0], c1 = f->crc[1];
        stage_text(f, txt);
        if (len < 13) { f->status = 2; continue; }                         // acars.c:124
        txt[12] = (unsigned char)((txt[12] & (ETX | STX)) | (ETX & STX));   // acars.c:132-133
        int pn = 0, pr[MAXPERR];
        unsigned short crc = 0;
        for (int i = 0; i < len; ++i) {                                     // acars.c:136-144, 159-163
            const unsigned int c = txt[i];
            if ((__popc(c) & 1) == 0) {
                if (pn < MAXPERR) pr[pn] = i;
                ++pn;
            }
            crc = crc_upd(crctab, crc, c);
        }
        if (pn > MAXPERR) { f->status = 2; continue; }                      // acars.c:145
        crc = crc_upd(crctab, crc, c0);
        crc = crc_upd(crctab, crc, c1);
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
                    for (int d = 0; d < pn; ++d) txt[pr[d]] ^= (unsigned char)(1 << ((comb >> (3 * (pn - 1 - d))) & 7));
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
                            txt[k] ^= (unsigned char)((1 << i) | (1 << j));
                            ok = true;
                        }
                    }
            }
        }
        if (!ok) { f->status = 2; continue; }
        int pn2 = 0;                                                         // acars.c:195-207
        for (int i = 0; i < len; ++i) {
            const unsigned int c = txt[i];
            if ((__popc(c) & 1) == 0) ++pn2;
            txt[i] = (unsigned char)(c & 0x7f);
        }
        {   // back into the ring: the text as vectors (whole 16-byte groups up to len: the bytes behind len are the block's own)
            uint4* dst = (uint4*)f->txt;
            const int nv = (len + 15) >> 4;
            for (int j = 0; j < nv; ++j) dst[j] = ((const uint4*)txt)[j];
        }
        f->err = pn;                                                        // acars.c:156
        f->status = pn2 ? 2 : 1;
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
                                     const unsigned short* crctab, void* stream)
{
    hipLaunchKernelGGL(blk_repair_kernel, dim3(128), dim3(BLK_T), 0, (hipStream_t)stream, frames, cap, upto, done_upto, done_ctr, synd, crctab);
    return (int)hipGetLastError();
}
