// blk.hip -- the block thread's work (acars.c:93-215) batched on the device: parity check, CRC
// check, recursive parity-error repair (fixprerr, acars.c:39-64), two-bits-in-a-byte repair
// (fixdberr, acars.c:66-90), parity strip.  One thread per queued block: blocks are rare (a few per
// second per channel) and each costs at most a few thousand table look-ups, so this is not a
// bandwidth or latency problem -- it only removes the last per-message host loop when tens of
// thousands of channels deliver blocks.  Results are written back in place into the block queue.
#include <hip/hip_runtime.h>
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

__global__ void blk_repair_kernel(AcgFrameRec* frames, unsigned int cap, const unsigned int* count,
                                  const unsigned int* done_upto, const unsigned short* synd,
                                  const unsigned short* crctab)
{
    // at most one lap of the ring: if the demodulator queued more than `cap` blocks since the last repair pass
    // (the host reports that as ACG_EOVERFLOW), the surviving newest `cap` are each processed exactly once
    const unsigned int hi = *count;
    const unsigned int lo = (hi - *done_upto > cap) ? hi - cap : *done_upto;
    for (unsigned int q = lo + blockIdx.x * blockDim.x + threadIdx.x; q - lo < hi - lo; q += gridDim.x * blockDim.x) {
        AcgFrameRec* f = frames + (q % cap);
        const int len = f->len;
        unsigned char* txt = f->txt;
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
        f->err = pn;                                                        // acars.c:156
        crc = crc_upd(crctab, crc, f->crc[0]);
        crc = crc_upd(crctab, crc, f->crc[1]);
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
            if ((__popc((unsigned int)txt[i]) & 1) == 0) ++pn2;
            txt[i] &= 0x7f;
        }
        f->status = pn2 ? 2 : 1;
    }
}

__global__ void blk_advance_kernel(unsigned int* done_upto, const unsigned int* count) { *done_upto = *count; }

extern "C" int acg_launch_blk_repair(AcgFrameRec* frames, unsigned int cap, const unsigned int* count,
                                     unsigned int* done_upto, const unsigned short* synd,
                                     const unsigned short* crctab, void* stream)
{
    hipLaunchKernelGGL(blk_repair_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, frames, cap, count, done_upto, synd, crctab);
    hipLaunchKernelGGL(blk_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, done_upto, count);
    return (int)hipGetLastError();
}
