// msk.hip -- demodMSK() (msk.c:67-137) + putbit() (msk.c:53-63) + the decodeAcars() framing FSM
// (acars.c:239-375) for thousands of channels on gfx950.
//
// The loop is a strict serial recurrence per channel (the PLL feeds the VCO that produced the
// sample it is fed from; the framing FSM writes MskDf / MskS back into it, acars.c:242,259,274),
// so the bulk parallelism is across channels (state resident in HBM between launches); inside a
// channel only the <= 6 mixer evaluations between two bit decisions are independent, and those are
// spread over LPC lanes (see msk_demod_kernel).  Arithmetic follows the reference's C promotions
// operation for operation (f64 VCO/PLL, f32 clock and filter, f64 divide): this translation unit
// is compiled with -ffp-contract=off so no multiply-add is fused that the reference's IEEE build
// keeps separate.
//
// SIMT scheduling: a bit is decided every 5 or 6 input samples and the per-bit work (matched
// filter, normalise, PLL, FSM) costs about as much as one sample's VCO + sincos.  Testing for a
// bit after every sample would make the wave execute the bit branch on nearly every sample (some
// lane always fires).  Instead each lane runs up to 6 samples until ITS bit fires, then all lanes
// run the bit branch together: same per-channel operation sequence, ~1/3 fewer issued
// instructions.  Lanes drift apart by a few samples; each stops at its own len.
#include <hip/hip_runtime.h>
#include "acg_internal.h"

#include "msk_common.h"

// LPC lanes cooperate on one channel ("replicated state machine, distributed sincos"): every lane
// of a group carries an identical copy of the channel's scalar state and executes the same bit
// logic, so nothing has to be broadcast; only the expensive per-sample work (f64 sin/cos + mix) of
// the <= 6 samples between two bit decisions is spread over the group's lanes, and the ring buffer
// lives in LDS where the whole group reads it back for the matched filter.  The VCO phase and the
// bit clock of those samples are first advanced sequentially (they are cheap adds and their exact
// rounding order matters), which also tells how many samples this bit period consumes.
//   LPC = 8: latency mode (few thousand channels: one sincos per lane per bit)
//   LPC = 1: throughput mode (tens of thousands of channels: no redundant work)
// Workgroup shape: the waves are independent, but placement is not: the dispatcher puts the waves of ONE
// workgroup on different SIMDs of a CU, whereas consecutive single-wave workgroups can land two to a SIMD
// while other SIMDs of the same CU stay empty (measured with profiles/probe/cumask_probe.hip), which
// doubles the per-bit time of both.  On a CU-masked stream (few channels, the demodulator owns a
// handful of CUs) four waves therefore form a workgroup; unmasked, single-wave workgroups spread over
// the whole chip and are ~3 % faster (less LDS sharing).

// ACG_MSK_STAMP (measurement build only, lib/libacarsdec_amd_stamp.so): s_memtime at the phase boundaries of
// the per-bit loop, summed per wave, written to MskArgs::stamp at the end -- the cycle breakdown of the serial
// chain (profiles/probe/msk_phase_stamps.py).  The product build compiles none of it.
#ifdef ACG_MSK_STAMP
#define STAMP_DECL unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define STAMP_DECL
#define STAMP(k) do { } while (0)
#endif

// VEC: the launch's dm rows are 16-byte aligned and len is a multiple of 32 (every in_callback launch): the refill reads
// 16 bytes per load instead of four clamped words.
// POLY (run-time flag ACG_F_PRECISE_MIXER, a verification mode like ACG_F_EXACT_FIR): the mixer's sin/cos as the < 1 ulp
// Cody-Waite + fdlibm-kernel evaluation instead of the 128-entry table + rotation (<= 2.1 ulp).  What the loop keeps are the
// float-rounded products, and those are the same for both (tests); the flag lets a maintainer see that on his own input.
// A/B builds only (profiles/probe/build_ab.py; VERDICT r04 item 7, results in profiles/LEDGER.md round 5):
//   ACG_MSK_AB_EU         the register budget of one wave per SIMD (the kernel never runs more): amdgpu_waves_per_eu(1, 1) -- the same
//                         instruction stream in higher registers, nothing alone, -6 % at 4096 channels beside the down-converter
//   ACG_MSK_AB_UNCOUNTED  round 4's loop: the window test and the wave-wide `any lane left` test before EVERY bit period
//   ACG_MSK_AB_EXPECT     branch weights on the three branches of the per-bit path (quick period: likely; bit fired: likely; framing
//                         state machine beyond its two fast paths: unlikely), so that the block placement keeps the common path
//                         fall-through
#ifdef ACG_MSK_AB_EXPECT
#define MSK_LIKELY(x) __builtin_expect(!!(x), 1)
#define MSK_UNLIKELY(x) __builtin_expect(!!(x), 0)
#else
#define MSK_LIKELY(x) (x)
#define MSK_UNLIKELY(x) (x)
#endif
#ifdef ACG_MSK_AB_EU
#define MSK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))
#else
#define MSK_KERNEL_ATTR
#endif
template <int LPC, int WPG, bool VEC, bool POLY = false>
__global__ __launch_bounds__(64 * WPG) MSK_KERNEL_ATTR void msk_demod_kernel(const MskArgs a)
{
    constexpr int CPW = 64 / LPC;                  // channels per wave
    constexpr int SPL = (6 + LPC - 1) / LPC;       // mixer samples per lane per bit period
    // dm samples per refill block = how far ahead of its use a block is requested (64 samples = 12 bit periods ~ 10 us:
    // with 32, a demodulator whose dm comes from HBM beside the streaming down-converter -- calls too large for the
    // last-level cache -- waited for its refills: 1024 channels, 36 callbacks per call: 1.04 us per bit instead of 0.89)
#ifndef ACG_MSK_WB
#define ACG_MSK_WB 64
#endif
    constexpr int WB = LPC >= 4 ? ACG_MSK_WB : 32;
    constexpr int SPB = WB / LPC;                  // dm samples per lane per refill
    constexpr int WSTR = 2 * WB + 4;               // rows stay 16-byte aligned (b128 refill stores); 4 mod 32 banks apart
    // inb[] of the wave's channels, every sample stored twice (k and k+FLEN) so that the 11 taps of
    // the matched filter are always 11 consecutive rows starting at idx: no wrap, constant offsets
    // one struct = one layout: h[] first, so that its 11 reads h[o + 12 j] are immediate offsets (< 1 KiB) from the one
    // address o * 4 (placed after the ring, every pair of reads needed its own base register)
    struct alignas(16) Lds {
        float hs[(FLEN * MFLTOVER + 1 + 3) & ~3];
        double sc[2 * ACG_SINCOS_N];               // (cos, sin)(j * 2 pi / 128): the mixer's table, 2 KiB
        float2 ring_all[WPG][3 * FLEN + 1][CPW];   // rows 0..21: inb[] twice; rows 22 and 33: where lanes without a sample write
        float win_all[WPG][CPW][WSTR];             // sliding window of dm: blocks j and j+1
    };
    __shared__ Lds lds;
    float* hs = lds.hs;

    for (int i = threadIdx.x; i < FLEN * MFLTOVER + 1; i += 64 * WPG) hs[i] = a.h[i];
    for (int i = threadIdx.x; i < 2 * ACG_SINCOS_N; i += 64 * WPG) lds.sc[i] = a.sctab[i];

    const int wv = threadIdx.x >> 6;               // the waves of a workgroup never talk to each other
    const int tid = threadIdx.x & 63;
    float2 (*ring)[CPW] = lds.ring_all[wv];
    float (*win)[WSTR] = lds.win_all[wv];
    const int slot = tid / LPC;                    // channel slot inside the wave
    const int g = tid - slot * LPC;                // lane inside the group
    const bool leader = g == 0;
    const int ch = (blockIdx.x * WPG + wv) * CPW + slot;
    const bool active = ch < a.nch;
    const int chc = active ? ch : a.nch - 1;
    AcgChan* st = a.st + chc;

    Lane L;
    L.phi = st->phi; L.df = st->df; L.lvlsum = st->lvlsum;
    L.clk = st->clk; L.bitcount = st->bitcount; L.S = st->S; L.idx = st->idx;
    L.nbits = st->nbits; L.astate = st->astate; L.blen = st->blen; L.berr = st->berr;
    L.outbits = st->outbits; L.crc0 = st->crc0; L.nbit_total = st->nbit_total;
    const long long samp0 = st->nsamp_total;
    if (leader) {
#pragma unroll
        for (int j = 0; j < FLEN; ++j) {
            const float2 x = make_float2(st->inb[2 * j], st->inb[2 * j + 1]);
            ring[j][slot] = x;
            ring[j + FLEN][slot] = x;
        }
    }

    const float* __restrict__ dm = a.dm + (size_t)chc * a.dm_pitch;
    unsigned char* txt = a.txt + (size_t)chc * 256;
    // every lane of a group stores the (identical) bit record and text byte: no exec-mask branches on the per-bit path
    // no bit log: the record goes to one scratch slot in the tail of the text buffer (bytes 248..255; blen <= 241).  A
    // store that is always there keeps the bit decision one basic block (a branch around it cost 2.7 % per bit)
    float2* bits = a.bits ? a.bits + (size_t)chc * a.bit_cap : (float2*)(txt + 248);
    const int bit_cap = a.bits ? a.bit_cap : 1;
    const int len = active ? a.len : 0;
    int nb = (a.bit_append && active) ? a.nbits_out[ch] : 0;
    int n = 0;
    unsigned int idx = L.idx;
    double p = L.phi;

    // ---- dm window: lane g owns samples [g*SPB, (g+1)*SPB) of every 32-sample block.  Blocks 0 and 1
    // go straight into the window, block 2 waits in registers; from then on, whenever consumption
    // enters block j, the registers (block j+1, requested 32 samples earlier) replace block j-1 in the
    // window and block j+2 is requested.  Global-memory latency never sits on the per-bit chain.
    float pend[SPB];
    // branch-free on purpose: any control flow around these loads makes the compiler wait for them
    // on the spot, which would put the whole HBM round trip back on the per-bit chain.  Samples at or
    // beyond len are never consumed, so out-of-range indices are simply clamped into the row.
    const int lim = a.len > 0 ? a.len - 1 : 0;
    const int limv = a.len >= SPB ? a.len - SPB : 0;
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto fetch_block = [&](int blk) {
        const int base = blk * WB + g * SPB;
        if constexpr (VEC) {
            static_assert(SPB % 4 == 0 && SPB <= 32, "refill share of a lane");
            // len is a multiple of 32 (VEC): a lane's share of a block never straddles len, so one clamp of its start does
            const f4v* src = (const f4v*)(dm + (base < limv ? base : limv));
#pragma unroll
            for (int q = 0; q < SPB; q += 4) {
                const f4v v = src[q / 4];
                pend[q] = v.x; pend[q + 1] = v.y; pend[q + 2] = v.z; pend[q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SPB; ++q) {
                const int i = base + q;
                pend[q] = dm[i < lim ? i : lim];
            }
        }
    };
    // one LDS address per lane, formed once: the two halves of the window are immediate offsets from it
    f4v* const wrow = (f4v*)&win[slot][g * SPB];
    auto store_block = [&](int blk) {
        f4v* w = wrow + (blk & 1) * (WB / 4);
#pragma unroll
        for (int q = 0; q < SPB; q += 4) w[q / 4] = f4v{pend[q], pend[q + 1], pend[q + 2], pend[q + 3]};
    };
    fetch_block(0);
    store_block(0);
    fetch_block(1);
    store_block(1);
    fetch_block(2);
    int pend_blk = 2;
    int refill_at = WB;                            // = (pend_blk - 1) * WB: consumption enters the newest block of the window
    __syncthreads();

    // few channels: this serial chain is the critical path of the whole job -> win issue arbitration;
    // many channels: the down-converter is, and these waves have slack -> stay at normal priority
    if (a.high_prio) __builtin_amdgcn_s_setprio(3);
#ifdef ACG_MSK_AB_PICK_EXEC
    unsigned long long pick_mask[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pick_mask[u] = __builtin_amdgcn_ballot_w64(g == u);
#endif
    STAMP_DECL
#ifdef ACG_MSK_STAMP
    unsigned long long stamp_iters = 0, stamp_bits = 0;
#endif

    while (__any(n < len)) {
        // ---- window upkeep (rare: once per 32 samples per channel)
        // (measured and left out: looking at the refill condition only every 8th pass, so that the channels of a wave
        //  refill together -- 8 instructions fewer per pass, 0.8 % SLOWER; dropping the per-bit counters that can be derived
        //  from the record index, 32-bit store indices, the reciprocal of s formed early, the loop-filter products formed
        //  early, a three-level phase wrap: all within +-0.5 %.  What moved the time were branches inside the serial part.)
        if (n >= refill_at && n < len) {
            store_block(pend_blk);
            ++pend_blk;
            refill_at += WB;
            fetch_block(pend_blk);
        }
        // ---- K bit periods between two looks at the window and at `is any lane left` (round 5: 0.762 -> 0.725 us per bit alone,
        // +6 % on the headline).  A period consumes <= 6 samples and reads <= 8 ahead of its start, the window holds two blocks of
        // WB samples: with 6 K <= WB - 1 the reads of K periods that start in the older block end inside the newer one, and after
        // a refill (n at most 6 K into the newer block) inside the block just stored (12 K <= 2 WB - 1).  The loop counter and
        // its branch are scalar; the per-period `v_cmp -> vcc -> branch` round trip of the old loop is gone.  Lanes past their
        // len idle through the remaining periods (cnt = 0: no sample, no bit, the mixer writes its scratch row).
#ifndef ACG_MSK_AB_UNCOUNTED
        constexpr int PERIODS = (WB - 8) / 6;          // 9 (WB 64), 4 (WB 32)
        static_assert(6 * PERIODS <= WB - 1 && 12 * PERIODS <= 2 * WB - 1, "the counted periods must stay inside the dm window");
        // Round 6: the counted loop is UNROLLED (all 9 periods straight-line, ~6 400 instructions, no spill: 106 VGPRs): no loop
        // counter / branch between periods and the tail of a period scheduled against the head of the next.  Same box, alone:
        // 0.724 -> 0.720 (3 x) -> 0.714 us per bit at 1024 channels, 0.730 -> 0.717 at 2048, 0.990 -> 0.951 at 16 384 (4 lanes per
        // channel), headline 1.426 -> 1.441 -> 1.448 M channel*Msps (profiles/r06_msk_unroll_ab.txt).  ACG_MSK_AB_UNROLL=n: A/B builds.
#define ACG_MSK_PRAGMA_(x) _Pragma(#x)
#define ACG_MSK_PRAGMA(x) ACG_MSK_PRAGMA_(x)
#ifdef ACG_MSK_AB_UNROLL
        ACG_MSK_PRAGMA(unroll ACG_MSK_AB_UNROLL)
#else
        ACG_MSK_PRAGMA(unroll)
#endif
        for (int period_ = 0; period_ < PERIODS; ++period_) {
#endif
#ifdef ACG_MSK_STAMP
        ++stamp_iters;                                                     // (periods, not looks at the window)
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // this period's mixer inputs: lane g takes samples n+g, n+g+LPC, ... (issued now, used in B)
        float in_cur[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) in_cur[j] = win[slot][(n + g + j * LPC) & (2 * WB - 1)];

        STAMP(0);                                                          // window upkeep + loop control
        // ---- A: advance VCO phase and bit clock over this bit period (replicated, sequential)
        const double s = K_VCO + L.df;                                     // msk.c:81
        const double thr = K_3PI2 - s / 2;                                 // msk.c:96
        double myp[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) myp[j] = p;
        int cnt = 0;
        bool fired = false;
        // Common case first: a bit period is 5 or 6 samples (the clock advances ~0.905 rad per
        // sample from within +-s/2 of zero to 3pi/2 - s/2), so the first four steps neither fire
        // nor hit the end of the buffer.  With s > 0 the clock is monotonic, so "none of the first
        // four fired" is decided by the fourth value alone.  Same operations, same order, fewer
        // predicated instructions; anything unusual takes the one-sample pass below.
        double p4 = p;
        float c4 = L.clk;
        double pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p4 += s;
            p4 = wrap_2pi(p4);
            c4 = (float)((double)c4 + s);
            pq[u] = p4;
        }
        // (six samples left in the buffer: the last few samples of a call go through the one-sample pass, so that the
        //  fifth step below needs no end-of-buffer predicate on the serial phase / clock chains: 3.5 % per bit)
        const bool quick = (s > 0) && !((double)c4 >= thr) && (n + 6 <= len);
        if (MSK_LIKELY(n < len && quick)) {
#ifdef ACG_MSK_AB_PICK_EXEC
                // A/B build only: the per-lane phase pick as an EXEC-masked 64-bit move (1 VALU + 2 SALU) instead of the two
                // v_cndmask_b32 the compiler makes of the select
                if constexpr (LPC == 8) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        unsigned long long sv_;
                        asm volatile("s_and_saveexec_b64 %1, %2\n\tv_mov_b64 %0, %3\n\ts_mov_b64 exec, %1"
                                     : "+v"(myp[0]), "=&s"(sv_) : "s"(pick_mask[u]), "v"(pq[u]));
                    }
                } else
#endif
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((u % LPC) == g) myp[u / LPC] = pq[u];
                p = p4;
                L.clk = c4;
                cnt = 4;
                {
                    // the fifth sample always belongs to the period (six samples are left in the buffer)
                    double pn = p + s;                                     // msk.c:82-83
                    pn = wrap_2pi(pn);
                    const float cn = (float)((double)L.clk + s);           // msk.c:95
                    p = pn;
                    L.clk = cn;
                    cnt = 5;
                    fired = (double)cn >= thr;
                    if ((4 % LPC) == g) myp[4 / LPC] = pn;
                }
#pragma unroll
                for (int u = 5; u < 6; ++u) {
                    const bool go = !fired;

                    double pn = p + s;                                     // msk.c:82-83
                    pn = wrap_2pi(pn);
                    const float cn = (float)((double)L.clk + s);           // msk.c:95
                    if (go) {
                        p = pn;
                        L.clk = cn;
                        cnt = u + 1;
                        fired = (double)cn >= thr;
                    }
                    if ((u % LPC) == g) myp[u / LPC] = pn;
                }
        } else if (n < len) {
            // anything unusual (buffer tail, a clock that is not where a locked loop keeps it): THIS channel
            // advances one sample in this pass (its bit period is then simply spread over several passes)
            // while the others do their normal period; it is back in step as soon as its bit fires
            double pn = p + s;                                             // msk.c:82-83
            pn = wrap_2pi(pn);
            const float cn = (float)((double)L.clk + s);                   // msk.c:95
            p = pn;
            L.clk = cn;
            cnt = 1;
            fired = (double)cn >= thr;
            if (g == 0) myp[0] = pn;
        }
        // ---- C0: everything of the matched filter that does not depend on this period's mixer outputs, issued before
        // them so that its LDS round trips and the f64 divide run beside the sin/cos work (msk.c:100-107):
        // the tap phase o needs only the clock (known since phase A), h[] is read-only, and with at most 6 new
        // samples per period the five oldest ring entries inb[(j + idx) % 11], j = 0..4, are already there.
        // The sum keeps the reference's order (oldest first): j = 0..4 here, j = 5..10 after the mixer.
        unsigned int idx_n = idx + (unsigned int)cnt;
        if (idx_n >= FLEN) idx_n -= FLEN;
        const float clk_f = fired ? (float)((double)L.clk - K_3PI2) : L.clk;   // msk.c:100
#ifdef ACG_MSK_AB_TAPPHASE_FAST
        // A/B build only (profiles/probe/build_ab.py; never in the product): the quotient from rcp + ONE Newton step, no
        // remainder step -- 4 instructions fewer, and no longer the correctly rounded quotient: o can differ from the
        // reference's where 12 (q + 0.5) lies within an ulp or two of an integer.  What it would buy is in profiles/LEDGER.md.
        int o;
        {
            double r_ = __builtin_amdgcn_rcp(s);
            r_ = __builtin_fma(r_, __builtin_fma(-s, r_, 1.0), r_);
            o = (int)(MFLTOVER * ((double)clk_f * r_ + 0.5));
        }
#else
        int o = (int)(MFLTOVER * (div1_rcp((double)clk_f, s) + 0.5));           // msk.c:103
#endif
        if (o > MFLTOVER) o = MFLTOVER;
        if (o < 0) o = 0;          // memory safety only: the reference indexes h[] out of bounds here
        typedef float f2v __attribute__((ext_vector_type(2)));
        // (broadcasting the odd taps out of the high word of their ds_read2 pair with v_pk_mul_f32 op_sel saves five
        //  v_mov per bit and measured 2 % SLOWER: the moves fill the latency slots of the serial v_pk_add chain)
#define MSK_TAP(j, x) (f2v{(j & 1) ? hv2[j / 2].y : hv2[j / 2].x, (j & 1) ? hv2[j / 2].y : hv2[j / 2].x} * x)
        f2v hv2[(FLEN + 1) / 2];                                           // (h[o + 12 j], h[o + 12 (j + 1)]), j even
        f2v acc = {0.f, 0.f};
        {
            const float* hp = &hs[o];
#pragma unroll
            for (int j = 0; j < FLEN; j += 2) {
                hv2[j / 2].x = hp[j * MFLTOVER];
                hv2[j / 2].y = hp[j + 1 < FLEN ? (j + 1) * MFLTOVER : j * MFLTOVER + 1];    // (last pair: the neighbour rides along unused)
            }
        }
        float2 xo[5];
        {
            const float2* rp0 = &ring[idx_n][slot];
#pragma unroll
            for (int j = 0; j < 5; ++j) xo[j] = rp0[j * CPW];
        }
        STAMP(1);                                                          // A + C0
        // ---- B: mixer for the cnt samples, spread over the group's lanes (msk.c:86-91)
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int u = g + j * LPC;
            {
                // every lane runs the sin/cos (lanes without a sample of this period on a phase that is lying around) and
                // only the ring row differs: no branch around the mixer, so it schedules as one block with the tap-phase
                // divide and the first filter taps around it (0.8 % per bit)
                double sn, cs;
#ifdef ACG_MSK_SINCOS_POLY
                sincos_2pi(myp[j], &sn, &cs);                              // (checking build: the < 1 ulp polynomial version everywhere)
#else
                if constexpr (POLY) sincos_2pi(myp[j], &sn, &cs);          // ACG_F_PRECISE_MIXER
                else sincos_tab(myp[j], lds.sc, &sn, &cs);
#endif
                const double in = (double)in_cur[j];
                unsigned int k = idx + (unsigned int)u;
                if (k >= FLEN) k -= FLEN;
                if (u >= cnt) k = 2 * FLEN;
                const float2 x = make_float2((float)(in * cs), (float)(in * (-sn)));
                ring[k][slot] = x;
                ring[k + FLEN][slot] = x;
            }
        }
        n += cnt;
        idx = idx_n;
        // the five oldest taps (read before the mixer ran).  (re, im) ride in one packed register pair: v_pk_mul_f32
        // then v_pk_add_f32 round exactly like the two scalar multiplies and adds of the reference (no fusion in this TU)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const f2v x = {xo[j].x, xo[j].y};
            acc = acc + MSK_TAP(j, x);
        }
        // one wave per block: LDS operations of a wave execute in order, so the reads below see the
        // writes above; only the compiler has to be told not to move them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        STAMP(2);                                                          // B
        // ---- C: bit decision (replicated; side effects by the group leader only)
#ifdef ACG_MSK_AB_FLAT
        // A/B build: the bit-decision part WITHOUT a branch around it -- every pass runs it and a pass without a bit (rare: the
        // one-sample passes at a call's tail or out of lock) masks its side effects -- so that the pass is one basic block and the
        // scheduler may run the decision / framing of this period (off the chain that feeds the next one) beside what follows
        const bool FB = fired;
        {
#else
        constexpr bool FB = true;
        if (MSK_LIKELY(fired)) {
#endif
#ifdef ACG_MSK_STAMP
            ++stamp_bits;
#endif
            L.clk = clk_f;                                                 // msk.c:100
            // matched filter, msk.c:103-107, second part: the taps whose samples the mixer just wrote
            {
                const float2* rp = &ring[idx][slot];
                float2 xs[FLEN - 5];
#pragma unroll
                for (int j = 5; j < FLEN; ++j) xs[j - 5] = rp[j * CPW];
#pragma unroll
                for (int j = 5; j < FLEN; ++j) {
                    const f2v x = {xs[j - 5].x, xs[j - 5].y};
                    acc = acc + MSK_TAP(j, x);
                }
            }
            float vr = acc.x, vi = acc.y;
#ifdef ACG_MSK_STAMP
            asm volatile("" : "+v"(vr), "+v"(vi));                          // the filter output exists before the stamp
#endif
            STAMP(3);                                                      // C1: tap phase (f64 divide), ring + h reads, matched filter
            // normalise, msk.c:110-113
            const float lvl = sqrtf_of_sum_of_squares((double)vr * (double)vr + (double)vi * (double)vi);   // cabsf, see fir.hip
            const double d = (double)lvl + 1e-8;
            // two IEEE quotients over one denominator: the compiler's own f64 division is rcp + two Newton steps on the
            // reciprocal, q0 = n * r, one remainder step q = fma(fma(-d, q0, n), r, q0), wrapped in div_scale / div_fixup
            // for operands near the ends of the exponent range.  Here d is in [1e-8, ~1e2] and |n| < ~1e2 (or 0), so the
            // scaling is the identity and the reciprocal can be shared: the same roundings, 8 instructions fewer.
            {
                double qr, qi;
                div2_shared_rcp((double)vr, (double)vi, d, &qr, &qi);
                vr = (float)qr;
                vi = (float)qi;
            }
            L.lvlsum += FB ? (double)(lvl * lvl / 4) : 0.0;                // (+ 0.0 is exact)
            L.bitcount += FB ? 1 : 0;
#ifdef ACG_MSK_STAMP
            asm volatile("" : "+v"(vr), "+v"(vi));
#endif
            STAMP(4);                                                      // C2: |v| (f64 sqrt), two f64 divides
            // decision + phase detector, msk.c:115-121, as sign-bit arithmetic (exact: only signs move):
            //   odd  S: vo = Im v, dphi = (vo >= 0) ? -Re v :  Re v
            //   even S: vo = Re v, dphi = (vo >= 0) ?  Im v : -Im v
            const bool odd = (L.S & 1) != 0;
            const float vo = odd ? vi : vr;
            const float ot = odd ? vr : vi;
            const unsigned int flip = ((vo >= 0) == odd) ? 0x80000000u : 0u;
            const double dphi = (double)__uint_as_float(__float_as_uint(ot) ^ flip);
            const float sv = __uint_as_float(__float_as_uint(vo) ^ ((L.S & 2u) << 30));   // msk.c:122-126
            bits[(FB && nb < bit_cap) ? nb : bit_cap - 1] = make_float2(sv, lvl);          // (no bit log: one scratch record in the text buffer's tail)
            nb += FB ? 1 : 0;
            // putbit, msk.c:53-63
            {
                unsigned int ob = (L.outbits >> 1) & 0x7fu;
                if (sv > 0) ob |= 0x80u;
                L.outbits = FB ? ob : L.outbits;
            }
            L.nbits -= FB ? 1 : 0;
            {
                // decodeAcars (acars.c:246-375) runs when nbits reaches 0.  Two cases cover nearly every call and are
                // taken without a branch: hunting for sync with no SYN in sight (acars.c:252-265, every bit of an idle
                // channel) and a plain text byte -- good parity, no terminator, room left (acars.c:303-341, every 8th
                // bit of a channel inside a block).  Everything else (sync, SOH, parity errors, ETX/ETB/DLE, CRC
                // bytes, resets: a few per block) goes through the full state machine.
                const bool ev = FB && L.nbits <= 0;
                const unsigned int r = L.outbits & 0xffu;
                const bool syn = (r == SYN) | (r == (0xffu & ~SYN));
                const bool hunt = ev & (L.astate == WSYN) & !syn;
                const bool term = (r == ETX) | (r == ETB) | (r == DLE);
                const bool plain = ev & (L.astate == TXT) & ((__popc(r) & 1) != 0) & !term & (L.blen < 240);
                txt[plain ? L.blen : 255] = (unsigned char)r;          // byte 255 of the 256-byte text buffer is scratch (blen <= 241)
                L.blen += plain ? 1 : 0;
                L.nbits = hunt ? 1 : (plain ? 8 : L.nbits);
                if (MSK_UNLIKELY(ev & !hunt & !plain)) decode_acars(L, a, ch, txt, samp0 + n - 1, leader, &st->soh32);
            }
            L.nbit_total += FB ? 1 : 0;
            L.S += FB ? 1u : 0u;
            STAMP(5);                                                      // C3: decision, bit record, putbit, framing FSM
            // PLL filter, msk.c:130 (float constants promoted to double)
            // (forming the products ahead -- 0.52 * MskDf at the top of the pass, the dphi term before decodeAcars -- and
            //  1 / s for the tap phase as soon as s exists: no measurable difference, the chain is bound by issue, not latency)
            {
                const double df_n = (double)0.52f * L.df + (1.0 - (double)0.52f) * (double)38e-4f * dphi;
                L.df = FB ? df_n : L.df;
            }
#ifdef ACG_MSK_STAMP
            asm volatile("" : "+v"(L.df));
#endif
            STAMP(6);                                                      // C4: loop filter
        }
#ifndef ACG_MSK_AB_UNCOUNTED
        }
#endif
    }
#ifdef ACG_MSK_STAMP
    if (a.stamp && tid == 0) {
        unsigned long long* o = a.stamp + (size_t)(blockIdx.x * WPG + wv) * 10;
        for (int k = 0; k < 8; ++k) o[k] = stamp_acc[k];
        o[8] = stamp_iters;
        o[9] = stamp_bits;
    }
#endif

    if (active && leader) {
        st->phi = p; st->df = L.df; st->lvlsum = L.lvlsum;
        st->clk = L.clk; st->bitcount = L.bitcount; st->S = L.S; st->idx = idx;
        st->nbits = L.nbits; st->astate = L.astate; st->blen = L.blen; st->berr = L.berr;
        st->outbits = L.outbits; st->crc0 = L.crc0; st->nbit_total = L.nbit_total;
        st->nsamp_total = samp0 + len;
#pragma unroll
        for (int j = 0; j < FLEN; ++j) {
            const float2 x = ring[j][slot];
            st->inb[2 * j] = x.x;
            st->inb[2 * j + 1] = x.y;
        }
        a.nbits_out[ch] = nb;
    }
    // the last workgroup out publishes the block-queue length of this launch to the host-mapped word of the
    // call (no copy packet on the launch chain; the host reads it after the call's event)
    if (a.snap) {
        if (WPG > 1) __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned int d = atomicAdd(a.done_ctr, 1u);
            if (d == gridDim.x - 1) {
                const unsigned int c = __hip_atomic_load(a.frame_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.snap, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}


// test hook: the device sin/cos used by the mixer, on n arguments
__global__ void sincos_selftest_kernel(const double* x, double* s, double* c, int n, const double* sctab)
{
    __shared__ double sc[2 * ACG_SINCOS_N];
    for (int i = threadIdx.x; i < 2 * ACG_SINCOS_N; i += blockDim.x) sc[i] = sctab[i];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_tab(x[i], sc, &s[i], &c[i]);
}

// test hook: the shared-reciprocal quotients of the normalisation against the compiler's IEEE division
__global__ void div2_selftest_kernel(const double* n0, const double* n1, const double* d, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    div2_shared_rcp(n0[i], n1[i], d[i], &a, &b);
    out[8 * i + 0] = a;
    out[8 * i + 1] = b;
    out[8 * i + 2] = n0[i] / d[i];
    out[8 * i + 3] = n1[i] / d[i];
    const double x = n0[i] * n0[i] + n1[i] * n1[i];
    out[8 * i + 4] = sqrt_rn_midrange(x);
    out[8 * i + 5] = __dsqrt_rn(x);
    out[8 * i + 6] = div1_rcp(n0[i], d[i]);
    out[8 * i + 7] = x;
}

extern "C" int acg_launch_div2_selftest(const double* n0, const double* n1, const double* d, double* out, int n, void* stream)
{
    hipLaunchKernelGGL(div2_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n0, n1, d, out, n);
    return (int)hipGetLastError();
}

extern "C" int acg_launch_sincos_selftest(const double* x, double* s, double* c, int n, const double* sctab, void* stream)
{
    hipLaunchKernelGGL(sincos_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, s, c, n, sctab);
    return (int)hipGetLastError();
}

extern "C" int acg_tune_has(const char* name);
extern "C" int acg_tune_get(const char* name, int dflt);

extern "C" int acg_launch_msk(const MskArgs* a, int lpc, void* stream)
{
    const int cpw = 64 / lpc;
    const int wpg = (a->waves_per_group == 4 && lpc >= 4) ? 4 : 1;
    const unsigned int waves = (unsigned int)((a->nch + cpw - 1) / cpw);
    const unsigned int grid = (waves + wpg - 1) / wpg;
    const dim3 blk(64 * wpg);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = ((uintptr_t)a->dm % 16 == 0) && (a->dm_pitch % 4 == 0) && (a->len % 32 == 0) && !acg_tune_has("ACG_MSK_NOVEC") && !a->precise_mixer;
    // Round 6: the in_callback-shaped launches go to msk_lean.hip (the framing state machine off the per-bit path: same bits,
    // blocks and state); ACG_MSK_NOLEAN=1 keeps this file's kernel for same-process A/B.  The stamp build measures this file's kernel.
#ifndef ACG_MSK_STAMP
    // (8 lanes per channel: 0.716 -> 0.643 us per bit alone, the headline 1.451 -> 1.574 M channel*Msps.  4 lanes per channel -- 16 channels
    //  share a wave's segments, two mixer evaluations per lane and period -- gain nothing alone with the bit log on and 0.5-1.4 % beside
    //  the down-converter in the same process (16 384 channels: streaming kernel +0.6 %, matrix-pipe kernel +0.6 ... +1.4 %, +4.4 % without
    //  the log; GPU calls 25, 27, 28): taken too; ACG_MSK_LEAN4=0 keeps this file's kernel for them.)
    if (vec && (lpc == 8 || (lpc == 4 && acg_tune_get("ACG_MSK_LEAN4", 1))) && !acg_tune_get("ACG_MSK_NOLEAN", 0)) return acg_launch_msk_lean(a, lpc, wpg, grid, stream);
#endif
    // (the verification mode is instantiated for the scalar-refill shape only: six more kernels, not twelve)
#define MSK_LAUNCH(L_, W_) do { if (a->precise_mixer) hipLaunchKernelGGL((msk_demod_kernel<L_, W_, false, true>), dim3(grid), blk, 0, s, *a); \
                                else if (vec) hipLaunchKernelGGL((msk_demod_kernel<L_, W_, true>), dim3(grid), blk, 0, s, *a); \
                                else hipLaunchKernelGGL((msk_demod_kernel<L_, W_, false>), dim3(grid), blk, 0, s, *a); } while (0)
    switch (lpc * 16 + wpg) {
    case 1 * 16 + 1: MSK_LAUNCH(1, 1); break;
    case 2 * 16 + 1: MSK_LAUNCH(2, 1); break;
    case 4 * 16 + 1: MSK_LAUNCH(4, 1); break;
    case 8 * 16 + 1: MSK_LAUNCH(8, 1); break;
    case 4 * 16 + 4: MSK_LAUNCH(4, 4); break;
    case 8 * 16 + 4: MSK_LAUNCH(8, 4); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef MSK_LAUNCH
    return (int)hipGetLastError();
}
