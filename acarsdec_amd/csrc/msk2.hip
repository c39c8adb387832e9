// msk2.hip -- demodMSK() + putbit() + decodeAcars() (msk.c:53-137, acars.c:239-375) for FEW channels (<= 2048), where the
// job is the per-bit latency of one channel: the instruction stream of msk_demod_kernel (msk.hip) split over TWO waves.
//
// Why: that kernel issues ~330 instructions per bit period from one wave, and a wave on gfx950 issues one instruction per
// ~5 cycles whatever it is -- the loop is issue-bound on its one SIMD while three quarters of the chip's SIMDs
// idle.  About 40 % of the stream is not on the true recurrence (df -> VCO phase -> mixer -> matched filter -> normalise ->
// decision -> df): the bit clock, the tap phase, the dm window upkeep, the bit record, putbit and the framing state machine.
//   wave M ("mixer"):  s = f(df); six VCO phase steps; sin/cos + mix of this period's samples (one per lane); ring write;
//                      matched filter; |v|, the two quotients; decision and phase detector; loop filter.
//   wave H ("helper"): the clock steps (-> samples in this period, bit fired, tap phase o); the dm window (global loads ->
//                      LDS); per bit: level sums, bit record, putbit, decodeAcars.
// Both waves sit in one workgroup on different SIMDs of a CU and meet at three workgroup barriers per bit period; what they
// hand each other goes through 32 bytes of LDS per channel:
//   B1  M has published {df after this bit's loop filter, soft bit vo, level}            -> H starts the next period's clock
//   B2  H has published {samples in the period, fired, tap phase o}                      -> M knows where its mixer outputs go
//   B3  H has published the framing VERDICT for this bit                                 -> M can run the loop filter
// The verdict is the one place where the framing state machine writes back into the loop: `MskDf = 0` (acars.c:242) happens
// inside putbit() -> decodeAcars(), BEFORE the loop filter of the same bit (msk.c:130).  Whether it happens depends on the
// seven bits H already holds and on the ONE bit M is about to decide, so H publishes ahead of time what would happen for
// either value, and M picks with the sign of its soft bit: one select on M's chain instead of a round trip.  (MskS ^= 2,
// acars.c:259,274, only changes the sign under which the bit enters putbit -- H's business; the phase detector looks at MskS & 1,
// which M tracks itself.)  H runs putbit / decodeAcars for bit i while M works on bit i + 1.
//
// Arithmetic: every operation of the reference, in the reference's order and roundings, exactly as in msk.hip (this TU is built
// with -ffp-contract=off as well): the two kernels are bit-identical in bits, state and blocks (tests/test_gpu_parity.py).
// The main loop handles whole bit periods while a channel has at least six samples left; the last <= 5 samples of a launch
// (a period that straddles two calls) go through a plain per-sample pass on wave M after the two waves have merged their
// halves of the state -- demodMSK() literally, one sample at a time.
#include <hip/hip_runtime.h>
#include "acg_internal.h"
#include "msk_common.h"

// the mixer's sin/cos: table + rotation, or (checking build, -DACG_MSK_SINCOS_POLY) the < 1 ulp polynomial version -- as in msk.hip
#ifdef ACG_MSK_SINCOS_POLY
#define MIXER_SINCOS(x, tab, sn, cs) sincos_2pi((x), (sn), (cs))
#else
#define MIXER_SINCOS(x, tab, sn, cs) sincos_tab((x), (tab), (sn), (cs))
#endif

namespace {

constexpr int LPC = 8;                         // lanes per channel
constexpr int CPW = 64 / LPC;                  // channels per wave pair
constexpr int WB = 64;                         // dm samples per refill block
constexpr int SPB = WB / LPC;                  // ... per lane
constexpr int WSTR = 2 * WB + 4;

// what the two waves hand each other, per channel
struct alignas(16) Msg {
    double df;          // M -> H (B1): MskDf after this bit's loop filter
    float vo, lvl;      // M -> H (B1): soft bit before the MskS & 2 sign (msk.c:115-121), |v| (msk.c:110)
    int per;            // H -> M (B2): samples in this period (bits 0-2) | fired << 3 | tap phase o << 4
    int verdict;        // H -> M (B3): decodeAcars would reset the loop on this bit if vo == 0 (bit 0), vo < 0 (bit 1), vo > 0 (bit 2)
    int pad[2];
};

// H's half of the channel state, handed to M for the per-sample tail
struct alignas(8) HState {
    double lvlsum;
    long long nbit_total;
    float clk;
    int bitcount;
    unsigned int S;
    int nbits, astate, blen, berr;
    unsigned int outbits, crc0;
    int nb;
};

struct alignas(16) PairLds {
    float2 ring[3 * FLEN + 1][CPW];            // rows 0..21: inb[] twice (no wrap in the filter); 22: where lanes without a sample write
    float win[CPW][WSTR];                      // sliding window of dm: blocks j and j+1
    Msg msg[CPW];
    HState hst[CPW];
};

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// the bit decision of msk.c:103-130 on a filled ring: shared by wave M's main loop and the per-sample tail
struct Decision {
    float vo;           // soft bit before the MskS & 2 sign
    float lvl;
    double dphi;
};

__device__ __forceinline__ Decision decide(const float2 (*ring)[CPW], int slot, unsigned int idx, const float* hs, int o, bool odd)
{
    // matched filter, msk.c:103-107: v = sum_j h[o + 12 j] * inb[(j + idx) % 11], j ascending, products and sums rounded separately
    const float* hp = &hs[o];
    const float2* rp = &ring[idx][slot];
    float hv[FLEN];
    float2 xs[FLEN];
#pragma unroll
    for (int j = 0; j < FLEN; ++j) {
        hv[j] = hp[j * MFLTOVER];
        xs[j] = rp[j * CPW];
    }
    f2v acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FLEN; ++j) {
        const f2v x = {xs[j].x, xs[j].y};
        const f2v t = {hv[j], hv[j]};
        acc = acc + t * x;
    }
    float vr = acc.x, vi = acc.y;
    // normalise, msk.c:110-111
    Decision d;
    d.lvl = sqrtf_of_sum_of_squares((double)vr * (double)vr + (double)vi * (double)vi);
    const double den = (double)d.lvl + 1e-8;
    double qr, qi;
    div2_shared_rcp((double)vr, (double)vi, den, &qr, &qi);
    vr = (float)qr;
    vi = (float)qi;
    // decision + phase detector, msk.c:115-121 (sign-bit arithmetic, see msk.hip)
    d.vo = odd ? vi : vr;
    const float ot = odd ? vr : vi;
    const unsigned int flip = ((d.vo >= 0) == odd) ? 0x80000000u : 0u;
    d.dphi = (double)__uint_as_float(__float_as_uint(ot) ^ flip);
    return d;
}

// ACG_MSK_STAMP (measurement build only): s_memtime around every barrier of the per-period loop, summed per wave:
// [0] B1 -> B2 work, [1] wait at B2, [2] B2 -> B3 work, [3] wait at B3, [4] B3 -> B1 work, [5] wait at B1; [8] periods
#ifdef ACG_MSK_STAMP
#define STAMP2_DECL unsigned long long st2_acc[6] = {0, 0, 0, 0, 0, 0}, st2_n = 0, st2_prev = __builtin_amdgcn_s_memtime();
#define STAMP2(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st2_acc[k] += t_ - st2_prev; st2_prev = t_; } while (0)
#define STAMP2_WRITE() do { if (a.stamp && tid == 0) { unsigned long long* o_ = a.stamp + (size_t)(blockIdx.x * 2 * PAIRS + wv) * 10; \
        for (int k_ = 0; k_ < 6; ++k_) o_[k_] = st2_acc[k_]; o_[6] = 0; o_[7] = 0; o_[8] = st2_n; o_[9] = st2_n; } } while (0)
#else
#define STAMP2_DECL
#define STAMP2(k) do { } while (0)
#define STAMP2_WRITE() do { } while (0)
#endif

// The per-period barriers only order LDS traffic between the waves of the pair: wait for this wave's LDS operations, then
// the hardware barrier.  (__syncthreads() also waits for the wave's GLOBAL operations -- wave H's bit-record and text stores,
// its window loads: hundreds of cycles of store acknowledgement per period that the other wave would spend waiting.)
__device__ __forceinline__ void pair_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// what decodeAcars (acars.c:246-375) would do to the loop if the byte register became r in this state: reset_acars() or not
__device__ __forceinline__ bool would_reset(int astate, int blen, int berr, unsigned int r)
{
    const bool syn = (r == SYN) | (r == (0xffu & ~SYN));
    const bool par_bad = (__popc(r) & 1) == 0;
    const bool term = (r == ETX) | (r == ETB);
    const int bl = blen + 1;
    const bool txt_reset = (par_bad & (berr + 1 > MAXPERR + 1)) |
                           (!(par_bad & (berr + 1 > MAXPERR + 1)) & !term & !((bl > 20) & (r == DLE)) & (bl > 240));
    return (astate == SYN2) ? !syn : (astate == SOH1) ? (r != SOH) : (astate == TXT) ? txt_reset :
           (astate == WSYN || astate == CRC1 || astate == CRC2) ? false : true;          // END and anything else: acars.c:370-373
}

}  // namespace

// PAIRS wave pairs per workgroup (1 or 2): wave 2 q is pair q's M, wave 2 q + 1 its H.  On a CU that holds one such
// workgroup every wave has a SIMD to itself.
// VEC: the launch's dm rows are 16-byte aligned and len is a multiple of 32 (every in_callback launch), as in msk.hip.
template <int PAIRS, bool VEC>
__global__ __launch_bounds__(128 * PAIRS) void msk_demod2_kernel(const MskArgs a)
{
    struct alignas(16) Lds {
        float hs[(FLEN * MFLTOVER + 1 + 3) & ~3];
        double sc[2 * ACG_SINCOS_N];
        PairLds pr[PAIRS];
        int cont[2 * PAIRS];                     // per wave: some channel of mine still has a whole period ahead
    };
    __shared__ Lds lds;
    for (int i = threadIdx.x; i < FLEN * MFLTOVER + 1; i += 128 * PAIRS) lds.hs[i] = a.h[i];
    for (int i = threadIdx.x; i < 2 * ACG_SINCOS_N; i += 128 * PAIRS) lds.sc[i] = a.sctab[i];

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wv >> 1;
    const bool is_m = (wv & 1) == 0;
    const int tid = threadIdx.x & 63;
    PairLds& P = lds.pr[pair];
    const int slot = tid / LPC;
    const int g = tid - slot * LPC;
    const bool leader = g == 0;
    const int ch = (blockIdx.x * PAIRS + pair) * CPW + slot;
    const bool active = ch < a.nch;
    const int chc = active ? ch : a.nch - 1;
    AcgChan* st = a.st + chc;
    const float* __restrict__ dm = a.dm + (size_t)chc * a.dm_pitch;
    unsigned char* txt = a.txt + (size_t)chc * 256;
    const int len = active ? a.len : 0;
    const long long samp0 = st->nsamp_total;
    Msg* mg = &P.msg[slot];

    if (a.high_prio) __builtin_amdgcn_s_setprio(3);
    // All waves of the workgroup pass the same barriers, so all of them run the same number of periods: each wave posts
    // whether it has work left just before B1, and everybody goes on while anybody has (a wave without work runs empty periods).
    auto cont_post = [&](bool more) {
        const int f = __any(more) ? 1 : 0;
        if (tid == 0) lds.cont[wv] = f;
    };
    auto cont_any = [&]() -> bool {
        int f = 0;
#pragma unroll
        for (int k = 0; k < 2 * PAIRS; ++k) f |= lds.cont[k];
        return f != 0;
    };

    if (is_m) {
        // =============================================================================================== wave M
        double p = st->phi, df = st->df;
        unsigned int idx = st->idx;
        bool odd = (st->S & 1u) != 0;
        if (leader) {
#pragma unroll
            for (int j = 0; j < FLEN; ++j) {
                const float2 x = make_float2(st->inb[2 * j], st->inb[2 * j + 1]);
                P.ring[j][slot] = x;
                P.ring[j + FLEN][slot] = x;
            }
        }
        int n = 0;
        cont_post(n + 6 <= len);
        __syncthreads();                                                   // P0: tables, ring, window (H) are in place
        STAMP2_DECL
        while (cont_any()) {
#ifdef ACG_MSK_STAMP
            ++st2_n;
#endif
            const bool act = n + 6 <= len;
            // this period's mixer input: lane g takes sample n + g (msk.c:86)
            const float in = P.win[slot][(n + g) & (2 * WB - 1)];
            // VCO, msk.c:81-83: six steps, every lane keeps the phase of ITS sample (lane g: after g + 1 steps)
            const double s = K_VCO + df;
            double q = p, myp = p, p5 = p, p6 = p;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                q = wrap_2pi(q + s);
                if (u == g) myp = q;
                if (u == 4) p5 = q;
                if (u == 5) p6 = q;
            }
            double sn, cs;
            MIXER_SINCOS(myp, lds.sc, &sn, &cs);
            const double ind = (double)in;
            float2 x = make_float2((float)(ind * cs), (float)(ind * (-sn)));                    // msk.c:90
            // (the mixer output exists BEFORE the barrier: this interval is where wave H computes the clock steps and the tap
            //  phase, about as long as the phase chain + sin/cos; left alone the scheduler sinks the sin/cos behind the barrier,
            //  into the interval that is M's longest)
            asm volatile("" : "+v"(x.x), "+v"(x.y));
            __builtin_amdgcn_sched_barrier(0);
            STAMP2(0);
            pair_barrier();                                                 // B2: H's {samples, fired, o} of this period
            STAMP2(1);
            const int per = act ? mg->per : 0;
            const int cnt = per & 7;
            const bool fired = (per & 8) != 0;
            const int o = (per >> 4) & 15;
            {
                unsigned int k = idx + (unsigned int)g;
                if (k >= FLEN) k -= FLEN;
                if (g >= cnt) k = 2 * FLEN;                                // not a sample of this period
                P.ring[k][slot] = x;
                P.ring[k + FLEN][slot] = x;
            }
            // the phase after cnt steps: 5 or 6 in a locked loop; anything else (the first period of a launch that began in
            // the previous call) is picked out of the lane that holds it
            double pn = cnt == 6 ? p6 : (cnt == 5 ? p5 : p);               // (0: a channel that has no whole period left)
            if (__any((cnt > 0) & (cnt < 5))) {
                const int src = (tid & ~(LPC - 1)) + (cnt > 0 ? cnt - 1 : 0);
                const double pl = __shfl(myp, src, 64);
                pn = ((cnt > 0) & (cnt < 5)) ? pl : pn;
            }
            p = pn;
            idx += (unsigned int)cnt;
            if (idx >= FLEN) idx -= FLEN;
            n += cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // the ring rows just written are read back below
            __builtin_amdgcn_wave_barrier();
            double dphi = 0.0;
            float vo = 0.f;
            if (fired) {
                const Decision d = decide(P.ring, slot, idx, lds.hs, o, odd);
                vo = d.vo;
                dphi = d.dphi;
                mg->vo = d.vo;
                mg->lvl = d.lvl;
                odd = !odd;                                                // MskS++ (msk.c:127); MskS ^= 2 leaves bit 0 alone
            }
            STAMP2(2);
            pair_barrier();                                                 // B3: H's verdict for this bit
            STAMP2(3);
            if (fired) {
                const int c = vo > 0 ? 2 : (vo < 0 ? 1 : 0);
                const bool reset = ((mg->verdict >> c) & 1) != 0;          // acars.c:242, inside putbit, before msk.c:130
                const double dfb = reset ? 0.0 : df;
                df = (double)0.52f * dfb + (1.0 - (double)0.52f) * (double)38e-4f * dphi;       // msk.c:130
            }
            mg->df = df;
            cont_post(n + 6 <= len);
            STAMP2(4);
            pair_barrier();                                                 // B1: H may start the next period
            STAMP2(5);
        }
        STAMP2_WRITE();
        __syncthreads();                                                   // T1: H's half of the state is in P.hst
        // ---- the last <= 5 samples of the launch: demodMSK() one sample at a time (msk.c:73-131), state merged
        Lane L;
        {
            const HState& h = P.hst[slot];
            L.phi = p; L.df = df; L.lvlsum = h.lvlsum; L.clk = h.clk; L.bitcount = h.bitcount; L.S = h.S; L.idx = idx;
            L.nbits = h.nbits; L.astate = h.astate; L.blen = h.blen; L.berr = h.berr; L.outbits = h.outbits; L.crc0 = h.crc0;
            L.nbit_total = h.nbit_total;
        }
        int nb = P.hst[slot].nb;
        float2* bits = a.bits ? a.bits + (size_t)chc * a.bit_cap : nullptr;
        while (__any(n < len)) {
            if (n < len) {
                const float in = dm[n];
                const double s = K_VCO + L.df;
                p = wrap_2pi(p + s);                                       // msk.c:82-83
                double sn, cs;
                MIXER_SINCOS(p, lds.sc, &sn, &cs);
                const double ind = (double)in;
                const float2 x = make_float2((float)(ind * cs), (float)(ind * (-sn)));
                P.ring[idx][slot] = x;                                     // (all lanes of the group write the same value)
                P.ring[idx + FLEN][slot] = x;
                idx = idx + 1 == FLEN ? 0 : idx + 1;
                ++n;
                L.clk = (float)((double)L.clk + s);                        // msk.c:95
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if ((double)L.clk >= K_3PI2 - s / 2) {                     // msk.c:96
                    L.clk = (float)((double)L.clk - K_3PI2);               // msk.c:100
                    int o = (int)(MFLTOVER * (div1_rcp((double)L.clk, s) + 0.5));               // msk.c:103
                    if (o > MFLTOVER) o = MFLTOVER;
                    if (o < 0) o = 0;
                    const Decision d = decide(P.ring, slot, idx, lds.hs, o, (L.S & 1u) != 0);
                    L.lvlsum += (double)(d.lvl * d.lvl / 4);               // msk.c:112-113
                    L.bitcount++;
                    const float sv = __uint_as_float(__float_as_uint(d.vo) ^ ((L.S & 2u) << 30));   // msk.c:122-126
                    if (bits) bits[nb < a.bit_cap ? nb : a.bit_cap - 1] = make_float2(sv, d.lvl);
                    ++nb;
                    L.outbits = (L.outbits >> 1) & 0x7fu;                  // putbit, msk.c:53-63
                    if (sv > 0) L.outbits |= 0x80u;
                    L.nbits--;
                    if (L.nbits <= 0) decode_acars(L, a, ch, txt, samp0 + n - 1, leader, &st->soh32);
                    L.nbit_total++;
                    L.S++;
                    L.df = (double)0.52f * L.df + (1.0 - (double)0.52f) * (double)38e-4f * d.dphi;   // msk.c:130
                }
            }
        }
        if (active && leader) {
            st->phi = p; st->df = L.df; st->lvlsum = L.lvlsum;
            st->clk = L.clk; st->bitcount = L.bitcount; st->S = L.S; st->idx = idx;
            st->nbits = L.nbits; st->astate = L.astate; st->blen = L.blen; st->berr = L.berr;
            st->outbits = L.outbits; st->crc0 = L.crc0; st->nbit_total = L.nbit_total;
            st->nsamp_total = samp0 + len;
#pragma unroll
            for (int j = 0; j < FLEN; ++j) {
                const float2 x = P.ring[j][slot];
                st->inb[2 * j] = x.x;
                st->inb[2 * j + 1] = x.y;
            }
            a.nbits_out[ch] = nb;
        }
    } else {
        // =============================================================================================== wave H
        Lane L;
        L.phi = 0; L.df = st->df; L.lvlsum = st->lvlsum;
        L.clk = st->clk; L.bitcount = st->bitcount; L.S = st->S; L.idx = 0;
        L.nbits = st->nbits; L.astate = st->astate; L.blen = st->blen; L.berr = st->berr;
        L.outbits = st->outbits; L.crc0 = st->crc0; L.nbit_total = st->nbit_total;
        float2* bits = a.bits ? a.bits + (size_t)chc * a.bit_cap : (float2*)(txt + 248);     // (no bit log: scratch slot, see msk.hip)
        const int bit_cap = a.bits ? a.bit_cap : 1;
        int nb = (a.bit_append && active) ? a.nbits_out[ch] : 0;
        // ---- dm window (as msk.hip): lane g owns samples [g * SPB, (g + 1) * SPB) of every 64-sample block
        float pend[SPB];
        const int lim = a.len > 0 ? a.len - 1 : 0;
        const int limv = a.len >= SPB ? a.len - SPB : 0;
        auto fetch_block = [&](int blk) {
            const int base = blk * WB + g * SPB;
            if constexpr (VEC) {
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
        f4v* const wrow = (f4v*)&P.win[slot][g * SPB];
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
        int refill_at = WB;
        int n = 0;
        double df = L.df;
        bool prev_fired = false;                 // the period before this one decided a bit whose putbit is still to run
        long long prev_end = 0;                  // sample index of that bit
        float pvo = 0.f, plvl = 0.f;             // ... its soft bit and level (read right after B1: M overwrites them after B2)
        cont_post(n + 6 <= len);
        __syncthreads();                                                   // P0
        STAMP2_DECL
        while (cont_any()) {
#ifdef ACG_MSK_STAMP
            ++st2_n;
#endif
            const bool act = n + 6 <= len;
            // ---- bit clock of this period, msk.c:95-100 (f32 accumulate, f64 compare) and the tap phase, msk.c:103
            const double s = K_VCO + df;
            const double thr = K_3PI2 - s / 2;
            float c = L.clk;
            int cnt = 0;
            bool fired = false;
            {
                float c4 = c;
#pragma unroll
                for (int u = 0; u < 4; ++u) c4 = (float)((double)c4 + s);
                // locked loop: the clock is monotonic (s > 0) and the first four steps cannot fire
                const bool quick = (s > 0) & !((double)c4 >= thr);
                if (__all(quick | !act)) {
                    const float c5 = (float)((double)c4 + s);
                    const float c6 = (float)((double)c5 + s);
                    const bool f5 = (double)c5 >= thr;
                    c = f5 ? c5 : c6;
                    cnt = f5 ? 5 : 6;
                    fired = f5 | ((double)c6 >= thr);
                } else {
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const float cn = (float)((double)c + s);
                        const bool go = !fired;
                        c = go ? cn : c;
                        cnt = go ? u + 1 : cnt;
                        fired = go ? ((double)cn >= thr) : fired;
                    }
                }
            }
            if (!act) { cnt = 0; fired = false; c = L.clk; }
            const float clk_f = fired ? (float)((double)c - K_3PI2) : c;   // msk.c:100
            int o = (int)(MFLTOVER * (div1_rcp((double)clk_f, s) + 0.5));  // msk.c:103
            if (o > MFLTOVER) o = MFLTOVER;
            if (o < 0) o = 0;
            L.clk = clk_f;
            mg->per = cnt | (fired ? 8 : 0) | (o << 4);
            STAMP2(0);
            pair_barrier();                                                 // B2
            STAMP2(1);
            // ---- the PREVIOUS period's bit: level sums, bit record, putbit, decodeAcars (msk.c:112-126, 53-63; acars.c:246-375)
            if (prev_fired) {
                const float lvl = plvl, vo = pvo;
                L.lvlsum += (double)(lvl * lvl / 4);
                L.bitcount++;
                const float sv = __uint_as_float(__float_as_uint(vo) ^ ((L.S & 2u) << 30));
                bits[nb < bit_cap ? nb : bit_cap - 1] = make_float2(sv, lvl);
                ++nb;
                L.outbits = (L.outbits >> 1) & 0x7fu;
                if (sv > 0) L.outbits |= 0x80u;
                L.nbits--;
                {
                    const bool ev = L.nbits <= 0;
                    const unsigned int r = L.outbits & 0xffu;
                    const bool syn = (r == SYN) | (r == (0xffu & ~SYN));
                    const bool hunt = ev & (L.astate == WSYN) & !syn;
                    const bool term = (r == ETX) | (r == ETB) | (r == DLE);
                    const bool plain = ev & (L.astate == TXT) & ((__popc(r) & 1) != 0) & !term & (L.blen < 240);
                    txt[plain ? L.blen : 255] = (unsigned char)r;
                    L.blen += plain ? 1 : 0;
                    L.nbits = hunt ? 1 : (plain ? 8 : L.nbits);
                    if (ev & !hunt & !plain) decode_acars(L, a, ch, txt, samp0 + prev_end - 1, leader, &st->soh32);
                }
                L.nbit_total++;
                L.S++;
            }
            // ---- verdict for THIS period's bit: would decodeAcars reset the loop (MskDf = 0) for either value of the bit?
            {
                int v = 0;
                if (fired & (L.nbits <= 1) & (L.astate != WSYN)) {
                    const unsigned int r0 = (L.outbits >> 1) & 0x7fu, r1 = r0 | 0x80u;
                    const bool z = would_reset(L.astate, L.blen, L.berr, r0);      // bit 0
                    const bool w = would_reset(L.astate, L.blen, L.berr, r1);      // bit 1
                    // bit = (sv > 0), sv = vo with its sign flipped when MskS & 2 (msk.c:122-126); vo == 0 is bit 0 either way
                    const bool inv = (L.S & 2u) != 0;
                    v = (z ? 1 : 0) | ((inv ? w : z) ? 2 : 0) | ((inv ? z : w) ? 4 : 0);
                }
                mg->verdict = v;
            }
            prev_fired = fired;
            n += cnt;
            prev_end = n;
            STAMP2(2);
            pair_barrier();                                                 // B3
            STAMP2(3);
            // ---- dm window upkeep (once per 64 samples per channel)
            if (n >= refill_at && n < len) {
                store_block(pend_blk);
                ++pend_blk;
                refill_at += WB;
                fetch_block(pend_blk);
            }
            cont_post(n + 6 <= len);
            STAMP2(4);
            pair_barrier();                                                 // B1: M's df for the next period, and the bit it just decided
            STAMP2(5);
            df = mg->df;
            pvo = mg->vo;
            plvl = mg->lvl;
        }
        STAMP2_WRITE();
        // the last period's bit
        if (prev_fired) {
            const float lvl = plvl, vo = pvo;
            L.lvlsum += (double)(lvl * lvl / 4);
            L.bitcount++;
            const float sv = __uint_as_float(__float_as_uint(vo) ^ ((L.S & 2u) << 30));
            bits[nb < bit_cap ? nb : bit_cap - 1] = make_float2(sv, lvl);
            ++nb;
            L.outbits = (L.outbits >> 1) & 0x7fu;
            if (sv > 0) L.outbits |= 0x80u;
            L.nbits--;
            if (L.nbits <= 0) decode_acars(L, a, ch, txt, samp0 + prev_end - 1, leader, &st->soh32);
            L.nbit_total++;
            L.S++;
        }
        if (leader) {
            HState& h = P.hst[slot];
            h.lvlsum = L.lvlsum; h.nbit_total = L.nbit_total; h.clk = L.clk; h.bitcount = L.bitcount; h.S = L.S;
            h.nbits = L.nbits; h.astate = L.astate; h.blen = L.blen; h.berr = L.berr; h.outbits = L.outbits; h.crc0 = L.crc0;
            h.nb = nb;
        }
        __syncthreads();                                                   // T1
    }
    // the last workgroup out publishes the block-queue length of this launch (see msk.hip)
    __syncthreads();                                                       // T2: M's tail (it may queue blocks) is done
    if (a.snap && threadIdx.x == 0) {
        __threadfence();
        const unsigned int d = atomicAdd(a.done_ctr, 1u);
        if (d == gridDim.x - 1) {
            const unsigned int c = __hip_atomic_load(a.frame_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.snap, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int acg_tune_has(const char* name);

extern "C" int acg_launch_msk2(const MskArgs* a, int pairs_per_group, void* stream)
{
    const unsigned int npair = (unsigned int)((a->nch + CPW - 1) / CPW);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = ((uintptr_t)a->dm % 16 == 0) && (a->dm_pitch % 4 == 0) && (a->len % 32 == 0) && !acg_tune_has("ACG_MSK_NOVEC");
    if (pairs_per_group == 2) {
        if (vec) hipLaunchKernelGGL((msk_demod2_kernel<2, true>), dim3((npair + 1) / 2), dim3(256), 0, s, *a);
        else hipLaunchKernelGGL((msk_demod2_kernel<2, false>), dim3((npair + 1) / 2), dim3(256), 0, s, *a);
    } else {
        if (vec) hipLaunchKernelGGL((msk_demod2_kernel<1, true>), dim3(npair), dim3(128), 0, s, *a);
        else hipLaunchKernelGGL((msk_demod2_kernel<1, false>), dim3(npair), dim3(128), 0, s, *a);
    }
    return (int)hipGetLastError();
}
