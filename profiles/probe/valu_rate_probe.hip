// valu_rate_probe: what one wave pays per instruction on gfx950, by instruction kind -- the cost model for the
// demodulator's per-bit loop (one wave per SIMD, a serial chain: msk.hip).  For every kind two numbers, in shader
// cycles per instruction (s_memtime around a loop of 64 x 16 instructions, one wave on an otherwise idle chip):
//   dep    a chain where every instruction reads the previous result (issue + result latency)
//   indep  four interleaved chains (issue rate of a single wave)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP4(x) x x x x

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)r_, __LINE__); return 1; } } while (0)

constexpr int ITERS = 64;

// KIND: an asm template with %0 = the chained register (read and written), %1 = a second operand.
#define PROBE_F64(NAME, ASM1)                                                                                              \
    __global__ void NAME##_dep(unsigned long long* t, double* sink, double x0, double y0)                                  \
    {                                                                                                                      \
        double a = x0 + threadIdx.x * 1e-9, y = y0;                                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        for (int i = 0; i < ITERS; ++i) { asm volatile(REP16(ASM1 "\n") : "+v"(a) : "v"(y)); }                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                                              \
        sink[threadIdx.x] = a;                                                                                             \
    }                                                                                                                      \
    __global__ void NAME##_indep(unsigned long long* t, double* sink, double x0, double y0)                                \
    {                                                                                                                      \
        double a = x0 + threadIdx.x * 1e-9, b = a + 1e-3, c = a + 2e-3, d = a + 3e-3, y = y0;                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        for (int i = 0; i < ITERS; ++i) {                                                                                  \
            REP4(asm volatile(ASM1 : "+v"(a) : "v"(y)); asm volatile(ASM1 : "+v"(b) : "v"(y));                             \
                 asm volatile(ASM1 : "+v"(c) : "v"(y)); asm volatile(ASM1 : "+v"(d) : "v"(y));)                            \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                                              \
        sink[threadIdx.x] = a + b + c + d;                                                                                 \
    }

#define PROBE_F32(NAME, ASM1)                                                                                              \
    __global__ void NAME##_dep(unsigned long long* t, double* sink, double x0, double y0)                                  \
    {                                                                                                                      \
        float a = (float)x0 + threadIdx.x * 1e-6f, y = (float)y0;                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        for (int i = 0; i < ITERS; ++i) { asm volatile(REP16(ASM1 "\n") : "+v"(a) : "v"(y)); }                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                                              \
        sink[threadIdx.x] = a;                                                                                             \
    }                                                                                                                      \
    __global__ void NAME##_indep(unsigned long long* t, double* sink, double x0, double y0)                                \
    {                                                                                                                      \
        float a = (float)x0 + threadIdx.x * 1e-6f, b = a + 1e-3f, c = a + 2e-3f, d = a + 3e-3f, y = (float)y0;              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        for (int i = 0; i < ITERS; ++i) {                                                                                  \
            REP4(asm volatile(ASM1 : "+v"(a) : "v"(y)); asm volatile(ASM1 : "+v"(b) : "v"(y));                             \
                 asm volatile(ASM1 : "+v"(c) : "v"(y)); asm volatile(ASM1 : "+v"(d) : "v"(y));)                            \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                                              \
        sink[threadIdx.x] = (double)(a + b + c + d);                                                                       \
    }

PROBE_F64(add_f64, "v_add_f64 %0, %0, %1")
PROBE_F64(mul_f64, "v_mul_f64 %0, %0, %1")
PROBE_F64(fma_f64, "v_fma_f64 %0, %0, %1, %1")
PROBE_F64(max_f64, "v_max_f64 %0, %0, %1")
PROBE_F64(rcp_f64, "v_rcp_f64 %0, %0")
PROBE_F64(rsq_f64, "v_rsq_f64 %0, %0")
PROBE_F64(rndne_f64, "v_rndne_f64 %0, %0")
PROBE_F64(mov_b64, "v_mov_b64 %0, %0")
PROBE_F32(add_f32, "v_add_f32 %0, %0, %1")
PROBE_F32(fma_f32, "v_fma_f32 %0, %0, %1, %1")
PROBE_F32(mov_b32, "v_mov_b32 %0, %0")
PROBE_F32(and_b32, "v_and_b32 %0, %0, %1")
PROBE_F32(add_u32, "v_add_u32 %0, %0, %1")
PROBE_F32(cvt_u8, "v_cvt_f32_ubyte0 %0, %0")

// f32 clock step of the demodulator (msk.c:95): c = (float)((double)c + s) -- cvt up, add, cvt down (3 instructions)
__global__ void clkstep3_dep(unsigned long long* t, double* sink, double x0, double y0)
{
    float a = (float)x0 + threadIdx.x * 1e-6f;
    double y = y0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a = (float)((double)a + y);
            asm volatile("" : "+v"(a));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = a;
}
__global__ void clkstep3_indep(unsigned long long* t, double* sink, double x0, double y0)
{
    float a = (float)x0 + threadIdx.x * 1e-6f, b = a + 1e-3f, c = a + 2e-3f, d = a + 3e-3f;
    double y = y0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a = (float)((double)a + y); b = (float)((double)b + y); c = (float)((double)c + y); d = (float)((double)d + y);
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = (double)(a + b + c + d);
}
// phase step (msk.c:82-83): p += s; wrap as compare + select of the high word + fma (4 instructions)
__global__ void phstep4_dep(unsigned long long* t, double* sink, double x0, double y0)
{
    double a = x0 + threadIdx.x * 1e-9, y = y0;
    const double twopi = 6.283185307179586;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a += y;
            const double k = __hiloint2double(a >= twopi ? (int)0xBFF00000 : (int)0x80000000, 0);
            a = __builtin_fma(k, twopi, a);
            asm volatile("" : "+v"(a));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = a;
}
__global__ void phstep4_indep(unsigned long long* t, double* sink, double x0, double y0)
{
    double a = x0 + threadIdx.x * 1e-9, b = a + 1e-3, c = a + 2e-3, d = a + 3e-3, y = y0;
    const double twopi = 6.283185307179586;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define PHS(v) { v += y; const double k = __hiloint2double(v >= twopi ? (int)0xBFF00000 : (int)0x80000000, 0); v = __builtin_fma(k, twopi, v); }
            PHS(a) PHS(b) PHS(c) PHS(d)
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d;
}

// LDS round trip: write then read back the same word (per pair), and a read whose address depends on the previous read
__global__ void lds_wr_rd_dep(unsigned long long* t, double* sink, double x0, double y0)
{
    __shared__ float buf[256];
    float a = (float)x0 + threadIdx.x;
    buf[threadIdx.x] = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned int addr = threadIdx.x * 4;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(REP16("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(addr) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = a;
}
__global__ void lds_rd_indep(unsigned long long* t, double* sink, double x0, double y0)
{
    __shared__ float buf[256];
    float a = (float)x0 + threadIdx.x, b, c, d;
    buf[threadIdx.x] = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned int addr = threadIdx.x * 4;
    for (int i = 0; i < ITERS; ++i) {
        REP4(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4\n s_waitcnt lgkmcnt(0)"
                          : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d;
}

typedef void (*kern_t)(unsigned long long*, double*, double, double);
struct Row { const char* name; kern_t dep; kern_t indep; int ninstr; double x0, y0; };

int main()
{
    unsigned long long* t;
    double* sink;
    CK(hipMalloc(&t, 8));
    CK(hipMalloc(&sink, 64 * 8));
    const Row rows[] = {
        {"v_add_f64", add_f64_dep, add_f64_indep, 1, 1.0, 1e-3},
        {"v_mul_f64", mul_f64_dep, mul_f64_indep, 1, 1.0, 1.0000001},
        {"v_fma_f64", fma_f64_dep, fma_f64_indep, 1, 1.0, 0.5},
        {"v_max_f64", max_f64_dep, max_f64_indep, 1, 1.0, 0.5},
        {"v_rcp_f64", rcp_f64_dep, rcp_f64_indep, 1, 1.5, 0.0},
        {"v_rsq_f64", rsq_f64_dep, rsq_f64_indep, 1, 1.5, 0.0},
        {"v_rndne_f64", rndne_f64_dep, rndne_f64_indep, 1, 1.5, 0.0},
        {"v_mov_b64", mov_b64_dep, mov_b64_indep, 1, 1.5, 0.0},
        {"v_add_f32", add_f32_dep, add_f32_indep, 1, 1.0, 1e-3},
        {"v_fma_f32", fma_f32_dep, fma_f32_indep, 1, 1.0, 0.5},
        {"v_mov_b32", mov_b32_dep, mov_b32_indep, 1, 1.0, 0.5},
        {"v_and_b32", and_b32_dep, and_b32_indep, 1, 1.0, 0.5},
        {"v_add_u32", add_u32_dep, add_u32_indep, 1, 1.0, 0.5},
        {"v_cvt_f32_ubyte0", cvt_u8_dep, cvt_u8_indep, 1, 1.0, 0.5},
        {"clock step: cvt_f64_f32 + add_f64 + cvt_f32_f64 (per 3)", clkstep3_dep, clkstep3_indep, 1, 0.1, 0.9},
        {"phase step: add_f64 + cmp + cndmask + fma_f64 (per 4)", phstep4_dep, phstep4_indep, 1, 0.1, 0.9},
        {"ds_write_b32 + ds_read_b32 + wait (per pair) | 4 ds_read_b32 + wait (per 4)", lds_wr_rd_dep, lds_rd_indep, 1, 0.1, 0.9},
    };
    printf("one wave, cycles (s_memtime) per instruction or per group: dep = serial chain, indep = four interleaved chains\n");
    for (const Row& r : rows) {
        double res[2];
        for (int m = 0; m < 2; ++m) {
            kern_t k = m == 0 ? r.dep : r.indep;
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t, sink, r.x0, r.y0);
                CK(hipDeviceSynchronize());
                unsigned long long h;
                CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
                if (h < best) best = h;
            }
            res[m] = (double)best / (ITERS * 16.0);
        }
        printf("%-82s dep %7.2f   indep %7.2f\n", r.name, res[0], res[1]);
    }
    return 0;
}
