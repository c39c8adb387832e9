// clock_probe.hip -- what shader clock does the chip actually run at, idle and beside a bandwidth-bound
// kernel?  One wave spins for a fixed real-time interval (s_memrealtime, 100 MHz) and counts shader
// cycles (s_memtime).  Not product code.
//   build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__global__ void clock_kernel(unsigned long long* out, unsigned long long ticks_100mhz)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    float x = (float)threadIdx.x;
    while (r1 - r0 < ticks_100mhz) {
        for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f;     // a dependent VALU chain, like the demodulator's
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (x == 123.456f) out[0] = 0;
}

template <bool FMA>
__global__ __launch_bounds__(256) void load_kernel(const u4v* __restrict__ src, size_t nvec, unsigned int* sink, int reps)
{
    u4v acc = {0, 0, 0, 0};
    float f = 0.f;
    for (int r = 0; r < reps; ++r)
        for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < nvec; t += (size_t)gridDim.x * 256) {
            const u4v v = __builtin_nontemporal_load(src + t);
            acc ^= v;
            if (FMA) {
#pragma unroll
                for (int k = 0; k < 40; ++k) f = f * 1.0001f + (float)(v.x & 0xff);
            }
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u || f == 1.5f) sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const size_t bytes = (size_t)1 << 31;
    unsigned char* d; unsigned int* sink; unsigned long long* out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 16 * 256));
    CK(hipMemset(d, 1, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto measure = [&](const char* what) {
        // 64 single-wave workgroups spin for 2 ms
        hipLaunchKernelGGL(clock_kernel, dim3(64), dim3(64), 0, s2, out, 200000ull);
        CK(hipStreamSynchronize(s2));
        unsigned long long h[128];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double mn = 1e9, mx = 0, av = 0;
        for (int i = 0; i < 64; ++i) {
            const double ghz = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);   // cycles per ns
            mn = ghz < mn ? ghz : mn; mx = ghz > mx ? ghz : mx; av += ghz / 64;
        }
        printf("%-44s shader clock %.3f GHz (min %.3f max %.3f)\n", what, av, mn, mx);
    };
    measure("idle");
    measure("idle (again)");
    hipLaunchKernelGGL((load_kernel<false>), dim3(256 * 4), dim3(256), 0, s1, (const u4v*)d, bytes / 16, sink, 40);
    measure("beside a pure nt reader");
    CK(hipStreamSynchronize(s1));
    hipLaunchKernelGGL((load_kernel<true>), dim3(256 * 5), dim3(256), 0, s1, (const u4v*)d, bytes / 16, sink, 40);
    measure("beside a reader with 40 FMA per 16 B");
    CK(hipStreamSynchronize(s1));
    measure("idle (after)");
    return 0;
}
