// cumask_probe.hip -- where do the waves of a CU-masked stream land?  N single-wave workgroups spin long
// enough to be co-resident and report (XCC, SE, CU, SIMD) from the hardware id registers.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void where_kernel(unsigned int* out, unsigned long long ticks)
{
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    unsigned int* out; CK(hipMalloc(&out, 8 * 4096));
    for (int nbits : {16, 32, 64, 128, 224, 256}) {
        for (int nwg : {nbits * 4}) {
            uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < nbits; ++i) m[i >> 5] |= 1u << (i & 31);
            hipStream_t s;
            CK(hipExtStreamCreateWithCUMask(&s, 8, m));
            hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, s, out, 20000ull);   // 200 us
            CK(hipStreamSynchronize(s));
            std::vector<unsigned int> h(2 * nwg);
            CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
            std::map<unsigned int, int> per_cu, per_simd, per_xcc;
            for (int i = 0; i < nwg; ++i) {
                const unsigned int hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
                const unsigned int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const unsigned int cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
                per_cu[cukey]++; per_simd[(cukey << 2) | simd]++; per_xcc[xcc]++;
            }
            int maxs = 0; for (auto& kv : per_simd) maxs = kv.second > maxs ? kv.second : maxs;
            printf("mask bits [0,%d): %d single-wave WGs -> %zu distinct CUs, %zu distinct SIMDs, max %d waves on one SIMD; per XCC:", nbits, nwg,
                   per_cu.size(), per_simd.size(), maxs);
            for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
            printf("\n");
            CK(hipStreamDestroy(s));
        }
    }
    return 0;
}
