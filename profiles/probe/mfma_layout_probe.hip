// Which lane / register holds what in v_mfma_f32_4x4x4_16b_f16 (16 blocks of 4x4x4)?  A[l][k] = l + 64 k, B[l][k] = (k == l % 4):
// then D[i][j] = A[row i][k = j] of the same block, and the printed value v decodes as (row lane = v % 64, k = v / 64).
// hipcc --offload-arch=gfx950 -O3 -o mfma_layout_probe mfma_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(f4* d)
{
    const int l = threadIdx.x;
    h4 a, b;
    for (int q = 0; q < 4; ++q) { a[q] = (_Float16)(l + 64 * q); b[q] = (_Float16)(q == (l & 3) ? 1.0f : 0.0f); }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    d[l] = c;
}
int main()
{
    f4* d; hipMalloc(&d, 64 * sizeof(f4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    f4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        for (int v = 0; v < 4; ++v) {
            const int val = (int)h[l][v];
            const int row = val % 64, kk = val / 64;
            if (row != 4 * (l / 4) + v || kk != (l & 3)) ok = 0;
            if (l < 8 || l >= 60) printf("lane %2d reg %d: row lane %2d, k %d\n", l, v, row, kk);
        }
    }
    printf("layout D[block = l/4][i = reg][j = l%%4] with A row i = lane 4*block + i, B column j = lane 4*block + j: %s\n", ok ? "CONFIRMED" : "NO");
    return 0;
}
