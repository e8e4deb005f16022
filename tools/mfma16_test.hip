// Layout check for v_mfma_f32_16x16x32_bf16 (gfx950): A lane l = row l%16, k = 8*(l/16)+j; B lane l = col l%16,
// k = 8*(l/16)+j; C lane l = col l%16, rows 4*(l/16)+reg.   hipcc --offload-arch=gfx950 tools/mfma16_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ uint16_t bf(float x) { return (uint16_t)(__float_as_uint(x) >> 16); }
__global__ void k(const float* A /*16x32*/, const float* B /*32x16*/, float* C /*16x16*/) {
    const int l = threadIdx.x;
    uint16_t a[8], b[8];
    for (int j = 0; j < 8; ++j) {
        a[j] = bf(A[(l % 16) * 32 + 8 * (l / 16) + j]);
        b[j] = bf(B[(8 * (l / 16) + j) * 16 + (l % 16)]);
    }
    uint4 av = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
    uint4 bv = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hC[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dC; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) if (fabsf(hC[i] - ref[i]) > 1e-3f) ++bad;
    printf("mfma16x16x32 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
