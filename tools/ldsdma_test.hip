// Hardware semantics check for gfx950 direct global->LDS loads (global_load_lds_dwordx4):
// lane i of a wave writes its 16 bytes to LDS[M0 + 16*i]; the data is visible to ds_read after s_waitcnt vmcnt.
// build: hipcc --offload-arch=gfx950 -O2 tools/ldsdma_test.hip -o tools/ldsdma_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory", "m0");
}
__global__ void k(const uint4* src, uint4* dst) {
    __shared__ __attribute__((aligned(16))) char smem[163840];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + 65536 + w * 4096;
    for (int q = 0; q < 4; ++q) {
        const int i = l + 64 * q, row = i >> 3, pos = i & 7, c = pos ^ ((row >> 1) & 7);
        dma16(src + w * 256 + row * 8 + c, base + q * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int q = 0; q < 4; ++q)
        dst[w * 256 + l + 64 * q] = *reinterpret_cast<uint4*>(smem + 65536 + w * 4096 + 16 * (l + 64 * q));
}
int main() {
    const int n = 4 * 256;
    std::vector<uint4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(i, 2 * i, 3 * i, 0xabc00000u + i);
    uint4 *s, *d; hipMalloc(&s, n * 16); hipMalloc(&d, n * 16);
    hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice); hipMemset(d, 0, n * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), d, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int i = 0; i < 256; ++i) {
        const int row = i >> 3, pos = i & 7, c = pos ^ ((row >> 1) & 7);
        const uint4 e4 = h[w * 256 + row * 8 + c], g = o[w * 256 + i];
        if (e4.x != g.x || e4.y != g.y || e4.z != g.z || e4.w != g.w) { if (bad < 8) printf("mismatch w%d i%d: got %u exp %u\n", w, i, g.x, e4.x); ++bad; }
    }
    printf("ldsdma: %s (%d mismatches), err=%d\n", bad ? "FAIL" : "ok", bad, (int)e);
    return bad != 0;
}
