// membench2.hip -- round 2: which LDS-DMA streaming shapes of W[V][ldw] reach the chip's bandwidth?
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/membench2.bin ; run on the GPU box.
// Questions:
//   k1dma<WC> : K1 shape.  A block owns WC columns (WC*4-byte row segments) x a K slice; every wave-instruction is one
//               global_load_lds_dwordx4 = 1 KB = (256/WC) row segments.  Narrow tiles need few K slices (small split-K
//               slabs): how much bandwidth do 128-B / 256-B / 512-B segments cost?
//   k2dma     : K2 shape.  A block owns TR whole rows; one wave-instruction = 1 KB contiguous of one row.
//   k2reg     : K2 shape of round 1 (lane = row, two adjacent float4 per lane) for comparison.
//   rw4p<TPB> : K3 shape, persistent (one block per CU, TPB tiles each, next tile's 32 float4 loads issued before the
//               current tile's 32 float4 stores) -- is K3's 45 us the access pattern or the MFMA/LDS work beside it?
//   rwlin     : elementwise float4 read-modify-write ceiling.
// Each pattern is timed "hot" (repeated on the same 60 MB: Infinity Cache) and "after rw" (alternating with the K3-shaped
// writer, minus the writer alone): the state a propagation kernel finds W in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int V = 10000, H = 1500, LDW = 1536;

__device__ __forceinline__ void dma16(const void* g, uint32_t lds_off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// ---- K1 shape through LDS-DMA ---------------------------------------------------------------------------------------
// grid = (ceil(H/WC), ks); block = 256; wave w takes row groups w, w+4, ... of the block's K slice; a row group = RPI rows
// (one instruction).  Ring: DEPTH slots of U instructions per wave.
template <int WC, int U, int DEPTH>
__global__ __launch_bounds__(256, 1) void k1dma(const float* __restrict__ W, int kchunk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPI = 256 / WC;                 // rows per instruction
    constexpr int LPR = WC / 4;                   // lanes per row
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n0 = blockIdx.x * WC;
    const int k0 = blockIdx.y * kchunk, k1 = min(k0 + kchunk, V);
    const int lr = l / LPR, lc = l % LPR;
    const int col = min(n0 + 4 * lc, LDW - 4);
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + w * (DEPTH * U * 1024);
    const uint32_t ubase = __builtin_amdgcn_readfirstlane(base);
    // groups of this wave: g = w + 4*i ; rows k0 + g*RPI
    const int ngroups = (k1 - k0 + RPI - 1) / RPI;
    const int my_groups = (ngroups - w + 3) / 4;
    float acc = 0.f;
    auto issue = [&](int slot, int gi) {          // U instructions: groups gi .. gi+U-1 of this wave
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gi + u < my_groups) {             // wave-uniform
                const int g = w + 4 * (gi + u);
                const int row = min(k0 + g * RPI + lr, V - 1);
                dma16(W + (size_t)row * LDW + col, ubase + (slot * U + u) * 1024);
            }
        }
    };
    const int steps = (my_groups + U - 1) / U;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d * U);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_vm<U * (DEPTH - 1)>();
            // consume one value per lane (keeps an LDS read in the loop, as a real kernel would have)
            acc += *reinterpret_cast<const float*>(smem + w * (DEPTH * U * 1024) + d * U * 1024 + 16 * l);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(d, (s + d + DEPTH) * U);        // clamped rows past the end: harmless re-reads of the last row
        }
    }
    wait_vm<0>();
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- K2 shape through LDS-DMA: block owns TR rows; instruction = 1 KB of one row --------------------------------------
template <int U, int DEPTH>
__global__ __launch_bounds__(256, 1) void k2dma(const float* __restrict__ W, int TR, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r0 = blockIdx.x * TR;
    const uint32_t ubase = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + w * (DEPTH * U * 1024));
    constexpr int CPR = 6;                        // 1-KB chunks per row (6000 B -> 6, the last one runs into the pitch padding)
    const int nchunks = TR * CPR;                 // chunk c: row c / CPR, piece c % CPR ; wave w takes chunks w, w+4, ...
    const int mine = (nchunks - w + 3) / 4;
    float acc = 0.f;
    auto issue = [&](int slot, int ci) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ci + u < mine) {                  // wave-uniform
                const int c = w + 4 * (ci + u);
                const int row = min(r0 + c / CPR, V - 1), pc = c % CPR;
                dma16(W + (size_t)row * LDW + pc * 256 + 4 * l, ubase + (slot * U + u) * 1024);
            }
        }
    };
    const int steps = (mine + U - 1) / U;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d * U);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_vm<U * (DEPTH - 1)>();
            acc += *reinterpret_cast<const float*>(smem + w * (DEPTH * U * 1024) + d * U * 1024 + 16 * l);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(d, (s + d + DEPTH) * U);
        }
    }
    wait_vm<0>();
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- K2 shape of round 1: lane = row, 8 consecutive k per lane as two float4, 4 waves interleave K blocks ----------------
__global__ __launch_bounds__(256, 2) void k2reg(const float* __restrict__ W, int TR, float* sink) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = min(l & 31, TR - 1), hh = l >> 5;
    const float* row = W + (size_t)min((int)blockIdx.x * TR + r, V - 1) * LDW;
    float s = 0.f;
    for (int g = 0; (4 * g + w) * 4 < H / 16; ++g) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int kb = 16 * ((4 * g + w) * 4 + d);
            const int k0 = min(kb + 8 * hh, H - 8);
            const float4 a = *(const float4*)(row + k0), b = *(const float4*)(row + k0 + 4);
            s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        }
    }
    if (s == 12345.678f) sink[0] = s;
}

// ---- K3 shape, persistent ---------------------------------------------------------------------------------------------
template <bool TWO>   // TWO: 2 blocks per CU (half the tiles each)
__global__ __launch_bounds__(256, TWO ? 2 : 1) void rw4p(float* __restrict__ W, float* __restrict__ M, int tpb) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    const int h0 = blockIdx.x * 128, tile0 = blockIdx.y * tpb;
    const int ntile = (V + 127) / 128;
    const int n_my = min(tpb, ntile - tile0);
    const int col = min(h0 + 4 * r, H - 4);
    float4 a[2][16], b[2][16];
    auto load = [&](int s, int t) {
        const int v0 = min(tile0 + t, ntile - 1) * 128 + 32 * w;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = min(v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, V - 1);
            a[s][reg] = *(const float4*)(W + (size_t)row * LDW + col); b[s][reg] = *(const float4*)(M + (size_t)row * LDW + col);
        }
    };
    auto store = [&](int s, int t) {
        const int v0 = (tile0 + t) * 128 + 32 * w;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (row < V && h0 + 4 * r < H) {
                const float4 x = a[s][reg], y = b[s][reg];
                float4 m = make_float4(y.x * .5f + x.x * 1e-4f, y.y * .5f + x.y * 1e-4f, y.z * .5f + x.z * 1e-4f, y.w * .5f + x.w * 1e-4f);
                *(float4*)(M + (size_t)row * LDW + col) = m;
                *(float4*)(W + (size_t)row * LDW + col) = make_float4(x.x + m.x, x.y + m.y, x.z + m.z, x.w + m.w);
            }
        }
    };
    load(0, 0);
    for (int t = 0; t < n_my; t += 2) {
        load(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        store(0, t);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < n_my) {
            load(0, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            store(1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// K3 shape with whole-KB row pieces: wave = 8 rows x 256 columns (lane = 16 B of a 1-KB row piece)
__global__ __launch_bounds__(256, 2) void rwrow(float* __restrict__ W, float* __restrict__ M, int rows_per_block) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block;
    // chunk c of the block: row c / 6, piece c % 6 ; wave w takes chunks w, w+4, ... 8 at a time
    const int nch = rows_per_block * 6;
    for (int c0 = w; c0 < nch; c0 += 32) {
        float4 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = min(c0 + 4 * i, nch - 1);
            const int row = min(r0 + c / 6, V - 1), col = min((c % 6) * 256 + 4 * l, H - 4);
            a[i] = *(const float4*)(W + (size_t)row * LDW + col); b[i] = *(const float4*)(M + (size_t)row * LDW + col);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + 4 * i;
            const int row = r0 + c / 6, col = (c % 6) * 256 + 4 * l;
            if (c < nch && row < V && col < H) {
                const float4 x = a[i], y = b[i];
                float4 m = make_float4(y.x * .5f + x.x * 1e-4f, y.y * .5f + x.y * 1e-4f, y.z * .5f + x.z * 1e-4f, y.w * .5f + x.w * 1e-4f);
                *(float4*)(M + (size_t)row * LDW + col) = m;
                *(float4*)(W + (size_t)row * LDW + col) = make_float4(x.x + m.x, x.y + m.y, x.z + m.z, x.w + m.w);
            }
        }
    }
}

__global__ __launch_bounds__(256) void rwlin(float4* __restrict__ W, float4* __restrict__ M, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = W[i], b = M[i];
        float4 m = make_float4(b.x * .5f + a.x * 1e-4f, b.y * .5f + a.y * 1e-4f, b.z * .5f + a.z * 1e-4f, b.w * .5f + a.w * 1e-4f);
        M[i] = m; W[i] = make_float4(a.x + m.x, a.y + m.y, a.z + m.z, a.w + m.w);
    }
}
__global__ __launch_bounds__(256) void lin4(const float4* __restrict__ p, size_t n4, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    const size_t n = (size_t)V * LDW, bytes = n * 4;
    float *Wb, *Mb, *sink;
    CK(hipMalloc(&Wb, bytes)); CK(hipMalloc(&Mb, bytes)); CK(hipMemset(Wb, 0, bytes)); CK(hipMemset(Mb, 0, bytes)); CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 30;
    auto timeit = [&](const std::function<void()>& body) {
        for (int i = 0; i < 3; ++i) body();
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int i = 0; i < REP; ++i) body();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / REP;
    };
    auto writer = [&] { hipLaunchKernelGGL(rw4p<false>, dim3(12, 20), dim3(256), 0, 0, Wb, Mb, 4); };
    const double t_writer = timeit(writer);
    auto report = [&](const char* name, double mb, const std::function<void()>& k) {
        const double hot = timeit(k);
        const double seq = timeit([&] { writer(); k(); }) - t_writer;
        printf("%-34s hot %7.1f us %6.0f GB/s | after rw4p %7.1f us %6.0f GB/s\n", name, hot, mb / hot * 1e3, seq, mb / seq * 1e3);
        fflush(stdout);
    };
    printf("writer rw4p<1/CU> tpb=4 (240 MB): %.1f us %.0f GB/s\n", t_writer, 240.0 / t_writer * 1e3);
    report("lin4 read 60MB", 60, [&] { hipLaunchKernelGGL(lin4, dim3(2048), dim3(256), 0, 0, (const float4*)Wb, n / 4, sink); });
#define K1(WC, U, D, KS) do { \
        const int tiles = (H + WC - 1) / WC; const int kch = ((V + KS - 1) / KS + 15) / 16 * 16; const int ks = (V + kch - 1) / kch; \
        char nm[96]; snprintf(nm, 96, "k1dma WC=%d U=%d D=%d grid=%dx%d", WC, U, D, tiles, ks); \
        CK(hipFuncSetAttribute((const void*)k1dma<WC, U, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * D * U * 1024)); \
        report(nm, 60, [&] { hipLaunchKernelGGL((k1dma<WC, U, D>), dim3(tiles, ks), dim3(256), 4 * D * U * 1024, 0, Wb, kch, sink); }); } while (0)
    K1(128, 4, 4, 20); K1(128, 8, 4, 20); K1(128, 4, 8, 20); K1(128, 4, 4, 40);
    K1(64, 4, 4, 10); K1(64, 8, 4, 10); K1(64, 4, 8, 10); K1(64, 4, 4, 20);
    K1(32, 4, 4, 5); K1(32, 8, 4, 5); K1(32, 4, 8, 5); K1(32, 4, 4, 10); K1(32, 4, 4, 16);
    K1(16, 4, 4, 3); K1(16, 4, 8, 3); K1(16, 4, 4, 8);
    K1(256, 4, 4, 40); K1(256, 4, 8, 40);
#define K2(U, D, TR) do { \
        char nm[96]; snprintf(nm, 96, "k2dma U=%d D=%d TR=%d grid=%d", U, D, TR, (V + TR - 1) / TR); \
        CK(hipFuncSetAttribute((const void*)k2dma<U, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * D * U * 1024)); \
        report(nm, 60, [&] { hipLaunchKernelGGL((k2dma<U, D>), dim3((V + TR - 1) / TR), dim3(256), 4 * D * U * 1024, 0, Wb, TR, sink); }); } while (0)
    K2(4, 4, 40); K2(8, 4, 40); K2(4, 8, 40); K2(2, 4, 40); K2(4, 4, 20); K2(4, 2, 20); K2(4, 4, 10);
    for (int tr : {20, 32, 40}) {
        char nm[96]; snprintf(nm, 96, "k2reg (round-1 shape) TR=%d", tr);
        report(nm, 60, [&] { hipLaunchKernelGGL(k2reg, dim3((V + tr - 1) / tr), dim3(256), 0, 0, Wb, tr, sink); });
    }
    // K3 shapes (240 MB each); not relative to the writer
    auto rep3 = [&](const char* name, const std::function<void()>& k) { const double t = timeit(k); printf("%-34s %7.1f us %6.0f GB/s\n", name, t, 240.0 / t * 1e3); fflush(stdout); };
    rep3("rw4p 1/CU tpb=4 (12x20)", [&] { hipLaunchKernelGGL(rw4p<false>, dim3(12, 20), dim3(256), 0, 0, Wb, Mb, 4); });
    rep3("rw4p 2/CU tpb=2 (12x40)", [&] { hipLaunchKernelGGL(rw4p<true>, dim3(12, 40), dim3(256), 0, 0, Wb, Mb, 2); });
    rep3("rw4p 2/CU tpb=4 (12x20)", [&] { hipLaunchKernelGGL(rw4p<true>, dim3(12, 20), dim3(256), 0, 0, Wb, Mb, 4); });
    rep3("rw4p 1/CU tpb=1 (12x79)", [&] { hipLaunchKernelGGL(rw4p<false>, dim3(12, 79), dim3(256), 0, 0, Wb, Mb, 1); });
    rep3("rw4p 2/CU tpb=1 (12x79)", [&] { hipLaunchKernelGGL(rw4p<true>, dim3(12, 79), dim3(256), 0, 0, Wb, Mb, 1); });
    for (int rpb : {20, 40, 10}) {
        char nm[96]; snprintf(nm, 96, "rwrow 1-KB row pieces rows/blk=%d", rpb);
        rep3(nm, [&] { hipLaunchKernelGGL(rwrow, dim3((V + rpb - 1) / rpb), dim3(256), 0, 0, Wb, Mb, rpb); });
    }
    rep3("rwlin float4 (2048 blocks)", [&] { hipLaunchKernelGGL(rwlin, dim3(2048), dim3(256), 0, 0, (float4*)Wb, (float4*)Mb, n / 4); });
    rep3("rwlin float4 (512 blocks)", [&] { hipLaunchKernelGGL(rwlin, dim3(512), dim3(256), 0, 0, (float4*)Wb, (float4*)Mb, n / 4); });
    return 0;
}
