// membench.hip -- what bandwidth does each weight-access pattern of the CD kernels get on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench ; run on the GPU box.
// Patterns (matrix W[V][H] fp32, V=10000, H=1500, 60 MB):
//   lin4     : fully coalesced float4 grid-stride read                     (ceiling)
//   up1      : K1 shape, dword/lane: half-wave reads one 128-B row segment (16 rows x 32 cols per block-iter)
//   up2/up4  : same with float2 / float4 per lane (256-B / 512-B segments)
//   down4    : K2 shape, float4/lane: lane = row, two adjacent 16-B pieces per row per instruction
//   rw1      : K3 shape, dword/lane read W + Wm then write both (C-layout 32x32 tiles)
//   rw4      : K3 shape with float4 per lane (4 interleaved column tiles)
// "cold" rotates over enough buffers to exceed the 256 MiB Infinity Cache; "hot" reuses one buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int V = 10000, H = 1500;

__global__ __launch_bounds__(256) void lin4(const float4* __restrict__ p, size_t n4, float* sink) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = p[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) sink[0] = s;
}

// K1 shape: block = 4 waves on one 64-col tile (VEC*32 cols per nt... here cols per wave = 32*VEC), K split over blocks+waves
template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void up(const float* __restrict__ W, int kchunk, float* sink) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * (32 * VEC);
    const int k_begin = blockIdx.y * kchunk, k_end = min(k_begin + kchunk, V);
    float s = 0.f;
    const int col = min(n0 + r * VEC, H - VEC);
    for (int kb = k_begin + 16 * w; kb < k_end; kb += 64 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = min(kb + 64 * u + 8 * hh + j, V - 1);
                const float* q = W + (size_t)k * H + col;
                if (VEC == 1) s += q[0];
                else if (VEC == 2) { float2 v = *(const float2*)q; s += v.x + v.y; }
                else { float4 v = *(const float4*)q; s += v.x + v.y + v.z + v.w; }
            }
    }
    if (s == 12345.678f) sink[0] = s;
}

// K2 shape: lane = row (v), 8 consecutive k per lane as two float4; block = 64 rows, 4 waves interleave K blocks
template <int UNROLL>
__global__ __launch_bounds__(256) void down4(const float* __restrict__ W, int kchunk, float* sink) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.y * kchunk, k_end = min(k_begin + kchunk, H);
    float s = 0.f;
    for (int kb = k_begin + 16 * w; kb < k_end; kb += 64 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = min(n0 + nt * 32 + r, V - 1);
                const int k0 = min(kb + 64 * u + 8 * hh, H - 8);
                const float4 a = *(const float4*)(W + (size_t)n * H + k0);
                const float4 b = *(const float4*)(W + (size_t)n * H + k0 + 4);
                s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
            }
    }
    if (s == 12345.678f) sink[0] = s;
}

// K3 shape, dword: wave tile 32(v) x 64(h): 32 (nt,reg) loads of W and Wm, then stores
__global__ __launch_bounds__(256) void rw1(float* __restrict__ W, float* __restrict__ M) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    const int v0 = blockIdx.y * 64 + (w >> 1) * 32, h0 = blockIdx.x * 128 + (w & 1) * 64;
    float a[2][16], b[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = min(v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, V - 1), col = min(h0 + nt * 32 + r, H - 1);
            a[nt][reg] = W[(size_t)row * H + col]; b[nt][reg] = M[(size_t)row * H + col];
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, col = h0 + nt * 32 + r;
            if (row < V && col < H) { const float m = b[nt][reg] * 0.5f + a[nt][reg] * 1e-4f; M[(size_t)row * H + col] = m; W[(size_t)row * H + col] = a[nt][reg] + m; }
        }
}

// K3 shape, float4: wave tile 32(v) x 128(h) as 4 interleaved column tiles (col = h0 + 4r + t): 16 float4 loads each
__global__ __launch_bounds__(256) void rw4(float* __restrict__ W, float* __restrict__ M) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 31, hh = l >> 5;
    const int v0 = blockIdx.y * 128 + w * 32, h0 = blockIdx.x * 128;
    float4 a[16], b[16];
    const int col = min(h0 + 4 * r, H - 4);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = min(v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, V - 1);
        a[reg] = *(const float4*)(W + (size_t)row * H + col); b[reg] = *(const float4*)(M + (size_t)row * H + col);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = v0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (row < V && h0 + 4 * r < H) {
            float4 m = make_float4(b[reg].x * .5f + a[reg].x * 1e-4f, b[reg].y * .5f + a[reg].y * 1e-4f, b[reg].z * .5f + a[reg].z * 1e-4f, b[reg].w * .5f + a[reg].w * 1e-4f);
            *(float4*)(M + (size_t)row * H + col) = m;
            *(float4*)(W + (size_t)row * H + col) = make_float4(a[reg].x + m.x, a[reg].y + m.y, a[reg].z + m.z, a[reg].w + m.w);
        }
    }
}

// plain elementwise float4 read-modify-write of W and M (apply_delta-like ceiling for K3)
__global__ __launch_bounds__(256) void rwlin(float4* __restrict__ W, float4* __restrict__ M, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = W[i], b = M[i];
        float4 m = make_float4(b.x * .5f + a.x * 1e-4f, b.y * .5f + a.y * 1e-4f, b.z * .5f + a.z * 1e-4f, b.w * .5f + a.w * 1e-4f);
        M[i] = m; W[i] = make_float4(a.x + m.x, a.y + m.y, a.z + m.z, a.w + m.w);
    }
}

int main() {
    const size_t n = (size_t)V * H, bytes = n * 4;
    const int NB = 6;                       // 6 x 60 MB = 360 MB > 256 MiB Infinity Cache
    std::vector<float*> Wb(NB), Mb(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&Wb[i], bytes)); CK(hipMalloc(&Mb[i], bytes)); CK(hipMemset(Wb[i], 0, bytes)); CK(hipMemset(Mb[i], 0, bytes)); }
    float* sink; CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 24;
    auto run = [&](const char* name, double mb, bool cold, auto launch) {
        for (int i = 0; i < 3; ++i) launch(cold ? i % NB : 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < REP; ++i) launch(cold ? i % NB : 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / REP;
        printf("%-28s %-4s %8.1f us  %7.0f GB/s\n", name, cold ? "cold" : "hot", us, mb * 1e6 / (us * 1e-6) / 1e9);
    };
    for (int cold = 1; cold >= 0; --cold) {
        run("lin4 read 60MB", 60, cold, [&](int b) { hipLaunchKernelGGL(lin4, dim3(2048), dim3(256), 0, 0, (const float4*)Wb[b], n / 4, sink); });
        for (int ksp : {8, 16, 32, 64}) {
            const int kch = ((V + ksp - 1) / ksp + 63) / 64 * 64; const int ks = (V + kch - 1) / kch;
            char nm[64];
            snprintf(nm, 64, "up1 u1 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((up<1, 1>), dim3((H + 31) / 32, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
            snprintf(nm, 64, "up2 u1 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((up<2, 1>), dim3((H + 63) / 64, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
            snprintf(nm, 64, "up4 u1 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((up<4, 1>), dim3((H + 127) / 128, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
            snprintf(nm, 64, "up2 u2 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((up<2, 2>), dim3((H + 63) / 64, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
        }
        for (int ksp : {1, 2, 4, 8}) {
            const int kch = ((H + ksp - 1) / ksp + 63) / 64 * 64; const int ks = (H + kch - 1) / kch;
            char nm[64];
            snprintf(nm, 64, "down4 u1 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((down4<1>), dim3((V + 63) / 64, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
            snprintf(nm, 64, "down4 u2 ks=%d", ks); run(nm, 60, cold, [&](int b) { hipLaunchKernelGGL((down4<2>), dim3((V + 63) / 64, ks), dim3(256), 0, 0, Wb[b], kch, sink); });
        }
        run("rw1  K3 dword   240MB", 240, cold, [&](int b) { hipLaunchKernelGGL(rw1, dim3((H + 127) / 128, (V + 63) / 64), dim3(256), 0, 0, Wb[b], Mb[b]); });
        run("rw4  K3 float4  240MB", 240, cold, [&](int b) { hipLaunchKernelGGL(rw4, dim3((H + 127) / 128, (V + 127) / 128), dim3(256), 0, 0, Wb[b], Mb[b]); });
        run("rwlin float4    240MB", 240, cold, [&](int b) { hipLaunchKernelGGL(rwlin, dim3(2048), dim3(256), 0, 0, (float4*)Wb[b], (float4*)Mb[b], n / 4); });
    }
    // ---- sequence experiment: does W / W_m stay in the Infinity Cache across the kernels of one CD step?
    // per iteration: 3 streaming reads of W (the three propagations) then the K3-shaped read-modify-write of W, W_m
    {
        auto seq = [&](bool with_rw, bool float4_rw) {
            for (int i = 0; i < 3; ++i) {
                for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(lin4, dim3(2048), dim3(256), 0, 0, (const float4*)Wb[0], n / 4, sink);
                if (with_rw) { if (float4_rw) hipLaunchKernelGGL(rw4, dim3((H + 127) / 128, (V + 127) / 128), dim3(256), 0, 0, Wb[0], Mb[0]);
                               else hipLaunchKernelGGL(rwlin, dim3(2048), dim3(256), 0, 0, (float4*)Wb[0], (float4*)Mb[0], n / 4); }
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < REP; ++i) {
                for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(lin4, dim3(2048), dim3(256), 0, 0, (const float4*)Wb[0], n / 4, sink);
                if (with_rw) { if (float4_rw) hipLaunchKernelGGL(rw4, dim3((H + 127) / 128, (V + 127) / 128), dim3(256), 0, 0, Wb[0], Mb[0]);
                               else hipLaunchKernelGGL(rwlin, dim3(2048), dim3(256), 0, 0, (float4*)Wb[0], (float4*)Mb[0], n / 4); }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return 1e3 * ms / REP;
        };
        const double t0 = seq(false, false), t1 = seq(true, true), t2 = seq(true, false);
        printf("sequence 3x read(W): %.1f us/iter ; + rw4(W,Wm): %.1f us (marginal %.1f us) ; + rwlin: %.1f us (marginal %.1f us)\n",
               t0, t1, t1 - t0, t2, t2 - t0);
    }
    // ---- step-shaped sequence: up4(W) down4(W) up4(W) rw4(W,Wm) per iteration, then leave one kernel out to get
    // each pattern's MARGINAL cost when its input was last touched by a DIFFERENT pattern (no lucky L2 hits).
    {
        const int kchu = ((V + 21 - 1) / 21 + 63) / 64 * 64; const int ksu = (V + kchu - 1) / kchu;      // K1: ~21 K slices
        auto L_up = [&] { hipLaunchKernelGGL((up<4, 1>), dim3((H + 127) / 128, ksu), dim3(256), 0, 0, Wb[0], kchu, sink); };
        auto L_dn = [&](int ksd) { const int kch = ((H + ksd - 1) / ksd + 63) / 64 * 64; const int ks = (H + kch - 1) / kch;
                                   hipLaunchKernelGGL((down4<1>), dim3((V + 63) / 64, ks), dim3(256), 0, 0, Wb[0], kch, sink); };
        auto L_lin = [&] { hipLaunchKernelGGL(lin4, dim3(2048), dim3(256), 0, 0, (const float4*)Wb[0], n / 4, sink); };
        auto L_rw = [&] { hipLaunchKernelGGL(rw4, dim3((H + 127) / 128, (V + 127) / 128), dim3(256), 0, 0, Wb[0], Mb[0]); };
        auto timeit = [&](auto&& body) {
            for (int i = 0; i < 3; ++i) body();
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
            for (int i = 0; i < REP; ++i) body();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / REP;
        };
        const double full = timeit([&] { L_up(); L_dn(1); L_up(); L_rw(); });
        const double no_dn = timeit([&] { L_up(); L_up(); L_rw(); });
        const double no_up = timeit([&] { L_up(); L_dn(1); L_rw(); });
        const double dn2 = timeit([&] { L_up(); L_dn(2); L_up(); L_rw(); });
        const double dn4 = timeit([&] { L_up(); L_dn(4); L_up(); L_rw(); });
        const double lin = timeit([&] { L_up(); L_lin(); L_up(); L_rw(); });
        const double no_rw = timeit([&] { L_up(); L_dn(1); L_up(); });
        printf("step sequence up4,down4,up4,rw4: %.1f us/iter; marginal: down4(ks1) %.1f  down4(ks2) %.1f  down4(ks4) %.1f  lin4-in-place-of-down4 %.1f  up4 %.1f  rw4 %.1f\n",
               full, full - no_dn, dn2 - no_dn, dn4 - no_dn, lin - no_dn, full - no_up, full - no_rw);
    }
    return 0;
}
