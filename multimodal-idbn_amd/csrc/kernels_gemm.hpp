// kernels_gemm.hpp -- the three MFMA kernels of the CD step (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
//   gemm_up_partial    K1  partial[ks][b][h] = sum_{v in chunk ks} A[b][v] * W[v][h]     (rbm.py:92,344)
//   gemm_down_partial  K2  partial[ks][b][v] = sum_{h in chunk ks} A[b][h] * W[v][h]     (rbm.py:96,350)
//   assoc_update       K3  W_m,W <- momentum/decay update with (X^T P+ - V'^T P-)/n      (rbm.py:200,209,212-213)
//
// Operand conventions
//   * W stays fp32 in HBM (the torch Parameter).  A weight tile is read ONCE per kernel, straight
//     into registers in the MFMA B-fragment shape, and split in registers into NW bf16 terms
//     (NW=3: hi+mid+lo == w exactly -> fp32-exact products; NW=1: round-to-nearest bf16).
//   * Activations arrive pre-split in bf16 "operand form" written by the finish/prep kernels:
//       "row-major" rm[t][Kpad/16][Bp][16], K16-blocked (A operand of K1/K2: 8 consecutive k per lane =
//                   one 16-B load, and a wave's 64 loads form one contiguous KB)
//       transposed  tr[t][N][Bp]      (A/B operands of K3: 8 consecutive batch rows per lane)
//     with t = 1 term for {0,1} samples and 3 terms for real-valued activations (a flag in device
//     memory decides for caller-supplied data), Bp = batch padded to 64 with zero rows.
//   * No LDS staging of operands: the fragment shapes above make every global access either a
//     128-B row segment per half-wave (W) or a 16-B piece of an L2-resident activation.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "kernels_ew.hpp"

namespace imdbn {

template <int NW>
__device__ __forceinline__ void make_w_frags(const float (&wv)[8], uint4 (&frag)[NW]) {
    if constexpr (NW == 3) {
        uint32_t hi[8], mid[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(wv[j], hi[j], mid[j], lo[j]);
        frag[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
        frag[1] = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
        frag[2] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    } else {
        uint32_t h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = bf16_rne(wv[j]);
        frag[0] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
}

// Cross-wave reduction of the 4 waves' accumulators (they split the K range) and slab store.
// acc[mt][nt]: 64(batch) x 64(n) tile in the 32x32 C layout.
__device__ __forceinline__ void reduce_store_tile(f32x16 (&acc)[2][2], float* __restrict__ slab /* [Bp][N] of this ks */,
                                                  int N, int mb, int n0, float (*red)[2][16][64]) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[w][nt][reg][l] = acc[mt][nt][reg];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = w * 8 + i, nt = c >> 4, reg = c & 15;
            const float s = ((red[0][nt][reg][l] + red[1][nt][reg][l]) + red[2][nt][reg][l]) + red[3][nt][reg][l];
            const int row = mb + mt * 32 + mfma_row(reg, l);
            const int col = n0 + nt * 32 + r;
            if (col < N) slab[(int64_t)row * N + col] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1 / K2 share one body.  UP: W is [K][N] (N contiguous, B-fragment = 8 dword loads of 128-B row
// segments).  DOWN: W is [N][K] (K contiguous, B-fragment = two float4 loads per lane).
// grid = (ceil(N/64), ksplit, Bp/64), block = 256: the 4 waves interleave 16-row K blocks.
//
// Pipeline: ALL operands of the next K block (weights and activation fragments) are issued as one
// group one iteration ahead; the vmcnt counter is in-order, so a wait for a younger activation load
// would otherwise drain the weight prefetch.  Weight loads are UNCONDITIONAL from clamped addresses
// (`cond ? load : 0` lowers to one predicated region per load); clamping is harmless because rows
// k >= K meet zero-padded activation columns and columns n >= N are never stored.
// ------------------------------------------------------------------------------------------
template <int NA>
struct GemmOperands {
    float wv[2][8];
    uint4 av[NA][2];
};

template <bool UP, bool VEC4, int NA>
__device__ __forceinline__ void gemm_load(GemmOperands<NA>& o, const float* const (&wbase)[2], int64_t ldw, int K,
                                          const bf16_t* Abase, int64_t a_term_stride, int64_t arow0, int64_t arow1,
                                          int kb, int hh, int Bp_) {
    const int k0 = kb + 8 * hh;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if constexpr (UP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.wv[nt][j] = wbase[nt][(int64_t)min(k0 + j, K - 1) * ldw];
        } else if constexpr (VEC4) {      // K % 4 == 0 here: a float4 is either fully inside or fully outside
            const float4 x0 = *reinterpret_cast<const float4*>(wbase[nt] + min(k0, K - 4));
            const float4 x1 = *reinterpret_cast<const float4*>(wbase[nt] + min(k0 + 4, K - 4));
            o.wv[nt][0] = x0.x; o.wv[nt][1] = x0.y; o.wv[nt][2] = x0.z; o.wv[nt][3] = x0.w;
            o.wv[nt][4] = x1.x; o.wv[nt][5] = x1.y; o.wv[nt][6] = x1.z; o.wv[nt][7] = x1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.wv[nt][j] = wbase[nt][min(k0 + j, K - 1)];
        }
    }
#pragma unroll
    for (int ta = 0; ta < NA; ++ta) {
        // K16-blocked operand: (row, kb) -> ((kb/16)*Bp + row)*16 ; arow0/arow1 carry Bp-relative row offsets
        o.av[ta][0] = *reinterpret_cast<const uint4*>(Abase + ta * a_term_stride + ((int64_t)(kb >> 4) * Bp_ + arow0) * 16 + 8 * hh);
        o.av[ta][1] = *reinterpret_cast<const uint4*>(Abase + ta * a_term_stride + ((int64_t)(kb >> 4) * Bp_ + arow1) * 16 + 8 * hh);
    }
}

template <bool UP, int NW, bool VEC4, int NA>
__device__ __forceinline__ void gemm_body(const float* __restrict__ W, int64_t ldw, int K, int N,
                                          const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
                                          float* __restrict__ partial, int Bp, int kchunk, float (*red)[2][16][64]) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y, mb = blockIdx.z * 64;
    const int k_begin = ks * kchunk;
    const int k_end = min(k_begin + kchunk, lda);

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const float* wbase[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = min(n0 + nt * 32 + r, N - 1);
        wbase[nt] = UP ? (W + n) : (W + (int64_t)n * ldw);
    }
    const int64_t arow0 = mb + r, arow1 = mb + 32 + r;

    auto compute = [&](const GemmOperands<NA>& o) {
        uint4 bf[2][NW];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) make_w_frags<NW>(o.wv[nt], bf[nt]);
#pragma unroll
        for (int ta = 0; ta < NA; ++ta)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(o.av[ta][mt]), as_frag(bf[nt][tw]), acc[mt][nt], 0, 0, 0);
    };
    const int kb0 = k_begin + 16 * w;
    GemmOperands<NA> cur, nxt;
    if (kb0 < k_end) gemm_load<UP, VEC4, NA>(cur, wbase, ldw, K, A, a_term_stride, arow0, arow1, kb0, hh, Bp);
    for (int kb = kb0; kb < k_end; kb += 64) {
        // next block's operands: one load group, pinned ahead of this block's math
        gemm_load<UP, VEC4, NA>(nxt, wbase, ldw, K, A, a_term_stride, arow0, arow1, min(kb + 64, lda - 16), hh, Bp);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    reduce_store_tile(acc, partial + (int64_t)ks * Bp * N, N, mb, n0, red);
}

// The activation term count (1 for {0,1} samples / exactly-bf16 data, 3 otherwise) is a wave-uniform
// run-time value (flag in device memory for caller data): dispatch once to a fully static body.
template <int NW>
__global__ __launch_bounds__(256) void gemm_up_partial(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    float* __restrict__ partial, int Bp, int kchunk) {
    __shared__ float red[4][2][16][64];
    const int na = operand_terms(a_flag, (lda + 63) / 64, Bp / 8, (blockIdx.y * kchunk) / 64, (blockIdx.y * kchunk + kchunk + 63) / 64, a_terms);
    if (na == 1) gemm_body<true, NW, false, 1>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red);
    else         gemm_body<true, NW, false, 3>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red);
}

template <int NW, bool VEC4>
__global__ __launch_bounds__(256) void gemm_down_partial(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    float* __restrict__ partial, int Bp, int kchunk) {
    __shared__ float red[4][2][16][64];
    const int na = operand_terms(a_flag, (lda + 63) / 64, Bp / 8, (blockIdx.y * kchunk) / 64, (blockIdx.y * kchunk + kchunk + 63) / 64, a_terms);
    if (na == 1) gemm_body<false, NW, VEC4, 1>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red);
    else         gemm_body<false, NW, VEC4, 3>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red);
}

// ------------------------------------------------------------------------------------------
// K1, fast path (N % 4 == 0, 16-B aligned weight rows): float4 weight loads.
// A wave owns 64(batch) x 128(columns) as 2 x 4 MFMA tiles whose columns are INTERLEAVED
// (column of tile t, lane r = n0 + 4r + t): one float4 per lane per K row -> 512-B contiguous row
// segments per half-wave (1.4x the bandwidth of the dword shape, tools/membench).  One wave per SIMD
// with a 2-deep register ring of operand blocks (8 float4 weights + activation fragments per slot);
// ~one block per CU (ksplit = CUs / column tiles), so the split-K slabs shrink to ~8 MB.
// grid = (ceil(N/128), ksplit, Bp/64), block = 256 (the 4 waves interleave 16-row K blocks).
// ------------------------------------------------------------------------------------------
template <int NA>
struct UpOperands {
    float4 wv[8];
    uint4 av[NA][2];
};

template <int NW, int NA>
__device__ __forceinline__ void up4_body(const float* __restrict__ W, int64_t ldw, int K, int N,
                                         const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
                                         float* __restrict__ partial, int Bp, int kchunk, float* red /*[4][4][16][64]*/, int dbg) {
    constexpr int D = 2;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * 128, ks = blockIdx.y, mb = blockIdx.z * 64;
    const int k_begin = ks * kchunk;
    const int k_end = min(k_begin + kchunk, lda);
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;
    const float* wcol = W + min(n0 + 4 * r, N - 4);
    const int64_t arow0 = mb + r, arow1 = mb + 32 + r;
    auto load = [&](UpOperands<NA>& o, int kb) {
        const int kbc = min(kb, lda - 16);                  // clamped: blocks past the end are loaded but never used
        const int k0 = kbc + 8 * hh;
        const int64_t ablk = (int64_t)(kbc >> 4) * Bp;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.wv[j] = *reinterpret_cast<const float4*>(wcol + (int64_t)min(k0 + j, K - 1) * ldw);
#pragma unroll
        for (int ta = 0; ta < NA; ++ta) {
            o.av[ta][0] = *reinterpret_cast<const uint4*>(A + ta * a_term_stride + (ablk + arow0) * 16 + 8 * hh);
            o.av[ta][1] = *reinterpret_cast<const uint4*>(A + ta * a_term_stride + (ablk + arow1) * 16 + 8 * hh);
        }
    };
    auto compute = [&](const UpOperands<NA>& o) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = t == 0 ? o.wv[j].x : (t == 1 ? o.wv[j].y : (t == 2 ? o.wv[j].z : o.wv[j].w));
            uint4 bf[NW];
            make_w_frags<NW>(x, bf);
#pragma unroll
            for (int ta = 0; ta < NA; ++ta)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(o.av[ta][mt]), as_frag(bf[tw]), acc[mt][t], 0, 0, 0);
        }
    };
    UpOperands<NA> ring[D];
    const int kb0 = k_begin + 16 * w;
    const int sblk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const bool st = (dbg & 4) != 0;
    stamp(st, sblk, 1);
#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d], kb0 + 64 * d);
    for (int kb = kb0; kb < k_end; kb += 64 * D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 64 * d < k_end) compute(ring[d]);                    // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            if (kb == kb0 && d == 0) stamp(st, sblk, 2);
            load(ring[d], kb + 64 * (d + D));                             // refill D blocks ahead
        }
    }
    stamp(st, sblk, 3);
    // cross-wave reduction (fixed order) and float4 slab stores: thread -> 4 interleaved columns of one row
    float* slab = partial + (int64_t)ks * Bp * N;
    const bool cok = (n0 + 4 * r) < N;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[((w * 4 + t) * 16 + reg) * 64 + l] = acc[mt][t][reg];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = w * 4 + i;
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                v[t] = ((red[((0 * 4 + t) * 16 + reg) * 64 + l] + red[((1 * 4 + t) * 16 + reg) * 64 + l]) +
                        red[((2 * 4 + t) * 16 + reg) * 64 + l]) + red[((3 * 4 + t) * 16 + reg) * 64 + l];
            const int row = mb + mt * 32 + mfma_row(reg, l);
            if (cok) *reinterpret_cast<float4*>(slab + (int64_t)row * N + n0 + 4 * r) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    stamp(st, sblk, 4);
}

template <int NW>
__global__ __launch_bounds__(256, 1) void gemm_up4_partial(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    float* __restrict__ partial, int Bp, int kchunk, int dbg) {
    __shared__ float red[4 * 4 * 16 * 64];      // 64 KB
    stamp((dbg & 4) != 0, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 0);
    const int na = operand_terms(a_flag, (lda + 63) / 64, Bp / 8, (blockIdx.y * kchunk) / 64, (blockIdx.y * kchunk + kchunk + 63) / 64, a_terms);
    if (na == 1) up4_body<NW, 1>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red, dbg);
    else         up4_body<NW, 3>(W, ldw, K, N, A, a_term_stride, lda, partial, Bp, kchunk, red, dbg);
}

// ------------------------------------------------------------------------------------------
// K2 fused with its epilogue (the hidden dimension K = H is short enough that no split-K is needed):
// block = 4 waves on a 32(v) x 64(batch) tile, two blocks per CU so that all ceil(V/32) tiles of the
// headline shape are resident in ONE round and two waves share each SIMD's VALU / matrix pipes; the waves
// interleave 16-wide K blocks, each behind a 4-deep REGISTER RING of operand blocks (all loads of a block are
// issued 4 blocks ahead, pinned with sched_barrier so the in-order vmcnt waits stay counted), reduce
// through LDS, and the SAME per-element epilogue as `finish` (bias, /T, noise, sigmoid, mu-pull, clamp-mix,
// Bernoulli sampling, operand forms, column sums, squared error) runs in-kernel: no slab round trip,
// one launch less.  Softmax-group columns still leave logits for finish_groups.
// grid = (ceil(V/32), 1, Bp/64), block = 256.
// ------------------------------------------------------------------------------------------
template <int NA>
struct DownOperands {
    float wv[8];
    uint4 av[NA][2];
    uint32_t ab[2];          // BITS: 32 activation bits (two K16 blocks x two halves) of batch rows r and 32 + r
};

// 8 activation bits -> the 8 bf16 values {0, 1} of an MFMA A fragment
__device__ __forceinline__ uint4 bits_to_frag(uint32_t byte) {
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)byte, 2 * j, 1) & 0x3F80u;
        d[j] = ((uint32_t)__builtin_amdgcn_sbfe((int)byte, 2 * j + 1, 1) & 0x3F800000u) | lo;
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
}

// MBB = 64-row batch chunks per block (1, 2 or 4): the four waves are dealt MBB chunks x (4 / MBB) K shares, so a 256-row
// batch takes ONE block per weight tile (every wave a chunk of its own over the whole K, no K shares to combine) instead
// of four blocks in four rounds of the grid.  MBB = 1 is the arrangement described above.
template <bool UP, int NW, bool VEC4, int NA, bool BITS = false, int MBB = 1>
__device__ __forceinline__ void down_fused_body(const float* __restrict__ W, int64_t ldw, int K, int N,
                                                const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
                                                const FinishArgs& fa, float* red /*[4][32][64]*/, float (*tile)[33], int TR,
                                                const uint8_t* __restrict__ abits, int ldbits,
                                                int bx, int bz, int nbx /* tile (bx of nbx, batch chunk bz): blockIdx in the plain kernels */) {
    static_assert(!BITS || NA == 1, "bit-packed activations are single-term");
    constexpr int D = 4;                     // operand ring depth: D x (2 KB weights + activations) in flight per wave
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, hh = l >> 5;
    // the block owns output columns [n0, n0 + TR), TR <= 32 (host: plan_down_rows); MFMA lanes >= TR repeat the last
    // column (same addresses -> no extra traffic) and their results are dropped
    const int r = l & 31, rc = min(r, TR - 1);
    constexpr int KW = 4 / MBB;              // waves sharing one batch chunk's K range
    const int kw = w % KW, cb = w / KW;
    const int n0 = bx * TR, mb = (bz * MBB + cb) * 64;
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    // UP: W is [K][N] (the tile's column, 8 K rows = 8 dword loads); DOWN: W is [N][K] (8 contiguous floats)
    const float* wrow = UP ? (W + min(n0 + rc, N - 1)) : (W + (int64_t)min(n0 + rc, N - 1) * ldw);
    const int64_t arow0 = mb + r, arow1 = mb + 32 + r;
    auto load = [&](DownOperands<NA>& o, int kb) {
        const int kbc = min(kb, lda - 16);              // clamped: blocks past the end are loaded but never used
        const int k0 = kbc + 8 * hh;
        const int64_t ablk = (int64_t)(kbc >> 4) * fa.Bp;
        if constexpr (UP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.wv[j] = wrow[(int64_t)min(k0 + j, K - 1) * ldw];
        } else if constexpr (VEC4) {
            const float4 x0 = *reinterpret_cast<const float4*>(wrow + min(k0, K - 4));
            const float4 x1 = *reinterpret_cast<const float4*>(wrow + min(k0 + 4, K - 4));
            o.wv[0] = x0.x; o.wv[1] = x0.y; o.wv[2] = x0.z; o.wv[3] = x0.w;
            o.wv[4] = x1.x; o.wv[5] = x1.y; o.wv[6] = x1.z; o.wv[7] = x1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.wv[j] = wrow[min(k0 + j, K - 1)];
        }
        if constexpr (BITS) {
            // byte-major bit plane: byte (k >> 3) of a batch row = the 8 k-values of this lane's fragment
            o.ab[0] = abits[(int64_t)((kbc >> 3) + hh) * fa.Bp + arow0];
            o.ab[1] = abits[(int64_t)((kbc >> 3) + hh) * fa.Bp + arow1];
        } else {
#pragma unroll
            for (int ta = 0; ta < NA; ++ta) {
                o.av[ta][0] = *reinterpret_cast<const uint4*>(A + ta * a_term_stride + (ablk + arow0) * 16 + 8 * hh);
                o.av[ta][1] = *reinterpret_cast<const uint4*>(A + ta * a_term_stride + (ablk + arow1) * 16 + 8 * hh);
            }
        }
    };
    auto compute = [&](const DownOperands<NA>& o) {
        uint4 bf[NW];
        make_w_frags<NW>(o.wv, bf);
        if constexpr (BITS) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint4 af = bits_to_frag(o.ab[mt]);
#pragma unroll
                for (int tw = 0; tw < NW; ++tw)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af), as_frag(bf[tw]), acc[mt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ta = 0; ta < NA; ++ta)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(o.av[ta][mt]), as_frag(bf[tw]), acc[mt], 0, 0, 0);
        }
    };
    // K blocks are dealt to the waves in groups of D CONSECUTIVE blocks (group g of wave w = blocks
    // (4g+w)*D .. +D-1), so one pass over the ring reads D*64 B = 256 contiguous bytes of every weight row
    // (whole 128-B lines per wave instead of lines split between waves).  Register ring, static slots.
    const int nblk = lda / 16;
    const int sblk = bz * nbx + bx;
    const bool st = (fa.dbg & 128) != 0;
    stamp(st, sblk, 1);
    DownOperands<NA> ring[D];
    // epilogue side inputs (bias, loss reference, clamp / mu planes) of this thread's column x 8 rows: issued now,
    // consumed after the K loop (their first-touch latency used to sit on every block's critical path)
    SideIn<8> side;
    if constexpr (MBB == 1) load_side<8>(fa, (tid & 31) < TR ? n0 + (tid & 31) : (1 << 30), bz * 64 + (tid >> 5) * 8, side);
#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d], 16 * (kw * D + d));
    for (int g = 0; (KW * g + kw) * D < nblk; ++g) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            __builtin_amdgcn_sched_barrier(0);
            if ((KW * g + kw) * D + d < nblk) compute(ring[d]);           // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            load(ring[d], 16 * ((KW * (g + 1) + kw) * D + d));            // refill the slot for the next group
        }
    }
    // (Tried for the multi-chunk form, 256 x 1500 -> 10000, 118 us as above: ring slots of TWO K16 blocks loaded together so that
    //  the four dwordx4 weight loads of a lane pair fall into one 128-byte line back to back: 125 us; the same with lane (r, hh)
    //  taking 16 consecutive floats and the activations from block 2j + hh: 139 us.  The 32-of-128-byte weight reads per
    //  instruction are not what bounds this kernel.)
    // cross-wave reduction (fixed order) into tile[batch row][column]
    stamp(st, sblk, 2);
    __syncthreads();
    stamp(st, sblk, 3);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[((w * 2 + mt) * 16 + reg) * 64 + l] = acc[mt][reg];
    __syncthreads();
    const int c = tid & 31, oct = tid >> 5;
    const int ecol = c < TR ? n0 + c : (1 << 30);          // columns past the tile: nothing stored
#pragma unroll 1
    for (int ch = 0; ch < MBB; ++ch) {                     // the block's batch chunks, one after the other through `tile`
        const int mbc = (bz * MBB + ch) * 64;
        if constexpr (MBB > 1) load_side<8>(fa, ecol, mbc + oct * 8, side);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cc = w * 8 + i, mt = cc >> 4, reg = cc & 15;
            float s;
            if constexpr (MBB == 1)
                s = ((red[((0 * 2 + mt) * 16 + reg) * 64 + l] + red[((1 * 2 + mt) * 16 + reg) * 64 + l]) +
                     red[((2 * 2 + mt) * 16 + reg) * 64 + l]) + red[((3 * 2 + mt) * 16 + reg) * 64 + l];
            else if constexpr (MBB == 2)
                s = red[(((2 * ch) * 2 + mt) * 16 + reg) * 64 + l] + red[(((2 * ch + 1) * 2 + mt) * 16 + reg) * 64 + l];
            else
                s = red[((ch * 2 + mt) * 16 + reg) * 64 + l];
            tile[mt * 32 + mfma_row(reg, l)][r] = s;
        }
        __syncthreads();
        // epilogue: 256 threads = 32 columns x 8 row-octets
        float xs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[i] = tile[oct * 8 + i][c];
        float lsum = 0.f;
        stamp(st, sblk, 4);
        // the reduction buffer is free again: it stages the K16-blocked operand tile [term][2 column groups][64 rows][16]
        // (single-chunk blocks only: the other chunks' sums are still in it otherwise; the host keeps such launches at MBB = 1)
        const bool staged = MBB == 1 && fa.rm_src && !fa.logits_only && TR == 32;      // needs 16-column-aligned tiles
        const RmStage stg{reinterpret_cast<bf16_t*>(red), 2, 64, n0, mbc};
        lsum = finish_rows8(fa, ecol, mbc + oct * 8, xs, (mbc >> 3) + oct, side, staged ? stg : RmStage{});
        stamp(st, sblk, 5);
        if (staged) {
            __syncthreads();
            flush_rm_stage(fa.op, stg);
        }
        stamp(st, sblk, 6);
        if (fa.loss_part) {
            __syncthreads();
            const float t = wave_sum(lsum);
            if (l == 0) tile[0][w] = t;                    // (the tile has been read; `red` may still hold the other chunks)
            __syncthreads();
            if (tid == 0) fa.loss_part[(bz * MBB + ch) * nbx + bx] = ((tile[0][0] + tile[0][1]) + tile[0][2]) + tile[0][3];
        }
        if constexpr (MBB > 1) __syncthreads();            // `tile` is rewritten by the next chunk
    }
}

// NAK = activation terms known to the host (1 or 3; 0 = decide per block from the exactness map), BITS = the
// activations are the bit-packed sampled states.  One body per instantiation: several bodies inlined into one kernel
// make the register allocator spill (scratch) in the hot loop.
template <int NW, bool VEC4, int NAK, bool BITS, int MBB = 1>
__global__ __launch_bounds__(256, 2) void gemm_down_fused(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    const FinishArgs fa, int tile_rows, const uint8_t* __restrict__ abits, int ldbits) {
    static_assert(MBB == 1 || (NAK == 0 && !BITS && VEC4), "several batch chunks per block: real-valued rows, aligned weights");
    __shared__ float red[4 * 32 * 64];
    __shared__ float tile[64][33];
    stamp((fa.dbg & 128) != 0, blockIdx.z * gridDim.x + blockIdx.x, 0);
    if constexpr (BITS) {
        down_fused_body<false, NW, VEC4, 1, true>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, abits, ldbits, blockIdx.x, blockIdx.z, gridDim.x);
    } else if constexpr (NAK == 1) {
        down_fused_body<false, NW, VEC4, 1>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
    } else if constexpr (NAK == 3) {
        down_fused_body<false, NW, VEC4, 3>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
    } else {
        const int na = operand_terms(a_flag, (lda + 63) / 64, fa.Bp / 8, 0, (lda + 63) / 64, a_terms);
        if (na == 1) down_fused_body<false, NW, VEC4, 1, false, MBB>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
        else         down_fused_body<false, NW, VEC4, 3, false, MBB>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
    }
}

// K1 fused with its epilogue for SHORT visible dimensions (K = V <= 1024: joint RBM, chains): same
// structure as gemm_down_fused with the [K][N] weight access; no split-K slabs, one launch per half step.
// The fused K2 of the first negative-phase step carrying the preparation of the NEXT batch (prep_item_body): the
// launch gets `prep_nbx` extra blocks per batch chunk behind its `main_nbx` weight tiles.  They become a third resident
// block on a CU (K2 runs two), load 16 values per thread, and are gone a few us into the 23 us launch.
// Own instantiations (single-term activations only: the CD step), so the plain kernels stay as tuned.
template <int NW, bool BITS>
__global__ __launch_bounds__(256, 2) void gemm_down_fused_next(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
    const FinishArgs fa, int tile_rows, const uint8_t* __restrict__ abits, int ldbits, const PrepArgs next, int main_nbx) {
    __shared__ float red[4 * 32 * 64];
    __shared__ float tile[64][33];
    if ((int)blockIdx.x >= main_nbx) {
        prep_item_body(next, blockIdx.x - main_nbx, blockIdx.z, reinterpret_cast<bf16_t*>(red));
        return;
    }
    stamp((fa.dbg & 128) != 0, blockIdx.z * main_nbx + blockIdx.x, 0);
    down_fused_body<false, NW, true, 1, BITS>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, tile_rows, abits, ldbits, blockIdx.x, blockIdx.z, main_nbx);
}

template <int NW>
__global__ __launch_bounds__(256, 2) void gemm_up_fused(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    const FinishArgs fa) {
    __shared__ float red[4 * 32 * 64];
    __shared__ float tile[64][33];
    const int na = operand_terms(a_flag, (lda + 63) / 64, fa.Bp / 8, 0, (lda + 63) / 64, a_terms);
    if (na == 1) down_fused_body<true, NW, false, 1>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, 32, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
    else         down_fused_body<true, NW, false, 3>(W, ldw, K, N, A, a_term_stride, lda, fa, red, tile, 32, nullptr, 0, blockIdx.x, blockIdx.z, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// K3: dW[v][h] = sum_b vpos[b][v]*hpos[b][h] - sum_b vneg[b][v]*hneg[b][h], fused with
//     W_m <- mom*W_m + lr*(dW/n - wd*W) ; W <- W + W_m    (one read + one write of W and W_m).
// MODE 1 (data parallel): store the un-normalised dW to `delta` instead and touch no parameter.
// grid = (ceil(H/128), ceil(V/64)), block = 256: wave (wv,wh) owns a 32(v) x 64(h) tile.
// ------------------------------------------------------------------------------------------
struct AssocArgs {
    float* W; float* Wm; int64_t ldw; int V, H;
    const bf16_t* vpos; const int* vpos_flag; int vpos_terms;
    const bf16_t* hpos; int hpos_terms;
    const bf16_t* vneg; const int* vneg_flag; int vneg_terms;
    const bf16_t* hneg; int hneg_terms;
    int64_t vts, hts; int Bp;
    float lr, mom, wd, n;
    float* delta;
    int dbg;
    // batches of more than one 64-row chunk: one launch per chunk, b0 = first batch row of this launch's chunk; the kernel's
    // PASS template argument = 0 single chunk | 1 first | 2 middle | 3 last.  The momentum update is linear in the statistics:
    //   first : W_m <- mom*W_m + lr*(d/n - wd*W)        (W untouched)
    //   middle: W_m <- W_m + lr*d/n
    //   last  : W_m <- W_m + lr*d/n ;  W <- W + W_m
    // (statistics mode: first writes delta, the others add to it)
    int b0;
};

// All operand fragments of one 16-row batch block for this wave's 32(v) x 64(h) tile.
template <int HT, int NAP, int NAN_>
struct AssocFrags {
    uint4 bp[2][HT], bn[2][HT], ap[NAP], an[NAN_];
};

template <int HT, int NAP, int NAN_>
__device__ __forceinline__ void assoc_load(AssocFrags<HT, NAP, NAN_>& f, const AssocArgs& a, int kb, int64_t vrow,
                                           int64_t hrow0, int64_t hrow1) {
    // rows are clamped by the caller (never predicated): out-of-range rows only feed outputs that are not stored
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        f.bp[0][t] = *reinterpret_cast<const uint4*>(a.hpos + t * a.hts + hrow0 + kb);
        f.bp[1][t] = *reinterpret_cast<const uint4*>(a.hpos + t * a.hts + hrow1 + kb);
        f.bn[0][t] = *reinterpret_cast<const uint4*>(a.hneg + t * a.hts + hrow0 + kb);
        f.bn[1][t] = *reinterpret_cast<const uint4*>(a.hneg + t * a.hts + hrow1 + kb);
    }
#pragma unroll
    for (int t = 0; t < NAP; ++t) f.ap[t] = *reinterpret_cast<const uint4*>(a.vpos + t * a.vts + vrow + kb);
#pragma unroll
    for (int t = 0; t < NAN_; ++t) {
        // the negative-phase HIDDEN planes are stored negated by their producer (OperandOut::tr_negate), so one
        // accumulator collects X^T P+ - V'^T P- without touching the fragments
        f.an[t] = *reinterpret_cast<const uint4*>(a.vneg + t * a.vts + vrow + kb);
    }
}

template <int MODE, int HT, int NAP, int NAN_>
__device__ __forceinline__ void assoc_body(const AssocArgs& a) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int v0 = blockIdx.y * 64 + (w >> 1) * 32;
    const int h0 = blockIdx.x * 128 + (w & 1) * 64;
    const int64_t vrow = (int64_t)min(v0 + r, a.V - 1) * a.Bp + 8 * hh;
    const int64_t hrow0 = (int64_t)min(h0 + r, a.H - 1) * a.Bp + 8 * hh;
    const int64_t hrow1 = (int64_t)min(h0 + 32 + r, a.H - 1) * a.Bp + 8 * hh;

    AssocFrags<HT, NAP, NAN_> cur, nxt;
    assoc_load(cur, a, 0, vrow, hrow0, hrow1);
    __builtin_amdgcn_sched_barrier(0);

    // Prefetch this wave's W / W_m tile (C layout: per (nt,reg) two 128-B row segments per wave): all 64
    // loads are in flight together, their HBM latency hides under the operand loads + MFMAs.
    float wold[2][16], mold[2][16];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = min(h0 + nt * 32 + r, a.H - 1);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t idx = (int64_t)min(v0 + mfma_row(reg, l), a.V - 1) * a.ldw + col;
                wold[nt][reg] = a.W[idx];
                mold[nt][reg] = a.Wm[idx];
            }
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;

    auto compute = [&](const AssocFrags<HT, NAP, NAN_>& f) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int tb = 0; tb < HT; ++tb) {
#pragma unroll
                for (int ta = 0; ta < NAP; ++ta)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(f.ap[ta]), as_frag(f.bp[nt][tb]), acc[nt], 0, 0, 0);
#pragma unroll
                for (int ta = 0; ta < NAN_; ++ta)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(f.an[ta]), as_frag(f.bn[nt][tb]), acc[nt], 0, 0, 0);
            }
        }
    };
    // Bp is a multiple of 64: an even number of 16-row blocks, processed ping-pong (no register copies)
    for (int kb = 0; kb < a.Bp; kb += 32) {
        assoc_load(nxt, a, kb + 16, vrow, hrow0, hrow1);
        __builtin_amdgcn_sched_barrier(0);      // keep the load group ahead of (not interleaved with) the MFMAs
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        assoc_load(cur, a, min(kb + 32, a.Bp - 16), vrow, hrow0, hrow1);      // redundant on the last pair
        __builtin_amdgcn_sched_barrier(0);
        compute(nxt);
        __builtin_amdgcn_sched_barrier(0);
    }

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = h0 + nt * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = v0 + mfma_row(reg, l);
            if (row < a.V && col < a.H) {
                const float d = acc[nt][reg];                                     // pos_assoc - neg_assoc
                if constexpr (MODE == 0) {
                    const int64_t idx = (int64_t)row * a.ldw + col;
                    const float w0 = wold[nt][reg];
                    float m = mold[nt][reg];
                    const float g = d / a.n - a.wd * w0;                          // rbm.py:212
                    m = m * a.mom;
                    m = m + a.lr * g;
                    a.Wm[idx] = m;
                    a.W[idx] = w0 + m;                                            // rbm.py:213
                } else {
                    a.delta[(int64_t)row * a.H + col] = d;
                }
            }
        }
    }
}

// HT = terms of the hidden-side operands (3 parity / 1 fast).  The visible-side term counts (1 for
// {0,1} samples or exactly-bf16 data, 3 otherwise) are wave-uniform run-time values: dispatch once to a
// fully static body so every fragment load can be issued ahead of its MFMAs.
template <int MODE, int HT>
__global__ __launch_bounds__(256) void assoc_update(const AssocArgs a) {
    const int ncb = (a.V + 15) / 16 * 16;      // prep's map covers ceil(Vpad/64) feature blocks
    const int v0b = blockIdx.y * 64;
    const int nap = operand_terms(a.vpos_flag, (ncb + 63) / 64, a.Bp / 8, v0b / 64, v0b / 64 + 1, a.vpos_terms);
    const int nan_ = a.vneg_terms;
    if (nap == 1) {
        if (nan_ == 1) assoc_body<MODE, HT, 1, 1>(a);
        else           assoc_body<MODE, HT, 1, 3>(a);
    } else {
        if (nan_ == 1) assoc_body<MODE, HT, 3, 1>(a);
        else           assoc_body<MODE, HT, 3, 3>(a);
    }
}

// ------------------------------------------------------------------------------------------
// K3, fast path (W rows 16-B aligned): bf16 operand planes staged through LDS, float4 weight tiles.
//
//   * Weight traffic is the whole cost (read + write W and W_m = 16 B per element), so the tile is
//     shaped for it: a wave owns 32(v) x 128(h) as FOUR INTERLEAVED 32x32 MFMA tiles
//     (column of tile t, lane r = h0 + 4r + t), i.e. one float4 per lane per accumulator row ->
//     512-B contiguous row segments per half-wave (measured 1.3-1.5x the dword shape, tools/membench).
//     All 32 float4 loads (W, W_m) are issued at kernel entry and land under the staging + MFMAs.
//   * The operands are the pre-split transposed bf16 planes written by finish/prep
//     (tr[t][feature][Bp], 8 batch rows = 16 B).  K = batch is tiny (64 per chunk), so a block copies
//     the planes of its 128 hidden and 128 visible features into LDS with plain coalesced 16-B
//     copies; every fragment is then ONE ds_read_b128 with no VALU work in the MFMA loop.  LDS rows
//     are padded to 144 B and hidden rows are stored permuted (feature 4r+t at row t*32+r) so that
//     consecutive lanes read consecutive rows: (9*row + chunk) mod 16 is a bijection -> conflict-free.
//   * History (profiles/r01_*): fragment loads straight from global were L2-latency bound (98 us);
//     an fp32-MFMA variant was matrix-pipe bound (91 us); fp32 in LDS + in-register splitting was
//     VALU bound at one wave per SIMD (75 us).
// grid = (ceil(H/128), ceil(V/128)), block = 256 (wave w: rows v0+32w..+31), 144 KB dynamic LDS.
// ------------------------------------------------------------------------------------------
constexpr int K3_ROWB = 128;                 // LDS row: 64 bf16 (one batch chunk), unpadded; 16-B chunk c of row r sits at
                                             // position c ^ ((r >> 1) & 7): 16 consecutive rows x one chunk = 64 distinct banks
constexpr int K3_PLANE = 128 * K3_ROWB;      // one plane of 128 hidden features (16 KB)
constexpr int K3_SLICE = 32 * K3_ROWB;       // one wave's 32 visible rows of one plane (4 KB)
constexpr int K3_VIS0 = 6 * K3_PLANE;        // 6 hidden planes (pos / neg x 3 terms), then 4 waves x 4 slices
constexpr int K3_LDS_BYTES = K3_VIS0 + 4 * 4 * K3_SLICE;      // 160 KB: the whole LDS of a CU

struct AssocPlanesArgs {
    float* W; float* Wm; int64_t ldw; int V, H;
    const bf16_t* vpos; const int* vpos_flag; int vpos_terms;
    const bf16_t* hpos;
    const bf16_t* vneg; int vneg_terms;
    const bf16_t* hneg;
    int64_t vts, hts; int Bp;
    float lr, mom, wd, n;
    float* delta;
    int dbg;
    // batches of more than one 64-row chunk: one launch per chunk, b0 = first batch row of this launch's chunk; the kernel's
    // PASS template argument = 0 single chunk | 1 first | 2 middle | 3 last.  The momentum update is linear in the statistics:
    //   first : W_m <- mom*W_m + lr*(d/n - wd*W)        (W untouched)
    //   middle: W_m <- W_m + lr*d/n
    //   last  : W_m <- W_m + lr*d/n ;  W <- W + W_m
    // (statistics mode: first writes delta, the others add to it)
    int b0;
};

__device__ __forceinline__ int k3_swz(int row, int c) { return row * K3_ROWB + ((c ^ ((row >> 1) & 7)) << 4); }

// copy NPL hidden planes of 128 features x 64 batch rows into LDS (whole block, once).  Feature 4r+t goes to
// row t*32+r so that lane r of an MFMA B fragment owns output columns 4r..4r+3 (float4 weight accesses).
// Per plane a thread moves 4 of the 1024 16-B chunks; all loads are issued before the first LDS write.
template <int NPL>
__device__ __forceinline__ void k3_stage(char* dst, const bf16_t* src, int64_t term_stride, int f0, int F,
                                         int Bp, bool negate) {
    const int tid = threadIdx.x;
    uint4 x[NPL][4];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q, row = i >> 3, c = i & 7;
            x[pl][q] = *reinterpret_cast<const uint4*>(src + pl * term_stride + (int64_t)min(f0 + row, F - 1) * Bp + 8 * c);
        }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q, row = i >> 3, c = i & 7;
            uint4 v = x[pl][q];
            if (negate) { v.x ^= 0x80008000u; v.y ^= 0x80008000u; v.z ^= 0x80008000u; v.w ^= 0x80008000u; }
            const int lrow = (row & 3) * 32 + (row >> 2);
            *reinterpret_cast<uint4*>(dst + pl * K3_PLANE + k3_swz(lrow, c)) = v;
        }
    __builtin_amdgcn_sched_barrier(0);     // keep the batches apart: at most NPL*4 staging registers live at once
}

// gfx950 direct global -> LDS load: lane i's 16 bytes land at LDS[lds_off + 16*i] (checked on hardware:
// tools/ldsdma_test.hip).  No destination registers, no ds_write.  Written as asm because the builtin form is a
// FLAT op: the compiler then flushes vmcnt AND lgkmcnt to 0 at the next dependent wait, which would drain the
// weight prefetch.  The compiler does not count these ops; every wait it computes is then at worst too strict
// (vmcnt retires in order), never too weak, and the "slices arrived" wait below is explicit.
__device__ __forceinline__ void k3_dma16(const void* g, uint32_t lds_off) {
    // M0 (the LDS destination base) is compiler-reserved and a clobber of it is not honoured: save and restore it inside
    // the same statement (cdna_hip_programming.md 5.7)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");      // wave-uniform by construction
}

// MFMAs of one visible plane slice (single term, this wave's 32 rows) against HT hidden planes
template <int HT>
__device__ __forceinline__ void k3_mfma_wave(f32x16 (&acc)[4], const char* sH, const char* sVw, int r, int kh) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int off = k3_swz(r, 2 * kb + kh);                  // (t*32 + r) >> 1 & 7 == r >> 1 & 7: same swizzle for every t
        const uint4 af = *reinterpret_cast<const uint4*>(sVw + off);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int tb = 0; tb < HT; ++tb) {
                const uint4 bf = *reinterpret_cast<const uint4*>(sH + tb * K3_PLANE + t * 32 * K3_ROWB + off);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af), as_frag(bf), acc[t], 0, 0, 0);
            }
    }
}

// Streaming form: block (bx, by) owns hidden columns [128*bx, +128) and the `tiles_per_block` visible tiles
// starting at tile by*tiles_per_block; wave w owns rows 32w..32w+31 of every tile.  The hidden planes (pos and
// NEGATED neg, HT terms: 96 KB) are staged once and stay in LDS.  nap / nan = number of bf16 terms (1 or 3) of the
// positive / negative visible operand (block-uniform run-time values), P = nap + nan planes per tile.
// Per tile and wave the VMEM order is
//   [DMA the first min(P,4) plane slices -> LDS] [next tile's W, W_m -> registers] ... [this tile's stores]
// vmcnt retires in order, so a wait for the slices leaves exactly the 32 prefetch loads in flight for the whole
// tile (MFMAs + epilogue); a plane load placed after the prefetch would drain it (only the 3+3-term case has
// such late planes).  No conditional sits between the loads and the wait: the prefetch of a non-existent next
// tile collapses onto one row, unused slices are still loaded (clamped plane index, L2 hits).
// After the hidden staging there is no block barrier: the four waves run independently and drift apart, which
// smooths the load / store bursts.  One wave per SIMD (two 128-register weight tiles in flight).
// One launch handles ONE 64-row batch chunk (a.b0); larger batches take one launch per chunk (template argument PASS).
// Requires 16-B aligned weight rows; otherwise the generic kernel runs.
template <int MODE, int HT, int PASS>
__device__ __forceinline__ void k3_body(const AssocPlanesArgs& a, char* smem, int bx, int by, int tiles_per_block,
                                        int nap, int nan_) {
    const int P = nap + nan_;                                       // 2, 4 or 6 planes per tile
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, kh = l >> 5;
    const int h0 = bx * 128;
    const int tile0 = by * tiles_per_block;
    const int n_vtiles = (a.V + 127) / 128;
    const int n_my = min(tiles_per_block, n_vtiles - tile0);
    const int colc = min(h0 + 4 * r, a.H - 4);
    const bool cok = (h0 + 4 * r) < a.H;          // H % 4 == 0: a float4 is entirely inside or outside
    const bool n_pow2 = (__float_as_uint(a.n) & 0x7fffffu) == 0u;      // d / n == d * (1/n) exactly
    const float inv_n = 1.0f / a.n;

    char* sHp = smem;
    char* sHn = smem + 3 * K3_PLANE;
    char* sV = smem + K3_VIS0 + w * (4 * K3_SLICE);                    // this wave's four slices
    const uint32_t sV_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)sV);

    // Register loads one load_tile issues (middle passes of a multi-chunk batch do not read W).  The hand-written
    // "planes / slices have landed" waits below are s_waitcnt vmcnt(NPF): vmcnt retires in order, the DMAs precede the
    // prefetch, so exactly the NPF prefetch loads may stay in flight.  The count must equal the loads the compiler really
    // emits: tests/test_isa_cpu.py checks it in the gfx950 ISA of every instantiation.
    constexpr int NPF = MODE == 0 ? (PASS == 2 ? 16 : 32) : (PASS >= 2 ? 16 : 0);
    auto load_tile = [&](float4 (&wo)[16], float4 (&mo)[16], int v0, bool valid) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = valid ? min(v0 + 32 * w + mfma_row(reg, l), a.V - 1) : tile0 * 128;
                const int64_t idx = (int64_t)row * a.ldw + colc;
                if constexpr (PASS != 2) wo[reg] = *reinterpret_cast<const float4*>(a.W + idx);
                mo[reg] = *reinterpret_cast<const float4*>(a.Wm + idx);
            }
        } else {
            if constexpr (PASS >= 2) {                           // statistics of a later batch chunk: add to delta
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = valid ? min(v0 + 32 * w + mfma_row(reg, l), a.V - 1) : tile0 * 128;
                    mo[reg] = *reinterpret_cast<const float4*>(a.delta + (int64_t)row * a.H + colc);
                }
            }
        }
    };
    // plane pl (positive terms first, then negative terms) of rows v0w..v0w+31 -> slice `slot`
    auto dma_plane = [&](int slot, int pl, int v0w) {
        pl = min(pl, P - 1);
        const bf16_t* src = pl < nap ? a.vpos + pl * a.vts : a.vneg + (pl - nap) * a.vts;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = l + 64 * q, row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);      // source chunk for LDS position i
            k3_dma16(src + (int64_t)min(v0w + row, a.V - 1) * a.Bp + a.b0 + 8 * c, sV_lds + slot * K3_SLICE + q * 1024);
        }
    };

    float4 wA[16], mA[16], wB[16], mB[16];
    const bool st = a.dbg != 0;
    const int sblk = by * gridDim.x + bx;
    stamp(st, sblk, 0);
    // hidden planes, once per block: 2 x HT planes x 16 KB straight into LDS (gfx950 LDS-DMA, no staging registers),
    // 16 wave-instructions per plane dealt to the four waves; THEN the first weight tile, so that "planes arrived"
    // is a vmcnt(32) that leaves the weight loads in flight.  LDS chunk i' = 8*lrow + pos holds chunk
    // pos ^ ((lrow >> 1) & 7) of feature row 4*(lrow & 31) + (lrow >> 5) (lane r of a B fragment owns columns 4r..4r+3).
    // The negative-phase planes are stored negated by their producer (OperandOut::tr_negate).
    const uint32_t sH_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int tb = 0; tb < HT; ++tb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * w + jj, i = 64 * j + l, lrow = i >> 3, pos = i & 7;
                const int row = 4 * (lrow & 31) + (lrow >> 5), c = pos ^ ((lrow >> 1) & 7);
                const bf16_t* src = (ph ? a.hneg : a.hpos) + tb * a.hts + (int64_t)min(h0 + row, a.H - 1) * a.Bp + a.b0 + 8 * c;
                k3_dma16(src, sH_lds + (3 * ph + tb) * K3_PLANE + j * 1024);
            }
    __builtin_amdgcn_sched_barrier(0);
    load_tile(wA, mA, tile0 * 128, true);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(MODE == 0 ? NPF : 0) : "memory");
    __syncthreads();                                                    // the ONLY block barrier
    stamp(st, sblk, 1);

    auto tile = [&](int it, float4 (&wc)[16], float4 (&mc)[16], float4 (&wn)[16], float4 (&mn)[16]) {
        const int v0 = (tile0 + it) * 128;
        // the previous tile's MFMAs have consumed their fragments (ds_reads retire before the MFMA issues)
#pragma unroll
        for (int p = 0; p < 4; ++p) dma_plane(p, p, v0 + 32 * w);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(wn, mn, v0 + 128, it + 1 < n_my);                      // next tile's weights: in flight for the whole tile
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(MODE == 0 ? NPF : 0) : "memory");     // slices arrived; prefetch still in flight
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if (p < P) {                                                  // block-uniform
                if (p >= 4) {                                             // 3+3 terms only: planes 4, 5 reuse slices 0, 1
                    if (p == 4) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // fragments of planes 0, 1 are in registers
                        dma_plane(0, 4, v0 + 32 * w);
                        dma_plane(1, 5, v0 + 32 * w);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                k3_mfma_wave<HT>(acc, p < nap ? sHp : sHn, sV + (p & 3) * K3_SLICE, r, kh);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // all fragment reads done before the next tile's DMA overwrites the slices
        auto epilogue = [&](auto pow2_tag) {
            constexpr bool POW2 = decltype(pow2_tag)::value;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = v0 + 32 * w + mfma_row(reg, l);
                if (row < a.V && cok) {
                    const float4 d = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);   // pos_assoc - neg_assoc
                    if constexpr (MODE == 0) {
                        const int64_t idx = (int64_t)row * a.ldw + h0 + 4 * r;
                        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (PASS != 2) w0 = wc[reg];                           // middle passes never load W
                        float4 m = mc[reg];
                        const float gx = POW2 ? d.x * inv_n : d.x / a.n, gy = POW2 ? d.y * inv_n : d.y / a.n;
                        const float gz = POW2 ? d.z * inv_n : d.z / a.n, gw = POW2 ? d.w * inv_n : d.w / a.n;
                        if constexpr (PASS <= 1) {                                       // single chunk, or the first of several
                            m.x = m.x * a.mom; m.x = m.x + a.lr * (gx - a.wd * w0.x);        // rbm.py:212
                            m.y = m.y * a.mom; m.y = m.y + a.lr * (gy - a.wd * w0.y);
                            m.z = m.z * a.mom; m.z = m.z + a.lr * (gz - a.wd * w0.z);
                            m.w = m.w * a.mom; m.w = m.w + a.lr * (gw - a.wd * w0.w);
                        } else {                                                         // later chunks: the statistics term only
                            m.x = m.x + a.lr * gx; m.y = m.y + a.lr * gy; m.z = m.z + a.lr * gz; m.w = m.w + a.lr * gw;
                        }
                        *reinterpret_cast<float4*>(a.Wm + idx) = m;
                        if constexpr (PASS == 0 || PASS == 3)
                            *reinterpret_cast<float4*>(a.W + idx) = make_float4(w0.x + m.x, w0.y + m.y, w0.z + m.z, w0.w + m.w);   // :213
                        // (non-temporal stores were tried here in round 2: 51.3 us against 49.4 event-timed -- the tail of this kernel is
                        //  not dirty lines lingering in L2.  Round 3: write-through (sc1) stores for the LAST tile of every block, so that
                        //  the launch ends with nothing dirty in the L2s: 46.7 us with and without, rocprofv3, same box)
                    } else {
                        float4 o = d;
                        if constexpr (PASS >= 2) { const float4 p = mc[reg]; o = make_float4(p.x + d.x, p.y + d.y, p.z + d.z, p.w + d.w); }
                        *reinterpret_cast<float4*>(a.delta + (int64_t)row * a.H + h0 + 4 * r) = o;
                    }
                }
            }
        };
        if (n_pow2) epilogue(std::true_type{}); else epilogue(std::false_type{});
        stamp(st, sblk, 2 + it);
    };
    for (int it = 0; it < n_my; it += 2) {
        tile(it, wA, mA, wB, mB);
        if (it + 1 < n_my) tile(it + 1, wB, mB, wA, mA);
    }
}

// ---- data-parallel factor exchange: the rank blocks are looped INSIDE the tile ------------------------------
// One launch per rank block (PASS 1/2/3 above) re-reads and re-writes W_m for every rank: 46 + 29 (R - 1) us.
// Here the statistics of all R gathered blocks are accumulated in the MFMA accumulators of one tile before its single
// read-modify-write: the weights move once whatever R is.  Per (tile, rank) the block restages that rank's hidden
// planes (96 KB from L2, LDS-DMA) and its visible slices; two block barriers per rank (the hidden planes are shared
// by the four waves), so the barrier-free schedule of the single-block kernel does not apply here.
// strides in bf16 elements between the rank blocks: _h for the hidden planes and the exactness map (ints: stride_h / 2),
// _v for the visible planes (they differ when the head of the blocks is read from the gathered wire blocks)
struct RankLoopArgs { int n_ranks; int64_t stride_h, stride_v; };

template <int HT>
__device__ __forceinline__ void k3_body_ranks(const AssocPlanesArgs& a, const RankLoopArgs& rl, char* smem, int bx, int by,
                                              int tiles_per_block, int nap, int nan_) {
    const int P = nap + nan_;                                       // planes per tile and rank: 2 or 4 (host guarantees <= 4)
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, kh = l >> 5;
    const int h0 = bx * 128;
    const int tile0 = by * tiles_per_block;
    const int n_vtiles = (a.V + 127) / 128;
    const int n_my = min(tiles_per_block, n_vtiles - tile0);
    const int colc = min(h0 + 4 * r, a.H - 4);
    const bool cok = (h0 + 4 * r) < a.H;
    const bool n_pow2 = (__float_as_uint(a.n) & 0x7fffffu) == 0u;
    const float inv_n = 1.0f / a.n;
    char* sHp = smem;
    char* sHn = smem + 3 * K3_PLANE;
    char* sV = smem + K3_VIS0 + w * (4 * K3_SLICE);
    const uint32_t sV_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)sV);
    const uint32_t sH_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);

    auto load_tile = [&](float4 (&wo)[16], float4 (&mo)[16], int v0, bool valid) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = valid ? min(v0 + 32 * w + mfma_row(reg, l), a.V - 1) : tile0 * 128;
            const int64_t idx = (int64_t)row * a.ldw + colc;
            wo[reg] = *reinterpret_cast<const float4*>(a.W + idx);
            mo[reg] = *reinterpret_cast<const float4*>(a.Wm + idx);
        }
    };
    auto dma_rank = [&](int rk, int v0w) {                          // hidden planes + this wave's visible slices of rank rk
        const int64_t ro = (int64_t)rk * rl.stride_h, rov = (int64_t)rk * rl.stride_v;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int tb = 0; tb < HT; ++tb)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = 4 * w + jj, i = 64 * j + l, lrow = i >> 3, pos = i & 7;
                    const int row = 4 * (lrow & 31) + (lrow >> 5), c = pos ^ ((lrow >> 1) & 7);
                    const bf16_t* src = (ph ? a.hneg : a.hpos) + ro + tb * a.hts + (int64_t)min(h0 + row, a.H - 1) * a.Bp + 8 * c;
                    k3_dma16(src, sH_lds + (3 * ph + tb) * K3_PLANE + j * 1024);
                }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int pl = min(p, P - 1);
            const bf16_t* src = (pl < nap ? a.vpos + pl * a.vts : a.vneg + (pl - nap) * a.vts) + rov;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = l + 64 * q, row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
                k3_dma16(src + (int64_t)min(v0w + row, a.V - 1) * a.Bp + 8 * c, sV_lds + p * K3_SLICE + q * 1024);
            }
        }
    };

    float4 wA[16], mA[16], wB[16], mB[16];
    load_tile(wA, mA, tile0 * 128, true);
    auto tile = [&](int it, float4 (&wc)[16], float4 (&mc)[16], float4 (&wn)[16], float4 (&mn)[16]) {
        const int v0 = (tile0 + it) * 128;
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int rk = 0; rk < rl.n_ranks; ++rk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                         // every wave is done with the previous rank's planes
            dma_rank(rk, v0 + 32 * w);
            __builtin_amdgcn_sched_barrier(0);
            if (rk == 0) {                                           // next tile's weights: behind this rank's planes in the VMEM queue
                load_tile(wn, mn, v0 + 128, it + 1 < n_my);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // later ranks' planes queue behind the prefetch: it has had a rank's time
            }
            __syncthreads();                                         // the hidden planes (staged by all four waves) are complete
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < P) k3_mfma_wave<HT>(acc, p < nap ? sHp : sHn, sV + p * K3_SLICE, r, kh);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto epilogue = [&](auto pow2_tag) {
            constexpr bool POW2 = decltype(pow2_tag)::value;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = v0 + 32 * w + mfma_row(reg, l);
                if (row < a.V && cok) {
                    const float4 d = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
                    const int64_t idx = (int64_t)row * a.ldw + h0 + 4 * r;
                    const float4 w0 = wc[reg];
                    float4 m = mc[reg];
                    const float gx = POW2 ? d.x * inv_n : d.x / a.n, gy = POW2 ? d.y * inv_n : d.y / a.n;
                    const float gz = POW2 ? d.z * inv_n : d.z / a.n, gw = POW2 ? d.w * inv_n : d.w / a.n;
                    m.x = m.x * a.mom; m.x = m.x + a.lr * (gx - a.wd * w0.x);        // rbm.py:212
                    m.y = m.y * a.mom; m.y = m.y + a.lr * (gy - a.wd * w0.y);
                    m.z = m.z * a.mom; m.z = m.z + a.lr * (gz - a.wd * w0.z);
                    m.w = m.w * a.mom; m.w = m.w + a.lr * (gw - a.wd * w0.w);
                    *reinterpret_cast<float4*>(a.Wm + idx) = m;
                    *reinterpret_cast<float4*>(a.W + idx) = make_float4(w0.x + m.x, w0.y + m.y, w0.z + m.z, w0.w + m.w);   // :213
                }
            }
        };
        if (n_pow2) epilogue(std::true_type{}); else epilogue(std::false_type{});
    };
    for (int it = 0; it < n_my; it += 2) {
        tile(it, wA, mA, wB, mB);
        if (it + 1 < n_my) tile(it + 1, wB, mB, wA, mA);
    }
}

// Rank loop OUTSIDE the tile loop (tiles_per_block <= 4): a rank's hidden planes are staged once and meet all of the
// block's visible tiles (one accumulator set per tile: 4 x 64 registers, AGPRs), the visible slices of the next tile
// are staged while the current one multiplies (two slot pairs when a tile needs <= 2 planes: binary data), and the
// weights are read-modify-written in a second phase, once.  The tile-wise order above restages 96 KB of hidden planes
// for every (tile, rank) pair with nothing to overlap them with: 8 rank blocks cost 189 us there.
template <int HT>
__device__ __forceinline__ void k3_body_ranks_acc(const AssocPlanesArgs& a, const RankLoopArgs& rl, char* smem, int bx, int by,
                                                  int tiles_per_block, int nap, int nan_) {
    const int P = nap + nan_;                                       // planes per tile and rank: 2 or 4
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, kh = l >> 5;
    const int h0 = bx * 128;
    const int tile0 = by * tiles_per_block;
    const int n_vtiles = (a.V + 127) / 128;
    const int n_my = min(tiles_per_block, n_vtiles - tile0);
    const int colc = min(h0 + 4 * r, a.H - 4);
    const bool cok = (h0 + 4 * r) < a.H;
    const bool n_pow2 = (__float_as_uint(a.n) & 0x7fffffu) == 0u;
    const float inv_n = 1.0f / a.n;
    char* sHp = smem;
    char* sHn = smem + 3 * K3_PLANE;
    char* sV = smem + K3_VIS0 + w * (4 * K3_SLICE);
    const uint32_t sV_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)sV);
    const uint32_t sH_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    const bool dbl = P <= 2;                                        // two slot pairs: stage tile t+1 under the MFMAs of tile t

    auto dma_hidden = [&](int rk) {                                  // 2 x HT planes of rank rk, all four waves: 8 HT ops per wave
        const int64_t ro = (int64_t)rk * rl.stride_h;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int tb = 0; tb < HT; ++tb)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = 4 * w + jj, i = 64 * j + l, lrow = i >> 3, pos = i & 7;
                    const int row = 4 * (lrow & 31) + (lrow >> 5), c = pos ^ ((lrow >> 1) & 7);
                    const bf16_t* src = (ph ? a.hneg : a.hpos) + ro + tb * a.hts + (int64_t)min(h0 + row, a.H - 1) * a.Bp + 8 * c;
                    k3_dma16(src, sH_lds + (3 * ph + tb) * K3_PLANE + j * 1024);
                }
    };
    // this wave's visible slices of (rank rk, rows v0w ..): planes 0..1 into slots base, base+1 (dbl) or planes 0..3 into slots 0..3;
    // always 8 (dbl) or 16 ops
    auto dma_slices = [&](int rk, int v0w, int base) {
        const int64_t ro = (int64_t)rk * rl.stride_v;
        const int np = dbl ? 2 : 4;
        for (int p = 0; p < np; ++p) {
            const int pl = min(p, P - 1);
            const bf16_t* src = (pl < nap ? a.vpos + pl * a.vts : a.vneg + (pl - nap) * a.vts) + ro;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = l + 64 * q, row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
                k3_dma16(src + (int64_t)min(v0w + row, a.V - 1) * a.Bp + 8 * c, sV_lds + (base + p) * K3_SLICE + q * 1024);
            }
        }
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[it][t][i] = 0.f;

    for (int rk = 0; rk < rl.n_ranks; ++rk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                             // every wave is done with the previous rank's planes and slices
        dma_hidden(rk);
        dma_slices(rk, tile0 * 128 + 32 * w, 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (it < n_my) {                                         // block-uniform
                const int base = dbl ? 2 * (it & 1) : 0;
                if (dbl) {
                    if (it + 1 < n_my) {
                        dma_slices(rk, (tile0 + it + 1) * 128 + 32 * w, 2 * ((it + 1) & 1));      // its previous readers (tile it-1) are done: same wave
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                           // everything but those 8 ops has landed
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                } else {
                    if (it > 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // tile it-1's LDS reads are done
                        dma_slices(rk, (tile0 + it) * 128 + 32 * w, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (it == 0) __syncthreads();                        // the hidden planes (staged by all four waves) are complete
                for (int p = 0; p < P; ++p)
                    k3_mfma_wave<HT>(acc[it], p < nap ? sHp : sHn, sV + (base + p) * K3_SLICE, r, kh);
                if (dbl) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the slot pair is restaged two tiles later
            }
        }
    }
    // phase 2: one read-modify-write of the block's weight tiles
    float4 wc[16], mc[16];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it < n_my) {
            const int v0 = (tile0 + it) * 128;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = min(v0 + 32 * w + mfma_row(reg, l), a.V - 1);
                const int64_t idx = (int64_t)row * a.ldw + colc;
                wc[reg] = *reinterpret_cast<const float4*>(a.W + idx);
                mc[reg] = *reinterpret_cast<const float4*>(a.Wm + idx);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = v0 + 32 * w + mfma_row(reg, l);
                if (row < a.V && cok) {
                    const float4 d = make_float4(acc[it][0][reg], acc[it][1][reg], acc[it][2][reg], acc[it][3][reg]);
                    const int64_t idx = (int64_t)row * a.ldw + h0 + 4 * r;
                    const float4 w0 = wc[reg];
                    float4 m = mc[reg];
                    const float gx = n_pow2 ? d.x * inv_n : d.x / a.n, gy = n_pow2 ? d.y * inv_n : d.y / a.n;
                    const float gz = n_pow2 ? d.z * inv_n : d.z / a.n, gw = n_pow2 ? d.w * inv_n : d.w / a.n;
                    m.x = m.x * a.mom; m.x = m.x + a.lr * (gx - a.wd * w0.x);        // rbm.py:212
                    m.y = m.y * a.mom; m.y = m.y + a.lr * (gy - a.wd * w0.y);
                    m.z = m.z * a.mom; m.z = m.z + a.lr * (gz - a.wd * w0.z);
                    m.w = m.w * a.mom; m.w = m.w + a.lr * (gw - a.wd * w0.w);
                    *reinterpret_cast<float4*>(a.Wm + idx) = m;
                    *reinterpret_cast<float4*>(a.W + idx) = make_float4(w0.x + m.x, w0.y + m.y, w0.z + m.z, w0.w + m.w);   // :213
                }
            }
        }
    }
}

// ACC = rank loop outside the tile loop (k3_body_ranks_acc; needs tiles_per_block <= 4); one body per kernel
template <int HT, bool ACC>
__global__ __launch_bounds__(256, 1) void assoc_update_planes_ranks(const AssocPlanesArgs a, const RankLoopArgs rl, int tiles_per_block,
                                                                    const BiasArgs bias, int bias_rows) {
    __shared__ __attribute__((aligned(16))) char smem[K3_LDS_BYTES];
    if (bias_rows > 0 && (int)blockIdx.y >= (int)gridDim.y - bias_rows) {
        bias_work(bias, (blockIdx.y - (gridDim.y - bias_rows)) * gridDim.x + blockIdx.x, bias_rows * gridDim.x,
                  reinterpret_cast<double*>(smem));
        return;
    }
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nbx = gridDim.x, nby = gridDim.y - bias_rows;
        int xa = 0;
        if (nbx % 2 == 0 && nby % 4 == 0) xa = 2; else if (nbx % 4 == 0 && nby % 2 == 0) xa = 4;
        else if (nby % 8 == 0) xa = 1; else if (nbx % 8 == 0) xa = 8;
        if (xa) {
            const int xc = 8 / xa, sa = nbx / xa, sc = nby / xc;
            const int p = blockIdx.y * nbx + blockIdx.x;
            const int xcd = p & 7, slot = p >> 3;
            bx = (xcd % xa) * sa + slot % sa;
            by = (xcd / xa) * sc + slot / sa;
        }
    }
    const int tile0 = by * tiles_per_block;
    const int ncbv = ((a.V + 15) / 16 * 16 + 63) / 64;
    int nap = a.vpos_terms;
    if (nap == 0) {                                                  // any rank with inexact data in these tiles -> three terms for all
        nap = 1;
        for (int rk = 0; rk < rl.n_ranks; ++rk)
            nap = max(nap, operand_terms(a.vpos_flag + rk * (rl.stride_h / 2), ncbv, a.Bp / 8, (tile0 * 128) / 64, (tile0 + tiles_per_block) * 2, 0));
    }
    if constexpr (ACC) k3_body_ranks_acc<HT>(a, rl, smem, bx, by, tiles_per_block, nap, a.vneg_terms);
    else               k3_body_ranks<HT>(a, rl, smem, bx, by, tiles_per_block, nap, a.vneg_terms);
}

template <int MODE, int HT, int PASS>
__global__ __launch_bounds__(256, 1) void assoc_update_planes(const AssocPlanesArgs a, int tiles_per_block,
                                                              const BiasArgs bias, int bias_rows) {
    __shared__ __attribute__((aligned(16))) char smem[K3_LDS_BYTES];      // 160 KB static: all of the CU's LDS
    // (Round 2 tried again to let the next batch's preparation ride here, on the CUs this ~one-block-per-CU grid leaves idle,
    //  now with the slim output set of a 0/1 batch (0.5 MB of stores instead of 7.7 MB) and with three items of loads in flight
    //  per worker: 68 us and 92 us for this kernel instead of 45.6.  Under this kernel's read + write stream nothing else makes
    //  progress; the preparation rides on a read-only stream instead: k1_stream's negative-phase launch.)
    // The last `bias_rows` block rows of the grid do the (tiny, independent) bias / loss update of
    // rbm.py:216-226 instead of a dependent launch of their own: they only touch the bias vectors.
    if (bias_rows > 0 && (int)blockIdx.y >= (int)gridDim.y - bias_rows) {
        bias_work(bias, (blockIdx.y - (gridDim.y - bias_rows)) * gridDim.x + blockIdx.x, bias_rows * gridDim.x,
                  reinterpret_cast<double*>(smem));
        return;
    }
    // XCD-aware block -> (hidden tile bx, visible chunk by) map.  Blocks are dealt round-robin to the 8 XCDs
    // (private L2 each), so give XCD x the sub-grid (xa of the hidden tiles) x (xc of the visible chunks),
    // xa*xc = 8: the operand planes a block stages were then already fetched into THIS L2 by a neighbour
    // (plane traffic ~60 MB -> ~11 MB at 10000x1500).  Placement only changes speed, never results.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nbx = gridDim.x, nby = gridDim.y - bias_rows;
        int xa = 0;
        if (nbx % 2 == 0 && nby % 4 == 0) xa = 2; else if (nbx % 4 == 0 && nby % 2 == 0) xa = 4;
        else if (nby % 8 == 0) xa = 1; else if (nbx % 8 == 0) xa = 8;
        if (xa) {
            const int xc = 8 / xa, sa = nbx / xa, sc = nby / xc;      // sub-grid of one XCD: sa x sc blocks
            const int p = blockIdx.y * nbx + blockIdx.x;              // dispatch order
            const int xcd = p & 7, slot = p >> 3;
            bx = (xcd % xa) * sa + slot % sa;
            by = (xcd / xa) * sc + slot / sa;
        }
    }
    const int tile0 = by * tiles_per_block;
    const int ncbv = ((a.V + 15) / 16 * 16 + 63) / 64;
    const int nap = operand_terms(a.vpos_flag, ncbv, a.Bp / 8, (tile0 * 128) / 64, (tile0 + tiles_per_block) * 2, a.vpos_terms);
    const int nan_ = a.vneg_terms;
    k3_body<MODE, HT, PASS>(a, smem, bx, by, tiles_per_block, nap, nan_);
}

// ------------------------------------------------------------------------------------------
// Fused K2 for MANY real-valued batch rows (decode / visible_probs of >= 128 rows; rbm.py:94-116,148-151, idbn.py:356-359).
// gemm_down_fused gives every 32-row weight tile its own block, and every block re-reads the activation terms of its batch rows
// from L2 (2.3 MB per tile at 256 x 1500 x 3 terms: 0.94 GB per launch).  Here a block owns 128 weight rows x 64 batch rows: wave w
// multiplies rows 32 w .. 32 w + 31 over the WHOLE K, its weight fragments come straight from global memory into registers (each
// weight is read by exactly one wave), and the activation terms of a K32 slab (NA x 4 KB) are staged ONCE per block by LDS-DMA (two
// slots) and read by all four waves (0.42 GB per launch).  Per iteration: everything requested has landed (vmcnt(0)) -> barrier ->
// fp32 weights of this slab -> bf16 terms (registers) -> request the next slab (weights to registers, A terms to the other slot) ->
// 2 x 2 x NA x NW MFMAs with A fragments from LDS.  No K split, no cross-wave sum; the epilogue is the one of gemm_down_fused, run
// once per 32-row wave tile.  grid = (ceil(Vpad / 128), 1, Bp / 64), block = 256.
// 256 x 1500 -> 10000: 105 us (one block per (tile, chunk): 130; chunks per block: 118).  Still 3.7x the MFMA time at peak: what is
// left is arithmetic on the SIMDs that host two waves -- per wave 54 k cycles of MFMA and ~30 k of splitting fp32 weights into bf16
// terms, redone by each of the four batch chunks.  Splitting W once per launch (as the chain kernel does) is the open step.
// (Two batch chunks per block -- 128 x 128 tiles, 158 blocks, one wave per SIMD with the accumulators partly in AGPRs -- was built
//  and was correct, and took 245 us.)
// ------------------------------------------------------------------------------------------
template <int NW, int NA>
__device__ __forceinline__ void down_tiled_body(const float* __restrict__ W, int64_t ldw, int K, int N,
                                                const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
                                                const FinishArgs& fa, char* smem) {
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, r = l & 31, hh = l >> 5;
    const int bx = blockIdx.x, bz = blockIdx.z;
    const int n0 = bx * 128, mb = bz * 64;
    const float* wrow = W + (int64_t)min(n0 + 32 * w + r, N - 1) * ldw;
    const int nblk = lda / 16, nslab = (nblk + 1) / 2;
    constexpr int SLOT = NA * 4096;                            // NA terms x 2 K16 blocks x (64 rows x 32 B)
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    // slab s -> slot s & 1: 4 NA instructions of 1 KB, instruction i = ((term * 2 + K16 block) * 2 + row half), dealt to the four waves
    auto issue_a = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int i = 4 * q + w;
            const int t = i >> 2, kb2 = (i >> 1) & 1, half = i & 1;
            const int kb = min(2 * s + kb2, nblk - 1);
            k3_dma16(A + t * a_term_stride + ((int64_t)kb * fa.Bp + mb + 32 * half) * 16 + 8 * l, ring_lds + (s & 1) * SLOT + i * 1024);
        }
    };
    float wv[16];                                              // this lane's weights of a slab: [K16 block][8 consecutive k]
    auto load_w = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
            const int k0 = 16 * min(2 * s + kb2, nblk - 1) + 8 * hh;
            const float4 x0 = *reinterpret_cast<const float4*>(wrow + min(k0, K - 4));
            const float4 x1 = *reinterpret_cast<const float4*>(wrow + min(k0 + 4, K - 4));
            wv[8 * kb2 + 0] = x0.x; wv[8 * kb2 + 1] = x0.y; wv[8 * kb2 + 2] = x0.z; wv[8 * kb2 + 3] = x0.w;
            wv[8 * kb2 + 4] = x1.x; wv[8 * kb2 + 5] = x1.y; wv[8 * kb2 + 6] = x1.z; wv[8 * kb2 + 7] = x1.w;
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    issue_a(0);
    load_w(0);
    // (Requesting TWO slabs ahead -- three slots, two register sets, a counted vmcnt -- changed nothing: 110 us against 105 at
    //  256 x 1500 -> 10000.  Neither the bytes in flight nor the re-read activation terms bound this launch: the SIMDs that host two
    //  waves (316 blocks on 256 CUs) spend ~84 k cycles per wave, 54 k of MFMA and ~30 k splitting fp32 weights into bf16 terms.)
    for (int s = 0; s < nslab; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's share of slab s has landed; its LDS reads of slab s - 1 are done
        __syncthreads();                                                // ... and so has everybody's: slot s & 1 is complete, the other one is free
        uint4 bf[2][NW];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = wv[8 * kb2 + j];
            make_w_frags<NW>(x, bf[kb2]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nslab) {                                             // block-uniform
            issue_a(s + 1);
            load_w(s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* slot = smem + (s & 1) * SLOT + r * 32 + 16 * hh;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
            if (2 * s + kb2 < nblk) {                                    // block-uniform: an odd number of K16 blocks
#pragma unroll
                for (int t = 0; t < NA; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const uint4 af = *reinterpret_cast<const uint4*>(slot + ((t * 2 + kb2) * 2 + mt) * 1024);
#pragma unroll
                        for (int tw = 0; tw < NW; ++tw)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af), as_frag(bf[kb2][tw]), acc[mt], 0, 0, 0);
                    }
            }
        }
    }
    // epilogue: the ring is free; red = [wave][2][16][64] raw sums, then one 32-row wave tile after the other through `tile`
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    float (*tile)[33] = reinterpret_cast<float (*)[33]>(smem + 32768);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[((w * 2 + mt) * 16 + reg) * 64 + l] = acc[mt][reg];
    __syncthreads();
    const int c = tid & 31, oct = tid >> 5;
    const int nt32 = 4 * gridDim.x;                         // 32-row tiles of the launch (= squared-error partials per batch chunk)
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        const int ecol = n0 + 32 * ch + c;                  // columns >= N: nothing stored
        SideIn<8> side;
        load_side<8>(fa, ecol, mb + oct * 8, side);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cc = (tid >> 6) * 8 + i, mt = cc >> 4, reg = cc & 15;
            tile[mt * 32 + mfma_row(reg, l)][r] = red[((ch * 2 + mt) * 16 + reg) * 64 + l];
        }
        __syncthreads();
        float xs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[i] = tile[oct * 8 + i][c];
        const float lsum = finish_rows8(fa, ecol, mb + oct * 8, xs, (mb >> 3) + oct, side, RmStage{});
        __syncthreads();
        if (fa.loss_part) {
            const float t = wave_sum(lsum);
            if (l == 0) tile[0][tid >> 6] = t;
            __syncthreads();
            if (tid == 0) fa.loss_part[bz * nt32 + 4 * bx + ch] = ((tile[0][0] + tile[0][1]) + tile[0][2]) + tile[0][3];
            __syncthreads();
        }
    }
}

template <int NW>
__global__ __launch_bounds__(256, 2) void gemm_down_tiled(const float* __restrict__ W, int64_t ldw, int K, int N,
                                                          const bf16_t* __restrict__ A, int64_t a_term_stride, int lda,
                                                          const int* __restrict__ a_flag, int a_terms, const FinishArgs fa) {
    __shared__ __attribute__((aligned(16))) char smem[32768 + 64 * 33 * 4];      // activation ring (2 x NA x 4 KB), then the epilogue's staging
    const int na = operand_terms(a_flag, (lda + 63) / 64, fa.Bp / 8, 0, (lda + 63) / 64, a_terms);
    if (na == 1) down_tiled_body<NW, 1>(W, ldw, K, N, A, a_term_stride, lda, fa, smem);
    else         down_tiled_body<NW, 3>(W, ldw, K, N, A, a_term_stride, lda, fa, smem);
}

}  // namespace imdbn
