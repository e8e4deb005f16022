// Row-parallel persistent chain kernel ("K4").
//
// The Gibbs / mean-field chains of the joint RBM (rbm.py:240-400 conditional_gibbs*, noisy_meanfield_annealed; the
// positive phase of train_epoch_clamped; imdbn.py:386-488 _cross_reconstruct) are 30-72 dependent half steps on a
// small RBM (532 <-> 256).  As one launch per half step they are pure launch / kernel-boundary latency (~49 us per
// step), and a grid-wide barrier costs as much as a launch on this part.  But batch rows never
// interact inside a chain, so here a block owns a few batch rows and runs the WHOLE chain for them:
//   * W is split once per call into fp16 hi/lo planes (k4_terms) in MFMA-fragment order for both directions
//     (k4_split_planes: plane[dir][term][n-tile][k-block] = one contiguous KB = one wave load);
//   * the visible / hidden activations of its rows live in LDS as the same two fp16 terms;
//   * every half step streams the planes from L2 through v_mfma_f32_16x16x32_f16 (8 waves split the 16-column
//     output tiles behind a 6-deep register ring; measured at the L2->CU fill rate, ~117 GB/s), stages the raw sums
//     in LDS and applies the same epilogue arithmetic as kernels_ew.hpp finish_rows / finish_groups, one element
//     per thread;
//   * only `rows` (4 or 8) of the 16 MFMA rows carry data: the element-wise work (Philox, Box-Muller, sigmoid) of a
//     half step on ONE CU costs more than the GEMM, so the batch is spread over 4x more CUs instead;
//   * the per-step schedule (temperature, noise, mu-pull, clamp, sampling, draw cursors) is a record array in the
//     workspace; the final visible state leaves as fp32 and the caller re-derives the operand forms from it.
// No inter-block communication, no barrier other than __syncthreads().
#pragma once
#include "kernels_ew.hpp"

#ifndef K4_NO_MFMA
#define K4_NO_MFMA 0         // 1: timing experiment (round 2): without any matrix work the GEMM phases still take 6.8 - 7.0 us (7.4 - 7.8 with it):
                              // the 544 KB of fragments reach a CU at ~80 GB/s (33 B/clk) whatever the ring depth or cache policy
#endif

namespace imdbn {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Term encodings of the chain kernel (template argument NW = number of terms):
//   NW == 2: hi + lo fp16 (11 + 11 significand bits; 4 bytes per weight and 4 MFMAs per product instead of 6 and 9
//            for the three bf16 terms of the propagation kernels -- the chain kernel is bound by the bytes a CU
//            streams; the ~2^-22 relative representation error is far below the 1e-4 parity budget and the
//            fixtures' Bernoulli margins) -- PARITY mode;
//   NW == 1: one bf16 (FAST mode, as everywhere else).
template <int NW>
__device__ __forceinline__ void k4_terms(float x, uint32_t (&t)[2]) {
    if constexpr (NW == 2) {
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        t[0] = (uint32_t)__builtin_bit_cast(unsigned short, hi);
        t[1] = (uint32_t)__builtin_bit_cast(unsigned short, lo);
    } else {
        t[0] = bf16_rne(x); t[1] = 0u;
    }
}
template <int NW>
__device__ __forceinline__ f32x4 k4_mfma(const uint4& a, const uint4& b, const f32x4& c) {
#if K4_NO_MFMA      // timing experiment only (wrong results): what the fragment stream costs without the matrix work
    f32x4 r = c; r[0] += __uint_as_float((a.x ^ b.x ^ b.y ^ b.z ^ b.w) & 0x3f800000u); return r;
#endif
    if constexpr (NW == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else                   return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), as_frag(b), c, 0, 0, 0);
}

struct ChainDraw { const float* tape; uint64_t draw; };
struct ChainRec {                 // one chain step (imdbn_chain_step + the draw cursors the host assigned to it)
    float T, sigma, eta;
    int flags;                    // bit 0 sample_h | bits 1-2 vmode | bit 3 clamp
    ChainDraw noise_h, uni_h, noise_v, uni_v, cat_uni;
    const int32_t* cat_tape;
};
constexpr int CHAIN_REC_BATCH = 32;          // records per writer launch (kernel arguments are limited to 4 KB)
constexpr int CHAIN_MAX_STEPS = 256;
constexpr int K4_ROWS = 16;                  // batch rows per block (M of the MFMA)
constexpr int K4_WAVES = 8;                  // 512 threads: two waves per SIMD, 256 registers each (no spills in the ring)
constexpr int K4_THREADS = 64 * K4_WAVES;
constexpr int K4_LDS_BYTES = 152 * 1024;
struct ChainRecBatch { ChainRec r[CHAIN_REC_BATCH]; };

__global__ void chain_write_recs(const ChainRecBatch b, ChainRec* dst, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = b.r[threadIdx.x];
}

// plane[dir][term]: NT x KB fragments of 64 lanes x 8 bf16.  dir 0 (h|v): n = hidden unit, k = visible unit;
// dir 1 (v|h): n = visible unit, k = hidden unit.  Lane l of fragment (nt, kb) holds k = 32*kb + 8*(l>>4) + j,
// n = 16*nt + (l&15): the B operand of v_mfma_f32_16x16x32_bf16.
template <int NW>
__global__ __launch_bounds__(64) void k4_split_planes(const float* __restrict__ W, int64_t ldw, int V, int H,
                                                      bf16_t* __restrict__ planes, int64_t plane_stride) {
    const int l = threadIdx.x, dir = blockIdx.z;
    const int NT = dir == 0 ? (H + 15) / 16 : (V + 15) / 16, KB = dir == 0 ? (V + 31) / 32 : (H + 31) / 32;
    const int nt = blockIdx.x, kb = blockIdx.y;
    if (nt >= NT || kb >= KB) return;
    const int n = 16 * nt + (l & 15), k0 = 32 * kb + 8 * (l >> 4);
    uint32_t t[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float x = 0.f;
        if (dir == 0) { if (k < V && n < H) x = W[(int64_t)k * ldw + n]; }
        else          { if (n < V && k < H) x = W[(int64_t)n * ldw + k]; }
        uint32_t tt[2];
        k4_terms<NW>(x, tt);
        t[0][j] = tt[0]; t[1][j] = tt[1];
    }
#pragma unroll
    for (int tw = 0; tw < NW; ++tw) {
        bf16_t* q = planes + (dir * 3 + tw) * plane_stride + ((int64_t)(nt * KB + kb) * 64 + l) * 8;
        *reinterpret_cast<uint4*>(q) = make_uint4(t[tw][0] | (t[tw][1] << 16), t[tw][2] | (t[tw][3] << 16),
                                                  t[tw][4] | (t[tw][5] << 16), t[tw][6] | (t[tw][7] << 16));
    }
}

// One chain of a launch: its rows, clamp inputs, mu-pull, schedule and state.  A launch runs one chain or TWO independent ones of the
// same RBM (imdbn_rbm_chain_pair: the IMG->TXT and TXT->IMG chains of iMDBN._cross_reconstruct, imdbn.py:419-449, share nothing but
// the read-only weights): blocks [0, nblk0) belong to chain 0, the rest to chain 1 -- twice the CUs busy for the time of one chain.
struct K4Seg {
    float* state; int64_t lds;                     // fp32 visible state [B][V]: v0 in, final v out
    const ChainRec* recs; int n_steps;
    const float* mu; int64_t ldmu; int Dz;
    const float* vk; const float* mask; int64_t ldk;
    int B;
};
struct K4Args {
    const bf16_t* planes; int64_t plane_stride;     // [2][3] planes
    int V, H, nw, rt;                              // nw weight terms, rt terms of a real-valued activation
    const float* hid_bias; const float* vis_bias;
    int n_groups; int gs[4]; int ge[4];
    int rows;                                      // batch rows per block (<= 16): fewer rows = more CUs share the element-wise work
    uint64_t seed; int64_t row0; const unsigned long long* draw_base;
    K4Seg s0, s1; int nblk0;                       // chain 0 owns blocks [0, nblk0), chain 1 the rest
    int dbg;                                       // 1: per-block stamps of chain step 2 (tools/stamps_probe.py)
};

__device__ __forceinline__ DrawSrc k4_src(const K4Args& a, const ChainDraw& d, int N) {
    DrawSrc s;
    s.tape = d.tape; s.seed = a.seed; s.draw = d.draw; s.row0 = a.row0; s.N = N; s.base = a.draw_base;
    return s;
}

// value -> its NW terms at act[t][row][col] (row pitch AK elements); exact1: the value is 0 or 1 (one term is exact)
template <int NW>
__device__ __forceinline__ void k4_put(bf16_t* act, int AK, int row, int col, float x, bool exact1) {
    uint32_t t[2];
    k4_terms<NW>(x, t);
    act[row * AK + col] = (bf16_t)t[0];
    if constexpr (NW == 2) act[(K4_ROWS + row) * AK + col] = exact1 ? (bf16_t)0 : (bf16_t)t[1];
}

// stage[16][SP] = act[16 x K] * plane[K x N].  Wave w owns the 16-column tiles w, w+16, ...; its (tile, k-block)
// items form ONE flat sequence behind a K4_RING-deep register ring of B fragments (a k-block of one tile is only
// 3-9 MFMAs: one block ahead is not enough to cover the L2 latency, measured 1.3 us per k-block).  na = activation
// terms multiplied (1: the input is exactly bf16).  Accumulators leave in the C layout (lane l: column l&15,
// rows 4*(l>>4)+r) at the end of each tile.
// NW / NA are compile-time: a run-time `if (tw < nw)` puts a branch around every load and MFMA and the compiler then
// waits vmcnt(0) before each MFMA (no overlap at all).
#ifndef K4_NT_LOADS
#define K4_NT_LOADS 0        // 1 (non-temporal fragment loads) measured in round 2: GEMM phases 7.6 / 8.0 us against 7.4 / 7.8
#endif
#ifndef K4_SKIP_LOLO
#define K4_SKIP_LOLO 0      // 1 was measured in round 2: GEMM phases 7.44 -> 7.28 us only (not MFMA bound)
#endif
constexpr int K4_RING = 6;           // round 2: 9 (242 VGPRs, no spill) -> GEMM phases 7.8 / 8.4 us, 12 (spills) -> 8.7 us, against 7.4 / 7.8 with 6: not bound by the bytes in flight
// one B fragment item (tile nt, k-block kb) of the wave's flat item sequence, clamped (surplus slots repeat the last item)
template <int NW>
__device__ __forceinline__ void k4_load_item(const K4Args& a, int dir, int KB, int n_items, int it, uint4 (&b)[NW]) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bf16_t* pl = a.planes + (int64_t)dir * 3 * a.plane_stride;
    const int itc = max(min(it, n_items - 1), 0);
    const int nt = w + K4_WAVES * (itc / KB), kb = itc - (itc / KB) * KB;
#pragma unroll
    for (int tw = 0; tw < NW; ++tw) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4* q = reinterpret_cast<const u32x4*>(pl + tw * a.plane_stride + ((int64_t)(nt * KB + kb) * 64 + l) * 8);
        const u32x4 x = K4_NT_LOADS ? __builtin_nontemporal_load(q) : *q;
        b[tw] = make_uint4(x.x, x.y, x.z, x.w);
    }
}
// The first K4_RING items of a GEMM do not depend on the activations: they are requested BEFORE the epilogue of the previous
// half step (k4_chain), so that the ~1 us of L2 latency at the start of every GEMM hides behind the epilogue's Philox work.
template <int NW>
__device__ __forceinline__ void k4_prefetch(const K4Args& a, int dir, int NT, int KB, uint4 (&ring)[K4_RING][NW]) {
    const int w = threadIdx.x >> 6;
    const int my_tiles = w < NT ? (NT - w + K4_WAVES - 1) / K4_WAVES : 0;
    const int n_items = my_tiles * KB;
    if (n_items == 0) return;
#pragma unroll
    for (int d = 0; d < K4_RING; ++d) k4_load_item<NW>(a, dir, KB, n_items, d, ring[d]);
}

template <int NW, int NA>
__device__ __forceinline__ void k4_gemm(const K4Args& a, int dir, int NT, int KB, const bf16_t* act, int AK,
                                        float* stage, int SP, uint4 (&ring)[K4_RING][NW]) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int my_tiles = w < NT ? (NT - w + K4_WAVES - 1) / K4_WAVES : 0;
    const int n_items = my_tiles * KB;
    if (n_items == 0) return;                        // (`ring` was filled by k4_prefetch)
    // The loads after the prefetched ones walk the item sequence with INCREMENTAL, wave-uniform address arithmetic: consecutive
    // k-blocks of a tile are 1 KB apart, the wave's next tile 8 KB tiles further.  (Recomputing tile and k-block of every item
    // from its index -- a division by the run-time KB per item -- kept the item loop busy with integer work: see DESIGN 9.3.)
    const char* pl = reinterpret_cast<const char*>(a.planes + (int64_t)dir * 3 * a.plane_stride) + l * 16;
    int lit = min(K4_RING, n_items - 1);                                 // item the next load fetches (clamped to the last one)
    int lkb = lit % KB;
    int64_t loff = ((int64_t)(w + K4_WAVES * (lit / KB)) * KB + lkb) * 1024;        // byte offset of that item inside a plane
    auto load = [&](uint4 (&b)[NW]) {
#pragma unroll
        for (int tw = 0; tw < NW; ++tw) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const u32x4 x = *reinterpret_cast<const u32x4*>(pl + tw * a.plane_stride * 2 + loff);
            b[tw] = make_uint4(x.x, x.y, x.z, x.w);
        }
        if (lit + 1 < n_items) {                                          // wave-uniform; past the end the last item is re-read (L2 hits)
            ++lit;
            if (++lkb == KB) { lkb = 0; loff += (int64_t)((K4_WAVES - 1) * KB + 1) * 1024; }
            else loff += 1024;
        }
    };
    f32x4 acc[NW];                   // one accumulator per weight term: NW independent MFMA chains instead of one of 3*NW
#pragma unroll
    for (int tw = 0; tw < NW; ++tw) acc[tw] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kb = 0, nt = w;
    // every ring slot is processed unconditionally (the item sequence is padded with repeats of the last item whose
    // products are never stored): a conditional around the loads would make the compiler wait for vmcnt(0) at every
    // use and serialise one L2 round trip per item (measured: 1.06 us per item)
    for (int it0 = 0; it0 < n_items; it0 += K4_RING) {
#pragma unroll
        for (int d = 0; d < K4_RING; ++d) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < NA; ++ta) {
                const uint4 af = *reinterpret_cast<const uint4*>(act + (ta * K4_ROWS + (l & 15)) * AK + 32 * min(kb, KB - 1) + 8 * (l >> 4));
#pragma unroll
                for (int tw = 0; tw < NW; ++tw) {
                    if (K4_SKIP_LOLO && ta == 1 && tw == 1) continue;     // lo x lo: 2^-22 of the product, below the hi/lo split's own error
                    acc[tw] = k4_mfma<NW>(af, ring[d][tw], acc[tw]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load(ring[d]);
            if (++kb == KB) {                                             // tile finished (padding items never get here with a real tile)
                if (it0 + d < n_items) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        float t = acc[NW - 1][rr];                         // smallest terms first
#pragma unroll
                        for (int tw = NW - 2; tw >= 0; --tw) t += acc[tw][rr];
                        stage[(4 * (l >> 4) + rr) * SP + 16 * nt + (l & 15)] = t;
                    }
                }
#pragma unroll
                for (int tw = 0; tw < NW; ++tw) acc[tw] = f32x4{0.f, 0.f, 0.f, 0.f};
                kb = 0; nt += K4_WAVES;
            }
        }
    }
}

template <int NW>          // terms of the weights and of a real-valued activation: 2 (fp16 hi + lo, PARITY) or 1 (bf16, FAST)
__global__ __launch_bounds__(K4_THREADS, 1) void k4_chain(const K4Args a) {
    __shared__ __attribute__((aligned(16))) char smem[K4_LDS_BYTES];
    const int tid = threadIdx.x;
    const int RB = a.rows;                                                          // valid rows of this block; MFMA rows >= RB are zero
    // this block's chain (block-uniform; selected member by member: an indexed struct would live in scratch)
    const bool second = (int)blockIdx.x >= a.nblk0;
    const int b0 = ((int)blockIdx.x - (second ? a.nblk0 : 0)) * RB;
    float* const c_state = second ? a.s1.state : a.s0.state; const int64_t c_lds = second ? a.s1.lds : a.s0.lds;
    const ChainRec* const c_recs = second ? a.s1.recs : a.s0.recs; const int c_nsteps = second ? a.s1.n_steps : a.s0.n_steps;
    const float* const c_mu = second ? a.s1.mu : a.s0.mu; const int64_t c_ldmu = second ? a.s1.ldmu : a.s0.ldmu; const int c_Dz = second ? a.s1.Dz : a.s0.Dz;
    const float* const c_vk = second ? a.s1.vk : a.s0.vk; const float* const c_mask = second ? a.s1.mask : a.s0.mask;
    const int64_t c_ldk = second ? a.s1.ldk : a.s0.ldk; const int c_B = second ? a.s1.B : a.s0.B;
    const int VK = (a.V + 31) / 32 * 32 + 8, HK = (a.H + 31) / 32 * 32 + 8;      // LDS row pitch (elements): 16-B skew against bank conflicts
    constexpr int at = NW;                                                          // activation terms kept (= a.rt)
    const int gs0 = a.gs[0], gwd = a.n_groups > 0 ? a.ge[0] - a.gs[0] : 0;          // the (single) softmax group
    const int GW = gwd + 1;
    const int SP = max((a.H + 15) / 16, (a.V + 15) / 16) * 16 + 1;                  // stage pitch (floats)
    bf16_t* vact = reinterpret_cast<bf16_t*>(smem);                                 // [at][16][VK]
    bf16_t* hact = vact + at * K4_ROWS * VK;                                        // [at][16][HK]
    float* stage = reinterpret_cast<float*>(hact + at * K4_ROWS * HK);              // [16][SP] raw sums of the current half step
    float* glog = stage + K4_ROWS * SP;                                             // [16][GW] group logits
    float* gaux = glog + K4_ROWS * GW;                                              // [16][4] group max, sum, sampled index
    float* gz = gaux + K4_ROWS * 4;                                                 // [8 row pairs][gwd][2] noise of the group columns, drawn ahead
    const int NTu = (a.H + 15) / 16, KBu = (a.V + 31) / 32, NTd = (a.V + 15) / 16, KBd = (a.H + 31) / 32;

    // v0 -> terms (pad columns and rows >= B are zero)
    for (int i = tid; i < K4_ROWS * (VK - 8); i += K4_THREADS) {
        const int row = i / (VK - 8), col = i - row * (VK - 8);
        const float x = (col < a.V && row < RB && b0 + row < c_B) ? c_state[(int64_t)(b0 + row) * c_lds + col] : 0.f;
        k4_put<NW>(vact, VK, row, col, x, false);
    }
    for (int i = tid; i < at * K4_ROWS * HK; i += K4_THREADS) hact[i] = 0;          // pad columns and rows >= RB stay zero for the whole chain
    __syncthreads();

    const int RP = (RB + 1) / 2;                                  // row pairs of the block
    uint4 ring[K4_RING][NW];                                      // B-fragment ring of the GEMMs, pre-filled one phase ahead
    k4_prefetch<NW>(a, 0, NTu, KBu, ring);
    for (int t = 0; t < c_nsteps; ++t) {
        const ChainRec r = c_recs[t];
        const bool st = a.dbg && t == 2;
        stamp(st, blockIdx.x, 0);
        const bool sample_h = (r.flags & 1) != 0, clamp = (r.flags & 8) != 0, last = t == c_nsteps - 1;
        const int vmode = (r.flags >> 1) & 3;
        const float T = fmaxf(r.T, 1e-6f);                       // max(1e-6, T)  rbm.py:92,96
        const bool pull_on = c_mu && r.eta != 0.f;
        // ---- h | v  (rbm.py:81-92) -----------------------------------------------------------------
        k4_gemm<NW, NW>(a, 0, NTu, KBu, vact, VK, stage, SP, ring);
        k4_prefetch<NW>(a, 1, NTd, KBd, ring);                    // the v|h GEMM's first fragments travel during the h epilogue
        __syncthreads();
        stamp(st, blockIdx.x, 1);
        {
            const DrawSrc nz = k4_src(a, r.noise_h, a.H), un = k4_src(a, r.uni_h, a.H);
            // thread = (row pair, column): the pair's draws come from one Philox block each (common.hpp)
            // The noise of the softmax-group columns of the COMING v half step depends on nothing computed here: the threads this
            // phase leaves idle (RP * H = 256 items on 512 threads with two rows per block) draw it now.  The v epilogue's
            // 532 = 512 + 20 items then no longer pay a second round of Philox + Box-Muller for 20 threads (3.2 -> 1.9 us).
            const int nh_items = RP * a.H, ng_items = (gwd > 0 && r.sigma > 0.f) ? RP * gwd : 0;
            const DrawSrc nzv = k4_src(a, r.noise_v, a.V);
            for (int i = tid; i < nh_items + ng_items; i += K4_THREADS) {
                if (i >= nh_items) {
                    const int j = i - nh_items, gp = j / gwd, gc = j - gp * gwd;
                    float zz[2] = {0.f, 0.f};
                    draw_normal_rows2(nzv, b0 + 2 * gp, c_B - 1, gs0 + gc, zz);
                    gz[(gp * gwd + gc) * 2 + 0] = zz[0]; gz[(gp * gwd + gc) * 2 + 1] = zz[1];
                    continue;
                }
                const int pr = i / a.H, col = i - pr * a.H;
                float z2[2] = {0.f, 0.f}, u2[2] = {0.f, 0.f};
                if (r.sigma > 0.f) draw_normal_rows2(nz, b0 + 2 * pr, c_B - 1, col, z2);
                if (sample_h) draw_uniform_rows<2>(un, b0 + 2 * pr, c_B - 1, col, u2);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int row = 2 * pr + hf;
                    if (row >= RB) break;
                    float x = stage[row * SP + col] + a.hid_bias[col];
                    if (T != 1.0f) x = x / T;
                    if (r.sigma > 0.f) x = x + z2[hf] * r.sigma;
                    float p = sigmoidf_ref(x);
                    if (sample_h) p = (p > u2[hf]) ? 1.f : 0.f;
                    k4_put<NW>(hact, HK, row, col, (b0 + row < c_B) ? p : 0.f, sample_h);
                }
            }
        }
        __syncthreads();
        stamp(st, blockIdx.x, 2);
        // ---- v | h  (rbm.py:94-135) ------------------------------------------------------------------
        if (sample_h) k4_gemm<NW, 1>(a, 1, NTd, KBd, hact, HK, stage, SP, ring);      // sampled states are exactly bf16: one term
        else          k4_gemm<NW, NW>(a, 1, NTd, KBd, hact, HK, stage, SP, ring);
        if (!last) k4_prefetch<NW>(a, 0, NTu, KBu, ring);         // the next step's h|v GEMM: during the v epilogue and the group passes
        __syncthreads();
        stamp(st, blockIdx.x, 3);
        {
            const DrawSrc nz = k4_src(a, r.noise_v, a.V), un = k4_src(a, r.uni_v, a.V);
            for (int i = tid; i < RP * a.V; i += K4_THREADS) {
                const int pr = i / a.V, col = i - pr * a.V;
                const bool in_group = gwd > 0 && col >= gs0 && col < gs0 + gwd;
                float z2[2] = {0.f, 0.f}, u2[2] = {0.f, 0.f};
                if (r.sigma > 0.f) {
                    if (in_group) { z2[0] = gz[(pr * gwd + (col - gs0)) * 2 + 0]; z2[1] = gz[(pr * gwd + (col - gs0)) * 2 + 1]; }   // drawn during the h epilogue
                    else draw_normal_rows2(nz, b0 + 2 * pr, c_B - 1, col, z2);
                }
                if (vmode != 0 && !in_group) draw_uniform_rows<2>(un, b0 + 2 * pr, c_B - 1, col, u2);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int row = 2 * pr + hf;
                    if (row >= RB) break;
                    const int b = b0 + row, bd = min(b, c_B - 1);
                    float x = stage[row * SP + col] + a.vis_bias[col];
                    if (T != 1.0f) x = x / T;
                    if (r.sigma > 0.f) x = x + z2[hf] * r.sigma;
                    if (in_group) {                                          // softmax group: logits now, the rest below
                        glog[row * GW + (col - gs0)] = x;
                        continue;
                    }
                    float p = sigmoidf_ref(x);
                    if (pull_on && col < c_Dz) p = (1.0f - r.eta) * p + r.eta * c_mu[(int64_t)bd * c_ldmu + col];
                    const float m = clamp ? c_mask[(int64_t)bd * c_ldk + col] : 0.f;
                    const float kn = clamp ? c_vk[(int64_t)bd * c_ldk + col] : 0.f;
                    const float mixed = clamp ? (p * (1.0f - m) + kn * m) : p;
                    float v;
                    if (vmode == 0) v = mixed;
                    else {
                        const float u = u2[hf];
                        if (vmode == 1) { const float smp = (p > u) ? 1.f : 0.f; v = clamp ? (smp * (1.0f - m) + kn * m) : smp; }
                        else v = (mixed > u) ? 1.f : 0.f;
                    }
                    if (b >= c_B) v = 0.f;
                    k4_put<NW>(vact, VK, row, col, v, false);
                    if (last && b < c_B) c_state[(int64_t)b * c_lds + col] = v;
                }
            }
        }
        stamp(st, blockIdx.x, 4);
        if (gwd > 0) {
            // softmax + categorical of the group (finish_groups_body arithmetic).  The exponentials and the clipped
            // probabilities are computed by all threads; only the ORDERED sums (softmax denominator, inverse-CDF
            // accumulation -- same order as the per-launch path and the oracle) stay with one thread per row.
            __syncthreads();
            if (tid < RB) {
                float mx = -INFINITY;
                // ordered one-thread loops over LDS: fetch 8 values at a time (independent loads) before the dependent chain --
                // one dependent LDS round trip per element made the three loops of a group 2.4 us of a 24 us chain step
                int j = 0;
                for (; j + 8 <= gwd; j += 8) {
                    float e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = glog[tid * GW + j + k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) mx = fmaxf(mx, e[k]);
                }
                for (; j < gwd; ++j) mx = fmaxf(mx, glog[tid * GW + j]);
                gaux[tid * 4 + 0] = mx;
            }
            __syncthreads();
            for (int i = tid; i < RB * gwd; i += K4_THREADS) {
                const int row = i / gwd, j = i - row * gwd;
                glog[row * GW + j] = expf(glog[row * GW + j] - gaux[row * 4 + 0]);          // e_j
            }
            __syncthreads();
            if (tid < RB) {
                float sum = 0.f;
                int j = 0;
                for (; j + 8 <= gwd; j += 8) {
                    float e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = glog[tid * GW + j + k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum += e[k];                     // same left-to-right order
                }
                for (; j < gwd; ++j) sum += glog[tid * GW + j];
                gaux[tid * 4 + 1] = sum;
            }
            __syncthreads();
            float* gt = stage;                                    // clipped sampling probabilities [RB][GW] (the stage is free now)
            if (vmode != 0 && !r.cat_tape) {
                for (int i = tid; i < RB * gwd; i += K4_THREADS) {
                    const int row = i / gwd, j = i - row * gwd, col = gs0 + j, bd = min(b0 + row, c_B - 1);
                    float tt = glog[row * GW + j] / gaux[row * 4 + 1];
                    if (pull_on && col < c_Dz) tt = (1.0f - r.eta) * tt + r.eta * c_mu[(int64_t)bd * c_ldmu + col];
                    if (vmode == 2 && clamp) {
                        const float m = c_mask[(int64_t)bd * c_ldk + col];
                        tt = tt * (1.0f - m) + c_vk[(int64_t)bd * c_ldk + col] * m;
                    }
                    gt[row * GW + j] = fminf(fmaxf(tt, 1e-8f), 1.0f);
                }
                __syncthreads();
            }
            if (tid < RB) {
                const int row = tid, bd = min(b0 + row, c_B - 1);
                int idx = -1;
                if (vmode != 0) {
                    if (r.cat_tape) idx = r.cat_tape[bd];
                    else {       // PHILOX: inverse CDF over the clipped probabilities (oracle/draws.py)
                        const DrawSrc cs = k4_src(a, r.cat_uni, 1);
                        const float thr = draw_uniform(cs, bd, 0);
                        float tot = 0.f;
                        int j = 0;
                        for (; j + 8 <= gwd; j += 8) {
                            float e[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) e[k] = gt[row * GW + j + k];
#pragma unroll
                            for (int k = 0; k < 8; ++k) tot += e[k];
                        }
                        for (; j < gwd; ++j) tot += gt[row * GW + j];
                        const float target = thr * tot;
                        float ac = 0.f;
                        for (j = 0; j + 8 <= gwd && idx < 0; j += 8) {
                            float e[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) e[k] = gt[row * GW + j + k];
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                ac += e[k];
                                if (idx < 0 && ac > target) idx = j + k;         // first crossing, as the sequential scan
                            }
                        }
                        for (; j < gwd && idx < 0; ++j) {
                            ac += gt[row * GW + j];
                            if (ac > target) idx = j;
                        }
                        if (idx < 0) idx = gwd - 1;
                    }
                }
                gaux[row * 4 + 2] = __int_as_float(idx);
            }
            __syncthreads();
            for (int i = tid; i < RB * gwd; i += K4_THREADS) {
                const int row = i / gwd, j = i - row * gwd, col = gs0 + j, b = b0 + row, bd = min(b, c_B - 1);
                float p = glog[row * GW + j] / gaux[row * 4 + 1];
                if (pull_on && col < c_Dz) p = (1.0f - r.eta) * p + r.eta * c_mu[(int64_t)bd * c_ldmu + col];
                const float m = clamp ? c_mask[(int64_t)bd * c_ldk + col] : 0.f;
                const float kn = clamp ? c_vk[(int64_t)bd * c_ldk + col] : 0.f;
                const int idx = __float_as_int(gaux[row * 4 + 2]);
                float v;
                if (vmode == 0) v = clamp ? (p * (1.0f - m) + kn * m) : p;
                else if (vmode == 1) { const float o = (j == idx) ? 1.f : 0.f; v = clamp ? (o * (1.0f - m) + kn * m) : o; }
                else v = (j == idx) ? 1.f : 0.f;
                if (b >= c_B) v = 0.f;
                k4_put<NW>(vact, VK, row, col, v, false);
                if (last && b < c_B) c_state[(int64_t)b * c_lds + col] = v;
            }
        }
        __syncthreads();
        stamp(st, blockIdx.x, 5);
    }
}

}  // namespace imdbn
