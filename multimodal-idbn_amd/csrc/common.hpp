// common.hpp -- device helpers shared by every kernel of the CD engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace imdbn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;     // raw bf16 bits in memory

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------
// Philox-4x32-10.  counter = (column, global row >> 1 [normal] or global row >> 2 [uniform], draw_lo, draw_hi), key = seed.
// numpy twin: oracle/draws.py:PhiloxStream.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// A source of one logical draw tensor [B][N]: either a replay tape or a Philox draw number.
struct DrawSrc {
    const float* tape;   // non-null: replay, element (b,n) at tape[b*N + n]
    uint64_t seed;
    uint64_t draw;       // Philox draw number
    int64_t row0;
    int N;               // row length of the logical tensor
    const unsigned long long* base;   // nullable: a device-resident counter added to `draw` (imdbn_rng.dev_offset: a captured graph
                                      // replays with fresh draws because a node of the graph advances the counter)
};

__device__ __forceinline__ uint64_t draw_no(const DrawSrc& s) { return s.draw + (s.base ? (uint64_t)*s.base : 0ull); }

__device__ __forceinline__ uint4 draw_block(const DrawSrc& s, int b, int n) {
    const uint64_t row = (uint64_t)(s.row0 + b);
    const uint64_t dn = draw_no(s);
    return philox4x32_10(make_uint4((uint32_t)n, (uint32_t)row, (uint32_t)dn, (uint32_t)(dn >> 32)),
                         make_uint2((uint32_t)s.seed, (uint32_t)(s.seed >> 32)));
}

// Uniform draws: the four rows of a global 4-row group share ONE Philox block (counter row = global row >> 2,
// component = global row & 3).  A Philox-4x32-10 call is ~40 quarter-rate integer multiplies (~900 cycles per wave):
// with one call per element it was 3.4 of the 3.9 us of the fused K2 epilogue (8 rows per thread -> 8 calls, now 2).
__device__ __forceinline__ uint4 draw_block4(const DrawSrc& s, uint64_t grow, int n) {
    const uint64_t dn = draw_no(s);
    return philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(grow >> 2), (uint32_t)dn, (uint32_t)(dn >> 32)),
                         make_uint2((uint32_t)s.seed, (uint32_t)(s.seed >> 32)));
}
__device__ __forceinline__ float u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }   // 2^-24
__device__ __forceinline__ uint32_t pick4(const uint4& x, int c) { return c == 0 ? x.x : (c == 1 ? x.y : (c == 2 ? x.z : x.w)); }

__device__ __forceinline__ float draw_uniform(const DrawSrc& s, int b, int n) {
    if (s.tape) return s.tape[(int64_t)b * s.N + n];
    const uint64_t g = (uint64_t)(s.row0 + b);
    return u24(pick4(draw_block4(s, g, n), (int)(g & 3)));
}

// R consecutive rows b0 .. b0+R-1 (clamped to bmax) of column n; R = 8 -> two Philox blocks, R = 2 -> one, when the
// rows line up with the 4-row groups (the usual case: b0 and row0 multiples of R), else element by element.
template <int R>
__device__ __forceinline__ void draw_uniform_rows(const DrawSrc& s, int b0, int bmax, int n, float (&u)[R]) {
    static_assert(R == 8 || R == 2, "row tiles of the epilogues");
    if (s.tape) {
#pragma unroll
        for (int i = 0; i < R; ++i) u[i] = s.tape[(int64_t)min(b0 + i, bmax) * s.N + n];
        return;
    }
    const uint64_t g0 = (uint64_t)(s.row0 + b0);
    if ((g0 & (R == 8 ? 3 : 1)) == 0 && b0 + R - 1 <= bmax) {
        if constexpr (R == 8) {
            const uint4 x = draw_block4(s, g0, n), y = draw_block4(s, g0 + 4, n);
            u[0] = u24(x.x); u[1] = u24(x.y); u[2] = u24(x.z); u[3] = u24(x.w);
            u[4] = u24(y.x); u[5] = u24(y.y); u[6] = u24(y.z); u[7] = u24(y.w);
        } else {
            const uint4 x = draw_block4(s, g0, n);
            const bool up = (g0 & 2) != 0;
            u[0] = u24(up ? x.z : x.x); u[1] = u24(up ? x.w : x.y);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) u[i] = draw_uniform(s, min(b0 + i, bmax), n);
}

// Normal draws (Box-Muller): the two rows of a global row PAIR share one Philox block (counter row = global row >> 1;
// the even row takes outputs x,y, the odd row z,w).  In the chain kernels Philox was ~80 % of the element-wise work.
__device__ __forceinline__ float normal_from(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-08f;
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
__device__ __forceinline__ uint4 draw_block2(const DrawSrc& s, uint64_t grow, int n) {
    const uint64_t dn = draw_no(s);
    return philox4x32_10(make_uint4((uint32_t)n, (uint32_t)(grow >> 1), (uint32_t)dn, (uint32_t)(dn >> 32)),
                         make_uint2((uint32_t)s.seed, (uint32_t)(s.seed >> 32)));
}
__device__ __forceinline__ float draw_normal(const DrawSrc& s, int b, int n) {
    if (s.tape) return s.tape[(int64_t)b * s.N + n];
    const uint64_t g = (uint64_t)(s.row0 + b);
    const uint4 x = draw_block2(s, g, n);
    return (g & 1) ? normal_from(x.z, x.w) : normal_from(x.x, x.y);
}
// rows b0, b0+1 (clamped to bmax) of column n: one Philox block when they are a global pair
__device__ __forceinline__ void draw_normal_rows2(const DrawSrc& s, int b0, int bmax, int n, float (&z)[2]) {
    if (s.tape) {
        z[0] = s.tape[(int64_t)min(b0, bmax) * s.N + n];
        z[1] = s.tape[(int64_t)min(b0 + 1, bmax) * s.N + n];
        return;
    }
    const uint64_t g0 = (uint64_t)(s.row0 + b0);
    if ((g0 & 1) == 0 && b0 + 1 <= bmax) {
        const uint4 x = draw_block2(s, g0, n);
        z[0] = normal_from(x.x, x.y);
        z[1] = normal_from(x.z, x.w);
        return;
    }
    z[0] = draw_normal(s, min(b0, bmax), n);
    z[1] = draw_normal(s, min(b0 + 1, bmax), n);
}

// ------------------------------------------------------------------------------------------
// fp32 -> bf16 pieces.  split3: x == hi + mid + lo exactly (truncation splits a 24-bit
// significand into 8+8+8 bits), so three bf16 MFMAs against an exactly-representable operand
// reproduce the fp32 products.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t h = __float_as_uint(x) & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    const uint32_t m = __float_as_uint(r) & 0xFFFF0000u;
    const float r2 = r - __uint_as_float(m);
    hi = h >> 16;
    mid = m >> 16;
    lo = __float_as_uint(r2) >> 16;
}

__device__ __forceinline__ uint32_t bf16_rne(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// C/D layout of v_mfma_f32_32x32x16_bf16: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Term count of a caller-supplied activation operand from prep's exactness map [P][ncb] (one entry per
// 8 batch rows x 64 features): OR of the entries of feature blocks [cb0, cb1) over all P row groups.
// Wave-uniform result; map == nullptr means the count is static.
__device__ __forceinline__ int operand_terms(const int* map, int ncb, int P, int cb0, int cb1, int static_terms) {
    if (static_terms) return static_terms;
    cb1 = min(cb1, ncb);
    const int w = cb1 - cb0, n = w * P;
    int any = 0;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) any |= map[(i / w) * ncb + cb0 + (i % w)] & 1;      // bit 0: inexact in bf16
    return __any(any) ? 3 : 1;
}

// Debug timeline (tools/stamps_probe.py): per-block wall-clock stamps (100 MHz), enabled by the `dbg` option bits.
__device__ long long g_stamps[4096 * 8];
__device__ __forceinline__ void stamp(bool on, int blk, int slot) {
    if (on && threadIdx.x == 0 && blk < 4096) g_stamps[blk * 8 + slot] = wall_clock64();
}

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace imdbn
