// kernels_ew.hpp -- element-wise kernels around the MFMA kernels.
//
//   finish          split-K slab sum + bias + /T + noise + sigmoid + mu-pull + clamp-mix + Bernoulli
//                   sampling; emits fp32 outputs, bf16 operand forms, column sums, squared-error partials
//   finish_groups   softmax groups of the visible layer (rbm.py:113-114,129-133): softmax, categorical
//   prep_operand    caller fp32 tensor -> bf16 operand forms (+ exactness flag); chain initial state
//   bias_update     rbm.py:216-226 from the column-sum partials
//   pack_stats / apply_delta / bias_from_packed   data-parallel split
//   bernoulli / categorical                       stand-alone sample_visible (rbm.py:118-135)
//
// Thread mapping of finish/prep: block = ONE wave = 64 columns; a thread owns one column and 8
// consecutive batch rows, so fp32 accesses are 256-B row segments per wave, the transposed operand
// form is one 16-B store per thread, column sums need no cross-thread step, and the grid
// (ceil(N/64), Bp/8) has enough blocks to cover the chip even for the 64x1500 hidden side.
#pragma once
#include "common.hpp"

namespace imdbn {

struct OperandOut {
    // "row-major" operand (A of K1/K2), stored K16-BLOCKED: element (b, k) of term t at
    //   rm[t*rm_ts + ((k>>4)*Bp + b)*16 + (k&15)],   k < ldrm (= feature count padded to 16)
    // so the MFMA A-fragment load of a wave (32 rows x 2 halves x 16 B) is ONE contiguous KB.
    bf16_t* rm; int64_t rm_ts; int ldrm; int rm_terms; int Bp;
    bf16_t* tr; int64_t tr_ts; int tr_terms;             // transposed  [t][N][Bp]
    int tr_negate;                                       // store the transposed planes with the sign flipped
    // bit-packed form of an exactly-{0,1} operand (sampled states, binary data), BYTE-major: bits[(k >> 3)*Bp + b] bit (k & 7):
    // the 8 k-values of one MFMA fragment lane are one byte, any 8-column-aligned tile can write its part, and a wave's loads
    // of one K block are contiguous.  16x smaller than the bf16 form.  bits_shape: how the lanes of a wave map to columns in
    // the writing kernel -- 0: 64 consecutive columns (finish, prep), 1: 32 consecutive columns x 2 row groups (lanes >= 32),
    // 2: 16 consecutive columns x 4 row groups
    uint8_t* bits; int bits_shape; int bits_cols;      // bits_cols (shape 1): valid columns of the 32-column run (multiple of 8)
};

// row `b` of a 0/1 operand for the columns the wave covers -> bit plane (one byte per 8 columns).  nz = this lane's value != 0.
__device__ __forceinline__ void store_bits_row(const OperandOut& o, bool nz, int col, int b, bool row_ok, int shape, int cols) {
    const unsigned long long m = __ballot(nz ? 1 : 0);
    const int lane = threadIdx.x & 63;
    int j, shift, colbase;
    bool w;
    if (shape == 0) { j = lane; w = lane < 8; shift = 8 * lane; colbase = col - lane; }
    else if (shape == 1) { j = lane & 31; w = j < (cols >> 3); shift = 32 * (lane >> 5) + 8 * j; colbase = col - j; }
    else { j = lane & 15; w = j < (cols >> 3); shift = 16 * (lane >> 4) + 8 * j; colbase = col - j; }
    if (w && row_ok) o.bits[(int64_t)((colbase >> 3) + j) * o.Bp + b] = (uint8_t)((m >> (shift & 63)) & 0xFFull);
}

// bf16 terms of R values: terms == 1 -> one round-to-nearest bf16 (exact for samples / exactly-bf16 data),
// terms == 3 -> hi/mid/lo truncation split (fp32-exact products).  Computed ONCE per element and shared by
// the two operand forms.
template <int R>
__device__ __forceinline__ void pieces(const float (&x)[R], int terms, uint32_t (&pc)[3][R]) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (terms == 1) { pc[0][i] = bf16_rne(x[i]); pc[1][i] = 0u; pc[2][i] = 0u; }
        else split3(x[i], pc[0][i], pc[1][i], pc[2][i]);
    }
}

// pc[t][i] = term t of the value at (row b0+i, col); rows >= B and cols >= N must already be 0.
template <int R>
__device__ __forceinline__ void store_rm_pc(const OperandOut& o, const uint32_t (&pc)[3][R], int b0, int col) {
    if (o.rm && col < o.ldrm) {
        bf16_t* q = o.rm + ((int64_t)(col >> 4) * o.Bp + b0) * 16 + (col & 15);
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < o.rm_terms) {
#pragma unroll
                for (int i = 0; i < R; ++i) q[t * o.rm_ts + i * 16] = (bf16_t)pc[t][i];
            }
    }
}
template <int R>
__device__ __forceinline__ void store_tr_pc(const OperandOut& o, const uint32_t (&pc)[3][R], int b0, int col, int N, int Bp) {
    static_assert(R == 8 || R == 4 || R == 2, "rows per thread");
    if (o.tr && col < N) {
        bf16_t* q = o.tr + (int64_t)col * Bp + b0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < o.tr_terms) {
                const uint32_t sg = o.tr_negate ? 0x80008000u : 0u;      // negative-phase planes are stored negated for K3
                if constexpr (R == 8)
                    *reinterpret_cast<uint4*>(q + t * o.tr_ts) =
                        make_uint4((pc[t][0] | (pc[t][1] << 16)) ^ sg, (pc[t][2] | (pc[t][3] << 16)) ^ sg,
                                   (pc[t][4] | (pc[t][5] << 16)) ^ sg, (pc[t][6] | (pc[t][7] << 16)) ^ sg);
                else if constexpr (R == 4)
                    *reinterpret_cast<uint2*>(q + t * o.tr_ts) = make_uint2((pc[t][0] | (pc[t][1] << 16)) ^ sg, (pc[t][2] | (pc[t][3] << 16)) ^ sg);
                else
                    *reinterpret_cast<uint32_t*>(q + t * o.tr_ts) = (pc[t][0] | (pc[t][1] << 16)) ^ sg;
            }
    }
}
// LDS staging of a block's K16-blocked operand tile: buf[t][kb][row][16] for the block's `nkb` 16-column groups
// x `rows` batch rows.  2-byte scattered global stores are processed about a lane at a time (24 of them per
// thread were ~5 us of the fused K2 epilogue); through LDS the tile leaves as coalesced 16-B stores.
// Passed BY VALUE (buf == nullptr: no stage): as `const RmStage*` selected with `staged ? &stg : nullptr` the struct lived in
// scratch memory (32 B per lane on every kernel with the general epilogue, reloaded inside the per-element code).
struct RmStage { bf16_t* buf = nullptr; int nkb = 0, rows = 0; int col0 = 0, row0 = 0; };

template <int R>
__device__ __forceinline__ void stage_rm_pc(const OperandOut& o, const RmStage& g, const uint32_t (&pc)[3][R], int b0, int col) {
    const int lc = col - g.col0, lr = b0 - g.row0;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < o.rm_terms) {
#pragma unroll
            for (int i = 0; i < R; ++i) g.buf[(((t * g.nkb) + (lc >> 4)) * g.rows + lr + i) * 16 + (lc & 15)] = (bf16_t)pc[t][i];
        }
}
// after a __syncthreads(): the whole block writes the staged tile (16 B per thread and step)
__device__ __forceinline__ void flush_rm_stage(const OperandOut& o, const RmStage& g) {
    if (!o.rm) return;
    const int per_t = g.nkb * g.rows * 2, n = o.rm_terms * per_t;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int t = i / per_t, j = i - t * per_t;
        const int half = j & 1, lr = (j >> 1) % g.rows, kb = (j >> 1) / g.rows;
        const int gkb = (g.col0 >> 4) + kb;
        if (gkb * 16 < o.ldrm)
            *reinterpret_cast<uint4*>(o.rm + t * o.rm_ts + ((int64_t)gkb * o.Bp + g.row0 + lr) * 16 + 8 * half) =
                *reinterpret_cast<const uint4*>(g.buf + ((t * g.nkb + kb) * g.rows + lr) * 16 + 8 * half);
    }
}

// both operand forms of one source vector (the common case shares the split); stg != nullptr: the K16-blocked
// form goes to the block's LDS stage (the caller flushes it), otherwise straight to memory
template <int R>
__device__ __forceinline__ void store_forms(const OperandOut& o, const float (&x)[R], bool rm, bool tr, int b0, int col, int N, int Bp,
                                            const RmStage stg = RmStage{}) {
    uint32_t pc[3][R];
    rm = rm && o.rm && col < o.ldrm;
    if (rm && tr && o.rm_terms == o.tr_terms) {
        pieces<R>(x, o.rm_terms, pc);
        if (stg.buf) stage_rm_pc<R>(o, stg, pc, b0, col); else store_rm_pc<R>(o, pc, b0, col);
        store_tr_pc<R>(o, pc, b0, col, N, Bp);
        return;
    }
    if (rm) { pieces<R>(x, o.rm_terms, pc); if (stg.buf) stage_rm_pc<R>(o, stg, pc, b0, col); else store_rm_pc<R>(o, pc, b0, col); }
    if (tr) { pieces<R>(x, o.tr_terms, pc); store_tr_pc<R>(o, pc, b0, col, N, Bp); }
}

struct FinishArgs {
    const float* partial; int ks; int64_t slab;    // partial[ks][Bp][N], slab = Bp*N
    int B, Bp, N;
    const float* bias; float T;
    float sigma; DrawSrc noise;
    int n_groups; int gs[4]; int ge[4];
    const float* mu; int64_t ldmu; int Dz; float eta;
    const float* vk; const float* mask; int64_t ldk; int clamp;
    int vmode; DrawSrc uni;                        // 0 mean-field, 1 sample(p), 2 sample(mix(p)) no re-mix
    const int32_t* cat_tape; DrawSrc cat_uni;      // categorical source for group g: cat_tape + g*B / draw+g
    int logits_only;
    float* out_prob; int64_t ld_prob;
    float* out_final; int64_t ld_final;
    OperandOut op; int rm_src, tr_src;             // 0 none, 1 prob, 2 final
    float* colsum_part; int colsum_src;            // [Bp/32][N]
    const float* loss_ref; int64_t ld_ref; int loss_src; float* loss_part;   // one per block (+ one per group)
    int simple;                                    // set by the host: none of T / noise / mu / clamp / groups / logits_only in use
    int lean;                                      // set by the host: `simple`, no fp32 outputs, no K16-blocked form, draws from a tape or from Philox with
                                                   // row0 % 4 == 0: the streaming kernels then run finish_lean8 (a few hundred instructions instead of
                                                   // the general epilogue's many inlined variants, whose code alone overflowed the instruction cache)
    int dbg;                                       // tuning aid: which kernels record per-block timeline stamps (common.hpp stamp)
};

// gs / ge are only ever indexed with compile-time constants: a dynamically indexed member pins the whole struct in
// memory (scratch, when it is a patched local copy as in the persistent chain kernel) instead of registers
__device__ __forceinline__ bool in_group(const FinishArgs& a, int col) {
    bool g = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) g |= (i < a.n_groups) && (col >= a.gs[i] && col < a.ge[i]);
    return g;
}
__device__ __forceinline__ void group_bounds(const FinishArgs& a, int g, int& s, int& e) {
    s = g == 0 ? a.gs[0] : (g == 1 ? a.gs[1] : (g == 2 ? a.gs[2] : a.gs[3]));
    e = g == 0 ? a.ge[0] : (g == 1 ? a.ge[1] : (g == 2 ? a.ge[2] : a.ge[3]));
}

__device__ __forceinline__ float wave_sum(float v) {
    // fixed butterfly order: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Epilogue of one column x R consecutive batch rows (R = 8, or 2 when a block spreads an 8-row group over its
// four waves), given the pre-bias sums xs[R].  Returns the squared-error partial; csum = this thread's share of
// the column sum (the caller stores / combines it).
// Side inputs of the epilogue of one column x R rows.  All are loaded up front, UNCONDITIONALLY, from clamped
// addresses (rows >= B and columns >= N are discarded later): a load inside the per-row branches would cost one
// dependent memory round trip per row.  A fused GEMM kernel issues them BEFORE its main loop.
template <int R>
struct SideIn { float bias; float ref[R], mk[R], kn[R], mu[R]; };

template <int R>
__device__ __forceinline__ void load_side(const FinishArgs& a, int col, int b0, SideIn<R>& s) {
    const int cc = min(col, a.N - 1);
    s.bias = a.bias[cc];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int bc = min(b0 + i, a.B - 1);
        s.ref[i] = a.loss_ref ? a.loss_ref[(int64_t)bc * a.ld_ref + cc] : 0.f;
        s.mk[i] = a.clamp ? a.mask[(int64_t)bc * a.ldk + cc] : 0.f;
        s.kn[i] = a.clamp ? a.vk[(int64_t)bc * a.ldk + cc] : 0.f;
        s.mu[i] = (a.mu && cc < a.Dz) ? a.mu[(int64_t)bc * a.ldmu + cc] : 0.f;
    }
}

// EX = false is the lean specialisation for the plain case (a.simple: T == 1, no noise, no mu-pull, no clamp, no
// softmax groups, not logits-only): the full version keeps so many FinishArgs fields live that the compiler
// re-fetches kernel arguments and walks a chain of scalar branches for every row (5 of the 7 us of the fused K2
// epilogue).  Sampling (vmode) stays a run-time switch in both.
template <int R, bool EX>
__device__ __forceinline__ float finish_rows_impl(const FinishArgs& a, int col, int b0, const float (&xs)[R], const SideIn<R>& sd, float& csum,
                                                  const RmStage stg, int bshape, int bcols) {
    const bool cok = col < a.N;
    const bool grp = EX && cok && in_group(a, col);
    const int cc = min(col, a.N - 1);
    const float bias = sd.bias;
    const bool pull = EX && a.mu && col < a.Dz;
    const bool clamp = EX && a.clamp;
    const bool raw = EX && (a.logits_only || grp);      // group columns: logits now, softmax etc. in finish_groups
    const int vmode = a.vmode, colsum_src = a.colsum_src, loss_src = a.loss_src;
    const bool has_ref = a.loss_ref != nullptr;
    float* const out_prob = a.out_prob; float* const out_final = a.out_final;
    float xp[R], xf[R], us[R];
    float lsum = 0.f;
    csum = 0.f;
    if (vmode != 0) draw_uniform_rows<R>(a.uni, b0, a.B - 1, cc, us);      // draws for padded rows / columns: clamped, discarded
    float nzs[R];
    if constexpr (EX) {
        if (a.sigma > 0.f) {
#pragma unroll
            for (int j = 0; j < R / 2; ++j) {
                float z[2];
                draw_normal_rows2(a.noise, b0 + 2 * j, a.B - 1, cc, z);
                nzs[2 * j] = z[0]; nzs[2 * j + 1] = z[1];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int b = b0 + i;
        const bool live = cok && b < a.B;
        const int bd = min(b, a.B - 1);             // draws for padded rows / columns: clamped, result discarded
        float x = xs[i] + bias;
        if constexpr (EX) {
            if (a.T != 1.0f) x = x / a.T;
            if (a.sigma > 0.f) x = x + nzs[i] * a.sigma;
        }
        float p = sigmoidf_ref(x);
        if (pull) p = (1.0f - a.eta) * p + a.eta * sd.mu[i];
        const float mixed = clamp ? (p * (1.0f - sd.mk[i]) + sd.kn[i] * sd.mk[i]) : p;
        float v;
        if (vmode == 0) {
            v = mixed;
        } else {
            const float u = us[i];
            if (vmode == 1) {
                const float smp = (p > u) ? 1.f : 0.f;
                v = clamp ? (smp * (1.0f - sd.mk[i]) + sd.kn[i] * sd.mk[i]) : smp;
            } else {
                v = (mixed > u) ? 1.f : 0.f;
            }
        }
        if (live) {
            if (out_prob) out_prob[(int64_t)b * a.ld_prob + col] = raw ? x : p;
            if (out_final && !raw) out_final[(int64_t)b * a.ld_final + col] = v;
        }
        const bool use = live && !raw;
        xp[i] = use ? p : 0.f;
        xf[i] = use ? v : 0.f;
        if (use) {
            csum += (colsum_src == 2 ? v : p);
            const float dlt = sd.ref[i] - (loss_src == 2 ? v : p);
            lsum += has_ref ? dlt * dlt : 0.f;
        }
    }
    if (!raw) {   // group columns get their operand forms from finish_groups
        if (a.rm_src == a.tr_src) {
            if (a.rm_src) store_forms<R>(a.op, a.rm_src == 2 ? xf : xp, true, true, b0, col, a.N, a.Bp, stg);
        } else {
            if (a.rm_src) store_forms<R>(a.op, a.rm_src == 2 ? xf : xp, true, false, b0, col, a.N, a.Bp, stg);
            if (a.tr_src) store_forms<R>(a.op, a.tr_src == 2 ? xf : xp, false, true, b0, col, a.N, a.Bp);
        }
    }
    // (softmax-group columns of a staged tile are flushed with whatever the stage holds and then rewritten by
    //  finish_groups, which runs after this kernel)
    if (a.op.bits) {                  // block-uniform; the wave's lanes cover aligned runs of 64 (or 2 x 32) consecutive columns
#pragma unroll
        for (int i = 0; i < R; ++i) store_bits_row(a.op, xf[i] != 0.f, col, b0 + i, b0 + i < a.Bp, bshape < 0 ? a.op.bits_shape : bshape, bshape < 0 ? a.op.bits_cols : bcols);
    }
    return lsum;
}

template <int R>
__device__ __forceinline__ float finish_rows(const FinishArgs& a, int col, int b0, const float (&xs)[R], const SideIn<R>& sd, float& csum,
                                             const RmStage stg = RmStage{}, int bshape = -1, int bcols = 0) {
    if (a.simple) return finish_rows_impl<R, false>(a, col, b0, xs, sd, csum, stg, bshape, bcols);
    return finish_rows_impl<R, true>(a, col, b0, xs, sd, csum, stg, bshape, bcols);
}

template <int R>
__device__ __forceinline__ float finish_rows(const FinishArgs& a, int col, int b0, const float (&xs)[R], float& csum) {
    SideIn<R> sd;
    load_side<R>(a, col, b0, sd);
    return finish_rows<R>(a, col, b0, xs, sd, csum);
}

// one column x 8 rows per thread: part_row = index of this 8-row group in the column-sum partials
__device__ __forceinline__ float finish_rows8(const FinishArgs& a, int col, int b0, const float (&xs)[8], int part_row, const SideIn<8>& sd,
                                              const RmStage stg = RmStage{}, int bshape = -1, int bcols = 0) {
    float csum;
    const float lsum = finish_rows<8>(a, col, b0, xs, sd, csum, stg, bshape, bcols);
    if (a.colsum_part && col < a.N) a.colsum_part[(int64_t)part_row * a.N + col] = csum;
    return lsum;
}
__device__ __forceinline__ float finish_rows8(const FinishArgs& a, int col, int b0, const float (&xs)[8], int part_row) {
    float csum;
    const float lsum = finish_rows<8>(a, col, b0, xs, csum);
    if (a.colsum_part && col < a.N) a.colsum_part[(int64_t)part_row * a.N + col] = csum;
    return lsum;
}

// ---- lean epilogue of the CD propagations: one column x 8 consecutive batch rows (b0 % 8 == 0) --------------------------
// Exactly the arithmetic (and summation order) of finish_rows_impl<8, false> for the configurations the CD step uses:
// T = 1, no noise / mu-pull / clamp / softmax groups, outputs = transposed operand plane(s), bit plane of the sample,
// column-sum partial, squared-error partial, fp32 probabilities (the fused forward of cd_step).  Side inputs: the bias and (with a loss reference) 8 reference values.
struct SideLean { float bias; float ref[8]; };

__device__ __forceinline__ void load_side_lean(const FinishArgs& a, int col, int b0, SideLean& s) {
    const int cc = min(col, a.N - 1);
    s.bias = a.bias[cc];
#pragma unroll
    for (int i = 0; i < 8; ++i) s.ref[i] = a.loss_ref ? a.loss_ref[(int64_t)min(b0 + i, a.B - 1) * a.ld_ref + cc] : 0.f;
}

__device__ __forceinline__ float finish_lean8(const FinishArgs& a, int col, int b0, const float (&xs)[8], int part_row, const SideLean& sd,
                                              int bshape, int bcols) {
    const bool cok = col < a.N;
    const int cc = min(col, a.N - 1);
    const int vmode = a.vmode, colsum_src = a.colsum_src, loss_src = a.loss_src;
    const bool has_ref = a.loss_ref != nullptr;
    float us[8];
    if (vmode != 0) {
        if (a.uni.tape) {
#pragma unroll
            for (int i = 0; i < 8; ++i) us[i] = a.uni.tape[(int64_t)min(b0 + i, a.B - 1) * a.uni.N + cc];
        } else {                                   // row0 % 4 == 0 (host): the 8 rows are two whole 4-row Philox groups
            const uint64_t g0 = (uint64_t)(a.uni.row0 + b0);
            const uint4 x = draw_block4(a.uni, g0, cc), y = draw_block4(a.uni, g0 + 4, cc);
            us[0] = u24(x.x); us[1] = u24(x.y); us[2] = u24(x.z); us[3] = u24(x.w);
            us[4] = u24(y.x); us[5] = u24(y.y); us[6] = u24(y.z); us[7] = u24(y.w);
        }
    }
    float xp[8], xf[8];
    float csum = 0.f, lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool live = cok && (b0 + i) < a.B;
        const float p = sigmoidf_ref(xs[i] + sd.bias);
        const float v = vmode == 0 ? p : ((p > us[i]) ? 1.f : 0.f);
        xp[i] = live ? p : 0.f;
        xf[i] = live ? v : 0.f;
        if (live) {
            csum += (colsum_src == 2 ? v : p);
            const float dlt = sd.ref[i] - (loss_src == 2 ? v : p);
            lsum += has_ref ? dlt * dlt : 0.f;
        }
    }
    if (a.tr_src) {
        uint32_t pc[3][8];
        pieces<8>(a.tr_src == 2 ? xf : xp, a.op.tr_terms, pc);
        store_tr_pc<8>(a.op, pc, b0, col, a.N, a.Bp);
    }
    if (a.op.bits) {
#pragma unroll
        for (int i = 0; i < 8; ++i) store_bits_row(a.op, xf[i] != 0.f, col, b0 + i, b0 + i < a.Bp, bshape, bcols);
    }
    if (a.out_prob && cok) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (b0 + i < a.B) a.out_prob[(int64_t)(b0 + i) * a.ld_prob + col] = xp[i];
    }
    if (a.colsum_part && cok) a.colsum_part[(int64_t)part_row * a.N + col] = csum;
    return lsum;
}

// The same for one column x FOUR rows per thread (k1_stream's last arriver runs its epilogue on all eight waves): the two threads of
// an 8-row group are lanes l and l + 32 of one wave (rows b0 .. b0 + 3 in the lower half-wave); the four rows are one whole Philox
// group (row0 % 4 == 0), and the column sum of the 8-row group is lower half + upper half.  No squared error here (K1 has none).
__device__ __forceinline__ void finish_lean4(const FinishArgs& a, int col, int b0, const float (&xs)[4], int part_row, float bias, int bshape, int bcols) {
    const bool cok = col < a.N;
    const int cc = min(col, a.N - 1);
    const int vmode = a.vmode, colsum_src = a.colsum_src;
    float us[4];
    if (vmode != 0) {
        if (a.uni.tape) {
#pragma unroll
            for (int i = 0; i < 4; ++i) us[i] = a.uni.tape[(int64_t)min(b0 + i, a.B - 1) * a.uni.N + cc];
        } else {
            const uint4 x = draw_block4(a.uni, (uint64_t)(a.uni.row0 + b0), cc);
            us[0] = u24(x.x); us[1] = u24(x.y); us[2] = u24(x.z); us[3] = u24(x.w);
        }
    }
    float xp[4], xf[4];
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool live = cok && (b0 + i) < a.B;
        const float p = sigmoidf_ref(xs[i] + bias);
        const float v = vmode == 0 ? p : ((p > us[i]) ? 1.f : 0.f);
        xp[i] = live ? p : 0.f;
        xf[i] = live ? v : 0.f;
        if (live) csum += (colsum_src == 2 ? v : p);
    }
    if (a.tr_src) {
        uint32_t pc[3][4];
        pieces<4>(a.tr_src == 2 ? xf : xp, a.op.tr_terms, pc);
        store_tr_pc<4>(a.op, pc, b0, col, a.N, a.Bp);
    }
    if (a.op.bits) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_bits_row(a.op, xf[i] != 0.f, col, b0 + i, b0 + i < a.Bp, bshape, bcols);
    }
    if (a.out_prob && cok) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (b0 + i < a.B) a.out_prob[(int64_t)(b0 + i) * a.ld_prob + col] = xp[i];
    }
    const float hi = __shfl_xor(csum, 32, 64);                 // (every lane takes part)
    if (a.colsum_part && cok && (threadIdx.x & 32) == 0) a.colsum_part[(int64_t)part_row * a.N + col] = csum + hi;
}

// block = 256 threads = 64 columns x 4 slab-quarters: thread (c, kq) sums slabs k = kq, kq+4, ... of its
// column for 8 rows (all loads of a quarter in flight together), the quarters are combined through LDS in
// a fixed order, and wave 0 runs the per-element epilogue.  grid = (ceil(N/64), Bp/8).
__global__ __launch_bounds__(256) void finish(const FinishArgs a) {
    __shared__ float part[4][8][64];
    __shared__ __attribute__((aligned(16))) bf16_t rmst[3 * 4 * 8 * 16];      // K16-blocked operand tile: [term][4 column groups][8 rows][16]
    const int c = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int b0 = blockIdx.y * 8;
    float xs[8];
    const bool st = (a.dbg & 256) != 0;
    const int sblk = blockIdx.y * gridDim.x + blockIdx.x;
    stamp(st, sblk, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[i] = 0.f;
    if (col < a.N) {
        const float* pp = a.partial + (int64_t)b0 * a.N + col;      // rows < Bp always exist
        // the slabs were written by other XCDs: every dependent batch of loads costs a full fabric round
        // trip, so a thread issues ALL its loads (up to 8 slabs x 8 rows) before the first add
        for (int k = kq; k < a.ks; k += 32) {
            float t[8][8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int ks_ = min(k + 4 * kk, a.ks - 1);          // clamped, masked below
#pragma unroll
                for (int i = 0; i < 8; ++i) t[kk][i] = pp[(int64_t)ks_ * a.slab + (int64_t)i * a.N];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float live = (k + 4 * kk < a.ks) ? 1.0f : 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[i] += live * t[kk][i];
            }
        }
    }
    // every wave contributes its slab quarter of all 8 rows, then takes TWO rows of the epilogue (the per-element
    // work -- Philox draws, sigmoid, bf16 splits -- on one wave alone was the critical path of this kernel)
#pragma unroll
    for (int i = 0; i < 8; ++i) part[kq][i][c] = xs[i];
    stamp(st, sblk, 1);
    __syncthreads();
    stamp(st, sblk, 2);
    float x2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = 2 * kq + i;
        x2[i] = ((part[0][rr][c] + part[1][rr][c]) + part[2][rr][c]) + part[3][rr][c];
    }
    float csum;
    SideIn<2> sd;
    load_side<2>(a, col, b0 + 2 * kq, sd);
    const bool staged = a.rm_src && !a.logits_only;
    const RmStage stg{rmst, 4, 8, (int)blockIdx.x * 64, b0};
    const float lsum = finish_rows<2>(a, col, b0 + 2 * kq, x2, sd, csum, staged ? stg : RmStage{});
    stamp(st, sblk, 3);
    __syncthreads();                                   // part[] is consumed: reuse it for the column / loss sums
    if (staged) flush_rm_stage(a.op, stg);
    part[kq][0][c] = csum;
    const float t = wave_sum(lsum);
    if (c == 0) part[kq][1][0] = t;
    __syncthreads();
    if (kq == 0) {
        if (a.colsum_part && col < a.N)
            a.colsum_part[(int64_t)blockIdx.y * a.N + col] = ((part[0][0][c] + part[1][0][c]) + part[2][0][c]) + part[3][0][c];
        if (a.loss_part && c == 0)
            a.loss_part[blockIdx.y * gridDim.x + blockIdx.x] = ((part[0][1][0] + part[1][1][0]) + part[2][1][0]) + part[3][1][0];
    }
}

// Softmax groups (rbm.py:113-114,129-133): grid = (n_groups, Bp/64), block = 256, group width <= 256.
//   stage : the block's 64 x width logits (left in out_prob by the main epilogue) -> LDS, coalesced
//   rows  : thread = batch row: max, sum of exp, categorical index -- from LDS, no global traffic
//   elems : thread = (column, 8-row octet): softmax value, mu-pull, clamp-mix, one-hot, all stores,
//           column-sum partial of the octet ([Bp/8][N] layout of the main epilogue), squared error
constexpr int GROUP_WMAX = 256;

// one softmax group g x one 64-row batch chunk rb (of nrb); a whole block
__device__ __forceinline__ void finish_groups_body(const FinishArgs& a, int loss_slot0, int g, int rb, int nrb) {
    __shared__ float sL[64][GROUP_WMAX + 1];
    __shared__ float rmx[64], rsum[64];
    __shared__ int ridx[64];
    __shared__ float sh[256];
    int s, e;
    group_bounds(a, g, s, e);
    const int wd = e - s;
    const int tid = threadIdx.x;
    for (int it = tid; it < 64 * wd; it += 256) {
        const int row = it / wd, j = it - row * wd;
        const int bc = min(rb * 64 + row, a.B - 1);
        sL[row][j] = a.out_prob[(int64_t)bc * a.ld_prob + s + j];
    }
    __syncthreads();
    if (tid < 64) {
        const int b = min(rb * 64 + tid, a.B - 1);
        // 8 LDS values at a time ahead of each dependent chain (a round trip per element otherwise); same summation order
        float mx = -INFINITY;
        int j0 = 0;
        for (; j0 + 8 <= wd; j0 += 8) {
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = sL[tid][j0 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) mx = fmaxf(mx, e[k]);
        }
        for (; j0 < wd; ++j0) mx = fmaxf(mx, sL[tid][j0]);
        float sum = 0.f;
        for (j0 = 0; j0 + 8 <= wd; j0 += 8) {
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = expf(sL[tid][j0 + k] - mx);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += e[k];
        }
        for (; j0 < wd; ++j0) sum += expf(sL[tid][j0] - mx);
        int idx = -1;
        if (a.vmode != 0) {
            if (a.cat_tape) idx = a.cat_tape[(int64_t)g * a.B + b];
            else {   // PHILOX: inverse CDF over clamp(t,1e-8,1), t = p (vmode 1) or mix(p) (vmode 2)  (oracle/draws.py)
                DrawSrc cs = a.cat_uni; cs.draw += g; cs.N = 1;
                const float thr = draw_uniform(cs, b, 0);
                float tot = 0.f;
                for (int pass = 0; pass < 2; ++pass) {
                    float acc = 0.f;
                    const float target = thr * tot;
                    for (int j = 0; j < wd; ++j) {
                        const int col = s + j;
                        float t = expf(sL[tid][j] - mx) / sum;
                        if (a.mu && col < a.Dz) t = (1.0f - a.eta) * t + a.eta * a.mu[(int64_t)b * a.ldmu + col];
                        if (a.vmode == 2 && a.clamp) {
                            const float m = a.mask[(int64_t)b * a.ldk + col];
                            t = t * (1.0f - m) + a.vk[(int64_t)b * a.ldk + col] * m;
                        }
                        acc += fminf(fmaxf(t, 1e-8f), 1.0f);
                        if (pass == 1 && acc > target) { idx = j; break; }
                    }
                    if (pass == 0) tot = acc; else if (idx < 0) idx = wd - 1;
                }
            }
        }
        rmx[tid] = mx; rsum[tid] = sum; ridx[tid] = idx;
    }
    __syncthreads();
    float lsum = 0.f;
    for (int it = tid; it < wd * 8; it += 256) {
        const int j = it % wd, oct = it / wd;
        const int col = s + j, b0 = rb * 64 + oct * 8;
        float xp[8], xf[8];
        float csum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = oct * 8 + i, bb = b0 + i;
            const bool ok = bb < a.B;
            const int bc = min(bb, a.B - 1);
            float p = expf(sL[row][j] - rmx[row]) / rsum[row];
            if (a.mu && col < a.Dz) p = (1.0f - a.eta) * p + a.eta * a.mu[(int64_t)bc * a.ldmu + col];
            const float m = a.clamp ? a.mask[(int64_t)bc * a.ldk + col] : 0.f;
            const float kn = a.clamp ? a.vk[(int64_t)bc * a.ldk + col] : 0.f;
            const float rv = a.loss_ref ? a.loss_ref[(int64_t)bc * a.ld_ref + col] : 0.f;
            float v;
            if (a.vmode == 0) v = a.clamp ? (p * (1.0f - m) + kn * m) : p;
            else if (a.vmode == 1) { const float o = (j == ridx[row]) ? 1.f : 0.f; v = a.clamp ? (o * (1.0f - m) + kn * m) : o; }
            else v = (j == ridx[row]) ? 1.f : 0.f;
            if (ok) {
                a.out_prob[(int64_t)bb * a.ld_prob + col] = p;
                if (a.out_final) a.out_final[(int64_t)bb * a.ld_final + col] = v;
                csum += (a.colsum_src == 2 ? v : p);
                const float d = rv - (a.loss_src == 2 ? v : p);
                lsum += a.loss_ref ? d * d : 0.f;
            }
            xp[i] = ok ? p : 0.f;
            xf[i] = ok ? v : 0.f;
        }
        if (a.rm_src == a.tr_src) {
            if (a.rm_src) store_forms<8>(a.op, a.rm_src == 2 ? xf : xp, true, true, b0, col, a.N, a.Bp);
        } else {
            if (a.rm_src) store_forms<8>(a.op, a.rm_src == 2 ? xf : xp, true, false, b0, col, a.N, a.Bp);
            if (a.tr_src) store_forms<8>(a.op, a.tr_src == 2 ? xf : xp, false, true, b0, col, a.N, a.Bp);
        }
        if (a.colsum_part) a.colsum_part[(int64_t)(b0 >> 3) * a.N + col] = csum;
    }
    if (a.loss_part) {
        sh[tid] = lsum;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int i = 0; i < 256; ++i) t += sh[i];
            a.loss_part[loss_slot0 + g * nrb + rb] = t;
        }
    }
}

__global__ __launch_bounds__(256) void finish_groups(const FinishArgs a, int loss_slot0) {
    finish_groups_body(a, loss_slot0, blockIdx.x, blockIdx.y, gridDim.y);
}

// Caller tensor -> operand forms.  With `mix`: x = vk*m + (1-m)*U  (chain initial state,
// rbm.py:271,333,392) and the mixed tensor is also written to out_f32.
struct PrepArgs {
    const float* in; int64_t ld; int B, Bp, N;
    int mix; const float* mask; int64_t ldm; DrawSrc uni;
    float* out_f32; int64_t ldo;
    OperandOut op; int* flag;      // exactness map [Bp/8][ceil(N/64)], per tile of 8 rows x 64 columns: bit 0 = some element needs
                                   // 3 bf16 terms, bit 1 = some element is neither 0 nor 1 (the bit plane op.bits does not describe it)
    float* colsum_part;            // [Bp/8][N] column sums over each 8-row group (sum data, rbm.py:223)
    int* zero; int n_zero;         // words block (0,0) of prep_operand clears (arrival counters of the split-K kernels of this call)
    int adaptive;                  // prep_item_* only: decide PER ITEM (64 columns x 64 rows) on the device what a batch of unknown content
                                   // needs -- an item whose values are all 0 / 1 leaves as bit plane + ONE bf16 plane (what the streaming K1
                                   // and the update kernel read of it), any other item with all three-term forms.  Nothing on the host has
                                   // to know (or ask: a device reduction + sync per fresh tensor) whether a batch is binary.
};
constexpr int FLAG_INEXACT = 1, FLAG_NONBINARY = 2;

// block = 64 columns x 8 rows, 256 threads = 64 columns x 4 row pairs: all loads unconditional (clamped) and
// issued before any use; the K16-blocked form leaves through the LDS stage as 16-B stores.
__global__ __launch_bounds__(256) void prep_operand(const PrepArgs a) {
    __shared__ float cs[4][64];
    __shared__ int fl[4];
    __shared__ __attribute__((aligned(16))) bf16_t rmst[3 * 4 * 8 * 16];
    const int c = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int g0 = blockIdx.y * 8, b0 = g0 + 2 * kq;
    const int cc = min(col, a.N - 1);
    if (a.zero && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < a.n_zero; i += 256) a.zero[i] = 0;
    float v[2], m[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int bc = min(b0 + i, a.B - 1);
        v[i] = a.in[(int64_t)bc * a.ld + cc];
        m[i] = a.mix ? a.mask[(int64_t)bc * a.ldm + cc] : 1.0f;
    }
    float x[2];
    float csum = 0.f;
    bool inexact = false, nonbin = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = b0 + i;
        const bool live = col < a.N && b < a.B;
        float t = v[i];
        if (a.mix) t = t * m[i] + (1.0f - m[i]) * draw_uniform(a.uni, min(b, a.B - 1), cc);
        if (live && a.out_f32) a.out_f32[(int64_t)b * a.ldo + col] = t;
        x[i] = live ? t : 0.f;
        csum += x[i];
        inexact |= (__float_as_uint(x[i]) & 0xFFFFu) != 0u;
        nonbin |= (x[i] != 0.f && x[i] != 1.0f);
        if (a.op.bits) store_bits_row(a.op, x[i] != 0.f, col, b, b < a.Bp, 0, 64);
    }
    const bool any = __any(inexact ? 1 : 0) != 0, anyb = __any(nonbin ? 1 : 0) != 0;
    if (c == 0) fl[kq] = (any ? FLAG_INEXACT : 0) | (anyb ? FLAG_NONBINARY : 0);
    cs[kq][c] = csum;
    const RmStage stg{rmst, 4, 8, (int)blockIdx.x * 64, g0};
    store_forms<2>(a.op, x, true, true, b0, col, a.N, a.Bp, stg);
    __syncthreads();
    flush_rm_stage(a.op, stg);
    if (kq == 0) {
        // plain store, rewritten by every call: no zeroing / atomics; consumers OR the entries they cover
        if (a.flag && c == 0) a.flag[blockIdx.y * gridDim.x + blockIdx.x] = (fl[0] | fl[1] | fl[2] | fl[3]);
        if (a.colsum_part && col < a.N) a.colsum_part[(int64_t)blockIdx.y * a.N + col] = ((cs[0][c] + cs[1][c]) + cs[2][c]) + cs[3][c];
    }
}

// prep_operand of the NEXT batch as extra blocks of another launch of the step (the operand forms are a pure function
// of the batch): one block = one item (64-column tile tx, 64-row chunk mb); thread = (column, 16 consecutive rows = two
// 8-row groups), so column sums and exactness flags need no cross-thread step, the [col][row] planes leave as 16-B
// stores and the K16-blocked form goes through a 24 KB LDS stage (`rmst`, the host kernel's reduction buffer).
// The barrier is LDS-only (s_waitcnt lgkmcnt + s_barrier): a __syncthreads() also drains the global stores.
// Plain prep only (no chain mix).  Same outputs, bit for bit, as the prep_operand kernel (same pieces, same summation
// order of the column sums).
//
// Where it rides was measured (tools/prefetch_probe.py): on the 16 CUs the weight-update kernel K3 leaves idle it is
// hopeless -- under K3's ~5 TB/s read+write stream a store is acknowledged after ~10 us and a wave holds at most 64
// unacknowledged memory operations (vmcnt, shared by loads and stores on gfx950): 4-6 us per item, 90 us for 13 items
// per block, with one item of lookahead, three, or loader waves (LDS-DMA) separated from the storing waves.  As a
// third resident block per CU of the fused K2 (read-only stream, 23 us) the same work is done a few us into the launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RPT rows per thread: 16 (blocks of 256 threads: four waves x two 8-row groups) or 8 (blocks of 512 threads: eight waves x one group)
// the RPT values of this thread (column c of tile tx, rows RPT kq .. + RPT - 1 of chunk mb): requested here, used by prep_item_process
template <int RPT>
__device__ __forceinline__ void prep_item_load(const PrepArgs& a, int tx, int mb, float (&v)[RPT]) {
    const int c = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int cc = min(tx * 64 + c, a.N - 1);
#pragma unroll
    for (int i = 0; i < RPT; ++i) v[i] = a.in[(int64_t)min(mb * 64 + RPT * kq + i, a.B - 1) * a.ld + cc];
}

template <int RPT>
__device__ __forceinline__ void prep_item_process(const PrepArgs& a, int tx, int mb, const float (&v)[RPT], bf16_t* rmst) {
    static_assert(RPT == 16 || RPT == 8, "rows per thread");
    constexpr int NWV = 64 / RPT;                    // waves of the block
    const int c = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int ntx = (max(a.N, a.op.ldrm) + 63) / 64;
    const int col = tx * 64 + c;
    const RmStage stg{rmst, 4, 64, tx * 64, mb * 64};
    // adaptive: the item's forms follow its content (block-wide OR of "some value is neither 0 nor 1"; the words sit
    // behind the 24 KB stage, the caller's barrier after the item protects their reuse)
    OperandOut op = a.op;
    if (a.adaptive) {
        bool nb = false;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const float x = (col < a.N && mb * 64 + RPT * kq + i < a.B) ? v[i] : 0.f;
            nb |= (x != 0.f && x != 1.0f);
        }
        int* sw = reinterpret_cast<int*>(rmst + 3 * 4 * 64 * 16);
        const bool anyb = __any(nb ? 1 : 0) != 0;
        if (c == 0) sw[kq] = anyb ? 1 : 0;
        lds_barrier();
        int any_item = 0;
#pragma unroll
        for (int q = 0; q < NWV; ++q) any_item |= sw[q];
        if (!any_item) { op.rm = nullptr; op.rm_terms = 0; op.tr_terms = 1; }
    }
#pragma unroll
    for (int hf = 0; hf < RPT / 8; ++hf) {
        const int b0 = mb * 64 + RPT * kq + 8 * hf, by = b0 >> 3;
        float x[8];
        bool inexact = false, nonbin = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] = (col < a.N && b0 + i < a.B) ? v[8 * hf + i] : 0.f;
            inexact |= (__float_as_uint(x[i]) & 0xFFFFu) != 0u;
            nonbin |= (x[i] != 0.f && x[i] != 1.0f);
            if (op.bits) store_bits_row(op, x[i] != 0.f, col, b0 + i, true, 0, 64);
        }
        // the order of prep_operand: four row pairs, combined left to right
        const float csum = (((0.f + x[0] + x[1]) + (0.f + x[2] + x[3])) + (0.f + x[4] + x[5])) + (0.f + x[6] + x[7]);
        const bool any = __any(inexact ? 1 : 0) != 0, anyb = __any(nonbin ? 1 : 0) != 0;      // the wave = this group's 64 columns
        store_forms<8>(op, x, true, true, b0, col, a.N, a.Bp, stg);
        if (a.flag && c == 0) a.flag[by * ntx + tx] = (any ? FLAG_INEXACT : 0) | (anyb ? FLAG_NONBINARY : 0);
        if (a.colsum_part && col < a.N) a.colsum_part[(int64_t)by * a.N + col] = csum;
    }
    if (op.rm) {                    // the K16-blocked form leaves through the LDS stage (block-uniform)
        lds_barrier();
        flush_rm_stage(op, stg);
    }
}

template <int RPT = 16>
__device__ __forceinline__ void prep_item_body(const PrepArgs& a, int tx, int mb, bf16_t* rmst) {
    float v[RPT];
    prep_item_load<RPT>(a, tx, mb, v);
    prep_item_process<RPT>(a, tx, mb, v, rmst);
}

// Free energy of visible configurations (imdbn/utils/energy_utils.py:19-28):
//   F(v) = -sum_i v_i b_i - sum_j softplus(c_j + (v W)_j),  softplus as torch (x > 20 ? x : log1p(exp(x)))
// one block per batch row; x = pre-activations [B][H] from the K1 path (logits_only); fixed-order tree sums.
__global__ __launch_bounds__(256) void free_energy_rows(const float* __restrict__ v, int64_t ldv, const float* __restrict__ vis_bias, int V,
                                                        const float* __restrict__ x, int64_t ldx, int H, float* __restrict__ out) {
    __shared__ float sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int i = tid; i < V; i += 256) acc += v[(int64_t)b * ldv + i] * vis_bias[i];
    for (int j = tid; j < H; j += 256) {
        const float t = x[(int64_t)b * ldx + j];
        acc += t > 20.0f ? t : log1pf(expf(t));
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) sh[tid] += sh[tid + s2];
        __syncthreads();
    }
    if (tid == 0) out[b] = -sh[0];
}

// rbm.py:216-226.  parts are [P][len] column-sum partials; loss parts are summed in double.
struct BiasArgs {
    float* hid_bias; float* hb_m; int H; const float* hpos; const float* hneg;
    float* vis_bias; float* vb_m; int V; const float* vpos; const float* vneg;
    int P; float lr, mom, n; int sparsity; float target;
    const float* loss_part; int n_loss; float loss_den; float* loss_out;
    int R; int64_t rs;           // factor-exchange mode: the partials of R ranks, rs floats apart (R <= 1: one set)
    float* pack_tail;            // statistics mode (data-parallel all-reduce): no update -- the tail of the packed buffer
                                 // [dc (H) | db (V) | sum P+ (H) | squared error] is written instead (what the pack_stats launch did)
};

__device__ __forceinline__ float sum_parts(const float* p, int P, int len, int i) {
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= P; k += 8) {          // 8 independent loads in flight, fixed summation order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = p[(int64_t)(k + j) * len + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += t[j];
    }
    for (; k < P; ++k) s += p[(int64_t)k * len + i];
    return s;
}

__device__ __forceinline__ double loss_total_256(const float* part, int n, double* sh) {
    // thread t sums parts t, t+256, ... in order, then a fixed tree: deterministic
    const int tid = threadIdx.x;
    double t = 0.0;
    for (int k = tid; k < n; k += 256) t += (double)part[k];
    sh[tid] = t;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) sh[tid] += sh[tid + s2];
        __syncthreads();
    }
    return sh[0];
}

// Work of `nblk` 256-thread blocks: block `blk` == nblk-1 reduces the squared-error partials, the others
// stride over the bias vectors.  Used by the stand-alone kernel and by the extra block row of K3.
__device__ __forceinline__ void bias_work(const BiasArgs& a, int blk, int nblk, double* sh) {
    if (a.pack_tail) {
        if (blk == nblk - 1) {
            const double t = loss_total_256(a.loss_part, a.n_loss, sh);
            if (threadIdx.x == 0) a.pack_tail[2 * a.H + a.V] = (float)t;
            return;
        }
        for (int i = blk * 256 + threadIdx.x; i < max(a.V, a.H); i += (nblk - 1) * 256) {
            if (i < a.H) {
                const float sp = sum_parts(a.hpos, a.P, a.H, i), sn = sum_parts(a.hneg, a.P, a.H, i);
                a.pack_tail[i] = sp - sn;
                a.pack_tail[a.H + a.V + i] = sp;
            }
            if (i < a.V) a.pack_tail[a.H + i] = sum_parts(a.vpos, a.P, a.V, i) - sum_parts(a.vneg, a.P, a.V, i);
        }
        return;
    }
    if (blk == nblk - 1) {
        if (a.loss_out) {
            double t = 0.0;
            for (int rk = 0; rk < max(a.R, 1); ++rk) { t += loss_total_256(a.loss_part + rk * a.rs, a.n_loss, sh); __syncthreads(); }
            if (threadIdx.x == 0) a.loss_out[0] = (float)(t / (double)a.loss_den);      // rbm.py:226
        }
        return;
    }
    for (int i = blk * 256 + threadIdx.x; i < max(a.V, a.H); i += (nblk - 1) * 256) {
        if (i < a.H) {
            float sp = 0.f, sn = 0.f;
            for (int rk = 0; rk < max(a.R, 1); ++rk) { sp += sum_parts(a.hpos + rk * a.rs, a.P, a.H, i); sn += sum_parts(a.hneg + rk * a.rs, a.P, a.H, i); }
            float m = a.hb_m[i] * a.mom;
            m = m + (a.lr * (sp - sn)) / a.n;                                   // rbm.py:216
            if (a.sparsity) m = m + (-a.lr) * (sp / a.n - a.target);           // rbm.py:218-219
            a.hb_m[i] = m;
            a.hid_bias[i] += m;
        }
        if (i < a.V) {
            float sp = 0.f, sn = 0.f;
            for (int rk = 0; rk < max(a.R, 1); ++rk) { sp += sum_parts(a.vpos + rk * a.rs, a.P, a.V, i); sn += sum_parts(a.vneg + rk * a.rs, a.P, a.V, i); }
            float m = a.vb_m[i] * a.mom;
            m = m + (a.lr * (sp - sn)) / a.n;                                   // rbm.py:223
            a.vb_m[i] = m;
            a.vis_bias[i] += m;
        }
    }
}

// grid = ceil(max(V,H)/256) + 1
__global__ __launch_bounds__(256) void bias_update(const BiasArgs a) {
    __shared__ double sh[256];
    bias_work(a, blockIdx.x, gridDim.x, sh);
}

// ---- wire form of the factor block (data-parallel factor exchange) ----------------------------------------------
// The visible operands of a CD update are mostly bits: the negative visible state of train_epoch is a SAMPLE (always
// 0/1) and layer-1 data are binary images (the caller says so: imdbn.engine.dp.enable(binary_data=True)).  A bf16
// plane [V][Bp] then travels as V*Bp/8 bytes -- 16x smaller -- and the 7 MB block of a 10000 x 1500 layer becomes
// ~2 MB (5.8 MB when only the sample is packed).  pack: full block -> compact block; unpack: N gathered compact
// blocks -> N full blocks in the layout imdbn_rbm_apply_factors reads.  Everything before the visible planes
// ("head": exactness map, hidden planes, column-sum and error partials) is copied verbatim.  A plane that claims to
// be binary and is not sets the block's `bad` word; unpack then poisons the hidden-bias partials with NaN so that the
// update fails loudly instead of silently dropping information.
struct FactorWireArgs {
    const char* src; char* dst;            // pack: one full block -> one compact block; unpack: gathered compact -> gathered full
    size_t src_stride, dst_stride; int n_ranks;
    size_t head_bytes;                     // multiple of 16
    size_t f_vpos, f_vneg, f_cs_hpos;      // offsets inside the full block: vis_tr[0] (3 planes), vis_tr[1] (first plane), cs_hpos
    size_t c_vneg, c_vpos, c_bad;          // offsets inside the compact block
    int V, Bp, binary;
    int planes_only;                       // unpack: leave the head where it is (the update reads it from the wire blocks)
    int epoch;                             // pack: a value that differs from call to call (no memset of the `bad` word needed)
    // pack from a workspace whose data-side buffers were swapped for a prefetch slot (imdbn_rbm_cd_factors_wire): the
    // exactness map, the data's column-sum partials and its planes are then NOT inside the block (nullptr: they are)
    const char* alt_flags; const char* alt_cs_vpos; const char* alt_vpos;
    size_t f_flags, n_flags, f_cs_vpos, n_cs_vpos;      // where those two sit in the head, and their (256-B padded) sizes
};

// byte i of a bit plane <-> the 8 consecutive bf16 elements 8 i .. 8 i + 7 of the plane (one 16-B access per thread: the
// first version moved 64 B per thread in four strided pieces and ran 30 us for a 4.5 MB job)
__device__ __forceinline__ uint32_t pack8_bf16(const uint4 x, bool& bad) {
    const uint32_t e[4] = {x.x, x.y, x.z, x.w};
    uint32_t w = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo = e[j] & 0xFFFFu, hi = e[j] >> 16;
        bad |= (lo != 0u && lo != 0x3F80u) || (hi != 0u && hi != 0x3F80u);
        w |= (lo ? 1u : 0u) << (2 * j);
        w |= (hi ? 1u : 0u) << (2 * j + 1);
    }
    return w;
}
__device__ __forceinline__ uint4 unpack8_bf16(uint32_t w) {
    uint32_t e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        e[j] = (((w >> (2 * j)) & 1u) ? 0x3F80u : 0u) | (((w >> (2 * j + 1)) & 1u) ? 0x3F800000u : 0u);
    return make_uint4(e[0], e[1], e[2], e[3]);
}

// grid = (blocks, 1).  Trailer of the compact block: word 0 = `bad` (set to the call's epoch by any thread that meets a
// non-binary value), word 1 = the epoch itself; the block is bad when the two agree (stale marks of earlier calls do not)
__global__ __launch_bounds__(256) void factor_pack(const FactorWireArgs a) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
    for (int64_t i = tid; i < (int64_t)(a.head_bytes / 16); i += nthr) {
        const size_t o = (size_t)i * 16;
        const char* from = a.src + o;
        if (a.alt_flags && o >= a.f_flags && o < a.f_flags + a.n_flags) from = a.alt_flags + (o - a.f_flags);
        if (a.alt_cs_vpos && o >= a.f_cs_vpos && o < a.f_cs_vpos + a.n_cs_vpos) from = a.alt_cs_vpos + (o - a.f_cs_vpos);
        reinterpret_cast<uint4*>(a.dst)[i] = *reinterpret_cast<const uint4*>(from);
    }
    const int64_t nb = (int64_t)a.V * a.Bp / 8;                       // bytes of a bit plane
    const char* vpos = a.alt_vpos ? a.alt_vpos : a.src + a.f_vpos;
    bool bad = false;
    for (int64_t i = tid; i < nb; i += nthr)
        reinterpret_cast<uint8_t*>(a.dst + a.c_vneg)[i] = (uint8_t)pack8_bf16(reinterpret_cast<const uint4*>(a.src + a.f_vneg)[i], bad);
    if (a.binary) {
        for (int64_t i = tid; i < nb; i += nthr)
            reinterpret_cast<uint8_t*>(a.dst + a.c_vpos)[i] = (uint8_t)pack8_bf16(reinterpret_cast<const uint4*>(vpos)[i], bad);
    } else {
        const int64_t n16 = (int64_t)3 * a.V * a.Bp * 2 / 16;
        for (int64_t i = tid; i < n16; i += nthr)
            reinterpret_cast<uint4*>(a.dst + a.c_vpos)[i] = reinterpret_cast<const uint4*>(vpos)[i];
    }
    if (bad) *reinterpret_cast<volatile int*>(a.dst + a.c_bad) = a.epoch;
    if (tid == 0) reinterpret_cast<volatile int*>(a.dst + a.c_bad)[1] = a.epoch;
}

// grid = (blocks, n_ranks)
__global__ __launch_bounds__(256) void factor_unpack(const FactorWireArgs a) {
    const char* src = a.src + (size_t)blockIdx.y * a.src_stride;
    char* dst = a.dst + (size_t)blockIdx.y * a.dst_stride;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
    const bool bad = reinterpret_cast<const int*>(src + a.c_bad)[0] == reinterpret_cast<const int*>(src + a.c_bad)[1];
    if (a.planes_only) {
        // the head stays in the wire block (this process's own gather buffer): poison it there
        if (bad && tid == 0) *reinterpret_cast<uint32_t*>(const_cast<char*>(src) + a.f_cs_hpos) = 0x7FC00000u;
    } else {
        for (int64_t i = tid; i < (int64_t)(a.head_bytes / 16); i += nthr) {
            uint4 x = reinterpret_cast<const uint4*>(src)[i];
            if (bad && i == (int64_t)(a.f_cs_hpos / 16)) x.x = 0x7FC00000u;   // NaN into the first hidden column-sum partial
            reinterpret_cast<uint4*>(dst)[i] = x;
        }
    }
    const int64_t nb = (int64_t)a.V * a.Bp / 8;
    for (int64_t i = tid; i < nb; i += nthr)
        reinterpret_cast<uint4*>(dst + a.f_vneg)[i] = unpack8_bf16(reinterpret_cast<const uint8_t*>(src + a.c_vneg)[i]);
    if (a.binary) {
        for (int64_t i = tid; i < nb; i += nthr)
            reinterpret_cast<uint4*>(dst + a.f_vpos)[i] = unpack8_bf16(reinterpret_cast<const uint8_t*>(src + a.c_vpos)[i]);
    } else {
        const int64_t n16 = (int64_t)3 * a.V * a.Bp * 2 / 16;
        for (int64_t i = tid; i < n16; i += nthr)
            reinterpret_cast<uint4*>(dst + a.f_vpos)[i] = reinterpret_cast<const uint4*>(src + a.c_vpos)[i];
    }
}

// ---- data-parallel split ------------------------------------------------------------------
// packed = [dW V*H][dc H][db V][sumP+ H][sqerr 1][pad]
struct PackArgs {
    float* tail; int H, V; const float* hpos; const float* hneg; const float* vpos; const float* vneg; int P;
    const float* loss_part; int n_loss;
};
__global__ __launch_bounds__(256) void pack_stats(const PackArgs a) {
    __shared__ double sh[256];
    if (blockIdx.x == gridDim.x - 1) {
        const double t = loss_total_256(a.loss_part, a.n_loss, sh);
        if (threadIdx.x == 0) a.tail[2 * a.H + a.V] = (float)t;
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.H) {
        const float sp = sum_parts(a.hpos, a.P, a.H, i), sn = sum_parts(a.hneg, a.P, a.H, i);
        a.tail[i] = sp - sn;
        a.tail[a.H + a.V + i] = sp;
    }
    if (i < a.V) a.tail[a.H + i] = sum_parts(a.vpos, a.P, a.V, i) - sum_parts(a.vneg, a.P, a.V, i);
}

struct ApplyArgs {
    float* W; float* Wm; int64_t ldw; int V, H; const float* packed;
    float* hid_bias; float* hb_m; float* vis_bias; float* vb_m;
    float lr, mom, wd, n; int sparsity; float target; float* loss_out;
};
// grid.x >= ceil(max(V,H)/256); VEC4: H % 4 == 0 and 16-B aligned rows of W, W_m and packed.
template <bool VEC4>
__global__ __launch_bounds__(256) void apply_delta(const ApplyArgs a) {
    const int hq = VEC4 ? a.H / 4 : a.H;                         // work items per row
    const int64_t total = (int64_t)a.V * hq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / hq), q = (int)(i - (int64_t)row * hq);
        if constexpr (VEC4) {
            const int64_t idx = (int64_t)row * a.ldw + 4 * q;
            const float4 d = *reinterpret_cast<const float4*>(a.packed + (int64_t)row * a.H + 4 * q);
            const float4 w0 = *reinterpret_cast<const float4*>(a.W + idx);
            float4 m = *reinterpret_cast<const float4*>(a.Wm + idx);
            m.x = m.x * a.mom; m.x = m.x + a.lr * (d.x / a.n - a.wd * w0.x);
            m.y = m.y * a.mom; m.y = m.y + a.lr * (d.y / a.n - a.wd * w0.y);
            m.z = m.z * a.mom; m.z = m.z + a.lr * (d.z / a.n - a.wd * w0.z);
            m.w = m.w * a.mom; m.w = m.w + a.lr * (d.w / a.n - a.wd * w0.w);
            *reinterpret_cast<float4*>(a.Wm + idx) = m;
            *reinterpret_cast<float4*>(a.W + idx) = make_float4(w0.x + m.x, w0.y + m.y, w0.z + m.z, w0.w + m.w);
        } else {
            const int64_t idx = (int64_t)row * a.ldw + q;
            const float wold = a.W[idx];
            float m = a.Wm[idx] * a.mom;
            m = m + a.lr * (a.packed[(int64_t)row * a.H + q] / a.n - a.wd * wold);
            a.Wm[idx] = m;
            a.W[idx] = wold + m;
        }
    }
    const float* tail = a.packed + (int64_t)a.V * a.H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 < (unsigned)max(a.V, a.H)) {
        if (i < a.H) {
            float m = a.hb_m[i] * a.mom;
            m = m + (a.lr * tail[i]) / a.n;
            if (a.sparsity) m = m + (-a.lr) * (tail[a.H + a.V + i] / a.n - a.target);
            a.hb_m[i] = m;
            a.hid_bias[i] += m;
        }
        if (i < a.V) {
            float m = a.vb_m[i] * a.mom;
            m = m + (a.lr * tail[a.H + i]) / a.n;
            a.vb_m[i] = m;
            a.vis_bias[i] += m;
        }
        if (i == 0 && a.loss_out) a.loss_out[0] = tail[2 * a.H + a.V] / (a.n * (float)a.V);
    }
}

// ---- stand-alone sample_visible (rbm.py:125-135) -------------------------------------------
__global__ __launch_bounds__(256) void bernoulli_rows(const float* p, int64_t ldp, int B, int N, DrawSrc uni, float* out, int64_t ldo) {
    const int64_t total = (int64_t)B * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / N), n = (int)(i - (int64_t)b * N);
        out[(int64_t)b * ldo + n] = (p[(int64_t)b * ldp + n] > draw_uniform(uni, b, n)) ? 1.f : 0.f;
    }
}
__global__ __launch_bounds__(64) void categorical_rows(const float* p, int64_t ldp, int B, int s, int e,
                                                       const int32_t* cat_tape, DrawSrc cu, float* out, int64_t ldo) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int wd = e - s;
    int idx;
    if (cat_tape) idx = cat_tape[b];
    else {
        float tot = 0.f;
        for (int j = 0; j < wd; ++j) tot += fminf(fmaxf(p[(int64_t)b * ldp + s + j], 1e-8f), 1.0f);
        cu.N = 1;
        const float target = draw_uniform(cu, b, 0) * tot;
        float acc = 0.f;
        idx = wd - 1;
        for (int j = 0; j < wd; ++j) {
            acc += fminf(fmaxf(p[(int64_t)b * ldp + s + j], 1e-8f), 1.0f);
            if (acc > target) { idx = j; break; }
        }
    }
    for (int j = 0; j < wd; ++j) out[(int64_t)b * ldo + s + j] = (j == idx) ? 1.f : 0.f;
}

}  // namespace imdbn
