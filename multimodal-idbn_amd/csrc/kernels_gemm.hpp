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
//       row-major   rm[t][Bp][Kpad]   (A operand of K1/K2: 8 consecutive k per lane = one 16-B load)
//       transposed  tr[t][N][Bp]      (A/B operands of K3: 8 consecutive batch rows per lane)
//     with t = 1 term for {0,1} samples and 3 terms for real-valued activations (a flag in device
//     memory decides for caller-supplied data), Bp = batch padded to 64 with zero rows.
//   * No LDS staging of operands: the fragment shapes above make every global access either a
//     128-B row segment per half-wave (W) or a 16-B piece of an L2-resident activation.
#pragma once
#include "common.hpp"

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
// K1: W is [K][N] (N contiguous).  grid = (ceil(N/64), ksplit, Bp/64), block = 256 (4 waves split K).
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void gemm_up_partial(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    float* __restrict__ partial, int Bp, int kchunk) {
    __shared__ float red[4][2][16][64];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y, mb = blockIdx.z * 64;
    const int k_begin = ks * kchunk;
    const int k_end = min(k_begin + kchunk, lda);
    const int na = a_terms ? a_terms : (*a_flag ? 3 : 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    for (int kb = k_begin + 16 * w; kb < k_end; kb += 64) {
        float wv[2][8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + nt * 32 + r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb + 8 * hh + j;
                wv[nt][j] = (k < K && n < N) ? W[(int64_t)k * ldw + n] : 0.f;
            }
        }
        uint4 bf[2][NW];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) make_w_frags<NW>(wv[nt], bf[nt]);
        for (int ta = 0; ta < na; ++ta) {
            const bf16_t* Ap = A + ta * a_term_stride + kb + 8 * hh;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint4 av = *reinterpret_cast<const uint4*>(Ap + (int64_t)(mb + mt * 32 + r) * lda);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(bf[nt][tw]), acc[mt][nt], 0, 0, 0);
            }
        }
    }
    reduce_store_tile(acc, partial + (int64_t)ks * Bp * N, N, mb, n0, red);
}

// ------------------------------------------------------------------------------------------
// K2: W is [N][K] (K contiguous).  Same grid shape with N = V, K = H.
// ------------------------------------------------------------------------------------------
template <int NW, bool VEC4>
__global__ __launch_bounds__(256) void gemm_down_partial(
    const float* __restrict__ W, int64_t ldw, int K, int N,
    const bf16_t* __restrict__ A, int64_t a_term_stride, int lda, const int* __restrict__ a_flag, int a_terms,
    float* __restrict__ partial, int Bp, int kchunk) {
    __shared__ float red[4][2][16][64];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y, mb = blockIdx.z * 64;
    const int k_begin = ks * kchunk;
    const int k_end = min(k_begin + kchunk, lda);
    const int na = a_terms ? a_terms : (*a_flag ? 3 : 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    for (int kb = k_begin + 16 * w; kb < k_end; kb += 64) {
        float wv[2][8];
        const int k0 = kb + 8 * hh;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + nt * 32 + r;
            const float* wp = W + (int64_t)n * ldw + k0;
            if constexpr (VEC4) {
                float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
                if (n < N && k0 + 3 < K) x0 = *reinterpret_cast<const float4*>(wp);
                if (n < N && k0 + 7 < K) x1 = *reinterpret_cast<const float4*>(wp + 4);
                wv[nt][0] = x0.x; wv[nt][1] = x0.y; wv[nt][2] = x0.z; wv[nt][3] = x0.w;
                wv[nt][4] = x1.x; wv[nt][5] = x1.y; wv[nt][6] = x1.z; wv[nt][7] = x1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[nt][j] = (n < N && k0 + j < K) ? wp[j] : 0.f;
            }
        }
        uint4 bf[2][NW];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) make_w_frags<NW>(wv[nt], bf[nt]);
        for (int ta = 0; ta < na; ++ta) {
            const bf16_t* Ap = A + ta * a_term_stride + k0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint4 av = *reinterpret_cast<const uint4*>(Ap + (int64_t)(mb + mt * 32 + r) * lda);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(bf[nt][tw]), acc[mt][nt], 0, 0, 0);
            }
        }
    }
    reduce_store_tile(acc, partial + (int64_t)ks * Bp * N, N, mb, n0, red);
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
};

template <int MODE>
__global__ __launch_bounds__(256) void assoc_update(const AssocArgs a) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, hh = l >> 5;
    const int v0 = blockIdx.y * 64 + (w >> 1) * 32;
    const int h0 = blockIdx.x * 128 + (w & 1) * 64;
    const int nap = a.vpos_terms ? a.vpos_terms : (*a.vpos_flag ? 3 : 1);
    const int nan_ = a.vneg_terms ? a.vneg_terms : (*a.vneg_flag ? 3 : 1);
    const bool vok = (v0 + r) < a.V;
    const bool hok0 = (h0 + r) < a.H, hok1 = (h0 + 32 + r) < a.H;
    const uint4 z4 = make_uint4(0, 0, 0, 0);

    f32x16 accp[2], accn[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accp[nt][i] = 0.f; accn[nt][i] = 0.f; }

    const int64_t vrow = (int64_t)(v0 + r) * a.Bp + 8 * hh;
    const int64_t hrow0 = (int64_t)(h0 + r) * a.Bp + 8 * hh;
    const int64_t hrow1 = (int64_t)(h0 + 32 + r) * a.Bp + 8 * hh;

    for (int kb = 0; kb < a.Bp; kb += 16) {
        for (int tb = 0; tb < a.hpos_terms; ++tb) {
            const uint4 b0 = hok0 ? *reinterpret_cast<const uint4*>(a.hpos + tb * a.hts + hrow0 + kb) : z4;
            const uint4 b1 = hok1 ? *reinterpret_cast<const uint4*>(a.hpos + tb * a.hts + hrow1 + kb) : z4;
            for (int ta = 0; ta < nap; ++ta) {
                const uint4 av = vok ? *reinterpret_cast<const uint4*>(a.vpos + ta * a.vts + vrow + kb) : z4;
                accp[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(b0), accp[0], 0, 0, 0);
                accp[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(b1), accp[1], 0, 0, 0);
            }
        }
        for (int tb = 0; tb < a.hneg_terms; ++tb) {
            const uint4 b0 = hok0 ? *reinterpret_cast<const uint4*>(a.hneg + tb * a.hts + hrow0 + kb) : z4;
            const uint4 b1 = hok1 ? *reinterpret_cast<const uint4*>(a.hneg + tb * a.hts + hrow1 + kb) : z4;
            for (int ta = 0; ta < nan_; ++ta) {
                const uint4 av = vok ? *reinterpret_cast<const uint4*>(a.vneg + ta * a.vts + vrow + kb) : z4;
                accn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(b0), accn[0], 0, 0, 0);
                accn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(av), as_frag(b1), accn[1], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = h0 + nt * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = v0 + mfma_row(reg, l);
            if (row < a.V && col < a.H) {
                const float d = accp[nt][reg] - accn[nt][reg];                  // pos_assoc - neg_assoc
                if constexpr (MODE == 0) {
                    const int64_t idx = (int64_t)row * a.ldw + col;
                    const float wold = a.W[idx];
                    float m = a.Wm[idx];
                    const float g = d / a.n - a.wd * wold;                        // rbm.py:212
                    m = m * a.mom;
                    m = m + a.lr * g;
                    a.Wm[idx] = m;
                    a.W[idx] = wold + m;                                          // rbm.py:213
                } else {
                    a.delta[(int64_t)row * a.H + col] = d;
                }
            }
        }
    }
}

}  // namespace imdbn
