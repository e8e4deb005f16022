// kernels_stream.hpp -- round-2 propagation kernels for BINARY activations: the weight matrix streams through LDS by
// LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, no staging registers) and the activations are bit planes.
//
//   k1_stream   K1  h = sigmoid(v W + c) for a 0/1 visible operand (the negative-phase sample always; the data when the
//                   caller says they are binary): 32-column tiles x a few K slices, split-K combined by the LAST block to
//                   arrive at a tile (no spin, no second launch), epilogue fused                    (rbm.py:92, :199-208)
//
// Why 32-column tiles: the split-K slab volume is (#blocks) x 64 x (columns per block) x 4 B.  With the 128-column tiles
// of gemm_up4_partial, 240 blocks leave 7.7 MB of slabs (20 K slices) for a second launch (`finish`, 7 us) to re-read.
// LDS-DMA decouples the load shape from the MFMA fragment shape: a 32-column tile is read as 8 rows x 128 B per
// instruction at the same rate as 512-B segments (tools/membench2.hip: 9.4-10.3 us for the 60 MB of a 10000 x 1500
// layer against 9.0-9.4), needs only 5 K slices (1.9 MB of slabs) and the last arriver of a tile sums 5 x 8 KB.
#pragma once
#include "common.hpp"
#include "kernels_ew.hpp"
#include "kernels_gemm.hpp"

namespace imdbn {

// LDS-DMA of 16 B per lane: lane i's bytes land at LDS[lds_off + 16 i].  M0 (the LDS base) is saved and restored inside
// the statement (cdna_hip_programming.md 5.7); the compiler does not count these operations: every wait is explicit.
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");      // wave-uniform by construction
}
// the same with 4 B per lane: lane i's dword lands at LDS[lds_off + 4 i]
__device__ __forceinline__ void dma4(const void* g, uint32_t lds_off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// fp32 -> NW bf16 terms with round-to-nearest conversions (v_cvt_pk_bf16_f32: two values and the packing in ONE instruction):
// hi = rne(w), r = w - hi (exact: <= 16 significant bits), mid = rne(r), lo = r - mid (exact in bf16: <= 8 bits), so hi + mid + lo
// == w exactly, as with the truncation split of make_w_frags -- in ~36 instead of ~56 instructions per 8 weights, which matters in
// k1_stream's issue-bound K loop.  (Other kernels keep the truncation split; the products are exact either way.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
template <int NW>
__device__ __forceinline__ void make_w_frags_rne(const float (&wv)[8], uint4 (&frag)[NW]) {
    uint32_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = cvt_pk_bf16(wv[2 * j], wv[2 * j + 1]);
    frag[0] = make_uint4(h[0], h[1], h[2], h[3]);
    if constexpr (NW == 3) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[2 * j] = wv[2 * j] - __uint_as_float(h[j] << 16);
            r[2 * j + 1] = wv[2 * j + 1] - __uint_as_float(h[j] & 0xFFFF0000u);
        }
        uint32_t m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
        frag[1] = make_uint4(m[0], m[1], m[2], m[3]);
        uint32_t lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            lo[j] = cvt_pk_bf16(r[2 * j] - __uint_as_float(m[j] << 16), r[2 * j + 1] - __uint_as_float(m[j] & 0xFFFF0000u));
        frag[2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

constexpr int K1S_WAVES = 8;                      // waves per block = two per SIMD.  With one wave per SIMD (rounds 2 / 3a) the K loop was bound by
                                                  // instruction issue, not memory: per K16 step ~85 dependent VALU instructions (fp32 -> 3 bf16 terms, bits ->
                                                  // fragments), the LDS latency of the fragment reads and 6 MFMAs back to back took ~870 cycles for 2 KB per wave
                                                  // (4.9 TB/s chip-wide against 6.4 for the bare LDS-DMA stream); two waves hide each other's latencies
constexpr int K1S_D = 2;                          // ring slots per wave; a slot = 2 K16 steps = 4 instructions = 32 rows x 128 B
constexpr int K1S_RING = K1S_D * 4 * 1024;        // 8 KB per wave (64 KB per block in flight, as with 4 waves x 16 KB)
constexpr int K1S_REGION_REAL = 16 * 1024;        // per-wave LDS of the kernels that can read bf16 terms: weight ring (2 K16 steps, 4 KB) + A ring (2 steps x 2 NA KB)
constexpr int K1S_WSTEPS = 2, K1S_ASTEPS = 2;     // ring depths (K16 steps) of that loop
constexpr int K1S_MAX_KCHUNK = 3520;              // LDS: rings + 8 B per K row of activation bits (8 x 16 KB + 27.5 KB + K1S_LDS_EXTRA <= 160 KB)
constexpr int K1S_LDS_EXTRA = 64 + 4096;          // behind the bits (adaptive operands): 16 mask / scratch words, 2 x 512 exactness-map entries

// How the activation operand of k1_stream is read (K1sArgs::amode).  Whatever the mode, a K16 step multiplies the same
// fragments in the same order -- for a 0/1 value the bit plane and the first bf16 term are the same number and the further
// terms are exact zeros -- so the result does not depend on which mode (or which per-item choice) served an element: the
// choice is speed only, and nothing on the host has to know what a batch contains.
constexpr int K1S_BITS = 0;        // 0/1 by construction (a sample): the bit plane
constexpr int K1S_ASSERTED = 1;    // the caller says 0/1: the bit plane; checked against the exactness map, NaN when false
constexpr int K1S_ADAPTIVE = 2;    // unknown content: per 64-column item the bit plane (item all 0/1 by the exactness map) or the bf16 terms
constexpr int K1S_REAL = 3;        // real values: the bf16 terms (K16-blocked operand form), no bit plane

struct K1sArgs {
    const float* W; int64_t ldw; int K, N;        // W[K][ldw] fp32, N valid columns (K = visible, N = hidden units)
    const uint8_t* abits; int Bp;                 // activations: byte-major bit plane [rup(K,64)/8][Bp]
    const int* aflag; int ncb, P;                 // exactness map of caller data ([P][ncb], FLAG_NONBINARY): nullptr = binary by construction
    float* slabs;                                 // [Bp/64][ks][tiles][64][32] partial sums
    int* counters;                                // [Bp/64][tiles] arrival counters, zero at launch, zero again at exit
    int kchunk, ks;                               // rows per K slice (multiple of 64), number of slices
    int amode;                                    // K1S_*
    int region;                                   // LDS bytes per wave: K1S_RING, or K1S_REGION_REAL where the loop over bf16 terms may run
    const bf16_t* arm; int64_t arm_ts;            // K16-blocked operand form [term][ceil(K/16)][Bp][16] (ADAPTIVE: valid for non-binary items)
    // ADAPTIVE operands written item by item (prep_item_process, PrepArgs::adaptive): an all-0/1 item has ONE transposed plane, but
    // the update kernel decides 1 or 3 planes per block over `fix_span` items -- where a span mixes both kinds, the binary items'
    // planes 1, 2 are zeroed here (what a three-term split of 0/1 values is).  Cold path; nullptr = off.
    bf16_t* fix_tr; int64_t fix_ts; int fix_span, fix_ranges;
};

// `next` / block rows >= a.ks (RIDER): the launch can carry the preparation of the NEXT batch of the training loop
// (prep_item_body: imdbn_cd_opts.next_data) as extra workgroups.  This kernel is a read-only stream with one workgroup per CU
// and exactly half of a CU's LDS (64 KB of rings + 16 KB of bits at the headline shape): a second workgroup fits beside each
// streaming one, so the extra blocks start at once and are gone a few us into the launch.  (The update kernel's idle CUs were
// tried first: under its read + write stream the same work took 25-45 us longer than the kernel itself.)
//
// NA = 0: bit-plane operands only (K1S_BITS / K1S_ASSERTED).  NA = 1 / NW: the kernel can also read NA bf16 terms per element
// (K1S_ADAPTIVE / K1S_REAL): the A fragments of a K16 step are 2 NA contiguous KB of the K16-blocked form (L2-resident) and travel
// by LDS-DMA into a second per-wave ring beside the weight ring -- no registers, every wait hand-counted (register loads written
// in inline asm were tried: the compiler moves their destination registers around before the wait that makes them valid; plain
// loads make it wait vmcnt(0) at every use beside LDS-DMA).  A block whose K slice holds only 0/1 items runs the bit-plane loop.
// GE = the general epilogue (temperature, noise, clamp, fp32 outputs, K16-blocked form ...) is compiled in; the launches of a
// CD pass only need the lean one (FinishArgs::lean), and a kernel without the general epilogue's ~25 000 instructions (and
// without the preparation blocks) measurably starts and runs faster: code size is not free here.
template <int NW, int NA, bool GE, bool RIDER>
__global__ __launch_bounds__(64 * K1S_WAVES, (RIDER && !GE) ? 4 : 2) void k1_stream(const K1sArgs a, const FinishArgs fa, const PrepArgs next) {
    constexpr bool REAL = NA > 0;
    static_assert(!(RIDER && REAL), "preparation blocks ride on the bit-plane instantiation");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [8 rings][activation bits][mask words, map entries]; no static LDS (keeps the base 16-B aligned)
    if (RIDER && (int)blockIdx.y >= a.ks) {
        const int nworkers = (gridDim.y - a.ks) * gridDim.x, wid = (blockIdx.y - a.ks) * gridDim.x + blockIdx.x;
        const int ntx = (max(next.N, next.op.ldrm) + 63) / 64;
        for (int it = wid; it < ntx; it += nworkers) {
            prep_item_body<8>(next, it, blockIdx.z, reinterpret_cast<bf16_t*>(smem));      // eight waves x one 8-row group each
            if (next.op.rm || next.adaptive) lds_barrier();
        }
        return;
    }
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 31, kg = l >> 5;      // w: provably wave-uniform (scalar branches)
    const int tile = blockIdx.x, sl = blockIdx.y, z = blockIdx.z, ntiles = gridDim.x;
    const int n0 = tile * 32, mb = z * 64;
    const int k0 = sl * a.kchunk;
    const int k_end = min(k0 + a.kchunk, (a.K + 15) / 16 * 16);
    const int nsteps = (k_end - k0) / 16;                 // K16 steps of this slice; wave w takes steps w, w + 8, ...
    const int my_steps = (nsteps - w + K1S_WAVES - 1) / K1S_WAVES;
    const int n_slots = (my_steps + 1) / 2;
    const int REGION = a.region;                          // LDS per wave
    char* ring = smem + w * REGION;
    const uint8_t* abl = reinterpret_cast<const uint8_t*>(smem + K1S_WAVES * REGION);      // [kchunk/8][64] bytes
    uint32_t* smask = reinterpret_cast<uint32_t*>(smem + K1S_WAVES * REGION + 8 * a.kchunk);      // 16 words
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)ring);
    const uint32_t abl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem) + K1S_WAVES * REGION;

    const bool st = (fa.dbg & 64) != 0;                   // tuning aid: per-block timeline (tools/stamps_probe.py)
    const int sblk = (z * gridDim.y + sl) * ntiles + tile;
    stamp(st, sblk, 0);
    const int cb0 = k0 / 64, cb1 = min((k_end + 63) / 64, a.ncb), wd = cb1 - cb0;      // 64-column items of this slice
    // ---- ADAPTIVE (host: at most 32 items per slice, at most 256 map entries per span of the update kernel): thread (p = tid >> 5,
    // j = tid & 31), tid < 256, fetches the exactness-map entry of row group p of item j, and entry `tid` of the span the update
    // kernel ORs for one of its blocks (waves 4-7 fetch the same again: every wave has the same two loads in flight).  The two
    // loads are requested BEHIND the bits and the first ring slots and looked at after the bit-plane loop: a batch of 0/1 images --
    // the case that matters -- runs exactly the instruction stream of the bit-plane kernel, and a slice that turns out to hold
    // other values is done again from its bf16 terms.  They are LDS-DMA like everything else in flight here (no destination
    // registers: register loads written in inline asm were tried -- the compiler copies their destinations around before the
    // wait that makes them valid -- and its own loads it waits for with vmcnt(0)).
    const bool adaptive = REAL && a.amode == K1S_ADAPTIVE;
    const int fix_r = sl * ntiles + tile;
    const bool fixer = REAL && adaptive && a.fix_tr && z == 0 && fix_r < a.fix_ranges;
    constexpr int NF = 2;                                 // those loads, per wave
    const uint32_t* sfl = smask + 16;                     // [512] entries of the slice's items (twice), [512] of the span (twice)
    // ---- activation bits of the slice -> LDS (16 byte-rows = 1 KB per instruction, dealt to the waves)
    const bool have_bits = !REAL || a.amode != K1S_REAL;
    if (have_bits) {
        const int nrows8 = a.kchunk / 8, last_row = (a.K + 63) / 64 * 8 - 1;
        for (int q = w; q * 16 < nrows8; q += K1S_WAVES) {
            const int br = min((k0 >> 3) + 16 * q + (l >> 2), last_row);
            dma16(a.abits + (int64_t)br * a.Bp + mb + 16 * (l & 3), abl_lds + q * 1024);
        }
    }
    // ---- weight ring
    const int colc = min(n0 + 4 * (l & 7), a.N - 4);      // 16-B chunk of the 128-B row segment (N % 4 == 0); clamped chunks feed columns >= N only
    const float* wsrc = a.W + colc;
    auto issue_slot = [&](int d, int si) __attribute__((always_inline)) {                // ring slot d <- the wave's steps 2 si, 2 si + 1
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int i = 2 * si + sub;
            if (i < my_steps) {                           // wave-uniform
                const int krow = k0 + 16 * (K1S_WAVES * i + w) + (l >> 3);
#pragma unroll
                for (int h = 0; h < 2; ++h)      // (non-temporal loads were tried in round 3: K1 unchanged, the update kernel 2-5 us slower -- W leaves the cache;
                                                  //  even the wave-uniform branch that selected them cost this loop 1.5 us: it is bound by instruction issue)
                    dma16(wsrc + (int64_t)min(krow + 8 * h, a.K - 1) * a.ldw, ring_lds + ((d * 2 + sub) * 2 + h) * 1024);
            }
        }
    };
    // epilogue side inputs (bias, ...) of this thread's column x 8 rows (threads 0..255 run the epilogue): requested now (before
    // the ring, so that the counted waits below still see the ring's instructions as the youngest), used by the tile's last
    // arriver -- their first-touch latency would otherwise sit behind the split-K combine
    // (the true column also past N: the epilogue stores nothing there but the zeros of the K16-blocked form's padding columns
    //  [N, ldrm), which the next propagation multiplies with clamped, non-zero weights; 32-column tiles never overlap)
    const int ecol = n0 + (tid & 31);
    SideIn<8> side;
    SideLean sl8;
    if (!GE || fa.lean) load_side_lean(fa, ecol, mb + 8 * ((tid >> 5) & 7), sl8);
    else if constexpr (GE) load_side<8>(fa, ecol, mb + 8 * ((tid >> 5) & 7), side);
    // the loop over bf16 terms works one K16 step at a time: weight ring of 2 steps (2 KB each) + A ring of 2 steps (2 NA KB)
    constexpr int AOPS = 2 * NA;                              // LDS-DMA per A step (2 row halves x NA terms, 1 KB each)
    const int nkb = (a.K + 15) / 16;
    const uint32_t aring_lds = ring_lds + K1S_WSTEPS * 2048;
    const char* aring = ring + K1S_WSTEPS * 2048;
    const bf16_t* abase = a.arm + (int64_t)mb * 16 + 8 * l;   // lane l: 16 B at [row l >> 1][half l & 1] of a 32-row half
    auto issue_wa = [&](int i) __attribute__((always_inline)) {      // step i: weights -> ring position i % 2, A fragments -> [i % 2][mt][term] KB
        if constexpr (REAL) {
            if (i < my_steps) {                                // wave-uniform
                const int krow = k0 + 16 * (K1S_WAVES * i + w) + (l >> 3);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    dma16(wsrc + (int64_t)min(krow + 8 * h, a.K - 1) * a.ldw, ring_lds + ((i % K1S_WSTEPS) * 2 + h) * 1024);
                const int kb = min((k0 >> 4) + K1S_WAVES * i + w, nkb - 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int t = 0; t < NA; ++t)
                        dma16(abase + t * a.arm_ts + ((int64_t)kb * a.Bp + 32 * mt) * 16, aring_lds + (((i % K1S_ASTEPS) * 2 + mt) * NA + t) * 1024);
            }
        }
    };
    const bool known_real = REAL && a.amode == K1S_REAL;
    if (known_real) {
        issue_wa(0);
        issue_wa(1);
    } else {
#pragma unroll
        for (int d = 0; d < K1S_D; ++d) issue_slot(d, d);
        if (adaptive) {
            const int i0 = fixer ? fix_r * a.fix_span : 0, nitem = fixer ? min(a.fix_span, a.ncb - i0) : 1, ne = fixer ? nitem * a.P : 1;
            const int t8 = tid & 255, ie = min(t8, ne - 1);
            const uint32_t sfl_lds = abl_lds + 8 * a.kchunk + 64 + w * 256;
            dma4(a.aflag + (int64_t)(z * 8 + (t8 >> 5)) * a.ncb + min(cb0 + (tid & 31), a.ncb - 1), sfl_lds);
            dma4(a.aflag + (int64_t)(ie / nitem) * a.ncb + i0 + (ie % nitem), sfl_lds + 2048);
        }
        // the bits (issued first) have landed once at most the ring's instructions (and the map loads behind them) are outstanding
        if (my_steps >= 2 * K1S_D) { if (adaptive) wait_vmcnt<4 * K1S_D + NF>(); else wait_vmcnt<4 * K1S_D>(); }
        else wait_vmcnt<0>();
    }
    __syncthreads();
    stamp(st, sblk, 1);

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    if (!known_real) {
        // a slot = two K16 steps: the fragments of BOTH are requested from LDS before the first one's arithmetic (the second step's
        // LDS latency hides under the first step's VALU / MFMA work; a partial last slot reads LDS garbage it does not use)
        auto math = [&](const float (&x)[8], const uint32_t (&bb)[2], int) __attribute__((always_inline)) {
            uint4 bf[NW];
            make_w_frags_rne<NW>(x, bf);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint4 af = bits_to_frag(bb[mt]);
#pragma unroll
                for (int tw = 0; tw < NW; ++tw)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(af), as_frag(bf[tw]), acc[mt], 0, 0, 0);
            }
        };
        for (int s0 = 0; s0 < n_slots; s0 += K1S_D) {
#pragma unroll
            for (int d = 0; d < K1S_D; ++d) {
                const int si = s0 + d;
                if (si < n_slots) {                           // wave-uniform
                    // slot si has landed when only the K1S_D - 1 younger slots are outstanding -- if they were all issued in full
                    // (ADAPTIVE, first pass over the ring: plus the map loads issued behind the first slots)
                    if (2 * (si + K1S_D - 1) + 1 < my_steps) { if (adaptive && s0 == 0) wait_vmcnt<4 * (K1S_D - 1) + NF>(); else wait_vmcnt<4 * (K1S_D - 1)>(); }
                    else wait_vmcnt<0>();
                    // (The two waves of a SIMD -- w and w + 4 -- are arbitrated by priority, then AGE: the older wave runs nearly unimpeded,
                    //  waves 4-7 leave this loop ~2.2 us after waves 0-3.  Alternating s_setprio slot by slot, opposite in the two partners,
                    //  evened them out (gap 1.0 us) and slowed the older ones by as much: zero-sum, not kept.)
                    float x0[8], x1[8];
                    uint32_t b0[2], b1[2];
                    const char* base = ring + (d * 2 * 2) * 1024 + (8 * kg) * 128 + 4 * n;
                    const int j0 = K1S_WAVES * (2 * si) + w, j1 = j0 + K1S_WAVES;      // K16 steps inside the slice: bytes 2 j, 2 j + 1 of a batch row
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x0[j] = *reinterpret_cast<const float*>(base + j * 128); x1[j] = *reinterpret_cast<const float*>(base + 2048 + j * 128); }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) { b0[mt] = abl[(2 * j0 + kg) * 64 + 32 * mt + n]; b1[mt] = abl[(2 * min(j1, nsteps - 1) + kg) * 64 + 32 * mt + n]; }
                    math(x0, b0, 0);
                    if (2 * si + 1 < my_steps) math(x1, b1, 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's fragments are in registers: refill it
                    issue_slot(d, si + K1S_D);
                }
            }
        }
    }
    uint32_t mlo = 0u;                                    // bit j: item j of the slice is read from its bf16 terms
    if constexpr (REAL) {
        if (adaptive) {
            // item j of the slice needs its bf16 terms when any of its 8 row groups holds a value that is neither 0 nor 1:
            // lanes 0-31 / 32-63 of wave w < 4 hold row groups 2w / 2w + 1 of item j
            wait_vmcnt<0>();
            const unsigned long long m = __ballot(((sfl[tid & 255] & FLAG_NONBINARY) != 0u && (tid & 31) < wd) ? 1 : 0);
            if (l == 0 && w < 4) smask[w] = (uint32_t)m | (uint32_t)(m >> 32);
        }
    }
    if (!known_real) wait_vmcnt<0>();                     // (announced real values: the rings requested in the prologue stay in flight)
    stamp(st, sblk, 2);
    if (st && tid == 64 * (K1S_WAVES - 1) && sblk < 4096) g_stamps[sblk * 8 + 7] = wall_clock64();      // (tuning aid: when the youngest wave leaves the loop)
    __syncthreads();                                      // every wave is done with its ring (the area becomes red[8][64][32]); the mask words are complete
    if constexpr (REAL) {
        if (adaptive) mlo = __builtin_amdgcn_readfirstlane(smask[0] | smask[1] | smask[2] | smask[3]);      // items 0 .. 31 of the slice
    }
    const bool real_loop = REAL && (known_real || mlo != 0u);
    if constexpr (REAL) if (real_loop) {
        // ---- operands with bf16 terms, one K16 step at a time.  Issue order: prologue W0 A0 W1 A1; iteration i, after its MFMAs:
        // W(i+2) A(i+2).  vmcnt retires in order, so before iteration i everything up to A(i) has landed once only the younger
        // W(i+1) A(i+1) are outstanding.  (ADAPTIVE: the slice was first run as bit planes -- all waves are past that loop, the
        // rings are free -- and is done again from scratch: the price of real values in a batch nobody announced.)
        if (!known_real) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
            issue_wa(0);
            issue_wa(1);
        }
        for (int i = 0; i < my_steps; ++i) {
            if (i + 1 < my_steps) wait_vmcnt<2 + AOPS>(); else wait_vmcnt<0>();
            const char* base = ring + ((i % K1S_WSTEPS) * 2) * 1024 + (8 * kg) * 128 + 4 * n;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float*>(base + j * 128);
            uint4 bf[NW];
            make_w_frags_rne<NW>(x, bf);
            const int jstep = K1S_WAVES * i + w, item = jstep >> 2;
            const bool isbin = a.amode == K1S_ADAPTIVE && ((mlo >> (item & 31)) & 1u) == 0u;      // wave-uniform
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                uint4 at[NA];
                if (isbin) {
                    at[0] = bits_to_frag(abl[(2 * jstep + kg) * 64 + 32 * mt + n]);
#pragma unroll
                    for (int t = 1; t < NA; ++t) at[t] = make_uint4(0u, 0u, 0u, 0u);
                } else {
#pragma unroll
                    for (int t = 0; t < NA; ++t)
                        at[t] = *reinterpret_cast<const uint4*>(aring + (((i % K1S_ASTEPS) * 2 + mt) * NA + t) * 1024 + n * 32 + kg * 16);
                }
#pragma unroll
                for (int ta = 0; ta < NA; ++ta)
#pragma unroll
                    for (int tw = 0; tw < NW; ++tw)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(at[ta]), as_frag(bf[tw]), acc[mt], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the step's fragments are in registers: refill its ring places
            __builtin_amdgcn_sched_barrier(0);
            issue_wa(i + 2);
        }
        wait_vmcnt<0>();
        __syncthreads();                                  // every wave is done with its rings
    }

    if constexpr (REAL) {
        // ---- ADAPTIVE slots: where the update kernel will read three planes (an inexact entry in its span of items), give the
        // all-0/1 items of the span their planes 1, 2 (zeros).  Nothing to do for batches that are 0/1 throughout or real throughout.
        auto block_or = [&](int v) __attribute__((always_inline)) {      // (the mask words are free now; no static LDS in this kernel)
            const bool wany = __any(v) != 0;                  // (all lanes: not inside the lane-0 branch)
            if (l == 0) smask[w] = wany ? 1u : 0u;
            __syncthreads();
            uint32_t r = 0u;
#pragma unroll
            for (int ww = 0; ww < K1S_WAVES; ++ww) r |= smask[ww];
            __syncthreads();
            return r != 0u;
        };
        if (fixer && block_or(sfl[512 + (tid & 255)] & FLAG_INEXACT)) {
            const int i0 = fix_r * a.fix_span, nitem = min(a.fix_span, a.ncb - i0);
            for (int it = 0; it < nitem; ++it)
                for (int zc = 0; zc < a.Bp / 64; ++zc) {
                    const int nb = (tid < 8) ? (a.aflag[(zc * 8 + tid) * a.ncb + i0 + it] & FLAG_NONBINARY) : 0;
                    if (block_or(nb)) continue;
                    for (int i = tid; i < 2 * 64 * 8; i += 64 * K1S_WAVES) {
                        const int pl = 1 + (i >> 9), c = (i >> 3) & 63, q = i & 7, col = (i0 + it) * 64 + c;
                        if (col < a.K) *reinterpret_cast<uint4*>(a.fix_tr + pl * a.fix_ts + (int64_t)col * a.Bp + zc * 64 + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
        }
    }

    // ---- cross-wave sum (fixed order) -> threads 0..255: column n0 + c, batch rows mb + 8 oct .. + 7
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(w * 64 + mt * 32 + mfma_row(reg, l)) * 32 + n] = acc[mt][reg];
    __syncthreads();
    if constexpr (!GE) {
        // ---- lean epilogue (the CD configuration): ALL EIGHT waves go on, one column x FOUR rows per thread -- the cross-wave sum, the
        // publish / combine of the split-K partials and the epilogue are each half as long per thread as with four waves x 8 rows
        auto sync8 = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        const int c = tid & 31, hex = tid >> 5;
        float xs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (4 * hex + i) * 32 + c;
            float t = red[o];
#pragma unroll
            for (int ww = 1; ww < K1S_WAVES; ++ww) t += red[ww * 2048 + o];
            xs[i] = t;
        }
        int* s_words = reinterpret_cast<int*>(smem + K1S_WAVES * 8192);
        if (a.amode == K1S_ASSERTED) {                    // promised 0/1 and is not: NaN, loudly (as below)
            int bad = 0;
            for (int i = tid; i < wd * 8; i += 64 * K1S_WAVES) bad |= a.aflag[(z * 8 + i / wd) * a.ncb + cb0 + (i % wd)] & FLAG_NONBINARY;
            const bool wbad = __any(bad) != 0;
            if (l == 0) s_words[4 + w] = wbad ? 1 : 0;
            sync8();
            int anyb = 0;
#pragma unroll
            for (int ww = 0; ww < K1S_WAVES; ++ww) anyb |= s_words[4 + ww];
            if (anyb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xs[i] = __uint_as_float(0x7FC00000u);
            }
        }
        stamp(st, sblk, 3);
        if (a.ks > 1) {
            typedef __attribute__((address_space(1))) uint32_t gu32;
            gu32* mine = (gu32*)(a.slabs + (((int64_t)z * a.ks + sl) * ntiles + tile) * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) __hip_atomic_store(mine + (4 * hex + i) * 32 + c, __float_as_uint(xs[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sync8();                                      // EVERY storing wave drains (vmcnt(0)) before the barrier, the counter add comes after it
            int* cnt = a.counters + z * ntiles + tile;
            int* s_last = s_words;
            if (tid == 0) *s_last = (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ks - 1) ? 1 : 0;
            sync8();
            stamp(st, sblk, 4);
            if (!*s_last) return;
            if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
            const gu32* s0p = (const gu32*)(a.slabs + ((int64_t)z * a.ks * ntiles + tile) * 2048) + (4 * hex) * 32 + c;
            for (int k = 0; k < a.ks; k += 8) {           // summed strictly in slice order, as below
                uint32_t t[8][4];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        t[q][i] = __hip_atomic_load(s0p + (int64_t)min(k + q, a.ks - 1) * ntiles * 2048 + i * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k + q < a.ks) {                       // block-uniform
#pragma unroll
                        for (int i = 0; i < 4; ++i) xs[i] = (k + q == 0) ? __uint_as_float(t[q][i]) : xs[i] + __uint_as_float(t[q][i]);
                    }
            }
        }
        stamp(st, sblk, 5);
        finish_lean4(fa, ecol, mb + 4 * hex, xs, (mb >> 3) + (hex >> 1), sl8.bias, 1, 32);
        stamp(st, sblk, 6);
        return;
    }
    if (tid >= 256) return;                               // general epilogue: the work of four waves (one column x 8 rows per thread)
    auto sync4 = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };      // the four waves left
    const int c = tid & 31, oct = tid >> 5;
    float xs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int o = (8 * oct + i) * 32 + c;
        float t = red[o];
#pragma unroll
        for (int ww = 1; ww < K1S_WAVES; ++ww) t += red[ww * 2048 + o];
        xs[i] = t;
    }
    // caller data that were promised to be 0/1 and are not: the bit plane does not describe them -> NaN, loudly
    int* s_words = reinterpret_cast<int*>(smem + K1S_WAVES * 8192);        // scratch words behind red[] (rings / bits are free now)
    if (a.amode == K1S_ASSERTED) {
        int bad = 0;
        for (int i = tid; i < wd * 8; i += 256) bad |= a.aflag[(z * 8 + i / wd) * a.ncb + cb0 + (i % wd)] & FLAG_NONBINARY;
        const bool wbad = __any(bad) != 0;
        if (l == 0) s_words[4 + w] = wbad ? 1 : 0;
        sync4();
        if (s_words[4] | s_words[5] | s_words[6] | s_words[7]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xs[i] = __uint_as_float(0x7FC00000u);
        }
    }
    stamp(st, sblk, 3);
    if (a.ks > 1) {
        // ---- publish the partial tile (write-through stores, every 128-B line whole from one instruction), count the
        // arrival; the block whose add comes last sums all slices in slice order (same result whoever is last)
        typedef __attribute__((address_space(1))) uint32_t gu32;
        gu32* mine = (gu32*)(a.slabs + (((int64_t)z * a.ks + sl) * ntiles + tile) * 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) __hip_atomic_store(mine + (8 * oct + i) * 32 + c, __float_as_uint(xs[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sync4();                                          // EVERY storing wave drains (vmcnt(0)) before the barrier, the counter add comes after it
        int* cnt = a.counters + z * ntiles + tile;
        int* s_last = s_words;
        if (tid == 0) *s_last = (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ks - 1) ? 1 : 0;
        sync4();
        stamp(st, sblk, 4);
        if (!*s_last) return;
        if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[i] = 0.f;
        const gu32* s0p = (const gu32*)(a.slabs + ((int64_t)z * a.ks * ntiles + tile) * 2048) + (8 * oct) * 32 + c;
        // up to 8 slices per round trip (5 at the headline shape: ONE dependent batch of loads instead of 4 + 1; surplus slots
        // re-read the last slice and are not added); summed strictly in slice order
        for (int k = 0; k < a.ks; k += 8) {
            uint32_t t[8][8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    t[q][i] = __hip_atomic_load(s0p + (int64_t)min(k + q, a.ks - 1) * ntiles * 2048 + i * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (k + q < a.ks) {                           // block-uniform
#pragma unroll
                    for (int i = 0; i < 8; ++i) xs[i] = (k + q == 0) ? __uint_as_float(t[q][i]) : xs[i] + __uint_as_float(t[q][i]);
                }
        }
    }
    stamp(st, sblk, 5);
    // ---- epilogue of `finish`, fused: bias, sigmoid, Bernoulli sample, operand forms, column sums
    if (!GE || fa.lean) (void)finish_lean8(fa, ecol, mb + 8 * oct, xs, (mb >> 3) + oct, sl8, 1, 32);
    else if constexpr (GE) (void)finish_rows8(fa, ecol, mb + 8 * oct, xs, (mb >> 3) + oct, side);
    stamp(st, sblk, 6);
}


// ------------------------------------------------------------------------------------------------------------------
//   k2_stream   K2  v = sample(sigmoid(h W^T + b)) for a 0/1 hidden operand (rbm.py:96-135, :205-206), epilogue fused.
//
// W is [N][K] here (N = visible units = rows of W, K = hidden units, contiguous): a block owns TR whole rows, TR chosen
// so that the tiles are dealt ONE per CU (TR = 40 -> 250 blocks at 10000 rows; the 32-wide tiles of gemm_down_fused
// put two 24-row tiles on most CUs and one on the rest: its K loop took 11 us on some CUs and 16 us on others).  TR is
// not a multiple of 32, so the MFMA is v_mfma_f32_16x16x32_bf16 with the WEIGHT rows as the M dimension (ceil(TR/16) tiles,
// the last one partly idle) and the batch as N (4 tiles of 16 rows): a lane's A fragment = 8 consecutive k of one weight
// row = two adjacent float4 loads, a wave-instruction = 16 rows x 128 B.  Four waves split K (32-wide steps dealt round
// robin) behind a 3-deep register ring; the activations are the byte-major bit plane, staged once in LDS.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int K2S_D = 3;                          // register ring depth (K32 steps) per wave
constexpr int K2S_W = 8;                          // waves per block: the 40 x 8 (column, row octet) items of the epilogue are ONE pass (5 waves)
                                                  // instead of two (with 4 waves the second pass cost 2.5 us)
constexpr int K2S_LW = 4;                         // waves that stream weights (one per SIMD; with all 8 the K loop ended 1.1 us later)
constexpr int K2S_LDR = 68;                       // row pitch (floats) of the reduction buffer red[K2S_W][16 MT][K2S_LDR]

struct K2sArgs {
    const float* W; int64_t ldw; int K, N;        // W[N][ldw]: N rows (visible units), K valid columns (hidden units), K % 4 == 0
    const uint8_t* abits; int Bp;                 // hidden sample: byte-major bit plane [rup(K,64)/8][Bp]
    int TR;                                       // rows per block: multiple of 8, <= 16 MT
};

template <int MT>
struct K2sOps { float4 w[MT][2]; };

// GE: the general epilogue is compiled in (see k1_stream: the CD pass only needs the lean one, and dead code costs time here)
template <int NW, int MT, bool GE>
__global__ __launch_bounds__(64 * K2S_W, 2) void k2_stream(const K2sArgs a, const FinishArgs fa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bx = blockIdx.x, bz = blockIdx.z, nbx = gridDim.x;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, m = l & 15, kg = l >> 4;
    const int TR = a.TR, v0 = bx * TR, mb = bz * 64;
    const int nsteps = (a.K + 31) / 32;                       // K32 steps; wave w takes steps w, w + K2S_W, ...
    const int my = w < K2S_LW ? (nsteps - w + K2S_LW - 1) / K2S_LW : 0;
    uint8_t* hbl = reinterpret_cast<uint8_t*>(smem);          // [nsteps * 4][64] activation bytes of this batch chunk
    float* red = reinterpret_cast<float*>(smem + ((nsteps * 4 * 64 + 255) & ~255));      // [K2S_LW][16 MT][K2S_LDR]

    // ---- hidden bits -> LDS: 16-B pieces, plain loads (12 KB at K = 1500)
    {
        const int n16 = nsteps * 4 * 4;                       // 16-B pieces: byte-row r, quarter q
        const int last_row = (a.K + 63) / 64 * 8 - 1;
        for (int i = tid; i < n16; i += 64 * K2S_W) {
            const int r = i >> 2, q = i & 3;
            *reinterpret_cast<uint4*>(hbl + r * 64 + 16 * q) =
                *reinterpret_cast<const uint4*>(a.abits + (int64_t)min(r, last_row) * a.Bp + mb + 16 * q);
        }
    }
    // ---- weight ring: all loads unconditional from clamped addresses (steps past the wave's last one re-read it: L2 hits)
    const float* wrow[MT];
    const int vlast = min(v0 + TR, a.N) - 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wrow[mt] = a.W + (int64_t)min(v0 + 16 * mt + m, vlast) * a.ldw;
    auto load = [&](K2sOps<MT>& o, int i) {                   // wave-local step i
        const int k0 = 32 * (K2S_LW * max(min(i, my - 1), 0) + (w & (K2S_LW - 1))) + 8 * kg;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            o.w[mt][0] = *reinterpret_cast<const float4*>(wrow[mt] + min(k0, a.K - 4));
            o.w[mt][1] = *reinterpret_cast<const float4*>(wrow[mt] + min(k0 + 4, a.K - 4));
        }
    };
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const K2sOps<MT>& o, int i) {
        const int step = K2S_LW * i + w;
        uint4 bf[4];                                          // batch fragments: k = 32 step + 8 kg + j of batch row 16 nt + m
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = bits_to_frag(hbl[(4 * step + kg) * 64 + 16 * nt + m]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float x[8] = {o.w[mt][0].x, o.w[mt][0].y, o.w[mt][0].z, o.w[mt][0].w, o.w[mt][1].x, o.w[mt][1].y, o.w[mt][1].z, o.w[mt][1].w};
            uint4 af[NW];
            make_w_frags<NW>(x, af);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int tw = 0; tw < NW; ++tw)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[tw]), as_frag(bf[nt]), acc[mt][nt], 0, 0, 0);
        }
    };
    K2sOps<MT> ring[K2S_D];
    const bool st = (fa.dbg & 128) != 0;
    const int sblk = bz * nbx + bx;
    stamp(st, sblk, 0);
    // epilogue item of this thread: pass A = columns 0 .. 31 (waves 0-3: lanes = 32 columns x 2 row octets), pass B = columns
    // 32 .. 47 (waves 4, 5: 16 columns x 4 octets); waves 6, 7 have none.  Side inputs are requested before the K loop.
    const bool isA = w < 4, isB = MT == 3 && TR > 32 && (w == 4 || w == 5);
    const int ec = isA ? (tid & 31) : 32 + (tid & 15), eoct = isA ? (tid >> 5) : ((tid - 256) >> 4);
    const int ecol = (ec < TR) ? v0 + ec : (1 << 30);
    SideIn<8> side;
    SideLean sl;
    if (isA || isB) {
        if (!GE || fa.lean) load_side_lean(fa, ecol, mb + 8 * eoct, sl);
        else if constexpr (GE) load_side<8>(fa, ecol, mb + 8 * eoct, side);
    }
    if (w < K2S_LW) {
#pragma unroll
        for (int d = 0; d < K2S_D; ++d) load(ring[d], d);
    }
    __syncthreads();                                          // the activation bytes are in LDS
    stamp(st, sblk, 1);
    for (int g = 0; g * K2S_D < my; ++g) {
#pragma unroll
        for (int d = 0; d < K2S_D; ++d) {
            const int i = g * K2S_D + d;
            __builtin_amdgcn_sched_barrier(0);
            if (i < my) compute(ring[d], i);                  // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            load(ring[d], i + K2S_D);
        }
    }
    stamp(st, sblk, 2);
    // ---- cross-wave reduction: C tile (mt, nt): lane holds batch row 16 nt + m, weight rows 16 mt + 4 kg + reg
    if (w < K2S_LW) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) red[(w * 16 * MT + 16 * mt + 4 * kg + reg) * K2S_LDR + 16 * nt + m] = acc[mt][nt][reg];
    }
    __syncthreads();
    stamp(st, sblk, 3);
    float lsum = 0.f;
    if (isA || isB) {                                         // wave-uniform
        float xs[8];
        const float* p = red + ec * K2S_LDR + 8 * eoct;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = p[i];
#pragma unroll
            for (int ww = 1; ww < K2S_LW; ++ww) t += p[ww * 16 * MT * K2S_LDR + i];      // fixed order
            xs[i] = t;
        }
        const int bshape = isA ? 1 : 2, bcols = isA ? min(TR, 32) : TR - 32;
        if (!GE || fa.lean) lsum = finish_lean8(fa, ecol, mb + 8 * eoct, xs, (mb >> 3) + eoct, sl, bshape, bcols);
        else if constexpr (GE) lsum = finish_rows8(fa, ecol, mb + 8 * eoct, xs, (mb >> 3) + eoct, side, RmStage{}, bshape, bcols);
    }
    stamp(st, sblk, 4);
    if (fa.loss_part) {
        __syncthreads();                                      // red is consumed
        const float t = wave_sum(lsum);
        if (l == 0) red[w] = t;
        __syncthreads();
        if (tid == 0) {
            float t2 = red[0];
#pragma unroll
            for (int ww = 1; ww < K2S_W; ++ww) t2 += red[ww];
            fa.loss_part[bz * nbx + bx] = t2;
        }
    }
    stamp(st, sblk, 5);
}

}  // namespace imdbn
