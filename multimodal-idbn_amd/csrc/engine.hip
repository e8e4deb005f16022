// engine.hip -- C ABI (include/imdbn_engine.h) and host-side launch plans of the CD engine.
//
// Everything here is launch orchestration: carve the caller's workspace, advance the draw cursor in
// the reference's draw order, and enqueue kernels on the caller's stream.
// No host synchronisation, no allocation, no global state besides tuning knobs, the thread-local
// error string and the optional profiling events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <vector>

#include "../../include/imdbn_engine.h"
#include "kernels_ew.hpp"
#include "kernels_gemm.hpp"
#include "kernels_chain.hpp"
#include "kernels_stream.hpp"

using namespace imdbn;

namespace {

thread_local char g_err[512] = "";
// Tuning / testing knobs.  The process-wide defaults are set by imdbn_set_option / imdbn_set_tuning; a caller that wants its own
// (two engines with different settings in one process) creates an imdbn_options handle and binds it to its thread with
// imdbn_use_options: every engine call made by that thread then reads the handle instead of the defaults.  Knobs that shape the
// workspace layout (split-K factors, tile heights) must be the same for all calls that share a workspace.
struct Tuning {
    int ks_up = 0, ks_down = 0;      // split-K factors of the generic propagation kernels (0 = automatic)
    bool no_fast_k3 = false;         // testing: force the unaligned-shape update kernel
    bool no_fast_k1 = false;
    bool no_fused_up = false;
    int k4_rows = 0;                 // tuning: batch rows per chain-kernel block (0 = automatic)
    int no_rank_loop = 0;            // testing: one update-kernel launch per gathered rank block
    int no_chain_kernel = 0;         // testing: run chains as one launch per half step
    int no_rank_acc = 0;             // testing: tile-wise rank loop (k3_body_ranks) even when the accumulating form applies
    int min_rank_loop = 2;           // apply_factors: rank blocks from which the single-launch rank loop is used (1 block: the plain update kernel, 46 vs 60 us)
    int no_prefetch = 0;             // testing: ignore imdbn_cd_opts.next_data
    int no_bits = 0;                 // testing: never use the bit-packed hidden operand
    int down_tr = 0;                 // tuning: rows per fused-K2 block (0 = automatic)
    int no_k1s = 0;                  // testing: never use the LDS-DMA streaming K1 for binary operands (k1_stream)
    int k1s_ks = 0;                  // tuning: K slices of k1_stream (0 = automatic)
    int no_k2s = 0;                  // testing: never use k2_stream (the fused K2 for a bit-plane hidden operand, one tile per CU)
    int k2s_tr = 0;                  // tuning: rows per k2_stream block (0 = automatic; multiple of 8, <= 48)
    int no_k1s_real = 0;             // testing: real-valued operands of K1 take the partial GEMM + finish launches, not k1_stream
    int no_adaptive = 0;             // testing: a prefetched batch of unknown content gets all three-term forms (no per-item choice)
    int k1s_lds_pad = 0;             // experiment: extra dynamic LDS (bytes) for the bit-plane k1_stream
    int no_chain_pair = 0;           // testing: imdbn_rbm_chain_pair runs its chains one after the other
    int no_down_tiled = 0;           // testing: multi-chunk real-valued K2 without the LDS-tiled kernel
    int no_down_chunks = 0;          // testing: the fused K2 of a multi-chunk batch runs one block per (tile, 64-row chunk)
    int k1s_force_na = 0;            // experiment: bit-plane operands run on the kernel instantiation that can also read bf16 terms
};
Tuning g_defaults;
thread_local const Tuning* t_bound = nullptr;
inline const Tuning& tune() { return t_bound ? *t_bound : g_defaults; }
#define g_ks_up (tune().ks_up)
#define g_ks_down (tune().ks_down)
#define g_no_fast_k3 (tune().no_fast_k3)
#define g_no_fast_k1 (tune().no_fast_k1)
#define g_no_fused_up (tune().no_fused_up)
#define g_k4_rows (tune().k4_rows)
#define g_no_rank_loop (tune().no_rank_loop)
#define g_no_chain_kernel (tune().no_chain_kernel)
#define g_no_rank_acc (tune().no_rank_acc)
#define g_min_rank_loop (tune().min_rank_loop)
#define g_no_prefetch (tune().no_prefetch)
#define g_no_bits (tune().no_bits)
#define g_down_tr (tune().down_tr)
#define g_no_k1s (tune().no_k1s)
#define g_k1s_ks (tune().k1s_ks)
#define g_no_k2s (tune().no_k2s)
#define g_k2s_tr (tune().k2s_tr)
#define g_no_k1s_real (tune().no_k1s_real)
#define g_no_adaptive (tune().no_adaptive)
#define g_k1s_lds_pad (tune().k1s_lds_pad)
#define g_k1s_force_na (tune().k1s_force_na)
#define g_no_chain_pair (tune().no_chain_pair)
#define g_no_down_chunks (tune().no_down_chunks)
#define g_no_down_tiled (tune().no_down_tiled)
int g_dbg = 0;    // tuning aid: kernels that record per-block timeline stamps (64 K1, 128 K2, 256 finish, 512 K3; tools/stamps_probe.py)

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail((int)e_, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)
#define CHK(expr)                   \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != 0) return rc_;   \
    } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- profiling of the update kernel (bench.py roofline leg) --------------------------------
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
    unsigned calls = 0;           // only every 8th update launch is bracketed (the 4th, 12th, ...): a bracket costs the stream ~10 us
                                  // (two event records: measured in the kernel trace as +5 us on the bracketed step and +5 on the next)
} g_prof;

// ---- split-K plan --------------------------------------------------------------------------
struct Split { int ks; int kchunk; };
Split plan_split(int Kpad, int n_tiles, int m_blocks, int forced, int cap) {
    int ks = forced;
    if (ks <= 0) {
        const int target = 768;    // ~3 workgroups per CU on 256 CUs
        ks = std::max(1, (int)((double)target / (double)(n_tiles * m_blocks) + 0.5));
    }
    ks = std::min(ks, cap);
    ks = std::min(ks, cdiv(Kpad, 64));
    ks = std::max(ks, 1);
    const int kchunk = rup(cdiv(Kpad, ks), 64);
    return {cdiv(Kpad, kchunk), kchunk};
}

// ---- workspace layout ----------------------------------------------------------------------
struct Layout {
    int V, H, B, Bp, Vpad, Hpad, P;
    Split up, down;
    bool up4;
    int* flags; int* flags_h;
    bf16_t* vis_rm[2];
    bf16_t* vis_tr[2];
    bf16_t* hid_rm;
    bf16_t* hid_tr[2];
    uint8_t* hid_bits; int ldbits;       // bit plane of the sampled hidden states, byte-major [Hpad64/8][Bp]
    uint8_t* vis_bits[2];                // bit planes of the visible operands (0: data, 1: negative-phase sample), [Vpad64/8][Bp]
    uint8_t* pf_bits[2];                 // ... of the prefetch slots
    float* partial;
    float* f_h;
    float* f_vp;
    float* f_v[2];
    float* cs_hpos; float* cs_hneg; float* cs_vpos; float* cs_vneg;
    float* loss_part; int n_loss_slots;
    size_t fb_off, fb_bytes;      // the factor block inside the workspace
    // prefetch slots 1 / 2: operand forms of a batch prepared ahead of its CD step (imdbn_cd_opts.next_data)
    bf16_t* pf_rm[2]; bf16_t* pf_tr[2]; int* pf_flags[2]; float* pf_cs[2];
    ChainRec* chain_recs;   // per-step schedule of the row-parallel chain kernel
    bf16_t* k4_planes; int64_t k4_plane_stride;      // fragment-ordered bf16 weight planes [2 directions][3 terms]
    int down_tr;            // visible rows per block of the fused K2 (<= 32): balances the row tiles over the CUs
    int k2s_tr;             // rows per k2_stream block: one tile per CU where the layer is large enough
    int k1s_tiles, k1s_ks, k1s_kchunk; int* k1s_cnt;      // k1_stream: 32-column tiles, K slices, arrival counters [Bp/64][tiles]
    size_t bytes;
};

int cu_count() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else return 256;
    }
    return cus;
}

// Fused K2 streams W once, one tile of `tr` visible rows per block, and every block costs the same; the kernel
// ends when the CU with the most rows ends (blocks are dealt round-robin).  10000 rows as 313 tiles of 32 put two
// tiles (64 rows) on 57 of the 256 CUs and one on the rest: the main loop ran 13 us at the median and 20 us on
// those 57.  20-row tiles (500 blocks, two per CU, 40 rows each) level it.  The MFMA tile stays 32 wide.
int plan_down_rows(int V) {
    if (g_down_tr > 0) return g_down_tr;
    const int cus = cu_count();
    int best = 32, best_cost = 1 << 30;
    for (int tr = 32; tr >= 16; tr -= 4) {
        const int cost = cdiv(cdiv(V, tr), cus) * tr;       // rows streamed by the busiest CU
        if (cost < best_cost) { best_cost = cost; best = tr; }
    }
    return best;
}

Layout make_layout(int V, int H, int B, char* base) {
    Layout L{};
    L.V = V; L.H = H; L.B = B;
    L.Bp = rup(std::max(B, 1), 64);
    L.Vpad = rup(V, 16); L.Hpad = rup(H, 16);
    L.P = L.Bp / 8;
    const int mb = L.Bp / 64;
    L.up4 = (H % 4 == 0) && H >= 4 && !g_no_fast_k1;      // float4 K1 (also needs 16-B aligned W: checked at launch)
    if (L.up4) {
        const int tiles = cdiv(H, 128) * mb;
        const int want = g_ks_up > 0 ? g_ks_up : std::max(1, 252 / std::max(1, tiles));
        L.up = plan_split(L.Vpad, cdiv(H, 128), mb, want, 64);
    } else {
        L.up = plan_split(L.Vpad, cdiv(H, 64), mb, g_ks_up, 64);
    }
    L.down = plan_split(L.Hpad, cdiv(V, 64), mb, g_ks_down, 16);
    L.down_tr = plan_down_rows(V);
    L.k2s_tr = g_k2s_tr > 0 ? g_k2s_tr : std::min(48, std::max(8, 8 * cdiv(V, 8 * cu_count())));
    size_t off = 0;
    auto take = [&](size_t nbytes) { char* p = base ? base + off : nullptr; off += (nbytes + 255) / 256 * 256; return p; };
    // exactness maps of caller-supplied operands (prep rewrites them every call): visible side, hidden side
    // ---- factor block: everything the weight / bias update needs from one CD pass, contiguous, so that the
    // data-parallel "factor exchange" can all-gather it in one piece (imdbn_factor_block): exactness map of the data,
    // hidden planes (pos, negated neg), column-sum and squared-error partials, visible planes (pos: 3 terms; neg: its
    // FIRST term only is inside the block -- the negative visible state of train_epoch is a sample, one term)
    L.fb_off = off;
    L.flags = (int*)take((size_t)L.P * cdiv(L.Vpad, 64) * 4);
    for (int i = 0; i < 2; ++i) L.hid_tr[i] = (bf16_t*)take((size_t)3 * H * L.Bp * 2);
    L.cs_hpos = (float*)take((size_t)L.P * H * 4);
    L.cs_hneg = (float*)take((size_t)L.P * H * 4);
    L.cs_vpos = (float*)take((size_t)L.P * V * 4);
    L.cs_vneg = (float*)take((size_t)L.P * V * 4);
    L.n_loss_slots = std::max(cdiv(std::max(V, H), 64) * L.P, (cdiv(V, 8) + 2) * (L.Bp / 64)) + IMDBN_MAX_GROUPS * (L.Bp / 64);
    L.loss_part = (float*)take((size_t)L.n_loss_slots * 4);
    for (int i = 0; i < 2; ++i) {
        L.vis_tr[i] = (bf16_t*)take((size_t)3 * V * L.Bp * 2);
        if (i == 1) L.fb_bytes = (off - (((size_t)3 * V * L.Bp * 2 + 255) / 256 * 256)) + ((size_t)V * L.Bp * 2 + 255) / 256 * 256 - L.fb_off;
    }
    // ---- the rest
    L.flags_h = (int*)take((size_t)L.P * cdiv(L.Hpad, 64) * 4);
    for (int i = 0; i < 2; ++i) L.vis_rm[i] = (bf16_t*)take((size_t)3 * L.Bp * L.Vpad * 2);
    L.hid_rm = (bf16_t*)take((size_t)3 * L.Bp * L.Hpad * 2);
    L.ldbits = 2 * cdiv(L.Hpad, 64);
    L.hid_bits = (uint8_t*)take((size_t)L.Bp * rup(H, 64) / 8);
    for (int i = 0; i < 2; ++i) L.vis_bits[i] = (uint8_t*)take((size_t)L.Bp * rup(V, 64) / 8);
    {   // k1_stream: ~one block per CU; a K slice is a multiple of 64 rows and at most K1S_MAX_KCHUNK (its bits sit in LDS)
        L.k1s_tiles = cdiv(H, 32);
        int ks = g_k1s_ks > 0 ? g_k1s_ks : std::max(1, (int)((double)cu_count() / (double)(L.k1s_tiles * mb) + 0.5));
        ks = std::min(ks, std::max(1, L.Vpad / 192));      // at least three K16 steps per wave and slice (1500 <-> 500: 8 slices of 192 rows, 45.5 us per update against 47.9 with 12 of 128)
        ks = std::max(ks, cdiv(L.Vpad, K1S_MAX_KCHUNK));
        L.k1s_kchunk = rup(cdiv(L.Vpad, ks), 64);
        L.k1s_ks = cdiv(L.Vpad, L.k1s_kchunk);
    }
    const size_t pf = std::max(std::max((size_t)L.up.ks * L.Bp * H, (size_t)L.down.ks * L.Bp * V), (size_t)L.k1s_ks * L.Bp * 32 * L.k1s_tiles);
    L.partial = (float*)take(pf * 4);
    L.k1s_cnt = (int*)take((size_t)mb * L.k1s_tiles * 4);
    L.f_h = (float*)take((size_t)L.Bp * H * 4);
    L.f_vp = (float*)take((size_t)L.Bp * V * 4);
    for (int i = 0; i < 2; ++i) L.f_v[i] = (float*)take((size_t)L.Bp * V * 4);
    L.chain_recs = (ChainRec*)take(sizeof(ChainRec) * CHAIN_MAX_STEPS);
    for (int i = 0; i < 2; ++i) {
        L.pf_rm[i] = (bf16_t*)take((size_t)3 * L.Bp * L.Vpad * 2);
        L.pf_tr[i] = (bf16_t*)take((size_t)3 * V * L.Bp * 2);
        L.pf_flags[i] = (int*)take((size_t)L.P * cdiv(L.Vpad, 64) * 4);
        L.pf_cs[i] = (float*)take((size_t)L.P * V * 4);
        L.pf_bits[i] = (uint8_t*)take((size_t)L.Bp * rup(V, 64) / 8);
    }
    L.k4_plane_stride = 0; L.k4_planes = nullptr;
    if (V <= 1024 && H <= 1024) {
        L.k4_plane_stride = (int64_t)std::max(cdiv(H, 16) * cdiv(V, 32), cdiv(V, 16) * cdiv(H, 32)) * 512;
        L.k4_planes = (bf16_t*)take((size_t)6 * L.k4_plane_stride * 2);
    }
    L.bytes = off;
    return L;
}

// ---- draw cursor -----------------------------------------------------------------------------
struct Rng {
    imdbn_rng* r;
    int64_t tpos = 0, cpos = 0;
    uint64_t draws = 0;
    bool bad = false;
    explicit Rng(imdbn_rng* r_) : r(r_) {}
    DrawSrc floats(int B, int N) {
        DrawSrc s{};
        s.N = N;
        if (!r) { bad = true; return s; }
        s.seed = r->seed; s.row0 = r->row0; s.draw = r->offset + draws; s.base = (const unsigned long long*)r->dev_offset;
        if (r->mode == IMDBN_RNG_REPLAY) {
            if (!r->tape || tpos + (int64_t)B * N > r->tape_len) { bad = true; return s; }
            s.tape = r->tape + tpos;
            tpos += (int64_t)B * N;
        }
        ++draws;
        return s;
    }
    void skip_floats(int B, int N) { (void)floats(B, N); }
    // categorical draws for all groups of one sample_visible call
    void cats(int B, int n_groups, const int32_t** tape, DrawSrc* uni) {
        *tape = nullptr;
        DrawSrc s{};
        s.N = 1;
        if (n_groups == 0) { *uni = s; return; }
        if (!r) { bad = true; *uni = s; return; }
        s.seed = r->seed; s.row0 = r->row0; s.draw = r->offset + draws; s.base = (const unsigned long long*)r->dev_offset;
        if (r->mode == IMDBN_RNG_REPLAY) {
            if (!r->cat_tape || cpos + (int64_t)B * n_groups > r->cat_len) { bad = true; *uni = s; return; }
            *tape = r->cat_tape + cpos;
            cpos += (int64_t)B * n_groups;
        }
        draws += n_groups;
        *uni = s;
    }
    int finish() {
        if (r) { r->tape_used = tpos; r->cat_used = cpos; r->draws_used = draws; }
        if (bad) return fail(IMDBN_E_RNG, "random draws requested but rng is null or the replay tape is exhausted");
        return 0;
    }
};

// ---- context -----------------------------------------------------------------------------------
struct Ctx {
    const imdbn_rbm_desc* d;
    Layout L;
    hipStream_t s;
    Rng rng;
    int nw;         // weight terms
    int rt;         // terms of a real-valued activation operand
    int ht;         // terms of the hidden-probability planes of the update kernel (K3) = rt.  Two terms (hi + mid, 16 bits) were tried in
                    // round 2: K3's staging and K1's plane stores shrink by a third, but the 2^-17 relative error of the statistics
                    // random-walks into the weights over thousands of small-batch updates (joint RBM, batch 8: ~1e-5 after 1500 updates)
                    // and flipped a sample of the train_joint fixture (margin 9e-6).  Exact products stay.
    bool hid_bits_ok = false;      // L.hid_bits describes the current contents of L.hid_rm
    bool data_prepped = false;     // cd_phases: the data-side operands are already in place (prefetch slot)
    int down_blocks = 0;           // blocks (per batch chunk) of the last K2 launch: the number of squared-error partials it left
    bool pos_phase = false;        // tuning aid: the propagation being launched is the positive phase of a CD pass (dbg bit 2048 stamps it, bit 64 the others)
    bool cnt_ok = false;           // the arrival counters of k1_stream are known to be zero: a launch of THIS call cleared them (prep / chain_init), or the
                                   // data-side operands come from a prefetch slot (an earlier call on this workspace ran, and every launch leaves them
                                   // zero).  Otherwise prop() clears them itself: a caller's workspace may hold anything
                                   // (tools/stress_chains.py on a NaN-filled workspace: chains of a layer wider than 1024 were all NaN)
    bool fix_slot = false;         // the data-side operands were written item by item (PrepArgs::adaptive): the next k1_stream that reads them
                                   // completes the planes of mixed spans for the update kernel (K1sArgs::fix_tr)
    Ctx(const imdbn_rbm_desc* d_, imdbn_rng* r, hipStream_t s_) : d(d_), s(s_), rng(r) {
        nw = d->mode == IMDBN_FAST_BF16 ? 1 : 3;
        rt = nw;
        ht = rt;
    }
};

int check_desc(const imdbn_rbm_desc* d, bool need_momentum) {
    if (!d) return fail(IMDBN_E_INVALID, "null descriptor");
    if (d->V <= 0 || d->H <= 0) return fail(IMDBN_E_INVALID, "bad shape V=%d H=%d", d->V, d->H);
    if (!d->W || !d->hid_bias || !d->vis_bias) return fail(IMDBN_E_INVALID, "null parameter pointer");
    if (d->ldw < d->H) return fail(IMDBN_E_INVALID, "ldw %lld < H %d", (long long)d->ldw, d->H);
    if (need_momentum && (!d->W_m || !d->hb_m || !d->vb_m)) return fail(IMDBN_E_INVALID, "null momentum buffer");
    if (d->n_groups < 0 || d->n_groups > IMDBN_MAX_GROUPS)
        return fail(IMDBN_E_UNSUPPORTED, "at most %d softmax groups supported, got %d", IMDBN_MAX_GROUPS, d->n_groups);
    for (int g = 0; g < d->n_groups; ++g) {
        if (d->group_start[g] < 0 || d->group_end[g] > d->V || d->group_start[g] >= d->group_end[g])
            return fail(IMDBN_E_INVALID, "softmax group %d = (%d,%d) outside [0,%d)", g, d->group_start[g], d->group_end[g], d->V);
        if (d->group_end[g] - d->group_start[g] > GROUP_WMAX)
            return fail(IMDBN_E_UNSUPPORTED, "softmax group %d is %d wide; at most %d supported", g, d->group_end[g] - d->group_start[g], GROUP_WMAX);
        for (int k = 0; k < g; ++k)
            if (d->group_start[g] < d->group_end[k] && d->group_start[k] < d->group_end[g])
                return fail(IMDBN_E_UNSUPPORTED, "overlapping softmax groups %d and %d", k, g);
    }
    if (d->mode != IMDBN_PARITY_F32 && d->mode != IMDBN_FAST_BF16) return fail(IMDBN_E_INVALID, "bad mode %d", d->mode);
    return 0;
}

// the CD step reads its data-side operands from prefetch slot `slot` (1 / 2) instead of the default buffers
void use_slot(Layout& L, int slot) {
    if (slot < 1 || slot > 2) return;
    std::swap(L.vis_rm[0], L.pf_rm[slot - 1]);
    std::swap(L.vis_tr[0], L.pf_tr[slot - 1]);
    std::swap(L.flags, L.pf_flags[slot - 1]);
    std::swap(L.cs_vpos, L.pf_cs[slot - 1]);
    std::swap(L.vis_bits[0], L.pf_bits[slot - 1]);
}

int setup(Ctx& c, int B, void* ws, size_t ws_bytes) {
    if (B <= 0) return fail(IMDBN_E_INVALID, "batch %d", B);
    if (!ws) return fail(IMDBN_E_WORKSPACE, "null workspace");
    if (((uintptr_t)ws & 255) != 0) return fail(IMDBN_E_INVALID, "workspace must be 256-byte aligned");
    c.L = make_layout(c.d->V, c.d->H, B, (char*)ws);
    if (c.L.bytes > ws_bytes)
        return fail(IMDBN_E_WORKSPACE, "workspace %zu < %zu bytes needed for V=%d H=%d B=%d", ws_bytes, c.L.bytes, c.d->V, c.d->H, B);
    return 0;
}

// An activation operand in row-major form: pointer + static term count (0 = look at flag)
// bits / binary: the operand also exists as a bit plane; binary = 1: it is 0/1 by construction (a sample), 2: the caller
// says so (checked on the device against the exactness map `flag`), 3: nobody knows -- the streaming K1 decides per
// 64-column item from the exactness map (bit plane where the item is all 0/1, the bf16 terms in `rm` elsewhere)
struct OpIn { const bf16_t* rm; int terms; const int* flag; const uint8_t* bits = nullptr; int binary = 0; };

// visible tiles (128 rows) per block of the streaming update kernel
int k3_tiles_per_block(int V, int H) {
    const int nh = cdiv(H, 128), nv = cdiv(V, 128);
    return std::max(1, cdiv(nh * nv, std::max(cu_count(), 1)));
}
// The per-item choice of k1_stream (K1S_ADAPTIVE) holds one exactness-map entry per thread for a K slice's items (<= 32) and one
// for a span of the update kernel (<= 256 entries); layers outside that read data of unknown content from the bf16 terms throughout
// (same numbers, all three-term forms prepared).
bool adaptive_shape_ok(const Layout& L) {
    // ... and one block of the (first batch chunk of the) positive-phase K1 per span of the update kernel for the fix-up of mixed
    // spans: a narrow hidden layer has too few (found by tools/stress_parity.py: 1576 x 12, 1437 x 32, 2116 x 28)
    const int tpb = k3_tiles_per_block(L.V, L.H);
    return L.k1s_kchunk <= 2048 && 2 * tpb * L.P <= 256 && cdiv(cdiv(L.V, 128), tpb) <= L.k1s_tiles * L.k1s_ks;
}

bool vec4_weights(const imdbn_rbm_desc* d) {
    return d->H % 4 == 0 && d->H >= 4 && d->ldw % 4 == 0 && (((uintptr_t)d->W) & 15) == 0;
}

void base_finish_args(Ctx& c, bool up, FinishArgs& f) {
    const Layout& L = c.L;
    f.partial = L.partial;
    f.ks = up ? L.up.ks : L.down.ks;
    f.B = L.B; f.Bp = L.Bp;
    f.N = up ? L.H : L.V;
    f.slab = (int64_t)L.Bp * f.N;
    f.bias = up ? c.d->hid_bias : c.d->vis_bias;
    f.n_groups = up ? 0 : c.d->n_groups;
    for (int g = 0; g < IMDBN_MAX_GROUPS; ++g) { f.gs[g] = up ? 0 : c.d->group_start[g]; f.ge[g] = up ? 0 : c.d->group_end[g]; }
    f.op.ldrm = up ? L.Hpad : L.Vpad;
    f.op.rm_ts = (int64_t)L.Bp * f.op.ldrm;
    f.op.Bp = L.Bp;
    f.op.tr_ts = (int64_t)f.N * L.Bp;
}

FinishArgs new_finish() {
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    f.T = 1.0f;
    return f;
}

// one propagation: partial GEMM + finish (+ group kernel)
int prop(Ctx& c, bool up, OpIn in, FinishArgs f, const PrepArgs* next = nullptr) {
    const Layout& L = c.L;
    const imdbn_rbm_desc* d = c.d;
    base_finish_args(c, up, f);
    if (f.T < 1e-6f) f.T = 1e-6f;                           // max(1e-6, T)  rbm.py:92,96
    // lean epilogue specialisation (kernels_ew.hpp finish_rows_impl<R, false>)
    f.simple = (f.T == 1.0f && !(f.sigma > 0.f) && !f.mu && !f.clamp && f.n_groups == 0 && !f.logits_only) ? 1 : 0;
    // lean epilogue of the streaming kernels (kernels_ew.hpp finish_lean8): decided once all outputs are known (below)
    auto lean_ok = [&](const FinishArgs& g) {
        return g.simple && (g.vmode == 0 || g.vmode == 1) && !g.out_final && !(g.rm_src && g.op.rm) &&
               (g.vmode == 0 || g.uni.tape || (g.uni.row0 & 3) == 0) ? 1 : 0;
    };
    const int mb = L.Bp / 64;
    // hidden samples (exactly 0/1, not mixed with clamped values) also leave as a bit plane for the fused K2
    const bool want_hbits = up && f.op.rm == L.hid_rm && f.rm_src == 2 && f.vmode == 1 && !f.clamp && f.n_groups == 0 && !f.logits_only;
    if (up && L.Vpad <= 1024 && !g_no_fused_up) {
        // short K: fused GEMM + epilogue, no split-K slabs (one launch per half step of a chain)
        dim3 grid(cdiv(L.H, 32), 1, mb);
        const int64_t ats = (int64_t)L.Bp * L.Vpad;
        f.dbg = 0;
        f.op.bits = want_hbits ? L.hid_bits : nullptr; f.op.bits_shape = 1; f.op.bits_cols = 32;      // epilogue lanes: 32 columns x 2 row octets
        if (f.op.rm == L.hid_rm) c.hid_bits_ok = want_hbits;
        if (c.nw == 3)
            hipLaunchKernelGGL(gemm_up_fused<3>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, f);
        else
            hipLaunchKernelGGL(gemm_up_fused<1>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, f);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // streaming K1: weights through LDS by LDS-DMA, split-K combined by the last arriver, epilogue fused.  The operand is a bit
    // plane (0/1 by construction or by the caller's word), bf16 terms (real values), or either per 64-column item (unknown)
    const bool k1s_bits = in.bits && (in.binary == 1 || in.binary == 2);
    const bool k1s_real = !k1s_bits && in.rm && !g_no_k1s_real && (in.binary == 0 || (in.binary == 3 && in.bits && in.flag));
    if (up && (k1s_bits || k1s_real) && vec4_weights(d) && !g_no_k1s && !f.logits_only) {
        if (!c.cnt_ok) {
            HIPCHK(hipMemsetAsync(L.k1s_cnt, 0, (size_t)mb * L.k1s_tiles * sizeof(int), c.s));
            c.cnt_ok = true;
        }
        K1sArgs a;
        memset(&a, 0, sizeof(a));
        a.W = d->W; a.ldw = d->ldw; a.K = L.V; a.N = L.H;
        a.abits = in.bits; a.Bp = L.Bp;
        a.aflag = in.binary >= 2 ? in.flag : nullptr; a.ncb = cdiv(L.Vpad, 64); a.P = L.P;
        a.slabs = L.partial; a.counters = L.k1s_cnt; a.kchunk = L.k1s_kchunk; a.ks = L.k1s_ks;
        a.amode = k1s_bits ? (in.binary == 2 ? K1S_ASSERTED : K1S_BITS) : ((in.binary == 3 && adaptive_shape_ok(L)) ? K1S_ADAPTIVE : K1S_REAL);
        a.arm = in.rm; a.arm_ts = (int64_t)L.Bp * L.Vpad;
        if (a.amode == K1S_ADAPTIVE && c.fix_slot && in.rm == L.vis_rm[0]) {
            const int tpb = k3_tiles_per_block(L.V, L.H);
            a.fix_tr = L.vis_tr[0]; a.fix_ts = (int64_t)L.V * L.Bp; a.fix_span = 2 * tpb; a.fix_ranges = cdiv(cdiv(L.V, 128), tpb);
            if (a.fix_ranges > L.k1s_tiles * a.ks) return fail(IMDBN_E_INVALID, "internal: k1_stream fix-up ranges");
            c.fix_slot = false;
        }
        // terms the kernel multiplies per element: the operand form carries `in.terms` of them (0 = prep's three, nw in FAST mode)
        const int na = k1s_bits ? ((g_k1s_force_na && !next) ? c.nw : 0) : ((in.terms == 1 || c.nw == 1) ? 1 : 3);
        f.dbg = c.pos_phase ? ((g_dbg & 2048) ? 64 : 0) : (g_dbg & ~2048);
        f.op.bits = want_hbits ? L.hid_bits : nullptr; f.op.bits_shape = 1; f.op.bits_cols = 32;
        if (f.op.rm == L.hid_rm) c.hid_bits_ok = want_hbits;
        if (want_hbits && !g_no_bits) f.op.rm = nullptr, f.rm_src = 0;      // the fused K2 reads the bit plane, nobody reads the bf16 form
        // 80 KB at the headline shape: two workgroups per CU (the bit-plane kernel carries the next batch's preparation blocks)
        a.region = k1s_bits ? K1S_RING : K1S_REGION_REAL;
        const size_t lds = (size_t)K1S_WAVES * a.region + (size_t)8 * a.kchunk + (k1s_bits ? (next ? 0 : g_k1s_lds_pad) : K1S_LDS_EXTRA);
        f.lean = lean_ok(f);
        // + block rows that prepare the next batch (one 64-column item each, or a few)
        PrepArgs pz;
        memset(&pz, 0, sizeof(pz));
        const int items = next ? cdiv(std::max(next->N, next->op.ldrm), 64) : 0;
        const int pr = next ? std::min(8, cdiv(items, L.k1s_tiles)) : 0;
        if (pr > 0 && na != 0) return fail(IMDBN_E_INVALID, "internal: preparation blocks ride on the bit-plane k1_stream only");
        dim3 grid(L.k1s_tiles, a.ks + pr, mb);
        const PrepArgs& pa = next ? *next : pz;
        hipError_t le = hipSuccess;
#define LAUNCH_K1S(NWV, NAV, GEV, RV) do { \
        static bool attr = false; \
        if (!attr) { le = hipFuncSetAttribute((const void*)k1_stream<NWV, NAV, GEV, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, K1S_WAVES * K1S_REGION_REAL + 8 * K1S_MAX_KCHUNK + K1S_LDS_EXTRA); attr = true; } \
        if (le == hipSuccess) hipLaunchKernelGGL((k1_stream<NWV, NAV, GEV, RV>), grid, dim3(64 * K1S_WAVES), lds, c.s, a, f, pa); } while (0)
#define LAUNCH_K1S_G(NWV, NAV, RV) do { if (f.lean) LAUNCH_K1S(NWV, NAV, false, RV); else LAUNCH_K1S(NWV, NAV, true, RV); } while (0)
        // (one instantiation per case: code that a launch does not run -- the general epilogue, the preparation blocks, the loop over
        //  bf16 terms -- still costs it time)
        if (c.nw == 3) { if (na == 0) { if (pr > 0) LAUNCH_K1S_G(3, 0, true); else LAUNCH_K1S_G(3, 0, false); } else if (na == 1) LAUNCH_K1S_G(3, 1, false); else LAUNCH_K1S_G(3, 3, false); }
        else           { if (na == 0) { if (pr > 0) LAUNCH_K1S_G(1, 0, true); else LAUNCH_K1S_G(1, 0, false); } else LAUNCH_K1S_G(1, 1, false); }
#undef LAUNCH_K1S_G
#undef LAUNCH_K1S
        HIPCHK(le);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (up) {
        const int64_t ats = (int64_t)L.Bp * L.Vpad;
        const bool fast = L.up4 && d->ldw % 4 == 0 && (((uintptr_t)d->W) & 15) == 0;
        if (fast) {
            dim3 grid(cdiv(L.H, 128), L.up.ks, mb);
            if (c.nw == 3)
                hipLaunchKernelGGL(gemm_up4_partial<3>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, L.partial, L.Bp, L.up.kchunk, g_dbg >> 4);
            else
                hipLaunchKernelGGL(gemm_up4_partial<1>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, L.partial, L.Bp, L.up.kchunk, g_dbg >> 4);
        } else {
            dim3 grid(cdiv(L.H, 64), L.up.ks, mb);
            if (c.nw == 3)
                hipLaunchKernelGGL(gemm_up_partial<3>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, L.partial, L.Bp, L.up.kchunk);
            else
                hipLaunchKernelGGL(gemm_up_partial<1>, grid, dim3(256), 0, c.s, d->W, d->ldw, L.V, L.H, in.rm, ats, L.Vpad, in.flag, in.terms, L.partial, L.Bp, L.up.kchunk);
        }
    } else {
        // K2: fused GEMM + epilogue (no split-K slabs)
        if (f.n_groups > 0 && !f.logits_only && !f.out_prob) f.out_prob = L.f_vp, f.ld_prob = L.V;
        if (f.n_groups > 0 && !f.logits_only && !f.out_final) f.out_final = L.f_v[1], f.ld_final = L.V;
        const uint8_t* hb = (c.hid_bits_ok && in.rm == L.hid_rm && in.terms == 1 && !g_no_bits) ? L.hid_bits : nullptr;
        if (hb && vec4_weights(d) && !g_no_k2s) {
            // 0/1 hidden operand as a bit plane: one tile of k2s_tr rows per CU, 16x16x32 MFMA (kernels_stream.hpp)
            K2sArgs a;
            memset(&a, 0, sizeof(a));
            a.W = d->W; a.ldw = d->ldw; a.K = L.H; a.N = L.V; a.abits = hb; a.Bp = L.Bp; a.TR = L.k2s_tr;
            const int nbx = cdiv(L.Vpad, a.TR), MT = cdiv(a.TR, 16);
            c.down_blocks = nbx;
            if ((nbx + IMDBN_MAX_GROUPS) * mb > L.n_loss_slots) return fail(IMDBN_E_INVALID, "internal: loss slots");
            f.dbg = g_dbg;
            if (f.op.bits && (f.n_groups > 0 || f.vmode == 0)) f.op.bits = nullptr;      // not a pure 0/1 sample
            f.lean = lean_ok(f);
            if (next) return fail(IMDBN_E_INVALID, "internal: k2_stream carries no next-batch blocks");
            const int nsteps = cdiv(L.H, 32);
            const size_t lds = (size_t)((nsteps * 256 + 255) & ~255) + (size_t)std::max(K2S_LW * 16 * MT * K2S_LDR * 4, 64);
            dim3 grid(nbx, 1, mb);
            hipError_t le = hipSuccess;
#define LAUNCH_K2S(NW, MTV, GEV) do { \
        static bool attr = false; \
        if (!attr) { le = hipFuncSetAttribute((const void*)k2_stream<NW, MTV, GEV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        if (le == hipSuccess) hipLaunchKernelGGL((k2_stream<NW, MTV, GEV>), grid, dim3(64 * K2S_W), lds, c.s, a, f); } while (0)
#define LAUNCH_K2S_G(NW, MTV) do { if (f.lean) LAUNCH_K2S(NW, MTV, false); else LAUNCH_K2S(NW, MTV, true); } while (0)
#define LAUNCH_K2S_M(NW) do { if (MT == 1) LAUNCH_K2S_G(NW, 1); else if (MT == 2) LAUNCH_K2S_G(NW, 2); else LAUNCH_K2S_G(NW, 3); } while (0)
            if (lds > 160 * 1024) return fail(IMDBN_E_UNSUPPORTED, "internal: k2_stream LDS");
            if (c.nw == 3) LAUNCH_K2S_M(3); else LAUNCH_K2S_M(1);
#undef LAUNCH_K2S_M
#undef LAUNCH_K2S_G
#undef LAUNCH_K2S
            HIPCHK(le);
            HIPCHK(hipGetLastError());
            if (f.n_groups > 0 && !f.logits_only) {
                hipLaunchKernelGGL(finish_groups, dim3(f.n_groups, L.Bp / 64), dim3(256), 0, c.s, f, (int)(nbx * mb));
                HIPCHK(hipGetLastError());
            }
            return 0;
        }
        // tiles cover [0, Vpad): the K16-blocked operand form must have its padding columns [V, Vpad) written (zeros) -- the
        // next K1 multiplies them with clamped (non-zero) weight rows.  Tiles of 20 / 24 / 28 rows (chosen for V in
        // (4096, 7168]) do not end on a multiple of 16 by themselves; found by tools/stress_parity.py.
        // sampled hidden states left by `finish` in bit-packed form: 16x less activation traffic per block
        const uint8_t* abits = (c.hid_bits_ok && in.rm == L.hid_rm && in.terms == 1 && !g_no_bits) ? L.hid_bits : nullptr;
        const int64_t ats = (int64_t)L.Bp * L.Hpad;
        const bool vec4 = (d->ldw % 4 == 0) && (((uintptr_t)d->W & 15) == 0) && (L.H % 4 == 0) && L.H >= 4;
        // several 64-row batch chunks: one block per weight tile takes 2 or 4 of them, a wave (pair) per chunk, on full 32-row
        // tiles (decode / visible_probs of a 256-row batch: 4 x 500 blocks of 20 rows in four rounds -> 313 blocks in one)
        const bool multi = mb >= 2 && !abits && !next && vec4 && !(f.rm_src && f.op.rm) && L.Vpad >= 128 * 32;      // (fewer than 128 weight tiles: the per-chunk grid fills the chip better)
        const int mbb = (multi && !g_no_down_chunks) ? (mb % 4 == 0 ? 4 : (mb % 2 == 0 ? 2 : 1)) : 1;
        const int down_tr = mbb > 1 ? 32 : L.down_tr;
        dim3 grid(cdiv(L.Vpad, down_tr), 1, mb / mbb);
        c.down_blocks = (int)grid.x;
        f.dbg = g_dbg;
        if (f.op.bits && (down_tr % 8 != 0 || f.n_groups > 0 || f.vmode == 0)) f.op.bits = nullptr;      // not a 0/1 plane the epilogue can write byte-wise
        f.op.bits_shape = 1; f.op.bits_cols = down_tr;
        // ... and, where the squared-error partials of 32-row tiles fit, the LDS-tiled kernel: 128 weight rows x 64 batch rows per block,
        // the activation terms staged once per block (kernels_gemm.hpp gemm_down_tiled)
        if (multi && !g_no_down_tiled && !g_no_down_chunks && (4 * cdiv(L.Vpad, 128) + IMDBN_MAX_GROUPS) * mb <= L.n_loss_slots) {
            dim3 gt(cdiv(L.Vpad, 128), 1, mb);
            c.down_blocks = 4 * (int)gt.x;
            f.op.bits_cols = 32;
            if (c.nw == 3) hipLaunchKernelGGL((gemm_down_tiled<3>), gt, dim3(256), 0, c.s, d->W, d->ldw, L.H, L.V, in.rm, ats, L.Hpad, in.flag, in.terms, f);
            else           hipLaunchKernelGGL((gemm_down_tiled<1>), gt, dim3(256), 0, c.s, d->W, d->ldw, L.H, L.V, in.rm, ats, L.Hpad, in.flag, in.terms, f);
            HIPCHK(hipGetLastError());
            if (f.n_groups > 0 && !f.logits_only) {
                hipLaunchKernelGGL(finish_groups, dim3(f.n_groups, L.Bp / 64), dim3(256), 0, c.s, f, (int)(c.down_blocks * mb));
                HIPCHK(hipGetLastError());
            }
            return 0;
        }
        if ((int)((grid.x + IMDBN_MAX_GROUPS) * mb) > L.n_loss_slots) return fail(IMDBN_E_INVALID, "internal: loss slots");
#define LAUNCH_DOWN_M(NW, MBBV) \
    hipLaunchKernelGGL((gemm_down_fused<NW, true, 0, false, MBBV>), grid, dim3(256), 0, c.s, d->W, d->ldw, L.H, L.V, in.rm, ats, L.Hpad, in.flag, in.terms, f, down_tr, abits, L.ldbits)
        if (mbb > 1) {
            if (c.nw == 3) { if (mbb == 4) LAUNCH_DOWN_M(3, 4); else LAUNCH_DOWN_M(3, 2); }
            else           { if (mbb == 4) LAUNCH_DOWN_M(1, 4); else LAUNCH_DOWN_M(1, 2); }
        } else
#define LAUNCH_DOWN(NW, V4, NAK, BITS) \
    hipLaunchKernelGGL((gemm_down_fused<NW, V4, NAK, BITS>), grid, dim3(256), 0, c.s, d->W, d->ldw, L.H, L.V, in.rm, ats, L.Hpad, in.flag, in.terms, f, down_tr, abits, L.ldbits)
#define LAUNCH_DOWN_A(NW, V4) \
    do { if (abits) LAUNCH_DOWN(NW, V4, 1, true); else if (in.terms == 1) LAUNCH_DOWN(NW, V4, 1, false); \
         else if (in.terms == 3) LAUNCH_DOWN(NW, V4, 3, false); else LAUNCH_DOWN(NW, V4, 0, false); } while (0)
        if (next) {
            // + one block per (64-column tile, batch chunk) of the NEXT batch behind the weight tiles (prep_item_body)
            if (!vec4 || in.terms != 1) return fail(IMDBN_E_INVALID, "internal: next-batch prep on an ineligible K2");
            const int main_nbx = (int)grid.x;
            dim3 gn(grid.x + cdiv(std::max(next->N, next->op.ldrm), 64), 1, mb);
#define LAUNCH_DOWN_N(NW, BITS) \
    hipLaunchKernelGGL((gemm_down_fused_next<NW, BITS>), gn, dim3(256), 0, c.s, d->W, d->ldw, L.H, L.V, in.rm, ats, L.Hpad, f, down_tr, abits, L.ldbits, *next, main_nbx)
            if (c.nw == 3) { if (abits) LAUNCH_DOWN_N(3, true); else LAUNCH_DOWN_N(3, false); }
            else           { if (abits) LAUNCH_DOWN_N(1, true); else LAUNCH_DOWN_N(1, false); }
#undef LAUNCH_DOWN_N
        }
        else if (c.nw == 3) { if (vec4) LAUNCH_DOWN_A(3, true); else LAUNCH_DOWN_A(3, false); }
        else                { if (vec4) LAUNCH_DOWN_A(1, true); else LAUNCH_DOWN_A(1, false); }
#undef LAUNCH_DOWN_A
#undef LAUNCH_DOWN
        HIPCHK(hipGetLastError());
        if (f.n_groups > 0 && !f.logits_only) {
            hipLaunchKernelGGL(finish_groups, dim3(f.n_groups, L.Bp / 64), dim3(256), 0, c.s, f, (int)(grid.x * mb));
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    HIPCHK(hipGetLastError());
    if (f.n_groups > 0 && !f.logits_only && !f.out_prob) f.out_prob = L.f_vp, f.ld_prob = L.V;
    if (f.n_groups > 0 && !f.logits_only && !f.out_final) f.out_final = L.f_v[1], f.ld_final = L.V;
    dim3 fgrid(cdiv(f.N, 64), L.P);
    if ((int)(fgrid.x * fgrid.y) + IMDBN_MAX_GROUPS * (L.Bp / 64) > L.n_loss_slots) return fail(IMDBN_E_INVALID, "internal: loss slots");
    f.dbg = g_dbg;
    f.op.bits = want_hbits ? L.hid_bits : nullptr; f.op.bits_shape = 0;
    if (up && f.op.rm == L.hid_rm) c.hid_bits_ok = want_hbits;
    hipLaunchKernelGGL(finish, fgrid, dim3(256), 0, c.s, f);
    HIPCHK(hipGetLastError());
    if (f.n_groups > 0 && !f.logits_only) {
        hipLaunchKernelGGL(finish_groups, dim3(f.n_groups, L.Bp / 64), dim3(256), 0, c.s, f, (int)(fgrid.x * fgrid.y));
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// k2_stream runs the K2 of a CD pass (its hidden operand is always a sample with a bit plane) whenever the weight rows allow
bool k2s_for_cd(const imdbn_rbm_desc* d) { return vec4_weights(d) && !g_no_k2s && !g_no_bits; }
int n_loss_used(const Ctx& c, bool up) {
    if (up) return cdiv(c.L.H, 64) * c.L.P;
    const int blocks = c.down_blocks > 0 ? c.down_blocks : cdiv(c.L.Vpad, k2s_for_cd(c.d) ? c.L.k2s_tr : c.L.down_tr);
    return (blocks + c.d->n_groups) * (c.L.Bp / 64);    // fused K2: one partial per block (+ per group block)
}

// caller fp32 tensor -> operand forms in the workspace
int prep(Ctx& c, const float* in, int64_t ld, int N, bf16_t* rm, int ldrm, bf16_t* tr, int* flag,
         float* colsum = nullptr, int terms = 3, uint8_t* bits = nullptr) {
    PrepArgs p;
    memset(&p, 0, sizeof(p));
    p.op.bits = bits; p.op.bits_shape = 0;
    p.zero = c.L.k1s_cnt; p.n_zero = (c.L.Bp / 64) * c.L.k1s_tiles;      // first launch of a call: arrival counters of k1_stream
    c.cnt_ok = true;
    p.in = in; p.ld = ld; p.B = c.L.B; p.Bp = c.L.Bp; p.N = N;
    p.op.rm = rm; p.op.ldrm = ldrm; p.op.rm_ts = (int64_t)c.L.Bp * ldrm; p.op.rm_terms = rm ? terms : 0; p.op.Bp = c.L.Bp;
    p.op.tr = tr; p.op.tr_ts = (int64_t)N * c.L.Bp; p.op.tr_terms = terms;
    p.flag = flag;
    p.colsum_part = colsum;
    hipLaunchKernelGGL(prep_operand, dim3(cdiv(std::max(N, ldrm), 64), c.L.P), dim3(256), 0, c.s, p);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_bias(Ctx& c, const BiasArgs& b);

// Next-batch preparation rides on the fused K2 of the first negative-phase step (gemm_down_fused_next: float4 weight
// rows, single-term hidden activations -- what imdbn_rbm_cd_step always launches when the weight rows are aligned).
bool prefetch_available(const imdbn_rbm_desc* d) {
    return !g_no_prefetch && d->H % 4 == 0 && d->H >= 4 && d->ldw % 4 == 0 && (((uintptr_t)d->W) & 15) == 0;
}

int launch_assoc(Ctx& c, int mode_stats, const imdbn_cd_opts* o, int vpos_terms, const int* vpos_flag, int vneg_terms,
                 float n, float* delta, const BiasArgs* bias = nullptr) {
    const Layout& L = c.L;
    AssocArgs a;
    memset(&a, 0, sizeof(a));
    a.W = c.d->W; a.Wm = c.d->W_m; a.ldw = c.d->ldw; a.V = L.V; a.H = L.H;
    a.vpos = L.vis_tr[0]; a.vpos_flag = vpos_flag; a.vpos_terms = vpos_terms;
    a.hpos = L.hid_tr[0]; a.hpos_terms = c.ht;
    a.vneg = L.vis_tr[1]; a.vneg_flag = vpos_flag; a.vneg_terms = vneg_terms;
    a.hneg = L.hid_tr[1]; a.hneg_terms = c.ht;
    a.vts = (int64_t)L.V * L.Bp; a.hts = (int64_t)L.H * L.Bp; a.Bp = L.Bp;
    a.lr = o->lr; a.mom = o->momentum; a.wd = o->weight_decay; a.n = n;
    a.delta = delta;
    const bool prof = g_prof.on && !mode_stats && (g_prof.calls++ % 8 == 3) && g_prof.used + 2 <= g_prof.ev.size();
    if (prof) HIPCHK(hipEventRecord(g_prof.ev[g_prof.used], c.s));
    // fast path: every W / W_m / delta row start 16-B aligned -> float4 weight tiles, LDS-staged planes
    const bool fast = L.H % 4 == 0 && L.H >= 4 && c.d->ldw % 4 == 0 && (((uintptr_t)c.d->W) & 15) == 0 &&
                      (mode_stats ? ((((uintptr_t)delta) & 15) == 0) : ((((uintptr_t)c.d->W_m) & 15) == 0)) && !g_no_fast_k3;
    if (fast) {
        AssocPlanesArgs f;
        memset(&f, 0, sizeof(f));
        f.W = a.W; f.Wm = a.Wm; f.ldw = a.ldw; f.V = a.V; f.H = a.H;
        f.vpos = a.vpos; f.vpos_flag = a.vpos_flag; f.vpos_terms = a.vpos_terms;
        f.hpos = a.hpos; f.vneg = a.vneg; f.vneg_terms = a.vneg_terms; f.hneg = a.hneg;
        f.vts = a.vts; f.hts = a.hts; f.Bp = a.Bp;
        f.lr = a.lr; f.mom = a.mom; f.wd = a.wd; f.n = a.n; f.delta = a.delta; f.dbg = (g_dbg & 512) ? 1 : 0;
        // ~one block per CU: each block streams `tpb` visible tiles with its hidden planes resident in LDS
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int nh = cdiv(L.H, 128), nv = cdiv(L.V, 128);
        const int tpb = std::max(1, cdiv(nh * nv, std::max(cus, 1)));
        // + one extra block row for the bias / loss update when the caller wants it fused
        const int brows = bias ? (nh >= 2 ? 1 : 2) : 0;     // >= 2 blocks: one reduces the loss, the rest stride the biases
        BiasArgs bz;
        memset(&bz, 0, sizeof(bz));
        const BiasArgs& bb = bias ? *bias : bz;
        // one launch per 64-row batch chunk (kernels_gemm.hpp AssocPlanesArgs::pass); the bias / loss blocks ride on the last
        const int nchunk = L.Bp / 64;
        for (int ch = 0; ch < nchunk; ++ch) {
            f.b0 = 64 * ch;
            const int pass = nchunk == 1 ? 0 : (ch == 0 ? 1 : (ch == nchunk - 1 ? 3 : 2));
            const int br = (ch == nchunk - 1) ? brows : 0;
            dim3 g(nh, cdiv(nv, tpb) + br);
#define LAUNCH_K3(M, HTV, PS) hipLaunchKernelGGL((assoc_update_planes<M, HTV, PS>), g, dim3(256), 0, c.s, f, tpb, bb, br)
#define LAUNCH_K3_P(M, HTV) do { if (pass == 0) LAUNCH_K3(M, HTV, 0); else if (pass == 1) LAUNCH_K3(M, HTV, 1); \
                                 else if (pass == 2) LAUNCH_K3(M, HTV, 2); else LAUNCH_K3(M, HTV, 3); } while (0)
            if (c.ht == 3) { if (mode_stats) LAUNCH_K3_P(1, 3); else LAUNCH_K3_P(0, 3); }
            else           { if (mode_stats) LAUNCH_K3_P(1, 1); else LAUNCH_K3_P(0, 1); }
#undef LAUNCH_K3_P
#undef LAUNCH_K3
        }
        HIPCHK(hipGetLastError());
        if (prof) { HIPCHK(hipEventRecord(g_prof.ev[g_prof.used + 1], c.s)); g_prof.used += 2; }
        return 0;
    } else {
        dim3 grid(cdiv(L.H, 128), cdiv(L.V, 64));
        if (c.ht == 3) {
            if (mode_stats) hipLaunchKernelGGL((assoc_update<1, 3>), grid, dim3(256), 0, c.s, a);
            else            hipLaunchKernelGGL((assoc_update<0, 3>), grid, dim3(256), 0, c.s, a);
        } else {
            if (mode_stats) hipLaunchKernelGGL((assoc_update<1, 1>), grid, dim3(256), 0, c.s, a);
            else            hipLaunchKernelGGL((assoc_update<0, 1>), grid, dim3(256), 0, c.s, a);
        }
    }
    HIPCHK(hipGetLastError());
    if (prof) { HIPCHK(hipEventRecord(g_prof.ev[g_prof.used + 1], c.s)); g_prof.used += 2; }
    if (bias) CHK(launch_bias(c, *bias));          // generic K3: bias update as its own launch
    return 0;
}

// imdbn_cd_opts.data_binary (0 unknown, 1 asserted 0/1, 2 real) -> OpIn::binary of the data operand
int data_operand_kind(int data_binary) { return data_binary == IMDBN_DATA_BINARY ? 2 : (data_binary == IMDBN_DATA_UNKNOWN ? 3 : 0); }

// rbm.py:199-209: positive phase, CD-k Gibbs, statistics left in the workspace operand buffers.
int cd_phases(Ctx& c, const float* data, int64_t ldd, const imdbn_cd_opts* o, const PrepArgs* next = nullptr) {
    const Layout& L = c.L;
    const int B = L.B;
    if (o->cd_k < 1) return fail(IMDBN_E_INVALID, "CD=%d (the reference needs CD>=1, rbm.py:204-209)", o->cd_k);
    if (!c.data_prepped) CHK(prep(c, data, ldd, L.V, L.vis_rm[0], L.Vpad, L.vis_tr[0], L.flags, L.cs_vpos, 3, L.vis_bits[0]));
    // the visible sample of the negative phase leaves the fused K2 as a bit plane too when its tiles are whole bytes wide
    const bool vbits = c.d->n_groups == 0 && (k2s_for_cd(c.d) || L.down_tr % 8 == 0);
    const bool k1s_neg = vec4_weights(c.d) && !g_no_k1s && L.Vpad > 1024;      // the negative-phase K1 will be k1_stream (prop())
    // the next batch's preparation rides on the old fused K2 (gemm_down_fused_next) or, with k2_stream (a 512-thread block per
    // CU: nothing fits beside it), on the negative-phase k1_stream
    const bool next_on_k1 = next && k2s_for_cd(c.d) && vbits && k1s_neg;
    // positive phase: P+ = up(data); h = 1[P+ > U]
    {
        FinishArgs f = new_finish();
        f.vmode = 1; f.uni = c.rng.floats(B, L.H);
        f.op.rm = L.hid_rm; f.op.rm_terms = 1; f.rm_src = 2;
        f.op.tr = L.hid_tr[0]; f.op.tr_terms = c.ht; f.tr_src = 1;
        f.colsum_part = L.cs_hpos; f.colsum_src = 1;
        c.pos_phase = true;
        const int rc = prop(c, true, OpIn{L.vis_rm[0], c.nw == 1 ? 1 : 0, L.flags, L.vis_bits[0], data_operand_kind(o->data_binary)}, f);
        c.pos_phase = false;
        CHK(rc);
    }
    for (int it = 0; it < o->cd_k; ++it) {
        const bool last = (it == o->cd_k - 1);
        {   // v_prob = down(h); v = sampleV(v_prob)
            FinishArgs f = new_finish();
            f.vmode = 1; f.uni = c.rng.floats(B, L.V);
            c.rng.cats(B, c.d->n_groups, &f.cat_tape, &f.cat_uni);
            // the fp32 copies of v_prob / v are only read back by the softmax-group kernel (prop() supplies
            // scratch for them when groups exist): without groups nobody needs them -> 5 MB of stores saved
            // the K16-blocked bf16 form only feeds a K1 that cannot read the bit plane (its 2-byte scattered stores were most
            // of the fused K2's 3.8 us epilogue)
            if (!(vbits && k1s_neg)) { f.op.rm = L.vis_rm[1]; f.op.rm_terms = 1; f.rm_src = 2; }
            f.op.tr = L.vis_tr[1]; f.op.tr_terms = 1; f.tr_src = 2;
            f.colsum_part = L.cs_vneg; f.colsum_src = 2;
            f.loss_ref = data; f.ld_ref = ldd; f.loss_src = 1; f.loss_part = L.loss_part;
            if (vbits) f.op.bits = L.vis_bits[1];
            CHK(prop(c, false, OpIn{L.hid_rm, 1, nullptr}, f, (it == 0 && !next_on_k1) ? next : nullptr));
        }
        {   // h_prob = up(v); h = 1[h_prob > U]  (the last draw is consumed but unused, rbm.py:208)
            FinishArgs f = new_finish();
            DrawSrc u = c.rng.floats(B, L.H);
            if (!last) { f.vmode = 1; f.uni = u; f.op.rm = L.hid_rm; f.op.rm_terms = 1; f.rm_src = 2; }
            f.op.tr = L.hid_tr[1]; f.op.tr_terms = c.ht; f.tr_src = 1; f.op.tr_negate = 1;
            f.colsum_part = L.cs_hneg; f.colsum_src = 1;
            CHK(prop(c, true, OpIn{L.vis_rm[1], 1, nullptr, vbits ? L.vis_bits[1] : nullptr, 1}, f, (it == 0 && next_on_k1) ? next : nullptr));
        }
    }
    return 0;
}

BiasArgs make_bias(Ctx& c, const imdbn_cd_opts* o, bool sparsity, float n, float* loss_out) {
    const Layout& L = c.L;
    BiasArgs b;
    memset(&b, 0, sizeof(b));
    b.hid_bias = c.d->hid_bias; b.hb_m = c.d->hb_m; b.H = L.H; b.hpos = L.cs_hpos; b.hneg = L.cs_hneg;
    b.vis_bias = c.d->vis_bias; b.vb_m = c.d->vb_m; b.V = L.V; b.vpos = L.cs_vpos; b.vneg = L.cs_vneg;
    b.P = L.P; b.lr = o->lr; b.mom = o->momentum; b.n = n;
    b.sparsity = sparsity ? 1 : 0; b.target = o->sparsity_target;
    b.loss_part = L.loss_part; b.n_loss = n_loss_used(c, false); b.loss_den = n * (float)L.V; b.loss_out = loss_out;
    return b;
}

int launch_bias(Ctx& c, const BiasArgs& b) {
    hipLaunchKernelGGL(bias_update, dim3(cdiv(std::max(c.L.V, c.L.H), 256) + 1), dim3(256), 0, c.s, b);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- row-parallel chain kernel (kernels_chain.hpp) -------------------------------------------
bool chain_kernel_ok(const Ctx& c, int n_steps) {
    const Layout& L = c.L;
    if (g_no_chain_kernel || !L.k4_planes || n_steps < 2 || n_steps > CHAIN_MAX_STEPS) return false;
    if (c.d->n_groups > 1) return false;                   // the kernel keeps ONE softmax group's logits in LDS
    const int gw = c.d->n_groups ? c.d->group_end[0] - c.d->group_start[0] : 0;
    const size_t lds = (size_t)(c.nw == 3 ? 2 : 1) * K4_ROWS * ((rup(L.V, 32) + 8) + (rup(L.H, 32) + 8)) * 2      // activation terms
                     + (size_t)K4_ROWS * (rup(std::max(L.V, L.H), 16) + 1) * 4                        // fp32 stage
                     + (size_t)K4_ROWS * (gw + 1) * 4 + K4_ROWS * 16                                   // group logits, row stats
                     + (size_t)(K4_ROWS / 2) * gw * 2 * 4;                                             // group-column noise drawn ahead
    return gw <= GROUP_WMAX && lds <= (size_t)K4_LDS_BYTES;
}

// One chain as the host sees it (imdbn_chain_spec + where its final state goes)
struct ChainSpec {
    const float* vk; const float* mask; int64_t ldk; int init_uniform; int n_steps; const imdbn_chain_step* st;
    const float* mu; int64_t ldmu; int Dz; float* out; int64_t ldo;
};

// v0 = vk*m + (1-m)*U   (rbm.py:271,333,392) into `out` (and, for the per-launch path, the operand forms of vis_rm[0])
int chain_init(Ctx& c, const ChainSpec& s, bool forms, bool stats_now) {
    const Layout& L = c.L;
    const int B = L.B;
    PrepArgs p;
    memset(&p, 0, sizeof(p));
    p.in = s.vk; p.ld = s.ldk; p.B = B; p.Bp = L.Bp; p.N = L.V;
    if (s.init_uniform) { p.mix = 1; p.mask = s.mask; p.ldm = s.ldk; p.uni = c.rng.floats(B, L.V); }
    p.out_f32 = s.out; p.ldo = s.ldo;
    p.op.ldrm = L.Vpad; p.op.rm_ts = (int64_t)L.Bp * L.Vpad; p.op.Bp = L.Bp;
    if (forms) { p.op.rm = L.vis_rm[0]; p.op.rm_terms = c.rt; }
    if (stats_now) {
        p.op.tr = L.vis_tr[0]; p.op.tr_ts = (int64_t)L.V * L.Bp; p.op.tr_terms = c.rt;
        p.colsum_part = L.cs_vpos;
    }
    p.zero = L.k1s_cnt; p.n_zero = (L.Bp / 64) * L.k1s_tiles;      // (as prep(): the per-launch chain of a wide layer runs k1_stream)
    c.cnt_ok = true;
    hipLaunchKernelGGL(prep_operand, dim3(cdiv(L.Vpad, 64), L.P), dim3(256), 0, c.s, p);
    HIPCHK(hipGetLastError());
    return 0;
}

// draw cursors of a chain's steps in exactly the order of the per-launch path, written as records [off, off + n_steps)
int chain_records(Ctx& c, const ChainSpec& s, int off) {
    const Layout& L = c.L;
    const imdbn_rbm_desc* d = c.d;
    const int B = L.B;
    ChainRecBatch batch;
    for (int t0 = 0; t0 < s.n_steps; t0 += CHAIN_REC_BATCH) {
        const int n = std::min(CHAIN_REC_BATCH, s.n_steps - t0);
        memset(&batch, 0, sizeof(batch));
        for (int i = 0; i < n; ++i) {
            const imdbn_chain_step& st = s.st[t0 + i];
            ChainRec& r = batch.r[i];
            r.T = st.T; r.sigma = st.sigma; r.eta = st.eta;
            r.flags = (st.sample_h ? 1 : 0) | ((st.vmode & 3) << 1) | (st.clamp ? 8 : 0);
            auto cd = [](const DrawSrc& x) { ChainDraw y; y.tape = x.tape; y.draw = x.draw; return y; };
            if (st.sigma > 0.f) r.noise_h = cd(c.rng.floats(B, L.H));
            if (st.sample_h) r.uni_h = cd(c.rng.floats(B, L.H));
            if (st.sigma > 0.f) r.noise_v = cd(c.rng.floats(B, L.V));
            if (st.vmode != 0) {
                r.uni_v = cd(c.rng.floats(B, L.V));
                DrawSrc cu; c.rng.cats(B, d->n_groups, &r.cat_tape, &cu);
                r.cat_uni = cd(cu);
            }
        }
        hipLaunchKernelGGL(chain_write_recs, dim3(1), dim3(64), 0, c.s, batch, L.chain_recs + off + t0, n);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the chain kernel for one chain (s1 == nullptr) or two independent ones of the same RBM in one launch
int launch_k4(Ctx& c, const ChainSpec& s0, int off0, const ChainSpec* s1, int off1) {
    const Layout& L = c.L;
    const imdbn_rbm_desc* d = c.d;
    const int B = L.B;
    {
        const dim3 sg(std::max(cdiv(L.H, 16), cdiv(L.V, 16)), std::max(cdiv(L.V, 32), cdiv(L.H, 32)), 2);
        if (c.nw == 3) hipLaunchKernelGGL(k4_split_planes<2>, sg, dim3(64), 0, c.s, d->W, d->ldw, L.V, L.H, L.k4_planes, L.k4_plane_stride);
        else           hipLaunchKernelGGL(k4_split_planes<1>, sg, dim3(64), 0, c.s, d->W, d->ldw, L.V, L.H, L.k4_planes, L.k4_plane_stride);
    }
    K4Args a;
    memset(&a, 0, sizeof(a));
    a.planes = L.k4_planes; a.plane_stride = L.k4_plane_stride;
    a.V = L.V; a.H = L.H; a.nw = c.nw; a.rt = c.rt;
    a.hid_bias = d->hid_bias; a.vis_bias = d->vis_bias;
    a.n_groups = d->n_groups;
    for (int g = 0; g < IMDBN_MAX_GROUPS; ++g) { a.gs[g] = d->group_start[g]; a.ge[g] = d->group_end[g]; }
    a.seed = c.rng.r ? c.rng.r->seed : 0; a.row0 = c.rng.r ? c.rng.r->row0 : 0;
    a.draw_base = c.rng.r ? (const unsigned long long*)c.rng.r->dev_offset : nullptr;
    auto seg = [&](const ChainSpec& s, int off) {
        K4Seg g;
        g.state = s.out; g.lds = s.ldo; g.recs = L.chain_recs + off; g.n_steps = s.n_steps;
        g.mu = s.mu; g.ldmu = s.ldmu; g.Dz = s.Dz; g.vk = s.vk; g.mask = s.mask; g.ldk = s.ldk; g.B = B;
        return g;
    };
    a.s0 = seg(s0, off0);
    a.s1 = s1 ? seg(*s1, off1) : a.s0;
    // rows per block: enough blocks to spread the per-element work (Philox, Box-Muller, sigmoid) over the CUs;
    // one block per CU at most (every block streams all of W from L2)
    a.dbg = (g_dbg & 1024) ? 1 : 0;
    const int nch = s1 ? 2 : 1, BT = B * nch;
    a.rows = g_k4_rows > 0 ? g_k4_rows : (BT <= 2 * cu_count() ? 2 : (BT <= 4 * cu_count() ? 4 : (BT <= 8 * cu_count() ? 8 : 16)));      // measured: 0.90 / 0.99 / 1.19 / 1.59 ms for 2 / 4 / 8 / 16 rows (30 steps, 532<->256)
    a.nblk0 = cdiv(B, a.rows);
    const dim3 grid(a.nblk0 * nch);
    if (c.nw == 3) hipLaunchKernelGGL(k4_chain<2>, grid, dim3(64 * K4_WAVES), 0, c.s, a);      // PARITY: fp16 hi + lo terms
    else           hipLaunchKernelGGL(k4_chain<1>, grid, dim3(64 * K4_WAVES), 0, c.s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// chain: init + steps.  The final state ends in fp32 `out` (ld ldo); with want_stats also as operand forms in vis_rm[0] (rt terms),
// the transposed form in vis_tr[0] and column sums in cs_vpos (the positive phase of the clamped update).
int run_chain(Ctx& c, const ChainSpec& s, bool want_stats) {
    const Layout& L = c.L;
    const int B = L.B;
    const float* vk = s.vk; const float* mask = s.mask; const int64_t ldk = s.ldk, ldo = s.ldo, ldmu = s.ldmu;
    const int n_steps = s.n_steps, Dz = s.Dz; const imdbn_chain_step* st = s.st; const float* mu = s.mu; float* out = s.out;
    if (n_steps < 0 || (n_steps > 0 && !st)) return fail(IMDBN_E_INVALID, "bad chain steps");
    if (mu && (Dz <= 0 || Dz > L.V)) return fail(IMDBN_E_INVALID, "mu-pull width %d outside (0,%d]", Dz, L.V);
    const bool k4 = chain_kernel_ok(c, n_steps);
    CHK(chain_init(c, s, !k4 || n_steps == 0, want_stats && n_steps == 0));
    if (k4) {
        CHK(chain_records(c, s, 0));
        CHK(launch_k4(c, s, 0, nullptr, 0));
        // operand forms of the final state (what the last v|h launch of the per-launch path leaves behind): only the clamped update reads them
        if (want_stats) CHK(prep(c, out, ldo, L.V, L.vis_rm[0], L.Vpad, L.vis_tr[0], nullptr, L.cs_vpos, c.rt));
        c.hid_bits_ok = false;
        return 0;
    }
    for (int t = 0; t < n_steps; ++t) {
        const imdbn_chain_step& s = st[t];
        const bool last = (t == n_steps - 1);
        {   // h | v
            FinishArgs f = new_finish();
            f.T = s.T; f.sigma = s.sigma;
            if (s.sigma > 0.f) f.noise = c.rng.floats(B, L.H);
            if (s.sample_h) { f.vmode = 1; f.uni = c.rng.floats(B, L.H); }
            f.op.rm = L.hid_rm; f.op.rm_terms = s.sample_h ? 1 : c.rt; f.rm_src = s.sample_h ? 2 : 1;
            CHK(prop(c, true, OpIn{L.vis_rm[0], c.rt, nullptr}, f));
        }
        {   // v | h
            FinishArgs f = new_finish();
            f.T = s.T; f.sigma = s.sigma;
            if (s.sigma > 0.f) f.noise = c.rng.floats(B, L.V);
            if (mu && s.eta != 0.f) { f.mu = mu; f.ldmu = ldmu; f.Dz = Dz; f.eta = s.eta; }
            if (s.clamp) { f.clamp = 1; f.vk = vk; f.mask = mask; f.ldk = ldk; }
            f.vmode = s.vmode;
            if (s.vmode != 0) { f.uni = c.rng.floats(B, L.V); c.rng.cats(B, c.d->n_groups, &f.cat_tape, &f.cat_uni); }
            f.out_prob = L.f_vp; f.ld_prob = L.V;
            f.out_final = out; f.ld_final = ldo;
            f.op.rm = L.vis_rm[0]; f.op.rm_terms = c.rt; f.rm_src = 2;
            if (want_stats && last) {
                f.op.tr = L.vis_tr[0]; f.op.tr_terms = c.rt; f.tr_src = 2;
                f.colsum_part = L.cs_vpos; f.colsum_src = 2;
            }
            CHK(prop(c, false, OpIn{L.hid_rm, s.sample_h ? 1 : c.rt, nullptr}, f));
        }
    }
    return 0;
}

hipStream_t S(imdbn_stream_t s) { return (hipStream_t)s; }

}  // namespace

// =================================================================================================
extern "C" {

int imdbn_version(void) { return IMDBN_ABI_VERSION; }

int imdbn_last_error(char* buf, size_t n) {
    if (buf && n) { strncpy(buf, g_err, n - 1); buf[n - 1] = 0; }
    return (int)strlen(g_err);
}

int imdbn_device_info(int* cu_count, char* arch, size_t n) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(IMDBN_E_NODEVICE, "no HIP device");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (arch && n) { strncpy(arch, p.gcnArchName, n - 1); arch[n - 1] = 0; }
    return 0;
}

size_t imdbn_ws_bytes(int V, int H, int B) {
    if (V <= 0 || H <= 0 || B <= 0) return 0;
    return make_layout(V, H, B, nullptr).bytes;
}

int imdbn_set_tuning(int ksplit_up, int ksplit_down) {
    g_defaults.ks_up = std::max(0, ksplit_up);
    g_defaults.ks_down = std::max(0, ksplit_down);
    return 0;
}

static int set_opt(Tuning& t, const char* name, int value) {
    if (!name) return fail(IMDBN_E_INVALID, "null option name");
    if (!strcmp(name, "ksplit_up")) t.ks_up = std::max(0, value);
    else if (!strcmp(name, "ksplit_down")) t.ks_down = std::max(0, value);
    else if (!strcmp(name, "generic_k3")) t.no_fast_k3 = value != 0;
    else if (!strcmp(name, "down_rows")) { if (value != 0 && (value < 4 || value > 32 || value % 4)) return fail(IMDBN_E_INVALID, "down_rows must be 0 or a multiple of 4 in [4, 32]"); t.down_tr = value; }
    else if (!strcmp(name, "no_rank_loop")) t.no_rank_loop = value;
    else if (!strcmp(name, "no_chain_kernel")) t.no_chain_kernel = value;
    else if (!strcmp(name, "chain_rows")) { if (value < 0 || value > 16) return fail(IMDBN_E_INVALID, "chain_rows must be in [0, 16]"); t.k4_rows = value; }
    else if (!strcmp(name, "no_bits")) t.no_bits = value;
    else if (!strcmp(name, "no_prefetch")) t.no_prefetch = value;
    else if (!strcmp(name, "no_rank_acc")) t.no_rank_acc = value;
    else if (!strcmp(name, "min_rank_loop")) t.min_rank_loop = value;
    else if (!strcmp(name, "dbg")) g_dbg = value;
    else if (!strcmp(name, "no_k1s")) t.no_k1s = value;
    else if (!strcmp(name, "k1s_ks")) t.k1s_ks = std::max(0, value);
    else if (!strcmp(name, "no_k2s")) t.no_k2s = value;
    else if (!strcmp(name, "k2s_rows")) { if (value != 0 && (value < 8 || value > 48 || value % 8)) return fail(IMDBN_E_INVALID, "k2s_rows must be 0 or a multiple of 8 in [8, 48]"); t.k2s_tr = value; }
    else if (!strcmp(name, "generic_k1")) t.no_fast_k1 = value != 0;
    else if (!strcmp(name, "no_k1s_real")) t.no_k1s_real = value;
    else if (!strcmp(name, "no_adaptive")) t.no_adaptive = value;
    else if (!strcmp(name, "k1s_force_na")) t.k1s_force_na = value;
    else if (!strcmp(name, "no_chain_pair")) t.no_chain_pair = value;
    else if (!strcmp(name, "no_down_chunks")) t.no_down_chunks = value;
    else if (!strcmp(name, "no_down_tiled")) t.no_down_tiled = value;
    else if (!strcmp(name, "k1s_lds_pad")) t.k1s_lds_pad = std::max(0, std::min(value, 64 * 1024));
    else if (!strcmp(name, "no_fused_up")) t.no_fused_up = value != 0;
    else return fail(IMDBN_E_INVALID, "unknown option %s", name);
    return 0;
}

int imdbn_set_option(const char* name, int value) { return set_opt(g_defaults, name, value); }

struct imdbn_options { Tuning t; };
imdbn_options* imdbn_options_create(void) { return new (std::nothrow) imdbn_options{g_defaults}; }      // starts as a copy of the defaults
void imdbn_options_destroy(imdbn_options* o) { if (o && t_bound == &o->t) t_bound = nullptr; delete o; }
int imdbn_options_set(imdbn_options* o, const char* name, int value) {
    if (!o) return fail(IMDBN_E_INVALID, "null options handle");
    if (name && !strcmp(name, "dbg")) return fail(IMDBN_E_INVALID, "dbg is process-wide (imdbn_set_option)");
    return set_opt(o->t, name, value);
}
int imdbn_use_options(const imdbn_options* o) { t_bound = o ? &o->t : nullptr; return 0; }

int imdbn_profile_enable(int on) {
    if (on && g_prof.ev.empty()) {
        g_prof.ev.resize(2 * 4096);
        for (auto& e : g_prof.ev) HIPCHK(hipEventCreate(&e));
    }
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.calls = 0;
    return 0;
}

int imdbn_debug_stamps(long long* out, int n) {
    if (!out || n <= 0 || n > 4096 * 8) return fail(IMDBN_E_INVALID, "imdbn_debug_stamps: bad buffer");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * (size_t)n, 0, hipMemcpyDeviceToHost));
    static const std::vector<long long> zeros(4096 * 8, 0);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros.data(), sizeof(long long) * zeros.size(), 0, hipMemcpyHostToDevice));
    return 0;
}

int imdbn_debug_ws_offset(int V, int H, int B, const char* name, size_t* offset) {
    if (V <= 0 || H <= 0 || B <= 0 || !name || !offset) return fail(IMDBN_E_INVALID, "debug_ws_offset: bad argument");
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 30);
    const Layout L = make_layout(V, H, B, fake);
    const void* p = nullptr;
    if (!strcmp(name, "vis_bits0")) p = L.vis_bits[0]; else if (!strcmp(name, "vis_bits1")) p = L.vis_bits[1];
    else if (!strcmp(name, "hid_bits")) p = L.hid_bits; else if (!strcmp(name, "vis_tr0")) p = L.vis_tr[0];
    else if (!strcmp(name, "vis_tr1")) p = L.vis_tr[1]; else if (!strcmp(name, "hid_tr0")) p = L.hid_tr[0];
    else if (!strcmp(name, "hid_tr1")) p = L.hid_tr[1]; else if (!strcmp(name, "cs_hpos")) p = L.cs_hpos;
    else if (!strcmp(name, "cs_hneg")) p = L.cs_hneg; else if (!strcmp(name, "cs_vpos")) p = L.cs_vpos;
    else if (!strcmp(name, "cs_vneg")) p = L.cs_vneg; else if (!strcmp(name, "flags")) p = L.flags;
    else if (!strcmp(name, "partial")) p = L.partial; else if (!strcmp(name, "vis_rm1")) p = L.vis_rm[1];
    else return fail(IMDBN_E_INVALID, "debug_ws_offset: unknown buffer %s", name);
    *offset = (size_t)((const char*)p - fake);
    return 0;
}

int imdbn_profile_read(double* total_ms, int* launches) {
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        HIPCHK(hipEventSynchronize(g_prof.ev[i + 1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int)(g_prof.used / 2);
    g_prof.used = 0;
    return 0;
}

// the device-resident draw counter of imdbn_rng.dev_offset: += n, as a node of the caller's stream (or captured graph)
__global__ void rng_advance_kernel(unsigned long long* p, unsigned long long n) { *p += n; }
int imdbn_rng_advance(uint64_t* dev_offset, uint64_t n, imdbn_stream_t stream) {
    if (!dev_offset) return fail(IMDBN_E_INVALID, "rng_advance: null counter");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, S(stream), (unsigned long long*)dev_offset, (unsigned long long)n);
    HIPCHK(hipGetLastError());
    return 0;
}

int imdbn_rbm_prop_up(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, float T, imdbn_rng* rng,
                      float* out_prob, int64_t ldo, float* out_sample, int64_t lds, void* ws, size_t ws_bytes,
                      imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v || !out_prob || ldv < d->V || ldo < d->H) return fail(IMDBN_E_INVALID, "prop_up: bad tensor argument");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(prep(c, v, ldv, d->V, c.L.vis_rm[0], c.L.Vpad, nullptr, c.L.flags));
    FinishArgs f = new_finish();
    f.T = T;
    f.out_prob = out_prob; f.ld_prob = ldo;
    if (out_sample) {
        if (lds < d->H) return fail(IMDBN_E_INVALID, "prop_up: bad sample ld");
        f.vmode = 1; f.uni = c.rng.floats(B, d->H); f.out_final = out_sample; f.ld_final = lds;
    }
    CHK(prop(c, true, OpIn{c.L.vis_rm[0], c.nw == 1 ? 1 : 0, c.L.flags}, f));
    return c.rng.finish();
}

// forward(v) at T = 1 on exactly the path the fused forward of imdbn_rbm_cd_step takes for the same batch (bit-identical):
// a 0/1 batch is read as a bit plane by k1_stream, anything else through the three-term operand form
int imdbn_rbm_forward(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, int data_binary, float* out_prob, int64_t ldo,
                      void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v || !out_prob || ldv < d->V || ldo < d->H) return fail(IMDBN_E_INVALID, "forward: bad tensor argument");
    Ctx c(d, nullptr, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    if (data_binary < 0 || data_binary > 2) return fail(IMDBN_E_INVALID, "forward: data_binary %d", data_binary);
    const bool k1s = vec4_weights(d) && !g_no_k1s && c.L.Vpad > 1024;       // prop() then launches k1_stream
    const bool bits = k1s && data_binary != IMDBN_DATA_REAL, only_bits = k1s && data_binary == IMDBN_DATA_BINARY;
    CHK(prep(c, v, ldv, d->V, only_bits ? nullptr : c.L.vis_rm[0], c.L.Vpad, nullptr, c.L.flags, nullptr, 3, bits ? c.L.vis_bits[0] : nullptr));
    FinishArgs f = new_finish();
    f.out_prob = out_prob; f.ld_prob = ldo;
    CHK(prop(c, true, OpIn{c.L.vis_rm[0], c.nw == 1 ? 1 : 0, c.L.flags, bits ? c.L.vis_bits[0] : nullptr, bits ? data_operand_kind(data_binary) : 0}, f));
    return 0;
}

int imdbn_rbm_free_energy(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, float* out_F, void* ws, size_t ws_bytes,
                          imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v || !out_F || ldv < d->V) return fail(IMDBN_E_INVALID, "free_energy: bad tensor argument");
    Ctx c(d, nullptr, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(prep(c, v, ldv, d->V, c.L.vis_rm[0], c.L.Vpad, nullptr, c.L.flags));
    FinishArgs f = new_finish();
    f.logits_only = 1;                                     // x = v W + c
    f.out_prob = c.L.f_h; f.ld_prob = d->H;
    CHK(prop(c, true, OpIn{c.L.vis_rm[0], c.nw == 1 ? 1 : 0, c.L.flags}, f));
    hipLaunchKernelGGL(free_energy_rows, dim3(B), dim3(256), 0, c.s, v, ldv, d->vis_bias, d->V, c.L.f_h, (int64_t)d->H, d->H, out_F);
    HIPCHK(hipGetLastError());
    return 0;
}

int imdbn_rbm_prop_down(const imdbn_rbm_desc* d, const float* h, int64_t ldh, int B, float T, int logits_only,
                        float* out_prob, int64_t ldo, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!h || !out_prob || ldh < d->H || ldo < d->V) return fail(IMDBN_E_INVALID, "prop_down: bad tensor argument");
    Ctx c(d, nullptr, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(prep(c, h, ldh, d->H, c.L.hid_rm, c.L.Hpad, nullptr, c.L.flags_h));
    FinishArgs f = new_finish();
    f.T = T; f.logits_only = logits_only;
    f.out_prob = out_prob; f.ld_prob = ldo;
    CHK(prop(c, false, OpIn{c.L.hid_rm, c.nw == 1 ? 1 : 0, c.L.flags_h}, f));
    return 0;
}

int imdbn_rbm_sample_visible(const imdbn_rbm_desc* d, const float* v_prob, int64_t ldp, int B, imdbn_rng* rng,
                             float* out, int64_t ldo, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v_prob || !out || ldp < d->V || ldo < d->V || B <= 0) return fail(IMDBN_E_INVALID, "sample_visible: bad argument");
    Rng r(rng);
    DrawSrc u = r.floats(B, d->V);
    const int32_t* ct; DrawSrc cu;
    r.cats(B, d->n_groups, &ct, &cu);
    CHK(r.finish());
    const int64_t total = (int64_t)B * d->V;
    hipLaunchKernelGGL(bernoulli_rows, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, S(stream),
                       v_prob, ldp, B, d->V, u, out, ldo);
    HIPCHK(hipGetLastError());
    for (int g = 0; g < d->n_groups; ++g) {
        DrawSrc cg = cu; cg.draw += g;
        hipLaunchKernelGGL(categorical_rows, dim3(cdiv(B, 64)), dim3(64), 0, S(stream), v_prob, ldp, B,
                           d->group_start[g], d->group_end[g], ct ? ct + (int64_t)g * B : nullptr, cg, out, ldo);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

int imdbn_rbm_gibbs_step(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, int sample_h, int sample_v,
                         imdbn_rng* rng, float* v_next, float* v_prob, float* h, float* h_prob, void* ws,
                         size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v || !v_next || !v_prob || !h || !h_prob || ldv < d->V) return fail(IMDBN_E_INVALID, "gibbs_step: bad argument");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    const Layout& L = c.L;
    CHK(prep(c, v, ldv, d->V, L.vis_rm[0], L.Vpad, nullptr, L.flags));
    {
        FinishArgs f = new_finish();
        f.out_prob = h_prob; f.ld_prob = d->H; f.out_final = h; f.ld_final = d->H;
        if (sample_h) { f.vmode = 1; f.uni = c.rng.floats(B, d->H); }
        f.op.rm = L.hid_rm; f.op.rm_terms = sample_h ? 1 : c.rt; f.rm_src = 2;
        CHK(prop(c, true, OpIn{L.vis_rm[0], c.nw == 1 ? 1 : 0, L.flags}, f));
    }
    {
        FinishArgs f = new_finish();
        f.out_prob = v_prob; f.ld_prob = d->V; f.out_final = v_next; f.ld_final = d->V;
        if (sample_v) { f.vmode = 1; f.uni = c.rng.floats(B, d->V); c.rng.cats(B, d->n_groups, &f.cat_tape, &f.cat_uni); }
        CHK(prop(c, false, OpIn{L.hid_rm, sample_h ? 1 : c.rt, nullptr}, f));
    }
    return c.rng.finish();
}

// Start of a CD pass (imdbn_rbm_cd_step, imdbn_rbm_cd_factors_wire): the next batch's preparation (imdbn_cd_opts.next_*) and
// the prefetch slot the data sit in.  `pn` = the preparation, `rides` = cd_phases carries it on one of its launches.
static int cd_prologue(Ctx& c, const imdbn_cd_opts* o, PrepArgs& pn, bool& rides, bool allow_compact = true) {
    const imdbn_rbm_desc* d = c.d;
    if (o->data_slot < 0 || o->data_slot > 2 || (o->next_data && (o->next_slot < 1 || o->next_slot > 2 || o->next_slot == o->data_slot || o->ld_next < d->V)))
        return fail(IMDBN_E_INVALID, "cd: bad prefetch slots (data %d, next %d)", o->data_slot, o->next_slot);
    const int next_rows = (o->next_data && prefetch_available(d)) ? 1 : 0;
    memset(&pn, 0, sizeof(pn));
    if (next_rows > 0) {            // target buffers are taken before the data slot is swapped in
        const Layout& L = c.L;
        const int t = o->next_slot - 1;
        pn.in = o->next_data; pn.ld = o->ld_next; pn.B = L.B; pn.Bp = L.Bp; pn.N = L.V;
        pn.op.rm = L.pf_rm[t]; pn.op.ldrm = L.Vpad; pn.op.rm_ts = (int64_t)L.Bp * L.Vpad; pn.op.rm_terms = 3; pn.op.Bp = L.Bp;
        pn.op.tr = L.pf_tr[t]; pn.op.tr_ts = (int64_t)L.V * L.Bp; pn.op.tr_terms = 3;
        pn.flag = L.pf_flags[t]; pn.colsum_part = L.pf_cs[t];
        pn.op.bits = L.pf_bits[t]; pn.op.bits_shape = 0;
    }
    // Where the next batch is prepared: as extra blocks of the fused K2 (gemm_down_fused_next) or of the negative-phase
    // k1_stream (cd_phases decides); where neither can carry them, a prep_operand launch of its own, first thing.
    rides = next_rows > 0 && (!k2s_for_cd(d) || (d->n_groups == 0 && vec4_weights(d) && !g_no_k1s && c.L.Vpad > 1024));
    if (next_rows > 0 && allow_compact && vec4_weights(d) && !g_no_k1s && c.L.Vpad > 1024) {
        if (o->next_binary == IMDBN_DATA_BINARY) {
            // a 0/1 batch: the positive phase reads the bit plane, the update kernel one bf16 plane (the exactness map says "one term")
            pn.op.rm = nullptr; pn.op.rm_terms = 0; pn.op.tr_terms = 1;
        } else if (o->next_binary == IMDBN_DATA_UNKNOWN && rides && !g_no_adaptive && !g_no_k1s_real && adaptive_shape_ok(c.L)) {
            pn.adaptive = 1;      // the same slim set for every 64-column item that turns out to be all 0/1, decided by the preparing block
        }
    }
    if (next_rows > 0 && !rides) {
        pn.zero = nullptr; pn.n_zero = 0;
        hipLaunchKernelGGL(prep_operand, dim3(cdiv(std::max(pn.N, pn.op.ldrm), 64), c.L.P), dim3(256), 0, c.s, pn);
        HIPCHK(hipGetLastError());
    }
    if (o->data_slot) { use_slot(c.L, o->data_slot); c.data_prepped = true; c.cnt_ok = true; c.fix_slot = o->data_binary == IMDBN_DATA_UNKNOWN; }
    return 0;
}

int imdbn_rbm_cd_step(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* o,
                      imdbn_rng* rng, float* loss_out, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    if (!data || !o || ldd < d->V) return fail(IMDBN_E_INVALID, "cd_step: bad argument");
    if (o->data_binary < 0 || o->data_binary > 2 || o->next_binary < 0 || o->next_binary > 2) return fail(IMDBN_E_INVALID, "cd_step: data_binary / next_binary outside 0..2");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    PrepArgs pn;
    bool rides = false;
    CHK(cd_prologue(c, o, pn, rides));
    CHK(cd_phases(c, data, ldd, o, rides ? &pn : nullptr));
    CHK(c.rng.finish());
    const BiasArgs bias = make_bias(c, o, o->sparsity != 0, (float)B, loss_out);
    CHK(launch_assoc(c, 0, o, c.nw == 1 ? 1 : 0, c.L.flags, 1, (float)B, nullptr, &bias));
    if (o->fwd_out) {
        // forward(data) under the updated weights (idbn.py:195-204: train_epoch(v); v = forward(v)): the positive-phase
        // propagation once more -- same operand forms, still in the workspace -- with the probabilities as the only output
        if (o->ld_fwd < d->H) return fail(IMDBN_E_INVALID, "cd_step: ld_fwd %lld < H %d", (long long)o->ld_fwd, d->H);
        FinishArgs f = new_finish();
        f.out_prob = o->fwd_out; f.ld_prob = o->ld_fwd;
        const bool bits = o->data_binary != IMDBN_DATA_REAL && vec4_weights(d) && !g_no_k1s && c.L.Vpad > 1024;      // as imdbn_rbm_forward
        CHK(prop(c, true, OpIn{c.L.vis_rm[0], c.nw == 1 ? 1 : 0, c.L.flags, bits ? c.L.vis_bits[0] : nullptr, bits ? data_operand_kind(o->data_binary) : 0}, f));
    }
    return 0;
}

int imdbn_rbm_prefetch_ok(const imdbn_rbm_desc* d, int B) {
    if (check_desc(d, true) != 0 || B <= 0) return 0;
    return prefetch_available(d) ? 1 : 0;
}

// bias / sparsity / error tail of the packed statistics buffer (after the V*H delta-W floats): written by the extra block row of the
// statistics kernel (BiasArgs::pack_tail) instead of a launch of its own
static BiasArgs make_pack(Ctx& c, float* packed) {
    const imdbn_rbm_desc* d = c.d;
    BiasArgs b;
    memset(&b, 0, sizeof(b));
    b.pack_tail = packed + (size_t)d->V * d->H; b.H = d->H; b.V = d->V;
    b.hpos = c.L.cs_hpos; b.hneg = c.L.cs_hneg; b.vpos = c.L.cs_vpos; b.vneg = c.L.cs_vneg; b.P = c.L.P;
    b.loss_part = c.L.loss_part; b.n_loss = n_loss_used(c, false);
    return b;
}

size_t imdbn_packed_delta_floats(int V, int H) {
    const size_t n = (size_t)V * H + (size_t)2 * H + V + 1;
    return (n + 3) / 4 * 4;
}

int imdbn_rbm_cd_stats(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* o,
                       imdbn_rng* rng, float* packed, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!data || !o || !packed || ldd < d->V) return fail(IMDBN_E_INVALID, "cd_stats: bad argument");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    PrepArgs pn;
    bool rides = false;
    CHK(cd_prologue(c, o, pn, rides));          // the next-batch preparation / prefetch slot of imdbn_rbm_cd_step
    CHK(cd_phases(c, data, ldd, o, rides ? &pn : nullptr));
    CHK(c.rng.finish());
    const BiasArgs pack = make_pack(c, packed);
    CHK(launch_assoc(c, 1, o, c.nw == 1 ? 1 : 0, c.L.flags, 1, 1.0f, packed, &pack));
    return 0;
}

// ---- data-parallel "factor exchange" ----------------------------------------------------------
// The statistics are linear in the per-row factors, and the factors of 64 rows (7 MB at 10000 x 1500) are 8x
// smaller than the fp32 delta-W (60 MB): every rank all-gathers the factor blocks and runs the streaming update
// kernel once per rank block (first / middle / last pass) -- the same kernel, bits and order on every rank.
int imdbn_factor_block(int V, int H, int B, size_t* offset, size_t* bytes) {
    if (V <= 0 || H <= 0 || B <= 0 || !offset || !bytes) return fail(IMDBN_E_INVALID, "factor_block: bad argument");
    const Layout L = make_layout(V, H, B, nullptr);
    *offset = L.fb_off; *bytes = L.fb_bytes;
    return 0;
}

static bool factor_mode_ok(const imdbn_rbm_desc* d, int B, bool with_momentum) {
    return B >= 1 && B <= 64 && d->H % 4 == 0 && d->H >= 4 && d->ldw % 4 == 0 && (((uintptr_t)d->W) & 15) == 0 &&
           (!with_momentum || (d->W_m && (((uintptr_t)d->W_m) & 15) == 0)) && d->n_groups == 0;
}

// offsets of the wire form (kernels_ew.hpp FactorWireArgs) for an (V, H, B) factor block
static FactorWireArgs wire_layout(int V, int H, int B, int binary, size_t* compact_bytes) {
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 30);          // only differences of the carved pointers are used
    const Layout L = make_layout(V, H, B, fake);
    const char* fb = fake + L.fb_off;
    FactorWireArgs w;
    memset(&w, 0, sizeof(w));
    w.V = V; w.Bp = L.Bp; w.binary = binary ? 1 : 0;
    w.f_vpos = (size_t)((const char*)L.vis_tr[0] - fb);
    w.f_vneg = (size_t)((const char*)L.vis_tr[1] - fb);
    w.f_cs_hpos = (size_t)((const char*)L.cs_hpos - fb);
    // (exact sizes, both multiples of 16: the alignment padding behind them is copied from the block itself either way)
    w.f_flags = (size_t)((const char*)L.flags - fb); w.n_flags = (size_t)L.P * cdiv(L.Vpad, 64) * 4;
    w.f_cs_vpos = (size_t)((const char*)L.cs_vpos - fb); w.n_cs_vpos = (size_t)L.P * V * 4;
    w.head_bytes = w.f_vpos;                                            // flags .. loss_part precede the visible planes
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t bits = up((size_t)V * L.Bp / 8);
    w.c_vneg = w.head_bytes;
    w.c_vpos = w.c_vneg + bits;
    w.c_bad = w.c_vpos + (binary ? bits : up((size_t)3 * V * L.Bp * 2));
    if (compact_bytes) *compact_bytes = w.c_bad + 256;
    return w;
}

int imdbn_factor_compact_bytes(int V, int H, int B, int binary_data, size_t* bytes) {
    if (V <= 0 || H <= 0 || B <= 0 || !bytes) return fail(IMDBN_E_INVALID, "factor_compact_bytes: bad argument");
    (void)wire_layout(V, H, B, binary_data, bytes);
    return 0;
}

// any value >= 1 that changes from call to call (a stale `bad` mark must not match; the caller zero-initialises a fresh compact buffer)
static int next_pack_epoch() {
    static std::atomic<int> epoch{0};
    return epoch.fetch_add(1) % 1000000 + 1;
}

int imdbn_rbm_pack_factors(int V, int H, int B, int binary_data, const void* block, void* compact, imdbn_stream_t stream) {
    if (V <= 0 || H <= 0 || B <= 0 || !block || !compact || (((uintptr_t)block | (uintptr_t)compact) & 15))
        return fail(IMDBN_E_INVALID, "pack_factors: bad argument (blocks must be 16-B aligned)");
    FactorWireArgs w = wire_layout(V, H, B, binary_data, nullptr);
    w.src = (const char*)block; w.dst = (char*)compact; w.n_ranks = 1;
    w.epoch = next_pack_epoch();
    hipLaunchKernelGGL(factor_pack, dim3(std::min(1024, cdiv((int)(w.head_bytes / 16), 256))), dim3(256), 0, S(stream), w);
    HIPCHK(hipGetLastError());
    return 0;
}

int imdbn_rbm_unpack_factors(int V, int H, int B, int binary_data, const void* compact, size_t compact_stride, int n_ranks,
                             void* gathered, size_t full_stride, int planes_only, imdbn_stream_t stream) {
    if (V <= 0 || H <= 0 || B <= 0 || !compact || !gathered || n_ranks < 1 || (((uintptr_t)compact | (uintptr_t)gathered | compact_stride | full_stride) & 15))
        return fail(IMDBN_E_INVALID, "unpack_factors: bad argument (blocks and strides must be 16-B aligned)");
    size_t cb = 0;
    FactorWireArgs w = wire_layout(V, H, B, binary_data, &cb);
    size_t off = 0, fbytes = 0;
    CHK(imdbn_factor_block(V, H, B, &off, &fbytes));
    if (compact_stride < cb || full_stride < fbytes) return fail(IMDBN_E_INVALID, "unpack_factors: strides smaller than the blocks");
    w.src = (const char*)compact; w.dst = (char*)gathered; w.src_stride = compact_stride; w.dst_stride = full_stride; w.n_ranks = n_ranks;
    w.planes_only = planes_only ? 1 : 0;
    hipLaunchKernelGGL(factor_unpack, dim3(std::min(512, cdiv((int)((planes_only ? (size_t)V * w.Bp / 8 * 16 : w.head_bytes) / 16), 256)), n_ranks), dim3(256), 0, S(stream), w);
    HIPCHK(hipGetLastError());
    return 0;
}

int imdbn_rbm_cd_factors(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* o,
                         imdbn_rng* rng, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!data || !o || ldd < d->V) return fail(IMDBN_E_INVALID, "cd_factors: bad argument");
    if (!factor_mode_ok(d, B, false)) return fail(IMDBN_E_UNSUPPORTED, "cd_factors: needs <= 64 rows per rank, 16-B aligned weight rows, no softmax groups");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(cd_phases(c, data, ldd, o));
    return c.rng.finish();
}

// The CD pass of this rank's rows straight into the wire form (cd_factors + pack_factors as one call), with the next-batch
// preparation of imdbn_rbm_cd_step (imdbn_cd_opts.next_* / data_slot): the data-side factors then sit in a prefetch slot and
// the pack kernel reads them from there.
int imdbn_rbm_cd_factors_wire(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* o, imdbn_rng* rng,
                              int binary_data, void* wire, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!data || !o || !wire || ldd < d->V || (((uintptr_t)wire) & 15)) return fail(IMDBN_E_INVALID, "cd_factors_wire: bad argument");
    if (!factor_mode_ok(d, B, false)) return fail(IMDBN_E_UNSUPPORTED, "cd_factors_wire: needs <= 64 rows per rank, 16-B aligned weight rows, no softmax groups");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    PrepArgs pn;
    bool rides = false;
    // (a wire form that carries the data as three bf16 planes needs all three written: no compact slot form then)
    CHK(cd_prologue(c, o, pn, rides, binary_data != 0));
    CHK(cd_phases(c, data, ldd, o, rides ? &pn : nullptr));
    CHK(c.rng.finish());
    FactorWireArgs w = wire_layout(d->V, d->H, B, binary_data, nullptr);
    w.src = (const char*)ws + c.L.fb_off; w.dst = (char*)wire; w.n_ranks = 1;
    if (o->data_slot) { w.alt_flags = (const char*)c.L.flags; w.alt_cs_vpos = (const char*)c.L.cs_vpos; w.alt_vpos = (const char*)c.L.vis_tr[0]; }
    w.epoch = next_pack_epoch();
    hipLaunchKernelGGL(factor_pack, dim3(std::min(1024, cdiv((int)(w.head_bytes / 16), 256))), dim3(256), 0, c.s, w);
    HIPCHK(hipGetLastError());
    return 0;
}

// The update from n_ranks factor blocks.  `head` / `planes`: where the blocks' head (exactness map, hidden planes, column-sum
// and error partials) and visible planes are read -- the same buffer and stride for full gathered blocks, or the gathered
// WIRE blocks (whose head is verbatim) plus the buffer imdbn_rbm_unpack_factors(planes_only) expanded the planes into.
static int apply_factors_impl(const imdbn_rbm_desc* d, const void* head, size_t head_stride, const void* planes, size_t planes_stride,
                              int n_ranks, int rows_per_rank, int global_B, const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    if (!head || !planes || !o || n_ranks < 1 || global_B <= 0) return fail(IMDBN_E_INVALID, "apply_factors: bad argument");
    if (!factor_mode_ok(d, rows_per_rank, true)) return fail(IMDBN_E_UNSUPPORTED, "apply_factors: needs <= 64 rows per rank, 16-B aligned weight rows, no softmax groups");
    Ctx c(d, nullptr, S(stream));
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 30);                   // only offsets inside the factor block are used
    c.L = make_layout(d->V, d->H, rows_per_rank, fake);
    const Layout& L = c.L;
    const char* fb = fake + L.fb_off;
    if (((head_stride | planes_stride) & 255) || ((((uintptr_t)head) | ((uintptr_t)planes)) & 255) || planes_stride < L.fb_bytes ||
        head_stride < (size_t)((const char*)L.vis_tr[0] - fb))
        return fail(IMDBN_E_INVALID, "apply_factors: the gathered blocks must be 256-B aligned and at least %zu bytes apart", L.fb_bytes);
    auto at_h = [&](const void* layout_ptr, int rk) { return (const char*)head + ((const char*)layout_ptr - fb) + (size_t)rk * head_stride; };
    auto at_v = [&](const void* layout_ptr, int rk) { return (const char*)planes + ((const char*)layout_ptr - fb) + (size_t)rk * planes_stride; };
    AssocPlanesArgs f;
    memset(&f, 0, sizeof(f));
    f.W = d->W; f.Wm = d->W_m; f.ldw = d->ldw; f.V = L.V; f.H = L.H;
    f.vpos_terms = c.nw == 1 ? 1 : 0; f.vneg_terms = 1;
    f.vts = (int64_t)L.V * L.Bp; f.hts = (int64_t)L.H * L.Bp; f.Bp = L.Bp;
    f.lr = o->lr; f.mom = o->momentum; f.wd = o->weight_decay; f.n = (float)global_B;
    BiasArgs b;
    memset(&b, 0, sizeof(b));
    b.hid_bias = d->hid_bias; b.hb_m = d->hb_m; b.H = L.H; b.vis_bias = d->vis_bias; b.vb_m = d->vb_m; b.V = L.V;
    b.hpos = (const float*)at_h(L.cs_hpos, 0); b.hneg = (const float*)at_h(L.cs_hneg, 0);
    b.vpos = (const float*)at_h(L.cs_vpos, 0); b.vneg = (const float*)at_h(L.cs_vneg, 0);
    b.P = L.P; b.lr = o->lr; b.mom = o->momentum; b.n = (float)global_B;
    b.sparsity = o->sparsity ? 1 : 0; b.target = o->sparsity_target;
    b.loss_part = (const float*)at_h(L.loss_part, 0); b.n_loss = n_loss_used(c, false);
    b.loss_den = (float)global_B * (float)L.V; b.loss_out = loss_out;
    b.R = n_ranks; b.rs = (int64_t)(head_stride / 4);
    BiasArgs bz;
    memset(&bz, 0, sizeof(bz));
    const int nh = cdiv(L.H, 128), nv = cdiv(L.V, 128);
    const int tpb = std::max(1, cdiv(nh * nv, std::max(cu_count(), 1)));
    const int brows = nh >= 2 ? 1 : 2;
    // (bench.py roofline at N > 1: the update kernel bracketed with HIP events on its stream, every 4th call)
    const bool prof = g_prof.on && (g_prof.calls++ % 8 == 3) && g_prof.used + 2 <= g_prof.ev.size();
    if (prof) HIPCHK(hipEventRecord(g_prof.ev[g_prof.used], c.s));
    auto prof_end = [&]() -> int { if (prof) { HIPCHK(hipEventRecord(g_prof.ev[g_prof.used + 1], c.s)); g_prof.used += 2; } return 0; };
    // all rank blocks inside one launch (the weights move once) when the visible operands need <= 4 plane slices
    if (n_ranks >= g_min_rank_loop && !g_no_rank_loop && f.vneg_terms == 1) {
        f.vpos = (const bf16_t*)at_v(L.vis_tr[0], 0); f.vpos_flag = (const int*)at_h(L.flags, 0);
        f.hpos = (const bf16_t*)at_h(L.hid_tr[0], 0);
        f.vneg = (const bf16_t*)at_v(L.vis_tr[1], 0); f.hneg = (const bf16_t*)at_h(L.hid_tr[1], 0);
        RankLoopArgs rl{n_ranks, (int64_t)(head_stride / 2), (int64_t)(planes_stride / 2)};
        dim3 g(nh, cdiv(nv, tpb) + brows);
        const bool acc = tpb <= 4 && !g_no_rank_acc;       // rank loop outside the tile loop: hidden planes staged once per rank
        if (c.ht == 3) { if (acc) hipLaunchKernelGGL((assoc_update_planes_ranks<3, true>), g, dim3(256), 0, c.s, f, rl, tpb, b, brows);
                         else     hipLaunchKernelGGL((assoc_update_planes_ranks<3, false>), g, dim3(256), 0, c.s, f, rl, tpb, b, brows); }
        else           { if (acc) hipLaunchKernelGGL((assoc_update_planes_ranks<1, true>), g, dim3(256), 0, c.s, f, rl, tpb, b, brows);
                         else     hipLaunchKernelGGL((assoc_update_planes_ranks<1, false>), g, dim3(256), 0, c.s, f, rl, tpb, b, brows); }
        HIPCHK(hipGetLastError());
        return prof_end();
    }
    for (int rk = 0; rk < n_ranks; ++rk) {
        f.vpos = (const bf16_t*)at_v(L.vis_tr[0], rk); f.vpos_flag = (const int*)at_h(L.flags, rk);
        f.hpos = (const bf16_t*)at_h(L.hid_tr[0], rk);
        f.vneg = (const bf16_t*)at_v(L.vis_tr[1], rk); f.hneg = (const bf16_t*)at_h(L.hid_tr[1], rk);
        const int pass = n_ranks == 1 ? 0 : (rk == 0 ? 1 : (rk == n_ranks - 1 ? 3 : 2));
        const int br = (rk == n_ranks - 1) ? brows : 0;
        const BiasArgs& bb = br ? b : bz;
        dim3 g(nh, cdiv(nv, tpb) + br);
#define LAUNCH_K3F(HTV, PS) hipLaunchKernelGGL((assoc_update_planes<0, HTV, PS>), g, dim3(256), 0, c.s, f, tpb, bb, br)
#define LAUNCH_K3F_P(HTV) do { if (pass == 0) LAUNCH_K3F(HTV, 0); else if (pass == 1) LAUNCH_K3F(HTV, 1); \
                               else if (pass == 2) LAUNCH_K3F(HTV, 2); else LAUNCH_K3F(HTV, 3); } while (0)
        if (c.ht == 3) LAUNCH_K3F_P(3); else LAUNCH_K3F_P(1);
#undef LAUNCH_K3F_P
#undef LAUNCH_K3F
    }
    HIPCHK(hipGetLastError());
    return prof_end();
}

int imdbn_rbm_apply_factors(const imdbn_rbm_desc* d, const void* gathered, int n_ranks, size_t rank_stride, int rows_per_rank,
                            int global_B, const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream) {
    return apply_factors_impl(d, gathered, rank_stride, gathered, rank_stride, n_ranks, rows_per_rank, global_B, o, loss_out, stream);
}

int imdbn_rbm_apply_factors_wire(const imdbn_rbm_desc* d, const void* wire, size_t wire_stride, const void* planes, size_t planes_stride,
                                 int n_ranks, int rows_per_rank, int global_B, const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream) {
    return apply_factors_impl(d, wire, wire_stride, planes, planes_stride, n_ranks, rows_per_rank, global_B, o, loss_out, stream);
}

// unpack(planes_only) + apply_factors_wire as one call: `planes` = scratch of n_ranks x planes_stride bytes (>= the full block size)
int imdbn_rbm_apply_wire(const imdbn_rbm_desc* d, const void* wire, size_t wire_stride, int n_ranks, int rows_per_rank, int global_B,
                         int binary_data, void* planes, size_t planes_stride, const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    CHK(imdbn_rbm_unpack_factors(d->V, d->H, rows_per_rank, binary_data, wire, wire_stride, n_ranks, planes, planes_stride, 1, stream));
    return apply_factors_impl(d, wire, wire_stride, planes, planes_stride, n_ranks, rows_per_rank, global_B, o, loss_out, stream);
}

int imdbn_rbm_apply_delta(const imdbn_rbm_desc* d, const float* packed, int global_B, const imdbn_cd_opts* o,
                          float* loss_out, imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    if (!packed || !o || global_B <= 0) return fail(IMDBN_E_INVALID, "apply_delta: bad argument");
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.W = d->W; a.Wm = d->W_m; a.ldw = d->ldw; a.V = d->V; a.H = d->H; a.packed = packed;
    a.hid_bias = d->hid_bias; a.hb_m = d->hb_m; a.vis_bias = d->vis_bias; a.vb_m = d->vb_m;
    a.lr = o->lr; a.mom = o->momentum; a.wd = o->weight_decay; a.n = (float)global_B;
    a.sparsity = o->sparsity; a.target = o->sparsity_target; a.loss_out = loss_out;
    const int64_t total = (int64_t)d->V * d->H;
    const bool vec4 = d->H % 4 == 0 && d->ldw % 4 == 0 && ((((uintptr_t)d->W) | ((uintptr_t)d->W_m) | ((uintptr_t)packed)) & 15) == 0;
    const int64_t items = vec4 ? total / 4 : total;
    const int grid = (int)std::max<int64_t>(std::min<int64_t>((items + 255) / 256, 8192), cdiv(std::max(d->V, d->H), 256));
    if (vec4) hipLaunchKernelGGL(apply_delta<true>, dim3(grid), dim3(256), 0, S(stream), a);
    else      hipLaunchKernelGGL(apply_delta<false>, dim3(grid), dim3(256), 0, S(stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}

int imdbn_rbm_chain(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                    int init_uniform, int n_steps, const imdbn_chain_step* steps, const float* mu, int64_t ldmu, int Dz,
                    imdbn_rng* rng, float* out_v, int64_t ldo, void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v_known || !mask || !out_v || ldk < d->V || ldo < d->V) return fail(IMDBN_E_INVALID, "chain: bad argument");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(run_chain(c, ChainSpec{v_known, mask, ldk, init_uniform, n_steps, steps, mu, ldmu, Dz, out_v, ldo}, false));
    return c.rng.finish();
}

// Two independent chains of the same RBM and batch size in one call (imdbn.py:419-449: the IMG->TXT and TXT->IMG chains of
// _cross_reconstruct share nothing but the read-only weights).  Draws are assigned in the order of two imdbn_rbm_chain calls (a, then b),
// so the results are those of the two calls, bit for bit; where the row-parallel chain kernel applies both run in ONE launch
// (chain a on the first half of the grid, chain b on the second).
int imdbn_rbm_chain_pair(const imdbn_rbm_desc* d, int B, const imdbn_chain_spec* a, const imdbn_chain_spec* b, imdbn_rng* rng,
                         void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!a || !b) return fail(IMDBN_E_INVALID, "chain_pair: null chain");
    for (const imdbn_chain_spec* s : {a, b})
        if (!s->v_known || !s->mask || !s->out_v || s->ldk < d->V || s->ldo < d->V || s->n_steps < 0 || (s->n_steps > 0 && !s->steps) ||
            (s->mu && (s->Dz <= 0 || s->Dz > d->V)))
            return fail(IMDBN_E_INVALID, "chain_pair: bad argument");
    if (a->out_v == b->out_v) return fail(IMDBN_E_INVALID, "chain_pair: the two chains need separate output buffers");
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    const ChainSpec sa{a->v_known, a->mask, a->ldk, a->init_uniform, a->n_steps, a->steps, a->mu, a->ldmu, a->Dz, a->out_v, a->ldo};
    const ChainSpec sb{b->v_known, b->mask, b->ldk, b->init_uniform, b->n_steps, b->steps, b->mu, b->ldmu, b->Dz, b->out_v, b->ldo};
    if (chain_kernel_ok(c, sa.n_steps) && chain_kernel_ok(c, sb.n_steps) && sa.n_steps + sb.n_steps <= CHAIN_MAX_STEPS && !g_no_chain_pair) {
        CHK(chain_init(c, sa, false, false));
        CHK(chain_records(c, sa, 0));
        CHK(chain_init(c, sb, false, false));
        CHK(chain_records(c, sb, sa.n_steps));
        CHK(launch_k4(c, sa, 0, &sb, sa.n_steps));
        c.hid_bits_ok = false;
    } else {
        CHK(run_chain(c, sa, false));
        CHK(run_chain(c, sb, false));
    }
    return c.rng.finish();
}

// rbm.py:443-471: v+ by conditional inference, H+, CD-k from v+ (optionally re-clamped / sampled), H-.
// Leaves the statistics operands in the workspace exactly as cd_phases does.
static int clamped_phases(Ctx& c, const float* v_known, const float* mask, int64_t ldk, int n_init,
                          const imdbn_chain_step* init_steps, const float* mu, int64_t ldmu, int Dz, const imdbn_cd_opts* o) {
    const imdbn_rbm_desc* d = c.d;
    const int B = c.L.B;
    const Layout& L = c.L;
    float* vplus = L.f_v[0];
    // positive phase: v+ by conditional inference (rbm.py:443-453), H+ = up(v+) (:455)
    CHK(run_chain(c, ChainSpec{v_known, mask, ldk, 1, n_init, init_steps, mu, ldmu, Dz, vplus, (int64_t)L.V}, true));
    for (int it = 0; it < o->cd_k; ++it) {
        {   // h_prob = up(v_neg) ; first iteration: v_neg == v+ so this is also H+
            FinishArgs f = new_finish();
            if (o->sample_h) { f.vmode = 1; f.uni = c.rng.floats(B, L.H); }
            f.op.rm = L.hid_rm; f.op.rm_terms = o->sample_h ? 1 : c.rt; f.rm_src = o->sample_h ? 2 : 1;
            if (it == 0) {
                f.op.tr = L.hid_tr[0]; f.op.tr_terms = c.ht; f.tr_src = 1;
                f.colsum_part = L.cs_hpos; f.colsum_src = 1;
            }
            CHK(prop(c, true, OpIn{it == 0 ? L.vis_rm[0] : L.vis_rm[1], (it > 0 && o->sample_v) ? 1 : c.rt, nullptr}, f));
        }
        {   // v_neg = down(h) [re-clamped] [sampled]   (rbm.py:463-469)
            const bool last = (it == o->cd_k - 1);
            FinishArgs f = new_finish();
            if (o->reclamp_negative) { f.clamp = 1; f.vk = v_known; f.mask = mask; f.ldk = ldk; }
            if (o->sample_v) { f.vmode = 2; f.uni = c.rng.floats(B, L.V); c.rng.cats(B, d->n_groups, &f.cat_tape, &f.cat_uni); }
            f.out_prob = L.f_vp; f.ld_prob = L.V;
            f.out_final = L.f_v[1]; f.ld_final = L.V;
            f.op.rm = L.vis_rm[1]; f.op.rm_terms = o->sample_v ? 1 : c.rt; f.rm_src = 2;
            if (last) {
                f.op.tr = L.vis_tr[1]; f.op.tr_terms = o->sample_v ? 1 : c.rt; f.tr_src = 2;
                f.colsum_part = L.cs_vneg; f.colsum_src = 2;
                f.loss_ref = vplus; f.ld_ref = L.V; f.loss_src = 2; f.loss_part = L.loss_part;
            }
            CHK(prop(c, false, OpIn{L.hid_rm, o->sample_h ? 1 : c.rt, nullptr}, f));
        }
    }
    {   // H- = up(v_neg)  (rbm.py:471)
        FinishArgs f = new_finish();
        f.op.tr = L.hid_tr[1]; f.op.tr_terms = c.ht; f.tr_src = 1; f.op.tr_negate = 1;
        f.colsum_part = L.cs_hneg; f.colsum_src = 1;
        CHK(prop(c, true, OpIn{L.vis_rm[1], o->sample_v ? 1 : c.rt, nullptr}, f));
    }
    return 0;
}

int imdbn_rbm_clamped_step(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                           int n_init, const imdbn_chain_step* init_steps, const float* mu, int64_t ldmu, int Dz,
                           const imdbn_cd_opts* o, imdbn_rng* rng, float* loss_out, void* ws, size_t ws_bytes,
                           imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    if (!v_known || !mask || !o || ldk < d->V) return fail(IMDBN_E_INVALID, "clamped_step: bad argument");
    if (o->cd_k < 1) return fail(IMDBN_E_INVALID, "CD=%d", o->cd_k);
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(clamped_phases(c, v_known, mask, ldk, n_init, init_steps, mu, ldmu, Dz, o));
    CHK(c.rng.finish());
    const BiasArgs bias = make_bias(c, o, false, (float)B, loss_out);
    CHK(launch_assoc(c, 0, o, c.rt, nullptr, o->sample_v ? 1 : c.rt, (float)B, nullptr, &bias));
    return 0;
}

// data-parallel half of the clamped update ("train_epoch_clamped shards the same way"): the rank's
// un-normalised statistics in the packed layout of imdbn_rbm_cd_stats; all-reduce, then imdbn_rbm_apply_delta
// (with sparsity off: the clamped update has no sparsity term, rbm.py:473-481).
int imdbn_rbm_clamped_stats(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                            int n_init, const imdbn_chain_step* init_steps, const float* mu, int64_t ldmu, int Dz,
                            const imdbn_cd_opts* o, imdbn_rng* rng, float* packed, void* ws, size_t ws_bytes,
                            imdbn_stream_t stream) {
    CHK(check_desc(d, false));
    if (!v_known || !mask || !o || !packed || ldk < d->V) return fail(IMDBN_E_INVALID, "clamped_stats: bad argument");
    if (o->cd_k < 1) return fail(IMDBN_E_INVALID, "CD=%d", o->cd_k);
    Ctx c(d, rng, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    CHK(clamped_phases(c, v_known, mask, ldk, n_init, init_steps, mu, ldmu, Dz, o));
    CHK(c.rng.finish());
    const BiasArgs pack = make_pack(c, packed);
    CHK(launch_assoc(c, 1, o, c.rt, nullptr, o->sample_v ? 1 : c.rt, 1.0f, packed, &pack));
    return 0;
}


// ---- K3 alone: rbm.py:209-224 from caller-supplied phase tensors ------------------------------------------------------
int imdbn_rbm_assoc_update(const imdbn_rbm_desc* d, const float* vpos, int64_t ldvp, const float* hpos, int64_t ldhp,
                           const float* vneg, int64_t ldvn, const float* hneg, int64_t ldhn, int B, const imdbn_cd_opts* o,
                           void* ws, size_t ws_bytes, imdbn_stream_t stream) {
    CHK(check_desc(d, true));
    if (!vpos || !hpos || !vneg || !hneg || !o || ldvp < d->V || ldvn < d->V || ldhp < d->H || ldhn < d->H)
        return fail(IMDBN_E_INVALID, "assoc_update: bad tensor argument");
    Ctx c(d, nullptr, S(stream));
    CHK(setup(c, B, ws, ws_bytes));
    const Layout& L = c.L;
    // operand planes + column sums of the four tensors (exact three-term planes wherever a value is not exactly bf16)
    CHK(prep(c, vpos, ldvp, L.V, nullptr, L.Vpad, L.vis_tr[0], L.flags, L.cs_vpos, c.rt));
    CHK(prep(c, vneg, ldvn, L.V, nullptr, L.Vpad, L.vis_tr[1], nullptr, L.cs_vneg, c.rt));
    CHK(prep(c, hpos, ldhp, L.H, nullptr, L.Hpad, L.hid_tr[0], L.flags_h, L.cs_hpos, c.ht));
    {
        PrepArgs p;
        memset(&p, 0, sizeof(p));
        p.in = hneg; p.ld = ldhn; p.B = L.B; p.Bp = L.Bp; p.N = L.H;
        p.op.ldrm = L.Hpad; p.op.Bp = L.Bp; p.op.tr = L.hid_tr[1]; p.op.tr_ts = (int64_t)L.H * L.Bp; p.op.tr_terms = c.ht; p.op.tr_negate = 1;
        p.colsum_part = L.cs_hneg;
        hipLaunchKernelGGL(prep_operand, dim3(cdiv(L.Hpad, 64), L.P), dim3(256), 0, c.s, p);
        HIPCHK(hipGetLastError());
    }
    BiasArgs bias = make_bias(c, o, o->sparsity != 0, (float)B, nullptr);
    bias.loss_part = nullptr; bias.n_loss = 0;
    CHK(launch_assoc(c, 0, o, c.nw == 1 ? 1 : 0, L.flags, c.rt, (float)B, nullptr, &bias));
    return 0;
}

// ---- C1: collectives over RCCL for callers that do not go through torch.distributed ------------------------------------
// librccl is opened on first use (dlopen: the engine has no link-time dependency on it; a process that already runs
// torch.distributed's nccl backend gets that same library).  One communicator per (process, GPU); the caller moves the
// 128-byte unique id from rank 0 to the other ranks by whatever channel it has.
namespace {
struct Id128 { char b[128]; };      // ncclUniqueId, passed by value
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
} g_rccl;

int rccl_load() {
    if (g_rccl.h) return 0;
    void* h = nullptr;
    for (const char* nm : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(IMDBN_E_UNSUPPORTED, "librccl not found: %s", dlerror());
    auto sym = [&](const char* n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.AllGather)
        return fail(IMDBN_E_UNSUPPORTED, "librccl lacks an expected symbol");
    g_rccl.h = h;
    return 0;
}
int rccl_fail(int rc, const char* what) {
    return fail(rc > 0 ? rc : IMDBN_E_INVALID, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

int imdbn_comm_unique_id(void* id128) {
    if (!id128) return fail(IMDBN_E_INVALID, "comm_unique_id: null buffer");
    CHK(rccl_load());
    const int rc = g_rccl.GetUniqueId(id128);
    return rc ? rccl_fail(rc, "ncclGetUniqueId") : 0;
}

int imdbn_comm_init(void** comm, int world, int rank, const void* id128) {
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return fail(IMDBN_E_INVALID, "comm_init: bad argument");
    CHK(rccl_load());
    Id128 id;
    memcpy(id.b, id128, 128);
    const int rc = g_rccl.CommInitRank(comm, world, id, rank);
    return rc ? rccl_fail(rc, "ncclCommInitRank") : 0;
}

int imdbn_comm_destroy(void* comm) {
    if (!comm) return 0;
    CHK(rccl_load());
    const int rc = g_rccl.CommDestroy(comm);
    return rc ? rccl_fail(rc, "ncclCommDestroy") : 0;
}

int imdbn_allreduce_sum_f32(void* comm, float* buf, size_t count, imdbn_stream_t stream) {
    if (!comm || !buf) return fail(IMDBN_E_INVALID, "allreduce: bad argument");
    CHK(rccl_load());
    const int rc = g_rccl.AllReduce(buf, buf, count, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, S(stream));
    return rc ? rccl_fail(rc, "ncclAllReduce") : 0;
}

int imdbn_allgather_bytes(void* comm, const void* send, void* recv, size_t bytes_per_rank, imdbn_stream_t stream) {
    if (!comm || !send || !recv) return fail(IMDBN_E_INVALID, "allgather: bad argument");
    CHK(rccl_load());
    const int rc = g_rccl.AllGather(send, recv, bytes_per_rank, /* ncclUint8 */ 1, comm, S(stream));
    return rc ? rccl_fail(rc, "ncclAllGather") : 0;
}

}  // extern "C"
