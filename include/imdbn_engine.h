/*
 * imdbn_engine.h -- C ABI of the MI355X (gfx950) contrastive-divergence engine.
 *
 * This is the drop-in boundary for the ONE hot path of francesco-cal98/multimodal-idbn
 * (SURVEY.md section 8).  The reference has no FFI: the path sits behind Python methods that
 * dispatch ATen ops.  Each entry point below replaces the ATen sequence of one reference
 * method (citations: file:line under /root/reference/imdbn/models/).  The Python classes in
 * multimodal-idbn_amd/imdbn/models/ keep the reference's signatures and call these through
 * ctypes (cffi, which BASELINE.json names, is not installed in the image).
 *
 * Conventions
 *   - extern "C", no exceptions cross the boundary; every function returns int:
 *       0 = ok, <0 = IMDBN_E_* below, >0 = hipError_t passed through.
 *     imdbn_last_error() returns the thread-local message of the last failure.
 *   - The caller (PyTorch) owns every buffer.  Pointers are raw DEVICE pointers
 *     (tensor.data_ptr()), fp32 row-major with an explicit leading dimension in ELEMENTS.
 *     The library allocates nothing persistent; scratch comes from the caller's workspace
 *     (imdbn_ws_bytes() tells how much; contents need not survive between calls).
 *   - Every launch goes to the hipStream_t passed in (torch.cuda.current_stream().cuda_stream);
 *     no call synchronises the host or touches another stream.  Graph-capturable.
 *   - Parameters are read through the descriptor at EVERY call -- the engine keeps no copy, so
 *     callers may mutate or re-bind W / biases between calls.
 *   - Randomness: imdbn_rng says where draws come from, in the reference's draw order
 *     (SURVEY.md Appendix B).  REPLAY consumes caller-recorded draws (parity tests);
 *     PHILOX generates Philox-4x32-10 keyed on (seed, draw number, GLOBAL row, column) so
 *     results do not depend on tiling or on data-parallel sharding.
 */
#ifndef IMDBN_ENGINE_H
#define IMDBN_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMDBN_ABI_VERSION 4

/* error codes (negative) */
#define IMDBN_E_INVALID   (-1)   /* bad argument (shape, null pointer, alignment) */
#define IMDBN_E_WORKSPACE (-2)   /* workspace too small */
#define IMDBN_E_RNG       (-3)   /* replay tape exhausted */
#define IMDBN_E_NODEVICE  (-4)   /* no gfx950 device */
#define IMDBN_E_UNSUPPORTED (-5)

/* arithmetic mode of the propagations / association products */
#define IMDBN_PARITY_F32 0   /* fp32 master weights split hi+mid+lo bf16 in registers: exact products */
#define IMDBN_FAST_BF16  1   /* one bf16 term per operand (throughput mode; not parity-grade) */

#define IMDBN_RNG_PHILOX 0
#define IMDBN_RNG_REPLAY 1

#define IMDBN_MAX_GROUPS 4

typedef void* imdbn_stream_t;    /* hipStream_t */

/* One RBM's parameters (rbm.py:41-79).  W is [V][ldw] with ldw >= H. */
typedef struct imdbn_rbm_desc {
    float*   W;
    int64_t  ldw;
    float*   hid_bias;      /* [H] */
    float*   vis_bias;      /* [V] */
    float*   W_m;           /* [V][ldw] momentum buffers; may be NULL for inference-only calls */
    float*   hb_m;          /* [H] */
    float*   vb_m;          /* [V] */
    int32_t  V, H;
    int32_t  mode;          /* IMDBN_PARITY_F32 | IMDBN_FAST_BF16 */
    int32_t  n_groups;      /* softmax groups over visible columns (rbm.py:66,113-114) */
    int32_t  group_start[IMDBN_MAX_GROUPS];
    int32_t  group_end[IMDBN_MAX_GROUPS];
} imdbn_rbm_desc;

/* Where random draws come from.  The *_used fields are OUTPUTS: how much the call consumed. */
typedef struct imdbn_rng {
    int32_t  mode;          /* IMDBN_RNG_PHILOX | IMDBN_RNG_REPLAY */
    int32_t  _pad;
    uint64_t seed;          /* PHILOX key */
    uint64_t offset;        /* PHILOX: number of the first draw tensor of this call */
    int64_t  row0;          /* PHILOX: global index of local row 0 (data-parallel shard offset) */
    const float*   tape;    /* REPLAY: device floats, uniform / normal draw tensors back to back */
    int64_t        tape_len;
    const int32_t* cat_tape;/* REPLAY: device int32, one index per (categorical draw, row) */
    int64_t        cat_len;
    int64_t  tape_used;     /* out */
    int64_t  cat_used;      /* out */
    uint64_t draws_used;    /* out: draw tensors consumed (advance `offset` by this) */
    const uint64_t* dev_offset; /* PHILOX, nullable: a device-resident counter the kernels ADD to `offset` when they run.  A call
                               sequence captured into a hipGraph bakes `offset` in; with dev_offset set and an
                               imdbn_rng_advance(dev_offset, draws_used, stream) node behind it, every replay of the graph draws
                               fresh numbers -- exactly those the same calls would draw issued one by one. */
} imdbn_rng;

/* One half-step pair v -> h -> v' of a conditional chain
 * (rbm.py:275-291 annealed Gibbs, :337-365 noisy mean-field, :393-399 plain Gibbs). */
typedef struct imdbn_chain_step {
    float   T;              /* temperature of both half steps (max(1e-6,T) applied by callee) */
    float   sigma;          /* std of Gaussian noise added to both logits; 0 = none, no draw */
    float   eta;            /* mu-pull weight on columns [0,Dz) (rbm.py:359-363); 0 = off */
    int32_t sample_h;       /* 1: h = 1[p_h > U] ; 0: h = p_h */
    int32_t vmode;          /* 0: v = p_v ; 1: v = sampleV(p_v) ; 2: v = sampleV(mix(p_v)) w/o re-mix */
    int32_t clamp;          /* 1: v = v*(1-mask) + v_known*mask (masks must be 0/1) */
} imdbn_chain_step;

enum { IMDBN_DATA_UNKNOWN = 0, IMDBN_DATA_BINARY = 1, IMDBN_DATA_REAL = 2 };      /* imdbn_cd_opts.data_binary / next_binary */

/* Options of one CD update (rbm.py:181-227, :403-483). */
typedef struct imdbn_cd_opts {
    int32_t cd_k;           /* Gibbs steps of the negative phase (>=1) */
    float   lr;             /* effective learning rate: lr/(1+0.01*epoch) [* aux_lr_mult] (rbm.py:194,476) */
    float   momentum;       /* momentum or final_momentum (rbm.py:195) */
    float   weight_decay;
    int32_t sparsity;       /* rbm.py:217-219 (train_epoch only) */
    float   sparsity_target;
    int32_t sample_h;       /* clamped step only (rbm.py:462) */
    int32_t sample_v;       /* clamped step only (rbm.py:468) */
    int32_t reclamp_negative; /* clamped step only (rbm.py:464) */
    /* -- next-batch prefetch (imdbn_rbm_cd_step only; all zero = off).  The operand forms of a batch are a pure
     * function of the batch: cd_step can prepare those of the FOLLOWING batch with extra blocks of its first
     * negative-phase launch, into prefetch slot 1 or 2 of the workspace, and a later cd_step on the same workspace is told that
     * its `data` already sits in that slot.  The caller guarantees the batch was not modified in between. */
    const float* next_data; /* [B][V] fp32 batch to prepare during this call (NULL: none); honoured only when
                               imdbn_rbm_prefetch_ok() says so for this descriptor and batch size */
    int64_t ld_next;
    int32_t next_slot;      /* 1 or 2: where to put it (must differ from data_slot) */
    int32_t data_slot;      /* 0: prepare `data` now (default); 1 / 2: `data` was prefetched into that slot */
    /* -- what the caller knows about the VALUES of `data` (IMDBN_DATA_*).  Nothing has to be known: with IMDBN_DATA_UNKNOWN (0)
     * the device decides, per 64-column piece of the batch and from the exactness map its own preparation kernel writes, whether the
     * positive phase reads that piece as a bit plane (all values 0 / 1: binary images) or as bf16 terms; the result is the same
     * number either way, so the caller never has to inspect a batch (no device reduction, no host synchronisation).
     * IMDBN_DATA_BINARY (1): the caller asserts every element is exactly 0 or 1; the assertion is checked on the device and a
     * batch that is not binary turns the update into NaN instead of being silently truncated.  IMDBN_DATA_REAL (2): real values
     * (the output of another layer): no bit plane is made. */
    int32_t data_binary;
    /* -- the same for `next_data`.  UNKNOWN: every 64-column x 64-row piece that turns out to be all 0 / 1 is prepared in the slim
     * form (bit plane, exactness map, column sums, one bf16 plane), any other piece with all three-term operand forms; BINARY: the
     * slim form throughout.  Honoured where the positive phase can read bit planes (16-B aligned weight rows, V > 1024); the later
     * cd_step on that batch must pass the same value in data_binary. */
    int32_t next_binary;
    /* -- forward pass fused behind the update (imdbn_rbm_cd_step only; NULL = off): after the weights are updated, the
     * hidden probabilities sigmoid(data @ W + hid_bias) of the SAME batch under the NEW weights are written to
     * fwd_out[B][H] (row stride ld_fwd floats) -- the `train_epoch(v); v = forward(v)` pair of the layer loop
     * (idbn.py:195-204) as one call: the batch's operand forms are still in the workspace, so nothing is prepared twice. */
    float*  fwd_out;
    int64_t ld_fwd;
} imdbn_cd_opts;

/* ---- plumbing ------------------------------------------------------------------------- */
int    imdbn_version(void);
int    imdbn_last_error(char* buf, size_t n);
/* cu_count / arch name of the current device; IMDBN_E_NODEVICE without a GPU */
int    imdbn_device_info(int* cu_count, char* arch, size_t n);
/* bytes of scratch any call below needs for an RBM of V x H at batch B */
size_t imdbn_ws_bytes(int V, int H, int B);
/* tuning knobs (split-K factors); 0 = automatic */
int    imdbn_set_tuning(int ksplit_up, int ksplit_down);
/* named tuning/testing knobs (process-wide defaults): "ksplit_up", "ksplit_down", "generic_k3" (1 = force the unaligned-shape
 * K3), "k1s_ks", "k2s_rows", "down_rows", "chain_rows", "no_k1s", "no_k2s", "no_bits", "no_prefetch", "no_chain_kernel", ...;
 * none of them changes results beyond fp32 summation order, none is needed for normal use */
int    imdbn_set_option(const char* name, int value);
/* The same knobs per caller instead of per process: a handle starts as a copy of the process defaults, takes
 * imdbn_options_set(name, value) (every name of imdbn_set_option but "dbg"), and imdbn_use_options(handle) binds it to the
 * CALLING THREAD: engine calls made by that thread read the handle (NULL = back to the process defaults).  Knobs that shape
 * the workspace layout (split-K factors, tile heights) must agree between calls that share a workspace. */
typedef struct imdbn_options imdbn_options;
imdbn_options* imdbn_options_create(void);
void   imdbn_options_destroy(imdbn_options* o);
int    imdbn_options_set(imdbn_options* o, const char* name, int value);
int    imdbn_use_options(const imdbn_options* o);
/* per-kernel timing of the update kernel with HIP events on the launch stream (bench.py roofline) */
int    imdbn_profile_enable(int on);
int    imdbn_profile_read(double* total_ms, int* launches);   /* synchronises the recorded events */
/* tuning aid: copies the per-block wall-clock stamps (100 MHz ticks, 8 per block, up to 4096 blocks) that the
 * propagation kernels record when the "dbg" option enables them; synchronises the device */
int    imdbn_debug_stamps(long long* out, int n);
/* test / tuning aid: byte offset of a named internal buffer inside the workspace of an (V, H, B) call ("vis_bits0/1", "hid_bits",
 * "vis_tr0/1", "hid_tr0/1", "cs_hpos/hneg/vpos/vneg", "flags", "partial", "vis_rm1"); the layout is NOT part of the ABI */
int    imdbn_debug_ws_offset(int V, int H, int B, const char* name, size_t* offset);

/* *dev_offset += n on `stream` (see imdbn_rng.dev_offset).  Every engine call is a plain sequence of kernel launches on the caller's
 * stream -- no host synchronisation, no allocation, no memcpy -- so it can be recorded with hipStreamBeginCapture. */
int imdbn_rng_advance(uint64_t* dev_offset, uint64_t n, imdbn_stream_t stream);

/* ---- K1: p(h|v)   replaces RBM.forward (rbm.py:81-92) ---------------------------------- */
/* out_prob[B][H] = sigmoid((v W + c)/T);  out_sample (nullable) = 1[out_prob > U] */
int imdbn_rbm_prop_up(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, float T,
                      imdbn_rng* rng, float* out_prob, int64_t ldo, float* out_sample, int64_t lds,
                      void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* forward(v) at T = 1: out_prob[B][H] = sigmoid(v W + c).  data_binary = IMDBN_DATA_* as in imdbn_cd_opts (UNKNOWN: the device
 * reads every 64-column piece of v as a bit plane or as bf16 terms, whichever describes it; BINARY: asserted, checked, NaN on a
 * false promise).  Bit-identical to the fused forward of imdbn_rbm_cd_step (imdbn_cd_opts.fwd_out) for the same batch and
 * weights, whatever data_binary says. */
int imdbn_rbm_forward(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, int data_binary,
                      float* out_prob, int64_t ldo, void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- free energy   F(v) = -v.b - sum_j softplus(c_j + (vW)_j)   (imdbn/utils/energy_utils.py:19-28; the
 *      `joint_rbm.free_energy` that imdbn.py:455-474 probes for and the reference never defines) -------- */
int imdbn_rbm_free_energy(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B, float* out_F,
                          void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- K2: p(v|h)   replaces RBM.visible_probs / backward (rbm.py:94-116,137-151) -------- */
/* out_prob[B][V] = sigmoid((h W^T + b)/T) with softmax over each group; if logits_only: raw logits */
int imdbn_rbm_prop_down(const imdbn_rbm_desc* d, const float* h, int64_t ldh, int B, float T,
                        int logits_only, float* out_prob, int64_t ldo,
                        void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- sample_visible (rbm.py:118-135) ---------------------------------------------------- */
int imdbn_rbm_sample_visible(const imdbn_rbm_desc* d, const float* v_prob, int64_t ldp, int B,
                             imdbn_rng* rng, float* out, int64_t ldo, imdbn_stream_t stream);

/* ---- one Gibbs step (rbm.py:158-178): outputs v_next, v_prob, h, h_prob ------------------ */
int imdbn_rbm_gibbs_step(const imdbn_rbm_desc* d, const float* v, int64_t ldv, int B,
                         int sample_h, int sample_v, imdbn_rng* rng,
                         float* v_next, float* v_prob, float* h, float* h_prob,
                         void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- whole RBM.train_epoch (rbm.py:180-227): K1 -> [K2 -> K1]^k -> K3 -> bias/loss -------- */
/* loss_out: device float[1] = mean((data - v_prob)^2) */
int imdbn_rbm_cd_step(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B,
                      const imdbn_cd_opts* o, imdbn_rng* rng, float* loss_out,
                      void* ws, size_t ws_bytes, imdbn_stream_t stream);
/* 1 when cd_step on this descriptor / batch size honours imdbn_cd_opts.next_data (16-B aligned weight rows: the
 * float4 fused K2 carries the prefetch blocks), else 0 -- then next_data is ignored and nothing may be assumed prefetched. */
int imdbn_rbm_prefetch_ok(const imdbn_rbm_desc* d, int B);

/* ---- data-parallel split of the same update ------------------------------ */
/* packed layout (floats): [dW V*H][dc H][db V][sum P+ H][sq-err sum 1][pad to 4] */
size_t imdbn_packed_delta_floats(int V, int H);
/* K3a: un-normalised local statistics of rbm.py:199-209 into `packed` (no parameter is touched) */
int imdbn_rbm_cd_stats(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B,
                       const imdbn_cd_opts* o, imdbn_rng* rng, float* packed,
                       void* ws, size_t ws_bytes, imdbn_stream_t stream);
/* K3b: rbm.py:212-226 from the (all-reduced) packed statistics with n = global batch */
int imdbn_rbm_apply_delta(const imdbn_rbm_desc* d, const float* packed, int global_B,
                          const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream);

/* ---- data-parallel "factor exchange" (alternative to cd_stats / all-reduce / apply_delta) -----------------
 * The factors of <= 64 rows (visible / hidden operand planes, column sums, error partials: ~7 MB at 10000 x 1500)
 * are 8x smaller than the fp32 delta-W (60 MB).  Every rank: cd_factors (the CD pass; the factor block stays in its
 * workspace at imdbn_factor_block's offset), all-gather of the blocks, apply_factors (the update kernel runs once
 * per rank block; identical arithmetic on every rank, replicas stay bit-identical).
 * Needs <= 64 rows per rank, 16-B aligned weight rows, no softmax groups (IMDBN_E_UNSUPPORTED otherwise). */
int    imdbn_factor_block(int V, int H, int B, size_t* offset, size_t* bytes);
int    imdbn_rbm_cd_factors(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* opts,
                            imdbn_rng* rng, void* ws, size_t ws_bytes, imdbn_stream_t stream);
int    imdbn_rbm_apply_factors(const imdbn_rbm_desc* d, const void* gathered, int n_ranks, size_t rank_stride,
                               int rows_per_rank, int global_B, const imdbn_cd_opts* opts, float* loss_out,
                               imdbn_stream_t stream);

/* Wire form of the factor block: the negative visible sample (always 0/1) and, with binary_data != 0, the data plane
 * travel as 1 bit per element (7.0 -> 5.8 / 2.1 MB per rank at 10000 x 1500).  pack: one full block (as left by
 * imdbn_rbm_cd_factors) -> one compact block of imdbn_factor_compact_bytes(); all-gather the compact blocks;
 * unpack: n_ranks compact blocks -> n_ranks full blocks for imdbn_rbm_apply_factors.  A plane that is declared binary
 * and is not poisons the update with NaN (fails loudly).  All pointers / strides 16-B aligned; a compact buffer must be
 * zero-initialised once before its first use. */
int    imdbn_factor_compact_bytes(int V, int H, int B, int binary_data, size_t* bytes);
int    imdbn_rbm_pack_factors(int V, int H, int B, int binary_data, const void* block, void* compact, imdbn_stream_t stream);
int    imdbn_rbm_unpack_factors(int V, int H, int B, int binary_data, const void* compact, size_t compact_stride, int n_ranks,
                                void* gathered, size_t full_stride, int planes_only, imdbn_stream_t stream);
/* The two halves of a data-parallel step as ONE call each (a binder's step is: cd_factors_wire, all-gather of the wire blocks,
 * apply_wire).  cd_factors_wire = cd_factors + pack_factors, and it honours the next-batch prefetch fields of imdbn_cd_opts
 * (next_data / next_slot / data_slot / next_binary) exactly as imdbn_rbm_cd_step does.  apply_wire = unpack_factors(planes_only)
 * + apply_factors_wire; `planes`: scratch of n_ranks x planes_stride bytes, planes_stride >= imdbn_factor_block's size. */
int    imdbn_rbm_cd_factors_wire(const imdbn_rbm_desc* d, const float* data, int64_t ldd, int B, const imdbn_cd_opts* opts,
                                 imdbn_rng* rng, int binary_data, void* wire, void* ws, size_t ws_bytes, imdbn_stream_t stream);
int    imdbn_rbm_apply_wire(const imdbn_rbm_desc* d, const void* wire, size_t wire_stride, int n_ranks, int rows_per_rank,
                            int global_B, int binary_data, void* planes, size_t planes_stride, const imdbn_cd_opts* opts,
                            float* loss_out, imdbn_stream_t stream);
/* apply_factors reading the blocks' head (everything before the visible planes: verbatim in the wire form) straight
 * from the gathered wire blocks and the visible planes from the buffer unpack(planes_only = 1) expanded them into
 * (full-block layout, planes_stride apart): no copy of the 1.9 MB head per rank. */
int    imdbn_rbm_apply_factors_wire(const imdbn_rbm_desc* d, const void* wire, size_t wire_stride, const void* planes,
                                    size_t planes_stride, int n_ranks, int rows_per_rank, int global_B,
                                    const imdbn_cd_opts* o, float* loss_out, imdbn_stream_t stream);

/* ---- K4: conditional chains (rbm.py:240-400) -------------------------------------------- */
/* v0 = v_known*mask + (1-mask)*U (init_uniform=1) or v_known (0); then n_steps steps; out_v[B][V].
 * mu (nullable) is the [B][Dz] pull target of rbm.py:359-363. */
int imdbn_rbm_chain(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                    int init_uniform, int n_steps, const imdbn_chain_step* steps,
                    const float* mu, int64_t ldmu, int Dz, imdbn_rng* rng,
                    float* out_v, int64_t ldo, void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* Two independent chains of the same RBM and batch size as ONE call (iMDBN._cross_reconstruct, imdbn.py:419-449: its IMG->TXT
 * and TXT->IMG chains share nothing but the read-only weights).  Equivalent, bit for bit, to imdbn_rbm_chain(a) followed by
 * imdbn_rbm_chain(b) on the same rng; where the row-parallel chain kernel applies the two run in one launch, side by side. */
typedef struct imdbn_chain_spec {
    const float* v_known; const float* mask; int64_t ldk;   /* [B][V] clamp values and 0/1 mask (same row stride) */
    int32_t init_uniform; int32_t n_steps; const imdbn_chain_step* steps;
    const float* mu; int64_t ldmu; int32_t Dz; int32_t _pad; /* mu-pull target [B][Dz] (nullable) */
    float* out_v; int64_t ldo;                              /* final visible state [B][V] */
} imdbn_chain_spec;
int imdbn_rbm_chain_pair(const imdbn_rbm_desc* d, int B, const imdbn_chain_spec* a, const imdbn_chain_spec* b,
                         imdbn_rng* rng, void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- whole RBM.train_epoch_clamped (rbm.py:402-483) -------------------------------------- */
/* positive phase = chain(n_init steps) ; negative = cd_k steps from v+ ; update with o->lr */
int imdbn_rbm_clamped_step(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                           int n_init, const imdbn_chain_step* init_steps,
                           const float* mu, int64_t ldmu, int Dz,
                           const imdbn_cd_opts* o, imdbn_rng* rng, float* loss_out,
                           void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* data-parallel half of the clamped update (reference statistics rbm.py:455-472): this rank's
 * un-normalised statistics in the packed layout of imdbn_rbm_cd_stats -> all-reduce (sum) ->
 * imdbn_rbm_apply_delta with o->sparsity = 0 (the clamped update has no sparsity term, rbm.py:473-481). */
int imdbn_rbm_clamped_stats(const imdbn_rbm_desc* d, const float* v_known, const float* mask, int64_t ldk, int B,
                            int n_init, const imdbn_chain_step* init_steps,
                            const float* mu, int64_t ldmu, int Dz,
                            const imdbn_cd_opts* o, imdbn_rng* rng, float* packed,
                            void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- K3 alone: the weight / bias update of rbm.py:209-224 from caller-supplied phase tensors ---------------------
 * W_m <- mom W_m + lr ((vpos^T hpos - vneg^T hneg)/B - wd W) ; W += W_m ; bias updates from the column sums (with the
 * sparsity term of rbm.py:217-219 when o->sparsity).  fp32 [B][V] / [B][H] inputs; no loss is computed. */
int imdbn_rbm_assoc_update(const imdbn_rbm_desc* d, const float* vpos, int64_t ldvp, const float* hpos, int64_t ldhp,
                           const float* vneg, int64_t ldvn, const float* hneg, int64_t ldhn, int B, const imdbn_cd_opts* o,
                           void* ws, size_t ws_bytes, imdbn_stream_t stream);

/* ---- C1: the data-parallel exchange over RCCL (xGMI) for binders that do not use torch.distributed ---------------
 * (the Python classes exchange through torch.distributed, whose "nccl" backend is the same RCCL).  One communicator
 * per process and GPU: rank 0 calls imdbn_comm_unique_id (128 bytes), the caller carries the id to every rank, every
 * rank calls imdbn_comm_init; then per CD step either imdbn_allreduce_sum_f32 on the packed statistics of
 * imdbn_rbm_cd_stats, or imdbn_allgather_bytes on the factor blocks.  librccl is opened on first use. */
int imdbn_comm_unique_id(void* id128);
int imdbn_comm_init(void** comm, int world, int rank, const void* id128);
int imdbn_comm_destroy(void* comm);
int imdbn_allreduce_sum_f32(void* comm, float* buf, size_t count, imdbn_stream_t stream);
int imdbn_allgather_bytes(void* comm, const void* send, void* recv, size_t bytes_per_rank, imdbn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMDBN_ENGINE_H */
