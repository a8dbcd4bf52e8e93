/* genie_hip.h -- C ABI of libgenie_hip.so: the MI355X (gfx950) GENIE forward / MaskGIT sampling path.
 *
 * The reference (1x-technologies/1xgpt) has no FFI of its own: its seam is the Python nn.Module
 * surface.  Each entry point below names the reference call it replaces
 * (file:line relative to the reference tree); the drop-in Python module in 1xgpt_amd/ binds them with
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every function returns 0 on success, a negative GENIE_E_* code otherwise; nothing throws or aborts;
 *    genie_last_error() returns a thread-local description of the last failure.
 *  - all pointers are DEVICE pointers unless the name ends in _host; they are borrowed for the call only.
 *  - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - no allocation: the caller owns the workspace (genie_workspace_bytes) and every output buffer.
 *  - token ids are int64 (torch.LongTensor), activations/logits float32, flags uint8.
 *  - activations are token-major (B, T, S, C) row-major; nothing is ever physically transposed to
 *    (B S) T C (reference st_transformer.py:77,82): the temporal kernel reads frame-strided rows.
 */
#ifndef GENIE_HIP_H
#define GENIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENIE_ABI_VERSION 3

enum {
    GENIE_OK = 0,
    GENIE_E_ARG = -1,         /* null pointer / bad size */
    GENIE_E_SHAPE = -2,       /* shape not supported by the kernels (see genie_check_config) */
    GENIE_E_UNSUPPORTED = -3, /* feature flag not supported */
    GENIE_E_LAUNCH = -4,      /* HIP reported a launch / runtime error */
    GENIE_E_ASSERT = -5       /* a reference `assert` would have fired (st_mask_git.py:154-156) */
};

enum { GENIE_PREC_EXACT = 0, /* f32 MFMA (v_mfma_f32_32x32x2_f32), f32 everywhere */
       GENIE_PREC_BF16 = 1,  /* bf16 MFMA operands, f32 accumulate/residual/LN/softmax: throughput mode */
       GENIE_PREC_F16X3 = 2  /* every Linear on the f16 matrix cores with split operands a = hi + lo/2048:
                                hi.hi + (hi.lo + lo.hi)/2048 in f32 = 22-bit operands, f32-class results at
                                1/3 of the f16 MFMA rate; attention, LN, softmax, residual stay f32 */ };

enum { GENIE_LAYOUT_TOKEN_MAJOR = 0, /* (B, nt, S, V)              */
       GENIE_LAYOUT_BCTHW = 1        /* (B, V, nt, S) == "B C T H W" (st_mask_git.py:264) */ };

enum { GENIE_UNMASK_RANDOM = 0, GENIE_UNMASK_GREEDY = 1 }; /* st_mask_git.py:200-209 */

/* GenieConfig (genie/config.py:7-55) reduced to what the kernels need. */
typedef struct genie_cfg {
    int32_t num_layers, num_heads, head_dim, d_model;
    int32_t T, S;
    int32_t hidden;            /* int(d_model * mlp_ratio) */
    int32_t factored_vocab;    /* 512 */
    int32_t num_factored;      /* 1 or 2 */
    int32_t image_vocab_size;  /* == mask token id (st_mask_git.py:51) */
    int32_t qk_norm, use_mup, qkv_bias, proj_bias, mlp_bias;
    float attn_scale;          /* use_mup ? 8/Dh : Dh^-0.5 (attention.py:26) */
    float readout_mult;        /* use_mup ? output_mult/width_mult : 1 (st_mask_git.py:316-323) */
    int32_t precision;         /* GENIE_PREC_* */
} genie_cfg;

/* SelfAttention parameters (genie/attention.py:27-34). *_w16 are bf16 copies made by
 * genie_pack_bf16 (only read when precision == GENIE_PREC_BF16). */
typedef struct genie_attn_weights {
    const float* qkv_w;  /* (3d, d) */
    const float* qkv_b;  /* (3d) or NULL */
    const float* proj_w; /* (d, d) */
    const float* proj_b; /* (d) or NULL */
    const float* norm_w; /* (Dh) qk-norm affine shared by q and k, or NULL */
    const float* norm_b;
    const uint16_t* qkv_w16;
    const uint16_t* proj_w16;
    /* GENIE_PREC_BF16, shipped geometry (d 256, 8 heads of 32), or NULL (unfused launches): in the TEMPORAL attention's struct
     * (T 16) the qkv and proj weights as the fragment stream of the fused qkv + attention + proj kernel
     * (genie_pack_temporal_fused_bf16); in the SPATIAL attention's struct (S 256) the proj weights as the fragment stream of
     * the fused attention + proj kernel (genie_pack_spatial_proj_fused_bf16). */
    const uint16_t* fused_w16;
    /* GENIE_PREC_F16X3 range flags of the packed tensors (bit 0: qkv_w16, bit 1: proj_w16): set when the tensor's hi plane
     * reaches |w| >= 32, see "range contract" below.  0 in the other precisions. */
    int32_t w16_wide;
    /* GENIE_PREC_F16X3, head_dim 64 or 32 (ABI 3), or NULL (the one-frame passes then run the row-major kernels): [qkv | proj] as split
     * f16 in FRAGMENT ORDER for the one-frame passes of generate (genie_pack_frame_w16, genie_frames_pass). */
    const uint16_t* frame_w16;
} genie_attn_weights;

/* STBlock parameters (genie/st_transformer.py:28-68). */
typedef struct genie_layer_weights {
    const float* norm1_w; /* (d) or NULL when qk_norm (Identity) */
    const float* norm1_b;
    genie_attn_weights spatial;
    genie_attn_weights temporal;
    const float* norm2_w;
    const float* norm2_b;
    const float* fc1_w; /* (hidden, d) */
    const float* fc1_b;
    const float* fc2_w; /* (d, hidden) */
    const float* fc2_b;
    const uint16_t* fc1_w16;
    const uint16_t* fc2_w16;
    /* GENIE_PREC_BF16, d 256 / hidden 1024: fc1 and fc2 as the fragment stream of the fused LayerNorm + MLP kernel
     * (genie_pack_mlp_fused_bf16), or NULL (unfused launches). */
    const uint16_t* mlp_fused_w16;
    int32_t w16_wide; /* f16x3 range flags: bit 0 fc1_w16, bit 1 fc2_w16 */
    const uint16_t* mlp_frame_w16; /* GENIE_PREC_F16X3 (ABI 3): [fc1 | fc2] in fragment order (genie_pack_frame_w16), or NULL */
} genie_layer_weights;

/* STMaskGIT parameters (genie/st_mask_git.py:36-61); `layers` is a HOST array of num_layers entries. */
typedef struct genie_weights {
    const float* pos_embed;  /* (T, S, d) */
    const float* mask_embed; /* (d) */
    const float* embed[4];   /* num_factored tables of (factored_vocab, d) */
    const float* out_w;      /* (num_factored*factored_vocab, d) */
    const float* out_b;
    const uint16_t* out_w16;
    const genie_layer_weights* layers_host;
    int32_t out_w16_wide; /* f16x3 range flag of out_w16 */
    const uint16_t* out_frame_w16; /* GENIE_PREC_F16X3 (ABI 3): out_w in fragment order (genie_pack_frame_w16), or NULL */
} genie_weights;

int genie_version(void);
/* The compiler's view of the POD structs above, for bindings to check their own declarations against (tests/test_abi_and_host.py):
 * out[0..8) = sizeof(genie_cfg), sizeof(genie_attn_weights), offsetof(.., fused_w16), offsetof(.., w16_wide),
 * sizeof(genie_layer_weights), offsetof(.., mlp_fused_w16), offsetof(.., w16_wide), sizeof(genie_weights),
 * offsetof(.., out_w16_wide), then (ABI 3) offsetof(genie_attn_weights, frame_w16), offsetof(genie_layer_weights, mlp_frame_w16),
 * offsetof(genie_weights, out_frame_w16).  Writes min(n, 12) entries, returns 12. */
int genie_abi_layout(size_t* out_host, int n);
const char* genie_last_error(void);
/* 0 if the kernels support this configuration, GENIE_E_SHAPE otherwise (message in genie_last_error). */
int genie_check_config(const genie_cfg* cfg);
/* Bytes of workspace genie_compute_* / genie_maskgit_generate need for a batch of B clips. */
size_t genie_workspace_bytes(const genie_cfg* cfg, int B);

/* f32 -> bf16 (round-to-nearest-even) weight packing for GENIE_PREC_BF16. */
int genie_pack_bf16(const float* src, uint16_t* dst, size_t n, void* stream);
/* f32 -> f16 split planes for GENIE_PREC_F16X3: dst[0..n) = hi, dst[n..2n) = lo with src ~ hi + lo/2048.
 * In that precision the *_w16 pointers of the weight tables point at such 2n-element buffers. */
int genie_pack_split_f16(const float* src, uint16_t* dst, size_t n, void* stream);
/* GENIE_PREC_F16X3 range contract per weight tensor.  The chip-filling split GEMM scales the weight's hi plane by 2^11 in f16
 * registers, which is exact for |w| < 32 only.  A caller that packs a tensor whose hi plane reaches 32 (finite in f16, i.e.
 * |w| < 65504) sets the tensor's bit in the `w16_wide` field of the struct that carries its pointer; every Linear reading that
 * tensor then runs on the two-accumulator kernels, which scale nothing (same f32-class result, lower rate for that tensor
 * only).  The flag travels with the weight table: there is no process-global state.
 * Reference counterpart: none (the reference's Linear layers are f32: st_transformer.py:16-25, attention.py:27-29). */
enum { GENIE_WIDE_QKV = 1, GENIE_WIDE_PROJ = 2, GENIE_WIDE_FC1 = 1, GENIE_WIDE_FC2 = 2,
       GENIE_FUSED_QKV_STREAM = 4 /* genie_attn_weights.w16_wide of a SPATIAL attention: fused_w16 also holds the qkv fragment stream (below) */ };

/* Fragment streams of the fused sub-block kernels (GENIE_PREC_BF16, magvit_n32_h8_d256 geometry; csrc/kernels_fused.hip).
 * temporal: qkv_w (768, 256) and proj_w (256, 256) f32 -> 262,144 bf16 values;  mlp: fc1_w (1024, 256) and fc2_w (256, 1024)
 * f32 -> 524,288 bf16 values.  Reference counterpart: the nn.Linear weights of attention.py:27-29 / st_transformer.py:16-25. */
#define GENIE_TEMPORAL_FUSED_ELEMS 262144
#define GENIE_MLP_FUSED_ELEMS 524288
#define GENIE_SPATIAL_PROJ_FUSED_ELEMS 65536 /* spatial: proj_w (256, 256) f32 -> the out-projection's fragments per head */
/* spatial, optional second part of the same buffer (at element GENIE_SPATIAL_PROJ_FUSED_ELEMS): qkv_w (768, 256) f32 -> 24 stages of 16
 * fragments.  With it (and GENIE_FUSED_QKV_STREAM set) the PREVIOUS block's fused MLP kernel also runs this block's norm1 and spatial qkv
 * Linear (st_transformer.py:74, attention.py:37) and writes the attention kernel's operand planes: no LayerNorm, no qkv GEMM launch. */
#define GENIE_SPATIAL_QKV_FUSED_ELEMS 196608
#define GENIE_SPATIAL_FUSED_ELEMS (GENIE_SPATIAL_PROJ_FUSED_ELEMS + GENIE_SPATIAL_QKV_FUSED_ELEMS)
int genie_pack_temporal_fused_bf16(const float* qkv_w, const float* proj_w, uint16_t* dst, void* stream);
/* GENIE_PREC_F16X3, same geometry (d 256, 8 heads of 32): the TEMPORAL attention's `fused_w16` holds the split-f16 fragment stream of its
 * qkv weights (768, 256) f32 -> 393,216 f16 values, and the temporal qkv Linear + temporal attention run as one kernel on the f32 rows of x
 * (csrc/kernels_fused_f16x3.hip; st_transformer.py:77-78, attention.py:36-58).  No |w| < 32 contract: nothing is scaled. */
#define GENIE_TEMPORAL_QKV_F16X3_ELEMS 393216
int genie_pack_temporal_qkv_f16x3(const float* qkv_w, uint16_t* dst, void* stream);
/* Unit entry points of the fused sub-blocks (parity tests, tuning).  All update the f32 residual stream x in place and return
 * GENIE_E_UNSUPPORTED outside the geometry above or below their size thresholds -- temporal / mlp: 2 clips' worth of rows (8,192), spatial:
 * 128 sequences; measured break-evens, DESIGN.md section 4 -- (the layer drivers then run the unfused launches).
 *   temporal: x (B,16,S,256) += proj(causal_attention_T(qkv(x16))), x16 = bf16 copy of x (B,16,S,256), or NULL: the kernel rounds its
 *             operands from x itself (same values, no second buffer read); needs aw->fused_w16.
 *             Reference: st_transformer.py:77-78 (the permute to (B S) T C is never materialised), attention.py:36-61.
 *   spatial:  x (n_seq*256, 256) += proj(softmax(q k^T) v) from the operand planes [Q scale log2e | K | V^T] of the spatial qkv GEMM
 *             (n_seq sequences of 256 tokens, 8 heads of 32), x16 = bf16 copy of the result (NULL: not written); needs aw->fused_w16.
 *             Reference: st_transformer.py:73-74, attention.py:48-60.  (The planes are an internal format of the block driver:
 *             this entry exists for tuning -- tools/bench_spatial_fused.py; tests/test_hip_fused.py checks the planes' formats
 *             element-wise at their producer, genie_mlp_fused_qkv_bf16, and this kernel through the block.)
 *   mlp:      x (rows,256) += fc2(gelu(fc1(LayerNorm(x; norm2)))), rows % 128 == 0; x16_out (or NULL) receives the bf16 copy
 *             of the result, or -- when next_norm_w / next_norm_b are given -- LayerNorm(result; next_norm_*) in bf16 (the next
 *             block's norm1 output, st_transformer.py:73); needs lw->mlp_fused_w16.  Reference: st_transformer.py:81, 16-25. */
int genie_temporal_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, const uint16_t* x16, float* x, int B, void* stream);
/* ... of the prefix-cache passes (cfg->T = the model's T, nframes < T the pass's frames; x (B, nframes, S, 256)):
 *   genie_temporal_prefix_fused_bf16: x += proj_t(attention(qkv_t(x))) in GENIE_PREC_BF16; mode 1 = causal + the K / V fragment images written to
 *     `kv` (B * S * 16 KB), mode 2 = query slot i attends the images' slots j < i + shift and its own key (evaluate.py:107-116).
 *   genie_temporal_qkv_attn_f16x3: a = attention(qkv_t(x)) in GENIE_PREC_F16X3 as operand planes a16 = [hi | lo'] `plane_elems` apart, rows
 *     (B nframes S) x 256; mode 0 = plain causal (nframes <= T, kv unused), 1 / 2 as above with the f32 k / v accumulators in `kv` (B * S * 32 KB).
 * Reference counterpart: st_transformer.py:77-78, attention.py:36-61.  Tests: tests/test_hip_fused.py, tests/test_hip_fused_f16x3.py. */
int genie_temporal_prefix_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, float* x, uint16_t* kv, int B, int nframes, int mode,
                                     int shift, void* stream);
int genie_temporal_qkv_attn_f16x3(const genie_cfg* cfg, const genie_attn_weights* aw, const float* x, uint16_t* a16, int64_t plane_elems,
                                  float* kv, int B, int nframes, int mode, int shift, void* stream);
int genie_spatial_attn_proj_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, const uint16_t* qkv_planes, float* x,
                                       uint16_t* x16, int64_t n_seq, void* stream);
int genie_mlp_fused_bf16(const genie_cfg* cfg, const genie_layer_weights* lw, float* x, uint16_t* x16_out, int64_t rows,
                         const float* next_norm_w, const float* next_norm_b, void* stream);
/* The same kernel in the form the block driver uses between two blocks: besides x += mlp(...) it runs the NEXT block's norm1 and spatial qkv
 * Linear (next->norm1_*, next->spatial.fused_w16 with GENIE_FUSED_QKV_STREAM) and writes that block's attention operand planes into
 * `planes` (3 * rows * 256 bf16 values: [Q | K | V^T]; sequences of 256 consecutive rows, 8 heads of 32):
 *   Q[((seq * 8 + head) * 256 + pos) * 32 + f]   = q * attn_scale * log2(e)          K likewise, unscaled
 *   V^T[((seq * 8 + head) * 32 + f) * 256 + p']  with p' = 16 * (pos / 16) + perm(pos % 16), perm = {0-3 -> 0-3, 8-11 -> 4-7, 4-7 -> 8-11, 12-15 -> 12-15}
 * (the formats genie_spatial_attn_proj_fused_bf16 consumes).  GENIE_E_UNSUPPORTED with qkv_bias or outside the geometry.
 * Reference: st_transformer.py:81 then :74 (norm1) and attention.py:37 (qkv) of the following block. */
int genie_mlp_fused_qkv_bf16(const genie_cfg* cfg, const genie_layer_weights* lw, const genie_layer_weights* next, float* x,
                             uint16_t* planes, int64_t rows, void* stream);
int genie_pack_mlp_fused_bf16(const float* fc1_w, const float* fc2_w, uint16_t* dst, void* stream);
int genie_pack_spatial_proj_fused_bf16(const float* proj_w, uint16_t* dst, void* stream);
int genie_pack_spatial_qkv_fused_bf16(const float* qkv_w, uint16_t* dst, void* stream);   /* dst: GENIE_SPATIAL_QKV_FUSED_ELEMS values */

/* ---- unit entry points (one reference op each; used by the parity tests) -------------------------- */

/* FactorizedEmbedding.forward + pos-embed add (factorization_utils.py:29-52, st_mask_git.py:257-261).
 * ids (B,T,S) int64 -> x (B,T,S,d). */
int genie_embed(const genie_cfg* cfg, const genie_weights* w, const int64_t* ids, int B, float* x, void* stream);

/* nn.LayerNorm(C, eps) over the last dim of (rows, C) (st_transformer.py:44,67). */
int genie_layer_norm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps,
                     void* stream);

/* nn.Linear: y = x W^T (+b) with optional fused erf-GELU and residual accumulate (y += ...).
 * x (M,K), W (N,K), y (M,N).  (attention.py:27,29; st_transformer.py:16-25) */
int genie_linear(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int gelu,
                 int accumulate, void* stream);

/* The same Linear on the 16-bit matrix cores with PRE-PACKED operands (unit entry for parity tests and tuning):
 * GENIE_PREC_BF16: x16 (M,K), W16 (N,K) bf16 (genie_pack_bf16);  GENIE_PREC_F16X3: split planes [hi | lo] of
 * each (genie_pack_split_f16).  y (M,N) f32. */
int genie_linear_lowp(int precision, const uint16_t* x16, const uint16_t* W16, const float* b, float* y, int M, int N,
                      int K, int gelu, int accumulate, void* stream);

/* softmax(scale q k^T [+causal]) v on a packed qkv buffer (attention.py:38-59).
 * qkv (B,T,S,3d) with feature index = which*d + head*Dh + i; out (B,T,S,d).
 * spatial: sequences are the S tokens of one (b,t), non-causal (st_transformer.py:73-74);
 * temporal: sequences are the T frames of one (b,s), causal (st_transformer.py:77-78). */
int genie_spatial_attention(const genie_cfg* cfg, const genie_attn_weights* aw, const float* qkv, float* out, int B,
                            void* stream);
int genie_temporal_attention(const genie_cfg* cfg, const genie_attn_weights* aw, const float* qkv, float* out, int B,
                             void* stream);

/* The same core on contiguous sequences: qkv (n_seq, N, 3*H*Dh) -> out (n_seq, N, H*Dh); this is the
 * body of SelfAttention.forward(x, causal) between the qkv and proj Linears (attention.py:38-59) and the
 * shape test_attention.py exercises.  norm_w/norm_b NULL = no qk-norm. */
int genie_attention_core(const float* qkv, float* out, int n_seq, int N, int num_heads, int head_dim, float scale,
                         int causal, const float* norm_w, const float* norm_b, void* stream);

/* STBlock.forward in place on x (B,T,S,d) (st_transformer.py:70-83). */
int genie_st_block_forward(const genie_cfg* cfg, const genie_layer_weights* lw_host, float* x, int B, void* workspace,
                           size_t workspace_bytes, void* stream);
/* STTransformerDecoder.forward in place (st_transformer.py:115-120). */
int genie_decoder_forward(const genie_cfg* cfg, const genie_weights* w, float* x, int B, void* workspace,
                          size_t workspace_bytes, void* stream);

/* out_x_proj (+ muP pre-scale) for frames [t0,t1) (st_mask_git.py:60-61,262-264,316-323). */
int genie_readout_logits(const genie_cfg* cfg, const genie_weights* w, const float* x, int B, int t0, int t1,
                         int layout, float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* ---- whole-path entry points ------------------------------------------------------------------------ */

/* STMaskGIT.compute_logits (st_mask_git.py:255-265): ids (B,T,S) -> logits of frames [t0,t1). */
int genie_compute_logits(const genie_cfg* cfg, const genie_weights* w, const int64_t* ids, int B, int t0, int t1,
                         int layout, float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* ---- teacher-forced prefix reuse (evaluate.py:107-116 recomputes frames < t in every timeline) -----------------
 * Temporal attention is causal and every other op is per-frame, so in the evaluator's timeline t the activations of
 * the ground-truth frames < t equal those of ONE forward over the ground-truth clip ("clean pass"), and frame t
 * itself only needs their temporal keys/values.
 *   genie_clean_pass: `ids` (B, nframes, S) = clip frames 0..nframes-1; runs that forward and stores every layer's
 *     temporal qkv f32 in `cache`, laid out (L, B, cache_frames, S, 3d) with cache_frames >= nframes (0 = nframes).  The
 *     evaluator needs nframes = cache_frames = T-1 (no timeline has the last frame as context); generate() fills the
 *     prompt's slots of its T-frame KV cache in one pass with nframes = P, cache_frames = T (then genie_frame_pass
 *     continues from slot P).  cache_frames != nframes at B > 1 needs 8 <= nframes <= 16 and head_dim 32/64
 *     (GENIE_E_UNSUPPORTED otherwise: fill the cache with genie_frame_pass instead).
 *   genie_masked_frames_logits: evaluates "frame t in timeline t" for nframes timelines at once: slot i of `frames`
 *     (B, nframes, S) holds the current tokens of clip frame frame0 + i in its own timeline (all-mask at MaskGIT step 0);
 *     it attends the cached keys of clip frames < frame0 + i and its own key.  `cache` comes from a clean pass with the
 *     same nframes; frame0 is 1 (slots = clip frames 1..nframes, the evaluator's case) or 0 (slot 0 = frame 0 seeing
 *     only itself).  logits: token-major (B, nframes, S, V).
 * Same per-row arithmetic as the full forwards: (1 + steps) passes over T-1 frames instead of 15*steps over T.
 * genie_prefix_cache_bytes is the size for nframes = T (an upper bound).
 * The cache is opaque to the caller and only valid for the precision that wrote it: GENIE_PREC_EXACT / _F16X3 store f32;
 * GENIE_PREC_BF16 stores bf16 values (models with T <= 16 and head_dim 32 / 64) in the first half of each layer's slice --
 * the slice strides stay those of the f32 layout, so the size above holds for every precision.
 * A clean pass with cache_frames == nframes < T makes a cache for genie_masked_frames_logits ONLY (a cache genie_frame_pass can
 * continue has T frame slots: cache_frames = T).  For the shipped geometry in GENIE_PREC_BF16 (d 256, 8 heads of 32, fused
 * streams present, 8 <= nframes < T, B * S >= 512) such a cache holds, per (clip, position, head), the K and V operand
 * fragments of the fused temporal kernel instead of qkv rows (csrc/kernels_fused_prefix.hip: both passes run the temporal
 * sub-block as one kernel), and in GENIE_PREC_F16X3 (same geometry, temporal fused_w16 = the genie_pack_temporal_qkv_f16x3
 * stream, 11 <= nframes < T) the f32 k / v accumulators of csrc/kernels_fused_f16x3.hip; producer and consumer decide by the
 * same predicate of (cfg, weights, B, nframes). */
size_t genie_prefix_cache_bytes(const genie_cfg* cfg, int B);
int genie_clean_pass(const genie_cfg* cfg, const genie_weights* w, const int64_t* ids, int B, int nframes, int cache_frames,
                     float* cache, size_t cache_bytes, void* workspace, size_t workspace_bytes, void* stream);
int genie_masked_frames_logits(const genie_cfg* cfg, const genie_weights* w, const int64_t* frames, int B, int frame0,
                               int nframes, const float* cache, size_t cache_bytes, float* logits, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Temporal KV cache for autoregressive generation (generate.py:81-95 re-runs the full 16-frame forward for every
 * MaskGIT step of every new frame): run ONE frame (frame_ids (B,S), frame index t) through the stack; each layer writes
 * the frame's temporal qkv into slot t of `cache` (same layout as genie_clean_pass) and attends slots 0..t.  Calling
 * it for t = 0..P-1 on the prompt frames fills the cache; for a new frame call it once per MaskGIT step on the current
 * (partially masked) tokens to get `logits` (B,S,V) token-major (NULL = not wanted), and once more on the final tokens
 * to commit them.  Same per-row arithmetic as the full forward restricted to frame t. */
int genie_frame_pass(const genie_cfg* cfg, const genie_weights* w, const int64_t* frame_ids, int B, int t, float* cache,
                     size_t cache_bytes, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* The same for nf CONSECUTIVE frames t0 .. t0 + nf - 1 in one pass (ABI 3): frame_ids (B, nf, S); every layer writes the nf
 * cache slots and frame t0 + i attends slots 0 .. t0 + i (those of this pass included); `logits` (B, S, V) token-major are those
 * of the LAST frame of the pass (NULL = not wanted).  generate()'s use: the pass that commits the final tokens of frame t also
 * carries the first MaskGIT step of frame t + 1 (all-mask tokens), so a new frame costs `steps` passes instead of `steps + 1`
 * (generate.py:81-95: the reference recomputes everything per step).  nf == 1 is genie_frame_pass; nf > 1 returns
 * GENIE_E_UNSUPPORTED (nothing enqueued) unless the fragment-order kernels cover the model (GENIE_PREC_F16X3, head_dim 64 or 32,
 * LayerNorm or qk-norm blocks, S 256, vocabulary columns % 64 == 0, frame_w16 streams present, B * nf * S <= 16,384 rows): the caller then
 * runs the frames one by one. */
int genie_frames_pass(const genie_cfg* cfg, const genie_weights* w, const int64_t* frame_ids, int B, int t0, int nf, float* cache,
                      size_t cache_bytes, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* generate.py:77-103 / STMaskGIT.generate (st_mask_git.py:65-113) on the temporal KV cache, the WHOLE loop enqueued by one call (no
 * host synchronisation, no host work between passes): ids (B, P + n_new, S) = the clip (frames [0, P) are the prompt; frames >= P are
 * read only when teacher_force_time); gen_out (B, n_new, S) receives the generated frames; logits0_out (B, n_new, S, V) f32
 * token-major (or NULL) the step-0 logits of every new frame (what maskgit_generate returns, st_mask_git.py:165,226).  P + n_new <= T.
 * Per new frame: `steps` MaskGIT steps of one-frame passes (genie_frames_pass), genie_sample, genie_mask_step; the commit of frame t
 * shares a two-frame pass with step 0 of frame t + 1 where the library covers it and merge_commit != 0.  noise: the unmasking draws
 * (n_new, steps - 1, B, S) f32 ('random' mode with steps > 1; st_mask_git.py:204-206 draws them with torch.rand_like); uniforms:
 * (n_new, steps, num_factored, B, S) for temperature > 0; both may be NULL otherwise.  cache: genie_prefix_cache_bytes(cfg, B);
 * workspace: genie_generate_workspace_bytes(cfg, B, P) (the loop keeps its token / sample / logits scratch behind the workspace of its
 * largest pass; for models whose T exceeds max(P, 2) genie_workspace_bytes(cfg, B) is that large too).  Same frames as the full-forward
 * schedule (generate.py:81-95) up to f32 accumulation order. */
size_t genie_generate_workspace_bytes(const genie_cfg* cfg, int B, int P);
int genie_generate_cached(const genie_cfg* cfg, const genie_weights* w, const int64_t* ids, int B, int P, int n_new, int steps,
                          float temperature, int unmask_mode, const float* noise, const float* uniforms, int teacher_force_time,
                          int merge_commit, int64_t* gen_out, float* logits0_out, float* cache, size_t cache_bytes, void* workspace,
                          size_t workspace_bytes, void* stream);
/* f32 (N, K) row-major weight -> split f16 in FRAGMENT ORDER (2 N K 16-bit values) for the one-frame kernels
 * (csrc/kernels_frame.hip): blocks of 32 rows x 64 k, per block [plane hi | lo'][MFMA step 0..3] fragments of 1 KB = the 64 lanes'
 * 16-byte operand pieces (lane 32 h + r: row r, k = 16 step + 8 h .. + 7), so that every operand load of those kernels is a
 * contiguous 1 KB read.  N % 32 == 0, K % 64 == 0.  Same split as genie_pack_split_f16.
 * Reference counterpart: none (nn.Linear weights are f32: st_transformer.py:16-25, attention.py:27-29, st_mask_git.py:60-61). */
int genie_pack_frame_w16(const float* src, uint16_t* dst, int N, int K, void* stream);
/* y (M, N) f32 = a . w^T + bias on the one-frame kernels, both operands split f16 in fragment order (genie_pack_frame_w16 packs
 * an (M, K) activation the same way as an (N, K) weight): the unit the parity tests and the micro-benchmark call.  mode 0 = the kernel
 * a frame pass of M rows would take, 1 = register-direct kernel (K <= 512), 2 = LDS-tiled kernel (M % 128 == 0, N % 64 == 0).
 * Reference counterpart: nn.Linear.forward (st_transformer.py:16-25, attention.py:27-29). */
int genie_frame_linear(const uint16_t* a_fr, const uint16_t* w_fr, const float* bias, float* y, int M, int N, int K, int mode,
                       void* stream);

/* Factored cross-entropy + accuracy partial sums from token-major or BCTHW logits of frames [t0,t1)
 * (st_mask_git.py:231-253; eval_utils.py:44-77).
 *   targets (B,T,S) int64 (full clip; frames [t0,t1) are read)
 *   weight_ids: if non-NULL (B,T,S) int64, a token counts only where weight_ids == mask id
 *               (STMaskGIT.forward's relevant_mask, st_mask_git.py:276); NULL = every token (compute_loss).
 *   sums_out: 3 doubles, ACCUMULATED: [sum CE, sum (all factors argmax-correct), n tokens counted].  */
int genie_factored_ce(const genie_cfg* cfg, const float* logits, int layout, const int64_t* targets,
                      const int64_t* weight_ids, int B, int t0, int t1, double* sums_out, void* stream);

/* The evaluator's metric vector (evaluate.py:167-179 / eval_utils.py:16-25) as six device-side f64 sums
 *   sums6 = [sum CE, n CE tokens, sum (sample == ground truth), n sampled tokens, n frames, n clips].
 * genie_metric_hits ADDS the number of equal entries of two (batch, n_per_batch) int64 views (batch strides in elements) to
 * sums6[2] (zero it first), writes sums6[3..5] = n_tokens, n_frames, n_clips, and -- when ce3 (the 3-vector
 * genie_factored_ce accumulated on the same stream) is not NULL -- sums6[0] = ce3[0], sums6[1] = ce3[2]. */
int genie_metric_hits(const int64_t* truth, int64_t truth_batch_stride, const int64_t* samples, int64_t samples_batch_stride,
                      int batch, int64_t n_per_batch, const double* ce3, double n_tokens, double n_frames, double n_clips,
                      double* sums6, void* stream);

/* Fused readout + CE for frames [t0,t1) without materialising logits for the caller. */
int genie_readout_ce(const genie_cfg* cfg, const genie_weights* w, const float* x, const int64_t* targets,
                     const int64_t* weight_ids, int B, int t0, int t1, double* sums_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* MaskGIT sampling half (st_mask_git.py:171-190) on one frame's logits.
 *   logits: (B,S,V) token-major or (B,V,S) BCTHW with nt == 1
 *   temperature <= 1e-8 -> argmax (first max wins); otherwise inverse-CDF sampling with caller-supplied
 *   uniforms (num_factored, B, S) float32, factor order = most significant first (flip(2), :179).
 *   samples (B,S) int64 = hi*Vf + lo;  conf (B,S) float32 = prod p[sample]. */
int genie_sample(const genie_cfg* cfg, const float* logits, int layout, int B, float temperature,
                 const float* uniforms, int64_t* samples, float* conf, void* stream);

/* MaskGIT mask half for one step (st_mask_git.py:192-223), one frame of B clips.
 *   keys (B,S) float32: caller's torch.rand draws ("random") or conf ("greedy"); ignored if last_step
 *   n: tokens to re-mask (ceil(cos(pi/2 (step+1)/steps) S)); unmasked (B,S) uint8 in/out;
 *   samples (B,S) int64 in/out; prompt_frame: pointer to prompt[0, out_t, 0], clip stride in elements.
 *   Writes the final samples back into the prompt frame (in-place semantics of :223). */
int genie_mask_step(const float* keys, int n, int last_step, int64_t mask_id, uint8_t* unmasked, int64_t* samples,
                    int64_t* prompt_frame, int64_t prompt_clip_stride, int B, int S, void* stream);

/* STMaskGIT.maskgit_generate (st_mask_git.py:123-229), whole loop on the device, no host sync.
 *   prompt (B,T,S) int64, frames >= out_t must be all-mask (checked on device: violation sets *status_flag
 *   (device int32, may be NULL) to GENIE_E_ASSERT); prompt[:, out_t] is overwritten in place.
 *   noise: (steps-1, B, S) float32 U[0,1) draws for GENIE_UNMASK_RANDOM (NULL allowed when steps == 1 or greedy)
 *   uniforms: (steps, num_factored, B, S) for temperature > 1e-8, else NULL
 *   samples_out (B,S) int64; logits0_out: step-0 logits of frame out_t in `layout` (NULL = not wanted). */
int genie_maskgit_generate(const genie_cfg* cfg, const genie_weights* w, int64_t* prompt, int B, int out_t, int steps,
                           float temperature, int unmask_mode, const float* noise, const float* uniforms,
                           int64_t* samples_out, float* logits0_out, int layout, int32_t* status_flag,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- optional per-launch timing (bench.py's roofline leg) ------------------------------------------------
 * When enabled, every launch of a kernel whose class bit is set in `class_mask` is bracketed by a pair of
 * HIP events recorded on the launch stream.  genie_profile_read synchronises those events and returns, for
 * one class, out[0..3] = {launches, total milliseconds, total algorithmic FLOPs, total algorithmic bytes}.
 * Process-global, not thread-safe; at most GENIE_PROFILE_MAX_LAUNCHES launches are timed between resets
 * (later ones run untimed and are not counted). */
enum { GENIE_KC_GEMM = 0, GENIE_KC_ATTN_SPATIAL = 1, GENIE_KC_ATTN_TEMPORAL = 2, GENIE_KC_LAYERNORM = 3,
       GENIE_KC_OTHER = 4, GENIE_KC_FUSED = 5 /* fused sub-block kernels (kernels_fused.hip) */, GENIE_KC_COUNT = 6 };
#define GENIE_PROFILE_MAX_LAUNCHES 32768
int genie_profile_enable(int class_mask); /* 0 disables */
int genie_profile_reset(void);
int genie_profile_read(int kernel_class, double* out4);
/* Which kernels the timed launches of one class actually were: one line per distinct kernel,
 * "name\tlaunches\tmilliseconds\tflops\n", written NUL-terminated into buf (truncated at buf_bytes).  bench.py derives
 * roofline.kernel from this instead of assuming the dispatch (a small batch or an override routes elsewhere). */
int genie_profile_kernels(int kernel_class, char* buf, size_t buf_bytes);
/* 1 when the library was compiled with -DGENIE_STUDY (ablation / reduced-precision knobs are live: measurements of such a
 * build are study data, never the product's), 0 for the shipping build, whose launch paths read no such knob. */
int genie_study_build(void);

/* LFQ.get_codebook_entry(...).flip(1) (lookup_free_quantize.py:181-194, visualize.py:115):
 * ids (n, hw) int64 -> z (n, bits, hw) float32 in {-1,+1}, channel c = bit c (LSB first). */
int genie_bits_from_tokens(const int64_t* ids, float* z, int n, int hw, int bits, void* stream);

/* rescale_magvit_output (visualize.py:84-92): u8 = trunc(clamp((x + 1) * 127.5, 0, 255)).  The reference applies
 * it to the decoder's bf16 output with bf16 arithmetic (each op rounds to bf16); `x` holds bf16 bits and the same
 * roundings are reproduced, so the bytes are identical for identical decoder outputs.  n elements. */
int genie_rescale_u8_bf16(const uint16_t* x, uint8_t* out, size_t n, void* stream);
/* f32 variant (no intermediate bf16 rounding), for an f32 decoder. */
int genie_rescale_u8_f32(const float* x, uint8_t* out, size_t n, void* stream);

/* Dataset-compatible LFQ index of an encoder output h (n, bits, hw) f32: id = sum_c [h_c > 0] << c (LSB first;
 * the inverse of genie_bits_from_tokens; cf. lookup_free_quantize.py:241-257 which packs MSB-first and is NOT
 * the dataset convention).  ids (n, hw) int64. */
int genie_tokens_from_bits(const float* h, int64_t* ids, int n, int hw, int bits, void* stream);

/* ---- MAGVIT2 decoder convolutions (improved_model.py:12-51, 124-237), activations NHWC bf16, f32 accumulate ------
 * genie_pack_conv_weight: (C_out, C_in, kh*kw) f32 torch layout -> (C_out, kh*kw, C_in) bf16 (tap-major K).
 * genie_conv3x3_bf16: 3x3 / pad 1 / stride 1 implicit GEMM on the bf16 matrix cores; C_in %% 64 == 0, C_out %% 8 == 0 (%% 32 with depth-to-space);
 *   residual (same shape as the output) is added before the bf16 store (ResBlock skip); depth_to_space != 0 writes
 *   the Upsampler's DCR-permuted (n, 2H, 2W, C_out/4) image; zero_page: >= 16 zero bytes for out-of-image taps.
 * genie_conv1x1_bf16: the 1x1 nin_shortcut (a plain Linear over pixels).
 * genie_conv_direct_bf16: direct 3x3 conv for the edge layers (C_in = 18 -> 512, 128 -> 3); out_mode 0 = NHWC bf16,
 *   1 = NCHW f32.
 * genie_group_norm_swish_bf16: GroupNorm(groups, eps) [+ x*sigmoid(x)] of an NHWC bf16 image (improved_model.py:24-34);
 *   stats_ws: genie_group_norm_scratch_floats(n, HW, groups) f32 of scratch.  The statistics are reduced in a fixed order
 *   (no atomics): the same input gives the same bytes on every call.
 * genie_bits_from_tokens_nhwc_bf16: tokens (n_pix) -> (n_pix, cpad) bf16: channel c < bits = +-1 for bit c (a18 in
 *   operand layout), channels >= bits zero (cpad %% 64 == 0 lets conv_in run on the implicit GEMM).
 * genie_rescale_u8_nhwc_bf16: decoder tail: (n, HW, cpad) bf16 -> (n, cout, HW) uint8 with the reference's bf16 rescale. */
int genie_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int taps, void* stream);
int genie_conv3x3_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, const uint16_t* residual, uint16_t* y,
                       const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, int depth_to_space, void* stream);
/* The encoder's downsample: 3x3 / pad 1 / stride 2 (improved_model.py:90); H, W are the OUTPUT size, input is (2H, 2W). */
int genie_conv3x3_s2_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, uint16_t* y,
                          const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, void* stream);
/* Encoder ends: uint8 frames (n, cin, HW) -> (n, HW, cpad) bf16 in [-1,1] (x/127.5 - 1; channels >= cin zero), and the
 * encoder code (n_pix, cpad) bf16 -> dataset-convention token ids, bit c = [h_c > 0]. */
int genie_frames_to_nhwc_bf16(const uint8_t* frames, uint16_t* x, int n, int HW, int cin, int cpad, void* stream);
int genie_tokens_from_code_nhwc_bf16(const uint16_t* h, int64_t* ids, int64_t n_pix, int bits, int cpad, void* stream);
int genie_conv1x1_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, uint16_t* y, int n_pix, int Cin,
                       int Cout, void* stream);
int genie_conv_direct_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, void* y, int n, int H, int W,
                           int Cin, int Cout, int out_mode, void* stream);
size_t genie_group_norm_scratch_floats(int n, int HW, int groups);
int genie_group_norm_swish_bf16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* stats_ws, int n,
                                int HW, int C, int groups, float eps, int apply_swish, void* stream);
/* GroupNorm statistics fused into the producing convolution ("fused GN+swish").
 * genie_conv3x3_gn_bf16 = genie_conv3x3_bf16 / genie_conv3x3_s2_bf16 (stride 1 or 2) that also writes, per 256-pixel x
 *   128-channel output tile, the (sum, sumsq) of the STORED bf16 output for each GroupNorm(groups) group of the tile
 *   (gn_part: genie_conv_gn_part_floats(n, H, W, Cout) floats; fixed reduction order, no atomics).  Needs H*W %% 256 == 0,
 *   C_out %% 128 == 0 and groups of 4, 8 or 16 channels; returns GENIE_E_UNSUPPORTED otherwise
 *   without launching, and the caller uses the separate-statistics path.
 * genie_group_norm_swish_fused_bf16: GroupNorm(groups, eps) [+ swish] of that convolution's output x from its partials:
 *   no statistics pass over x.  H, W, Cout, depth_to_space are the CONVOLUTION's; stats_ws: n * groups * 2 floats. */
size_t genie_conv_gn_part_floats(int n, int H, int W, int Cout);
int genie_conv3x3_gn_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, const uint16_t* residual, uint16_t* y,
                          const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, int depth_to_space, int stride,
                          float* gn_part, int groups, void* stream);
int genie_group_norm_swish_fused_bf16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y,
                                      const float* gn_part, float* stats_ws, int n, int H, int W, int Cout, int depth_to_space,
                                      int groups, float eps, int apply_swish, void* stream);
int genie_bits_from_tokens_nhwc_bf16(const int64_t* ids, uint16_t* z, int64_t n_pix, int bits, int cpad, void* stream);
int genie_rescale_u8_nhwc_bf16(const uint16_t* x, uint8_t* out, int n, int HW, int cpad, int cout, void* stream);

/* ---- training step (train.py:600-633) -------------------------------------------
 * All three precisions; both the LayerNorm (qk_norm = false, the shipped config) and the qk-norm block variants.
 * GENIE_PREC_EXACT: every contraction on the f32 matrix instruction.  GENIE_PREC_F16X3 / GENIE_PREC_BF16: every
 * Linear product (forward, input gradient, weight gradient) on the 16-bit matrix cores with f32 accumulation --
 * f16x3 keeps f32-class gradients (split operands), bf16 is what `accelerate --mixed_precision bf16` computes --
 * while LayerNorm, both attention cores, GELU, the residual stream, the loss and every gradient reduction stay f32.
 * The 16-bit precisions need 16-bit copies of the Linear weights in both orientations (genie_train_pack_weights,
 * again after every optimizer step); parameters, gradients and optimizer state are always f32.  Gradients travel in a second genie_weights table whose pointers address the caller's
 * gradient buffers (same shapes as the parameters; the *_w16 members are ignored).  `accumulate` = 0 overwrites the
 * gradients (optimizer.zero_grad() + backward), 1 adds to them (gradient accumulation, train.py:607-617).
 * Every reduction feeding a gradient has a fixed order: results are bit-reproducible run to run. */

/* Bytes of the saved-activation buffer / of the backward scratch for B clips. */
size_t genie_train_activation_bytes(const genie_cfg* cfg, int B);
size_t genie_train_workspace_bytes(const genie_cfg* cfg, int B);

/* STMaskGIT.forward in training mode (st_mask_git.py:267-279; dropout is 0 in every shipped config): embeds
 * input_ids (B,T,S), runs the L blocks keeping what the backward needs, the readout, and the masked factored CE of
 * compute_loss_and_acc (:231-253) against labels (B,T,S).  sums_out (3 doubles, overwritten) =
 * [sum CE over counted tokens, sum (all factors argmax-correct), n counted] with counted = frame >= 1 and
 * input id == mask id; loss = sums[0]/sums[2], acc = sums[1]/sums[2] (0/0 = nan as in the reference).
 * On return the logits slot of `acts` holds d loss / d logits. */
int genie_train_forward(const genie_cfg* cfg, const genie_weights* w, const int64_t* input_ids, const int64_t* labels,
                        int B, float* acts, size_t acts_bytes, double* sums_out, void* stream);

/* 16-bit precisions only: refresh the 16-bit copies of every Linear weight from the f32 parameters in `w`:
 * row-major (out, in) into the *_w16 members of `w16` (read by the forward) and transposed (in, out) into the *_w16
 * members of `w16T` (read by the input-gradient products; pass it as `wT` below, NULL for GENIE_PREC_EXACT).
 * bf16: one plane; f16x3: [hi | lo] planes as genie_pack_split_f16. */
int genie_train_pack_weights(const genie_cfg* cfg, const genie_weights* w, const genie_weights* w16,
                             const genie_weights* w16T, void* stream);

/* loss.backward() (train.py:617), split so that the caller can all-reduce finished gradients while earlier layers
 * are still running: head (readout weight/bias, d loss / d x_L into the workspace), then layers L-1 .. 0 (each
 * consumes and replaces the running d loss / d x in the workspace), then the embedding tables, mask embedding and
 * pos_embed.  Must be called in exactly that order on one stream after genie_train_forward. */
int genie_train_backward_head(const genie_cfg* cfg, const genie_weights* w, const genie_weights* wT,
                              const genie_weights* grads, int B, float* acts, void* workspace, size_t workspace_bytes,
                              int accumulate, void* stream);
int genie_train_backward_layer(const genie_cfg* cfg, const genie_weights* w, const genie_weights* wT,
                               const genie_weights* grads, int layer, int B, float* acts, void* workspace,
                               size_t workspace_bytes, int accumulate, void* stream);
int genie_train_backward_embed(const genie_cfg* cfg, const genie_weights* grads, const int64_t* input_ids, int B,
                               void* workspace, size_t workspace_bytes, int accumulate, void* stream);

/* *out += sum x[i]^2 in f64, two-stage with a fixed order (scratch: 1024 doubles).  The global gradient norm of
 * clip_grad_norm_ (train.py:628-629) is sqrt of this summed over all gradient buffers. */
int genie_sumsq(const float* x, size_t n, double* out, double* scratch, void* stream);

/* torch.optim.AdamW.step() on one flat range (train.py:440-441, 631): decoupled decay p *= 1 - lr*wd, then the
 * bias-corrected Adam update; `step` counts from 1.  The gradient is first multiplied by grad_mult (1/world_size
 * after a SUM all-reduce, 1/accumulation steps) and, when grad_sumsq != NULL and max_grad_norm > 0, by
 * min(1, max_grad_norm / (grad_mult * sqrt(*grad_sumsq) + 1e-6)) -- clip_grad_norm_ without a host sync.
 * The reference's grouping (train.py:426-437) is the caller's: weight_decay = 0 for names containing "bias". */
int genie_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_mult, const double* grad_sumsq,
                     float max_grad_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENIE_HIP_H */
