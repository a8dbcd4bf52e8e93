// Host-side launchers of the gfx950 kernels (implemented in kernels_exact.hip / kernels_bf16.hip).
#pragma once
#include "common.hpp"

namespace genie {

enum { GEMM_GELU = 1, GEMM_ACCUM = 2, GEMM_BIAS_ALONG_M = 4 };

// Workspace carving shared by api.hip and the precision-specific layer drivers (see carve() in api.hip).
struct Workspace {
    float* x;          // (M, d)   residual stream, f32 in every precision; offset 0 of the workspace
    void* xn;          // (M, d) x 4 bytes: exact = LayerNorm / attention output (f32);
                       //                   bf16  = [bf16 shadow of x | bf16 LayerNorm / attention output]
    void* big;         // (M, max(3d, hidden)) x 4 bytes: qkv, later the MLP hidden
    float* logits;     // (M, V)   token-major logits scratch
    int64_t* samples;  // (B, S)
    float* conf;       // (B, S)
    uint8_t* unmasked; // (B, S)
    void* aux;         // (M, d) x 4 bytes: f16x3 = split planes of the LayerNorm / attention output
    size_t total;
    // teacher-forced prefix reuse (set per layer by the prefix entry points, NULL otherwise):
    int model_T = 0;     // T of the MODEL's config (the passes below run on private copies with fewer frames): decides, once per
                         // model, whether GENIE_PREC_BF16 keeps its temporal qkv / KV cache in bf16 (temporal_qkv16)
    float* tqkv;         // where the temporal qkv GEMM writes (clean pass: this layer's slice of the cache)
    int tq_frames = 0;   // frames per clip in the layout of `tqkv` (0 = dense: cfg.T); > cfg.T when a short clean pass fills a
                         // full-length cache (generate: prompt frames into the T-frame KV cache)
    const float* tcache; // non-NULL: temporal attention takes keys j < i + tshift from this cached qkv (masked-frames pass)
    int tshift = 0;      // clip-frame offset of the masked-frames buffers against the cache (0 or 1)
    // single-frame decode (generate with a temporal KV cache): the block runs on ONE frame (cfg.T == 1, dense
    // (B,S,*) buffers); its temporal qkv is written into slot `frame_t` of this layer's cache slice and the
    // attention reads slots 0..frame_t.
    float* fcache;
    int frame_t;   // -1 = off
    int frame_T;   // frames per clip in the cache layout
    // 16-bit precisions: the block's last GEMM need not refresh the 16-bit shadow of x when the next consumer is a
    // LayerNorm (set by the layer loops for every layer but the last when the block has pre-norms)
    bool skip_shadow_mlp = false;
    // clean pass, last layer: only the temporal qkv (the cache entry) is needed -- the block returns right after that GEMM
    bool stop_after_tqkv = false;
    // GENIE_PREC_BF16, fused MLP kernel: the block that follows (NULL after the last one).  Its norm1 is applied in this block's
    // MLP epilogue and `ln1_done` tells that block to skip its own LayerNorm launch.
    const genie_layer_weights* next_layer = nullptr;
    bool ln1_done = false;
    bool qkv_planes_done = false;   // ... or even its spatial operand planes (in `big`): that block goes straight to its attention kernel
};

// GENIE_PREC_BF16: the temporal qkv buffer and the temporal KV cache hold bf16 values (half the bytes of the HBM-bound temporal
// attention kernels and of the qkv GEMM's output) whenever the bf16-input kernels cover every pass the model can run: T <= 16
// frames, head_dim 32 / 64.  The cache slices keep their f32-sized strides (a caller sizes the cache with
// genie_prefix_cache_bytes either way); only the first half of a slice is used.
inline bool temporal_qkv16(const genie_cfg& c, int model_T) {
    return c.precision == GENIE_PREC_BF16 && (model_T > 0 ? model_T : c.T) <= 16 && (c.head_dim == 32 || c.head_dim == 64);
}

// kernels_fused.hip: fused sub-blocks of the shipped geometry (GENIE_PREC_BF16, d 256); GENIE_E_UNSUPPORTED otherwise
int launch_pack_temporal_fused(const float* qkv_w, const float* proj_w, uint16_t* out, hipStream_t st);
int launch_pack_mlp_fused(const float* fc1_w, const float* fc2_w, uint16_t* out, hipStream_t st);
int launch_pack_spatial_proj(const float* proj_w, uint16_t* out, hipStream_t st);
int launch_spatial_attn_proj_bf16(const genie_cfg& c, const genie_attn_weights& aw, const uint16_t* qkv16, float* x, uint16_t* x16,
                                  long n_seq, hipStream_t st);
int launch_temporal_fused_bf16(const genie_cfg& c, const genie_attn_weights& aw, const uint16_t* x16, float* x, int B,
                               hipStream_t st);
int launch_mlp_fused_bf16(const genie_cfg& c, const genie_layer_weights& lw, float* x, uint16_t* x16_out, long rows, hipStream_t st,
                          const float* nx_g = nullptr, const float* nx_b = nullptr, const uint16_t* nx_qkv_stream = nullptr,
                          uint16_t* planes = nullptr);
int launch_pack_spatial_qkv(const float* qkv_w, uint16_t* out, hipStream_t st);
bool temporal_fused_takes(const genie_cfg& c, const genie_attn_weights& aw, int B);   // will launch_temporal_fused_bf16 run this problem?
// kernels_fused_prefix.hip: the same sub-block in the prefix-cache passes (mode 1 clean pass: K / V fragment images written to `kv`, the
// layer's cache slice; mode 2 masked pass: read back).  `takes` is the ONE predicate of producer and consumer.
bool temporal_prefix_fused_takes(const genie_cfg& c, const genie_attn_weights& aw, int B, int model_T);
int launch_temporal_prefix_fused_bf16(const genie_cfg& c, const genie_attn_weights& aw, float* x, uint16_t* kv, int B, int mode,
                                      int shift, int model_T, hipStream_t st);

// kernels_fused_f16x3.hip: temporal qkv Linear + temporal attention of the shipped geometry in GENIE_PREC_F16X3 (mode 0 plain, 1 clean pass
// writing the k / v accumulators to `kv`, 2 masked pass reading them); aw.fused_w16 = the stream of genie_pack_temporal_qkv_f16x3
bool temporal_qkv_attn_f16x3_takes(const genie_cfg& c, const genie_attn_weights& aw, int B, int model_T, bool cache_mode);
int launch_temporal_qkv_attn_f16x3(const genie_cfg& c, const genie_attn_weights& aw, const float* x, uint16_t* a16, long plane, float* kv, int B,
                                   int mode, int shift, int model_T, hipStream_t st);
int launch_pack_temporal_qkv_f16x3(const float* qkv_w, uint16_t* out, hipStream_t st);

// Study builds only (-DGENIE_STUDY): which Linear of the block the next GEMM launch is, and the layer it belongs to, so that
// tools/precision_study.py can run individual classes / layer ranges on 2 of the 3 split-f16 terms.
#ifdef GENIE_STUDY
extern int g_study_gemm_class;   // 0 qkv_s, 1 qkv_t, 2 proj_s, 3 proj_t, 4 fc1, 5 fc2, 6 readout
extern int g_study_layer;
int study_terms();                // 3, or 2 when GENIE_F16_TERMS2_CLASSES / _LAYER_LO / _LAYER_HI select the current launch
#define GENIE_STUDY_CLASS(k) (genie::g_study_gemm_class = (k))
#define GENIE_STUDY_LAYER(i) (genie::g_study_layer = (i))
#else
#define GENIE_STUDY_CLASS(k) ((void)0)
#define GENIE_STUDY_LAYER(i) ((void)0)
#endif

// Brackets one launch with HIP events when profiling of `cls` is enabled (see genie_profile_* in the ABI).
struct ProfScope {
    int slot;
    hipStream_t st;
    // kernel: a string LITERAL naming what is launched inside the scope (reported by genie_profile_kernels)
    ProfScope(int cls, double flops, double bytes, hipStream_t st, const char* kernel = nullptr);
    ~ProfScope();
};

int launch_embed(const genie_cfg& c, const genie_weights& w, const int64_t* ids, int B, float* x, hipStream_t st);
int launch_layer_norm(const float* x, const float* g, const float* b, float* y, long rows, int C, float eps,
                      hipStream_t st);
int launch_layer_norm_bf16(const float* x, const float* g, const float* b, uint16_t* y, long rows, int C, float eps,
                           hipStream_t st);
// C[batch][M,N] (+)= epi(alpha * A[batch][M,K] . W[batch][N,K]^T + bias)
int launch_gemm_f32(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, const float* bias,
                    float* C, long ldc, long strideC, int M, int N, int K, int batch, int flags, float alpha,
                    hipStream_t st);
int launch_attn_generic(const float* qkv, float* out, int N, long n_seq, int inner, long outer_stride,
                        long inner_stride, long pos_stride, int d, int H, int Dh, float scale, int causal,
                        const float* nw, const float* nb, hipStream_t st);
int launch_attn_generic_bf16(const uint16_t* qkv, uint16_t* out, int N, long n_seq, int inner, long outer_stride,
                             long inner_stride, long pos_stride, int d, int H, int Dh, float scale, int causal,
                             const float* nw, const float* nb, hipStream_t st);
// out16 != NULL: write the result 16-bit instead of f32 `out`: plane != 0 -> f16 split planes (hi at out16, lo at
// out16 + plane); plane == 0 -> bf16
int launch_attn_spatial_f32_mfma(const float* qkv, float* out, int S, long n_seq, int d, int H, int Dh, float scale,
                                 const float* nw, const float* nb, hipStream_t st, uint16_t* out16 = nullptr,
                                 size_t plane = 0);
int launch_attn_temporal_f32_mfma(const float* qkv, float* out, int B, int T, int S, int d, int H, int Dh, float scale,
                                  const float* nw, const float* nb, hipStream_t st, uint16_t* out16 = nullptr,
                                  size_t plane = 0, int Tq = 0, bool in16 = false);
int launch_attn_spatial_split(const float* qkv, float* out, int S, long n_seq, int d, int H, int Dh, float scale,
                              const float* nw, const float* nb, hipStream_t st, uint16_t* out16 = nullptr,
                              size_t plane = 0);
int launch_attn_temporal_single(const float* cache, float* out, int B, int T, int S, int t, int d, int H, int Dh,
                                float scale, const float* nw, const float* nb, hipStream_t st,
                                uint16_t* out16 = nullptr, size_t plane = 0, bool in16 = false);
int launch_attn_temporal_prefix(const float* cur, const float* cache, float* out, int B, int T, int S, int d, int H,
                                int Dh, float scale, const float* nw, const float* nb, hipStream_t st,
                                uint16_t* out16 = nullptr, size_t plane = 0, int sh = 0, bool in16 = false);
int launch_layer_norm_split(const float* x, const float* g, const float* b, uint16_t* y, size_t plane, long rows, int C,
                            float eps, hipStream_t st);
int launch_split_f16(const float* src, uint16_t* dst, size_t plane, size_t n, hipStream_t st);
int launch_transpose(const float* in, float* out, int batch, int rows, int cols, hipStream_t st);
int launch_count_equal(const int64_t* a, long sa, const int64_t* b, long sb, int batch, long n, const double* ce3, double* sums6,
                       double n_tokens, double n_frames, double n_clips, hipStream_t st);
int launch_factored_ce(const genie_cfg& c, const float* logits, int layout, const int64_t* targets,
                       const int64_t* weight_ids, int B, int t0, int t1, double* sums, hipStream_t st);
int launch_sample(const genie_cfg& c, const float* logits, int layout, int B, float temperature,
                  const float* uniforms, int64_t* samples, float* conf, hipStream_t st);
int launch_mask_step(const float* keys, int n, int last_step, int64_t mask_id, uint8_t* unmasked, int64_t* samples,
                     int64_t* prompt_frame, long clip_stride, int B, int S, hipStream_t st);
int launch_check_masked(const int64_t* prompt, int B, int T, int S, int out_t, int64_t mask_id, int32_t* flag,
                        hipStream_t st);
int launch_bits(const int64_t* ids, float* z, int n, int hw, int bits, hipStream_t st);
int launch_rescale_u8(const void* x, int is_bf16, uint8_t* out, size_t n, hipStream_t st);
int launch_tokens_from_bits(const float* h, int64_t* ids, int n, int hw, int bits, hipStream_t st);
int launch_conv3x3_igemm(const uint16_t* X, const uint16_t* Wt, const float* bias, const uint16_t* residual, uint16_t* Y,
                         const uint16_t* zero_page, int n_img, int H, int Wd, int Cin, int Cout, int d2s, hipStream_t st,
                         int stride = 1, float* gn_part = nullptr, int gn_groups = 0);
size_t conv_gn_part_floats(int n_img, int H, int Wd, int Cout);
int launch_gn_swish_tiles(const uint16_t* X, const float* gamma, const float* beta, uint16_t* Y, const float* part, float* stats,
                          int n_img, int H, int Wd, int Cout, int d2s, int groups, float eps, int apply_swish, hipStream_t st);
int launch_frames_to_nhwc(const uint8_t* f, uint16_t* x, long n_img, int HW, int cin, int cpad, hipStream_t st);
int launch_tokens_from_nhwc(const uint16_t* h, int64_t* ids, long n_pix, int bits, int cpad, hipStream_t st);
size_t gn_scratch_floats(int n_img, int HW, int groups);
int launch_gn_swish(const uint16_t* X, const float* gamma, const float* beta, uint16_t* Y, float* stats, int n_img, int HW,
                    int C, int groups, float eps, int apply_swish, hipStream_t st);
int launch_conv_direct(const uint16_t* X, const uint16_t* Wt, const float* bias, void* Y, int n_img, int H, int Wd, int Cin,
                       int Cout, int out_mode, hipStream_t st);
int launch_bits_nhwc(const int64_t* ids, uint16_t* z, long n_pix, int bits, int cpad, hipStream_t st);
int launch_rescale_nhwc_u8(const uint16_t* x, uint8_t* out, long n_img, int HW, int cpad, int cout, hipStream_t st);
int launch_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int taps, hipStream_t st);
int launch_gemm_bf16_out16(const uint16_t* A16, const uint16_t* W16, const float* bias, uint16_t* C16, int M, int N, int K,
                           hipStream_t st);
int launch_pack_bf16(const float* src, uint16_t* dst, size_t n, hipStream_t st);

// ---- training step (kernels_train.hip) ----
// C[b1][b2][M,N] = alpha * A.B (+bias) (+R); ta/tb: operand stored k-major; nsplit > 1: slabs C + s*sCsplit
int launch_gemm_f32_gen(bool ta, bool tb, const float* A, long lda, long sA1, long sA2, const float* W, long ldw, long sW1,
                        long sW2, const float* bias, const float* R, float* C, long ldc, long sC1, long sC2, int M, int N,
                        int K, int batch1, int batch2, int nsplit, long sCsplit, float alpha, hipStream_t st);
int launch_slab_reduce(const float* part, int ns, size_t n, float* out, float beta, hipStream_t st);
int launch_wgrad_f32(const float* dY, long ldy, const float* X, long ldx, float* dW, int Mtok, int N, int K, float alpha,
                     float beta, float* slabs, size_t slab_floats, hipStream_t st);
constexpr int COLSUM_CHUNKS = 256;  // row chunks of the two-stage column sums (= slabs of their scratch)
int launch_colsum(const float* Y, long ld, long rows, int N, float* out, float beta, float* part, hipStream_t st);
size_t ln_bwd_scratch_floats(int C);
int launch_ln_bwd(const float* x, const float* gamma, const float* dy, float* dxout, float* dgamma, float* dbeta,
                  long rows, int C, float eps, float beta, float* part, hipStream_t st);
int launch_gelu_fwd(const float* z, float* h, size_t n, hipStream_t st);
int launch_gelu_bwd(const float* z, float* g, size_t n, hipStream_t st);
int launch_softmax_rows(float* P, long rows, int N, hipStream_t st);
int launch_softmax_bwd_rows(const float* P, float* dP, long rows, int N, hipStream_t st);
// q, k rows are read from `qk` (leading dim qk_ld; q at column 0, k at column d), v from qkv
int launch_attn_temporal_bwd(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, int B, int T,
                             int S, int d, int H, int Dh, float scale, hipStream_t st);
int launch_attn_spatial_bwd_fused(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, long n_bt, int S,
                                  int d, int H, int Dh, float scale, hipStream_t st);
int launch_qk_norm_fwd(const float* qkv, float* qkn, const float* nw, const float* nb, long M, int H, int Dh, int d,
                       hipStream_t st);
int launch_qk_norm_bwd(const float* qkv, float* dqkv, const float* nw, float* dnw, float* dnb, long M, int H, int Dh,
                       int d, float beta, float* part, hipStream_t st);
int launch_ce_fwd_bwd(const genie_cfg& c, float* logits, const int64_t* ids, const int64_t* labels, int B, double* sums,
                      hipStream_t st);
int launch_embed_bwd(const genie_cfg& c, const float* dx, const int64_t* ids, int B, float* dpos, float* dmask,
                     float* const* tables_host, float beta, float* colpart, hipStream_t st);
int launch_sumsq(const float* x, size_t n, double* out, double* scratch, hipStream_t st);
// 16-bit operand copies (kernels_train16.hip); npl = 1 bf16, 2 = f16 split planes [hi | lo]
// colpart != NULL: also the column sums of the (post-gelu') values: slabs [rows/64][cols] for launch_slab_reduce
int launch_cast_transpose16(int npl, float* in, long ld, const float* z, uint16_t* out16, uint16_t* out16T, int rows,
                            int cols, hipStream_t st, float* colpart = nullptr);
int launch_transpose16(int npl, const uint16_t* in, uint16_t* outT, int rows, int cols, hipStream_t st);
// bf16 copy in the same orientation (+ gelu'(z)) and the [rows/64][cols] column-sum partials; no transposed copy
int launch_cast_rows16(const float* in, long ld, const float* z, uint16_t* out16, int rows, int cols, hipStream_t st,
                       float* colpart);
// dW[N,K] (beta*dW +)= alpha * dY^T . X from ROW-MAJOR bf16 dY (Mtok, N) and X (Mtok, K) (kernels_gemm_tn.hip)
int launch_wgrad16_tn(const uint16_t* dY, long ldy, const uint16_t* X, long ldx, float* dW, int Mtok, int N, int K, float alpha,
                      float beta, float* slabs, size_t slab_floats, hipStream_t st);
int launch_cast16(int npl, const float* src, uint16_t* dst, size_t n, hipStream_t st);
enum { G16X_GELU = 1, G16X_ACCUM = 2, G16X_OUT16 = 4, G16X_OUTF32 = 8, G16X_GELU16 = 16, G16X_NT = 32,
       G16X_QKV = 64 /* gemm16_pp only: the spatial-attention operand layout, see launch_gemm16_pp */,
       G16X_QKNORM = 128 /* with G16X_QKV: q and k leave through the per-head LayerNorm (qn_g, qn_b; attention.py:31-34, 42-47) */ };  // = the G16_* flags of kernels_bf16.hip
int launch_gemm16_ex(int npl, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw, long planeW,
                     const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M, int N,
                     int K, int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC);
// kernels_gemm_pp.hip: the 256x256 two-group phase-scheduled GEMM; GENIE_E_UNSUPPORTED when the shape does not fit it
int launch_gemm16_pp(int npl, int terms, int f16, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw,
                     long planeW, const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M,
                     int N, int K, int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC,
                     float qscale = 1.0f, int head_dim = 0, const float* qn_g = nullptr, const float* qn_b = nullptr);
// kernels_attn_bwd16.hip: spatial attention backward on the bf16 matrix cores (bf16 training precision)
int launch_attn_spatial_bwd_bf16(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, float* stats, long n_bt,
                                 int S, int d, int H, int Dh, float scale, hipStream_t st);
// kernels_gemm_sm.hip: small (latency-bound) problems; GENIE_E_UNSUPPORTED = not small / tiling does not fit
int launch_gemm16_sm(int npl, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw, long planeW,
                     const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M, int N, int K,
                     int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC);
int launch_gemm16_sm_ln(int npl, const float* x, long ldx, const float* ln_g, const float* ln_b, float eps, const uint16_t* W,
                        long ldw, long planeW, const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc,
                        int M, int N, int K, int flags, float alpha, hipStream_t st);
// kernels_frame.hip: the one-frame passes of generate on fragment-ordered operands (GENIE_PREC_F16X3)
int launch_pack_frame_w16(const float* src, uint16_t* dst, int N, int K, hipStream_t st);
bool frame_path_takes(const genie_cfg& c, const genie_layer_weights& lw, long rows);
int frame_prepare_f16x3(const genie_cfg& c, const float* x, Workspace& w, int B, int nf, hipStream_t st);
int st_block_frame_f16x3(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, int nf, bool want_xs,
                         hipStream_t st);
int launch_frame_linear(const uint16_t* A, const uint16_t* W, const float* bias, float* y, int M, int N, int K, int mode, hipStream_t st);
int readout_frame_f16x3(const genie_cfg& c, const genie_weights& wt, Workspace& w, int B, int nf, int f_out, float* logits,
                        hipStream_t st);
// kernels_attn_dma.hip: spatial attention over the operand planes written by launch_gemm16_pp(G16X_OUT16 | G16X_QKV)
int launch_attn_spatial_dma(int npl, const uint16_t* qkv16, long n_seq, int d, int H, int Dh, uint16_t* out16, size_t out_plane,
                            hipStream_t st);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_mult, const double* sumsq, float max_norm, hipStream_t st);

}  // namespace genie
