// C ABI of libgenie_hip.so (include/genie_hip.h): argument checking, workspace carving and the
// launch sequences of STBlock / decoder / readout / MaskGIT.  All work is enqueued on the caller's
// stream; nothing here allocates, synchronises or touches the host-side of any tensor.
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-launch event profiler ------------------------------------------------------------------
static int g_prof_mask = 0;
static int g_prof_n = 0;
static hipEvent_t* g_prof_ev = nullptr;  // 2 * GENIE_PROFILE_MAX_LAUNCHES events, created on first enable
static int g_prof_cls[GENIE_PROFILE_MAX_LAUNCHES];
static double g_prof_flops[GENIE_PROFILE_MAX_LAUNCHES], g_prof_bytes[GENIE_PROFILE_MAX_LAUNCHES];
static const char* g_prof_name[GENIE_PROFILE_MAX_LAUNCHES];

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s, const char* kernel) : slot(-1), st(s) {
    if (!(g_prof_mask & (1 << cls)) || g_prof_n >= GENIE_PROFILE_MAX_LAUNCHES || !g_prof_ev) return;
    slot = g_prof_n++;
    g_prof_cls[slot] = cls;
    g_prof_flops[slot] = flops;
    g_prof_bytes[slot] = bytes;
    g_prof_name[slot] = kernel ? kernel : "(unnamed)";
    (void)hipEventRecord(g_prof_ev[2 * slot], st);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof_ev[2 * slot + 1], st);
}

// Workspace carving (struct Workspace: kernels.hpp).  All offsets 256-byte aligned.
static Workspace carve(const genie_cfg& c, int B, void* base) {
    Workspace w;
    const size_t M = (size_t)B * c.T * c.S;
    const size_t wide = (size_t)(3 * c.d_model > c.hidden ? 3 * c.d_model : c.hidden);
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* r = p ? (void*)(p + off) : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    w.x = (float*)take(M * c.d_model * 4);
    w.xn = take(M * c.d_model * 4);
    w.big = take(M * wide * 4);
    w.logits = (float*)take(M * V * 4);
    w.samples = (int64_t*)take((size_t)B * c.S * 8);
    w.conf = (float*)take((size_t)B * c.S * 4);
    w.unmasked = (uint8_t*)take((size_t)B * c.S);
    w.aux = take(M * c.d_model * 4);
    w.total = off;
    w.tqkv = nullptr;
    w.tcache = nullptr;
    w.fcache = nullptr;
    w.frame_t = -1;
    w.frame_T = 0;
    w.model_T = c.T;
    return w;
}

static int check_cfg(const genie_cfg* c) {
    GENIE_CHECK_ARG(c != nullptr, "cfg is NULL");
    GENIE_CHECK_SHAPE(c->num_layers >= 1 && c->num_heads >= 1, "num_layers/num_heads must be >= 1");
    GENIE_CHECK_SHAPE(c->d_model == c->num_heads * c->head_dim, "d_model %d != num_heads %d * head_dim %d", c->d_model,
                      c->num_heads, c->head_dim);
    GENIE_CHECK_SHAPE(c->head_dim == 16 || c->head_dim == 32 || c->head_dim == 64,
                      "head_dim %d unsupported (16/32/64)", c->head_dim);
    GENIE_CHECK_SHAPE(c->d_model % 16 == 0 && c->d_model <= 2048, "d_model %d must be a multiple of 16, <= 2048",
                      c->d_model);
    GENIE_CHECK_SHAPE(c->hidden % 16 == 0 && c->hidden > 0, "hidden %d must be a positive multiple of 16", c->hidden);
    GENIE_CHECK_SHAPE(c->T >= 1 && c->T <= 64 && (c->T & (c->T - 1)) == 0, "T=%d must be a power of two <= 64", c->T);
    GENIE_CHECK_SHAPE(c->S >= 1 && c->S <= 1024, "S=%d must be in [1,1024]", c->S);
    GENIE_CHECK_SHAPE((size_t)c->S * c->head_dim * 8 <= 160 * 1024, "S=%d x head_dim=%d exceeds LDS", c->S,
                      c->head_dim);
    GENIE_CHECK_SHAPE(c->num_factored >= 1 && c->num_factored <= 4, "num_factored %d unsupported", c->num_factored);
    GENIE_CHECK_SHAPE(c->factored_vocab >= 1, "factored_vocab must be >= 1");
    GENIE_CHECK_SHAPE(c->precision == GENIE_PREC_EXACT || c->precision == GENIE_PREC_BF16 ||
                          c->precision == GENIE_PREC_F16X3,
                      "unknown precision %d", c->precision);
    if (c->precision == GENIE_PREC_BF16)
        GENIE_CHECK_SHAPE(c->d_model % 64 == 0 && c->hidden % 64 == 0, "bf16 precision needs d_model, hidden %% 64 == 0");
    if (c->precision == GENIE_PREC_F16X3)
        GENIE_CHECK_SHAPE(c->d_model % 32 == 0 && c->hidden % 32 == 0, "f16x3 precision needs d_model, hidden %% 32 == 0");
    return GENIE_OK;
}

static int check_ws(const genie_cfg& c, int B, void* ws, size_t bytes) {
    GENIE_CHECK_ARG(B >= 1, "B=%d must be >= 1", B);
    GENIE_CHECK_ARG(ws != nullptr, "workspace is NULL");
    size_t need = carve(c, B, nullptr).total;
    GENIE_CHECK_ARG(bytes >= need, "workspace too small: %zu < %zu bytes", bytes, need);
    return GENIE_OK;
}

// ---- one SelfAttention + residual: x += proj(attn(qkv(u)))  (attention.py:36-61, st_transformer.py:74,78)
static int attention_block(const genie_cfg& c, const genie_attn_weights& aw, const float* u, float* x, Workspace& w,
                           int B, bool temporal, hipStream_t st) {
    const int d = c.d_model, M = B * c.T * c.S;
    float* qkv = (temporal && w.tqkv) ? w.tqkv : (float*)w.big;
    float* ao = (float*)w.xn;  // u may alias w.xn: it is dead once qkv is computed
    const float* nw = c.qk_norm ? aw.norm_w : nullptr;
    const float* nb = c.qk_norm ? aw.norm_b : nullptr;
    if (temporal && w.frame_t >= 0) {  // single-frame decode: qkv -> cache slot frame_t, attend slots 0..frame_t
        float* slot = w.fcache + (size_t)w.frame_t * c.S * 3 * d;
        GENIE_TRY(launch_gemm_f32(u, d, (long)c.S * d, aw.qkv_w, d, 0, c.qkv_bias ? aw.qkv_b : nullptr, slot, 3 * d,
                                  (long)w.frame_T * c.S * 3 * d, c.S, 3 * d, d, B, 0, 1.0f, st));
        GENIE_TRY(launch_attn_temporal_single(w.fcache, ao, B, w.frame_T, c.S, w.frame_t, d, c.num_heads, c.head_dim,
                                              c.attn_scale, nw, nb, st));
        return launch_gemm_f32(ao, d, 0, aw.proj_w, d, 0, c.proj_bias ? aw.proj_b : nullptr, x, d, 0, M, d, d, 1,
                               GEMM_ACCUM, 1.0f, st);
    }
    const int Tq = (temporal && w.tqkv && w.tq_frames > c.T) ? w.tq_frames : c.T;  // frames per clip in qkv's layout
    if (Tq != c.T && B > 1)  // a short clean pass into a longer cache: one GEMM batch entry per clip
        GENIE_TRY(launch_gemm_f32(u, d, (long)c.T * c.S * d, aw.qkv_w, d, 0, c.qkv_bias ? aw.qkv_b : nullptr, qkv, 3 * d,
                                  (long)Tq * c.S * 3 * d, c.T * c.S, 3 * d, d, B, 0, 1.0f, st));
    else
    GENIE_TRY(launch_gemm_f32(u, d, 0, aw.qkv_w, d, 0, c.qkv_bias ? aw.qkv_b : nullptr, qkv, 3 * d, 0, M, 3 * d, d, 1,
                              0, 1.0f, st));
    if (temporal && w.stop_after_tqkv) return GENIE_OK;
    if (!temporal) {
        int rc = launch_attn_spatial_f32_mfma(qkv, ao, c.S, (long)B * c.T, d, c.num_heads, c.head_dim, c.attn_scale,
                                              nw, nb, st);
        if (rc == GENIE_E_UNSUPPORTED)  // no MFMA instantiation for this geometry: generic kernel
            rc = launch_attn_generic(qkv, ao, c.S, (long)B * c.T, 1, c.S, 0, 1, d, c.num_heads, c.head_dim,
                                     c.attn_scale, 0, nw, nb, st);
        GENIE_TRY(rc);
    } else if (w.tcache) {
        GENIE_TRY(launch_attn_temporal_prefix(qkv, w.tcache, ao, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale,
                                              nw, nb, st, nullptr, 0, w.tshift));
    } else {
        int rc = launch_attn_temporal_f32_mfma(qkv, ao, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale, nw, nb,
                                               st, nullptr, 0, Tq);
        if (rc == GENIE_E_UNSUPPORTED && !(Tq == c.T || B == 1)) {
            set_error("strided temporal qkv needs the MFMA temporal kernel (8 <= frames <= 16)");
            return GENIE_E_UNSUPPORTED;
        }
        if (rc == GENIE_E_UNSUPPORTED)
            rc = launch_attn_generic(qkv, ao, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, d, c.num_heads,
                                     c.head_dim, c.attn_scale, 1, nw, nb, st);
        GENIE_TRY(rc);
    }
    GENIE_TRY(launch_gemm_f32(ao, d, 0, aw.proj_w, d, 0, c.proj_bias ? aw.proj_b : nullptr, x, d, 0, M, d, d, 1,
                              GEMM_ACCUM, 1.0f, st));
    return GENIE_OK;
}

int st_block_exact(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st) {
    const int d = c.d_model, M = B * c.T * c.S;
    float* xn = (float*)w.xn;
    // spatial: x += SpAttn(norm1(x))  (st_transformer.py:73-74)
    const float* u = x;
    if (!c.qk_norm) {
        GENIE_TRY(launch_layer_norm(x, lw.norm1_w, lw.norm1_b, xn, M, d, 1e-5f, st));
        u = xn;
    }
    GENIE_TRY(attention_block(c, lw.spatial, u, x, w, B, false, st));
    // temporal: x += TmpAttn(x, causal), no pre-norm  (st_transformer.py:77-78)
    GENIE_TRY(attention_block(c, lw.temporal, x, x, w, B, true, st));
    if (w.stop_after_tqkv) return GENIE_OK;
    // MLP: x += fc2(gelu(fc1(norm2(x))))  (st_transformer.py:81, 16-25)
    u = x;
    if (!c.qk_norm) {
        GENIE_TRY(launch_layer_norm(x, lw.norm2_w, lw.norm2_b, xn, M, d, 1e-5f, st));
        u = xn;
    }
    float* hid = (float*)w.big;
    GENIE_TRY(launch_gemm_f32(u, d, 0, lw.fc1_w, d, 0, c.mlp_bias ? lw.fc1_b : nullptr, hid, c.hidden, 0, M, c.hidden,
                              d, 1, GEMM_GELU, 1.0f, st));
    GENIE_TRY(launch_gemm_f32(hid, c.hidden, 0, lw.fc2_w, c.hidden, 0, c.mlp_bias ? lw.fc2_b : nullptr, x, d, 0, M, d,
                              c.hidden, 1, GEMM_ACCUM, 1.0f, st));
    return GENIE_OK;
}

int st_block_bf16(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st);
int prepare_bf16(const genie_cfg& c, const float* x, Workspace& w, int B, hipStream_t st);
int st_block_f16x3(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st);
int prepare_f16x3(const genie_cfg& c, const float* x, Workspace& w, int B, hipStream_t st);
int readout_f16x3(const genie_cfg& c, const genie_weights& wt, const float* x, Workspace& w, int B, int t0, int t1,
                  int layout, float* logits, hipStream_t st);
int launch_pack_split(const float* src, uint16_t* dst, size_t n, hipStream_t st);
int launch_linear_lowp(int precision, const uint16_t* x16, const uint16_t* W16, const float* b, float* y, int M, int N,
                       int K, int gelu, int accumulate, hipStream_t st);
int readout_bf16(const genie_cfg& c, const genie_weights& wt, const float* x, Workspace& w, int B, int t0, int t1,
                 int layout, float* logits, hipStream_t st);

static int st_block(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st) {
    if (c.precision == GENIE_PREC_BF16) return st_block_bf16(c, lw, x, w, B, st);
    if (c.precision == GENIE_PREC_F16X3) return st_block_f16x3(c, lw, x, w, B, st);
    return st_block_exact(c, lw, x, w, B, st);
}

static int decoder(const genie_cfg& c, const genie_weights& wt, float* x, Workspace& w, int B, hipStream_t st) {
    if (c.precision == GENIE_PREC_BF16) GENIE_TRY(prepare_bf16(c, x, w, B, st));
    if (c.precision == GENIE_PREC_F16X3) GENIE_TRY(prepare_f16x3(c, x, w, B, st));
    w.ln1_done = w.qkv_planes_done = false;   // (hand-offs between consecutive blocks of ONE pass; a failed pass must not leave them set)
    for (int i = 0; i < c.num_layers; ++i) {
        w.skip_shadow_mlp = !c.qk_norm && i + 1 < c.num_layers;
        w.next_layer = i + 1 < c.num_layers ? &wt.layers_host[i + 1] : nullptr;
        GENIE_STUDY_LAYER(i);
        const int rc = st_block(c, wt.layers_host[i], x, w, B, st);
        w.skip_shadow_mlp = false;
        w.next_layer = nullptr;
        GENIE_TRY(rc);
    }
    return GENIE_OK;
}

// out_x_proj on frames [t0,t1): token-major (B,nt,S,V) or BCTHW (B,V,nt,S) via the operand-swapped GEMM
static int readout(const genie_cfg& c, const genie_weights& wt, const float* x, Workspace& w, int B, int t0, int t1,
                   int layout, float* logits, hipStream_t st) {
    if (c.precision == GENIE_PREC_BF16) return readout_bf16(c, wt, x, w, B, t0, t1, layout, logits, st);
    if (c.precision == GENIE_PREC_F16X3) return readout_f16x3(c, wt, x, w, B, t0, t1, layout, logits, st);
    const int d = c.d_model, nt = t1 - t0, V = c.factored_vocab * c.num_factored;
    const long rows = (long)nt * c.S;
    const float* xa = x + (size_t)t0 * c.S * d;
    const long strideX = (long)c.T * c.S * d;
    if (layout == GENIE_LAYOUT_TOKEN_MAJOR) {
        return launch_gemm_f32(xa, d, strideX, wt.out_w, d, 0, wt.out_b, logits, V, rows * V, (int)rows, V, d, B, 0,
                               c.readout_mult, st);
    }
    return launch_gemm_f32(wt.out_w, d, 0, xa, d, strideX, wt.out_b, logits, rows, rows * V, V, (int)rows, d, B,
                           GEMM_BIAS_ALONG_M, c.readout_mult, st);
}

static int mask_count(int step, int steps, int S) {
    // n = ceil(cos(pi/2 * (step+1)/steps) * S), python float math (st_mask_git.py:17-26,199)
    double u = (double)(step + 1) / (double)steps;
    return (int)ceil(cos(u * M_PI / 2.0) * (double)S);
}

}  // namespace genie

using namespace genie;

extern "C" {

int genie_version(void) { return GENIE_ABI_VERSION; }
int genie_abi_layout(size_t* out_host, int n) {
    const size_t v[12] = {sizeof(genie_cfg), sizeof(genie_attn_weights), offsetof(genie_attn_weights, fused_w16),
                          offsetof(genie_attn_weights, w16_wide), sizeof(genie_layer_weights),
                          offsetof(genie_layer_weights, mlp_fused_w16), offsetof(genie_layer_weights, w16_wide),
                          sizeof(genie_weights), offsetof(genie_weights, out_w16_wide), offsetof(genie_attn_weights, frame_w16),
                          offsetof(genie_layer_weights, mlp_frame_w16), offsetof(genie_weights, out_frame_w16)};
    for (int i = 0; i < n && i < 12 && out_host; ++i) out_host[i] = v[i];
    return 12;
}
const char* genie_last_error(void) { return g_err; }
int genie_check_config(const genie_cfg* cfg) { return check_cfg(cfg); }

size_t genie_workspace_bytes(const genie_cfg* cfg, int B) {
    if (check_cfg(cfg) != GENIE_OK || B < 1) return 0;
    return carve(*cfg, B, nullptr).total;
}

int genie_pack_bf16(const float* src, uint16_t* dst, size_t n, void* stream) {
    GENIE_CHECK_ARG(src && dst, "pack_bf16: NULL pointer");
    return launch_pack_bf16(src, dst, n, as_stream(stream));
}

int genie_pack_split_f16(const float* src, uint16_t* dst, size_t n, void* stream) {
    GENIE_CHECK_ARG(src && dst, "pack_split_f16: NULL pointer");
    return launch_pack_split(src, dst, n, as_stream(stream));
}

int genie_embed(const genie_cfg* cfg, const genie_weights* w, const int64_t* ids, int B, float* x, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(w && ids && x && B >= 1, "embed: bad argument");
    GENIE_CHECK_ARG(w->pos_embed && w->mask_embed && w->embed[0], "embed: weight table incomplete");
    return launch_embed(*cfg, *w, ids, B, x, as_stream(stream));
}

int genie_layer_norm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps,
                     void* stream) {
    GENIE_CHECK_ARG(x && gamma && beta && y && rows >= 0 && C >= 1, "layer_norm: bad argument");
    if (rows == 0) return GENIE_OK;
    return launch_layer_norm(x, gamma, beta, y, rows, C, eps, as_stream(stream));
}

int genie_linear(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int gelu,
                 int accumulate, void* stream) {
    GENIE_CHECK_ARG(x && W && y && M >= 0 && N >= 1 && K >= 1, "linear: bad argument");
    int flags = (gelu ? GEMM_GELU : 0) | (accumulate ? GEMM_ACCUM : 0);
    return launch_gemm_f32(x, K, 0, W, K, 0, b, y, N, 0, M, N, K, 1, flags, 1.0f, as_stream(stream));
}

int genie_spatial_attention(const genie_cfg* cfg, const genie_attn_weights* aw, const float* qkv, float* out, int B,
                            void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && qkv && out && B >= 1, "spatial_attention: bad argument");
    const genie_cfg& c = *cfg;
    const float* nw = c.qk_norm ? aw->norm_w : nullptr;
    const float* nb = c.qk_norm ? aw->norm_b : nullptr;
    int rc = launch_attn_spatial_f32_mfma(qkv, out, c.S, (long)B * c.T, c.d_model, c.num_heads, c.head_dim,
                                          c.attn_scale, nw, nb, as_stream(stream));
    if (rc != GENIE_E_UNSUPPORTED) return rc;
    return launch_attn_generic(qkv, out, c.S, (long)B * c.T, 1, c.S, 0, 1, c.d_model, c.num_heads, c.head_dim,
                               c.attn_scale, 0, nw, nb, as_stream(stream));
}

int genie_temporal_attention(const genie_cfg* cfg, const genie_attn_weights* aw, const float* qkv, float* out, int B,
                             void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && qkv && out && B >= 1, "temporal_attention: bad argument");
    const genie_cfg& c = *cfg;
    const float* nw = c.qk_norm ? aw->norm_w : nullptr;
    const float* nb = c.qk_norm ? aw->norm_b : nullptr;
    int rc = launch_attn_temporal_f32_mfma(qkv, out, B, c.T, c.S, c.d_model, c.num_heads, c.head_dim, c.attn_scale, nw,
                                           nb, as_stream(stream));
    if (rc != GENIE_E_UNSUPPORTED) return rc;
    return launch_attn_generic(qkv, out, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, c.d_model, c.num_heads,
                               c.head_dim, c.attn_scale, 1, nw, nb, as_stream(stream));
}

int genie_linear_lowp(int precision, const uint16_t* x16, const uint16_t* W16, const float* b, float* y, int M, int N,
                      int K, int gelu, int accumulate, void* stream) {
    GENIE_CHECK_ARG(x16 && W16 && y && M >= 0 && N >= 1 && K >= 1, "linear_lowp: bad argument");
    return launch_linear_lowp(precision, x16, W16, b, y, M, N, K, gelu, accumulate, as_stream(stream));
}

int genie_attention_core(const float* qkv, float* out, int n_seq, int N, int num_heads, int head_dim, float scale,
                         int causal, const float* norm_w, const float* norm_b, void* stream) {
    GENIE_CHECK_ARG(qkv && out && n_seq >= 0 && N >= 1 && num_heads >= 1, "attention_core: bad argument");
    GENIE_CHECK_ARG((norm_w == nullptr) == (norm_b == nullptr), "attention_core: norm_w/norm_b must come together");
    if (n_seq == 0) return GENIE_OK;
    if (!causal) {
        int rc = launch_attn_spatial_f32_mfma(qkv, out, N, n_seq, num_heads * head_dim, num_heads, head_dim, scale,
                                              norm_w, norm_b, as_stream(stream));
        if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    return launch_attn_generic(qkv, out, N, n_seq, 1, N, 0, 1, num_heads * head_dim, num_heads, head_dim, scale,
                               causal, norm_w, norm_b, as_stream(stream));
}

int genie_st_block_forward(const genie_cfg* cfg, const genie_layer_weights* lw_host, float* x, int B, void* workspace,
                           size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(lw_host && x, "st_block_forward: NULL pointer");
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    Workspace w = carve(*cfg, B, workspace);
    if (cfg->precision == GENIE_PREC_BF16) GENIE_TRY(prepare_bf16(*cfg, x, w, B, as_stream(stream)));
    if (cfg->precision == GENIE_PREC_F16X3) GENIE_TRY(prepare_f16x3(*cfg, x, w, B, as_stream(stream)));
    return st_block(*cfg, *lw_host, x, w, B, as_stream(stream));
}

int genie_decoder_forward(const genie_cfg* cfg, const genie_weights* wt, float* x, int B, void* workspace,
                          size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && wt->layers_host && x, "decoder_forward: NULL pointer");
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    Workspace w = carve(*cfg, B, workspace);
    return decoder(*cfg, *wt, x, w, B, as_stream(stream));
}

int genie_readout_logits(const genie_cfg* cfg, const genie_weights* wt, const float* x, int B, int t0, int t1,
                         int layout, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && x && logits && B >= 1, "readout_logits: bad argument");
    GENIE_CHECK_ARG(0 <= t0 && t0 <= t1 && t1 <= cfg->T, "readout_logits: bad frame range [%d,%d)", t0, t1);
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    Workspace w = carve(*cfg, B, workspace);
    if (t0 == t1) return GENIE_OK;
    return readout(*cfg, *wt, x, w, B, t0, t1, layout, logits, as_stream(stream));
}

int genie_compute_logits(const genie_cfg* cfg, const genie_weights* wt, const int64_t* ids, int B, int t0, int t1,
                         int layout, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && wt->layers_host && ids && logits, "compute_logits: NULL pointer");
    GENIE_CHECK_ARG(0 <= t0 && t0 <= t1 && t1 <= cfg->T, "compute_logits: bad frame range [%d,%d)", t0, t1);
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    Workspace w = carve(*cfg, B, workspace);
    hipStream_t st = as_stream(stream);
    GENIE_TRY(launch_embed(*cfg, *wt, ids, B, w.x, st));
    GENIE_TRY(decoder(*cfg, *wt, w.x, w, B, st));
    if (t0 == t1) return GENIE_OK;
    return readout(*cfg, *wt, w.x, w, B, t0, t1, layout, logits, st);
}

// ---- teacher-forced prefix reuse ---------------------------------------------------------------------
size_t genie_prefix_cache_bytes(const genie_cfg* cfg, int B) {
    if (check_cfg(cfg) != GENIE_OK || B < 1) return 0;
    return (size_t)cfg->num_layers * B * cfg->T * cfg->S * 3 * cfg->d_model * sizeof(float);
}

static int prefix_forward(const genie_cfg& c, const genie_weights& wt, const int64_t* ids, int B, float* cache,
                          bool clean, int tshift, Workspace& w, hipStream_t st, int cache_frames = 0) {
    if (cache_frames <= 0) cache_frames = c.T;  // frames per clip in the cache layout (L, B, cache_frames, S, 3d)
    const size_t per_layer = (size_t)B * cache_frames * c.S * 3 * c.d_model;
    GENIE_TRY(launch_embed(c, wt, ids, B, w.x, st));
    if (c.precision == GENIE_PREC_BF16) GENIE_TRY(prepare_bf16(c, w.x, w, B, st));
    if (c.precision == GENIE_PREC_F16X3) GENIE_TRY(prepare_f16x3(c, w.x, w, B, st));
    w.ln1_done = w.qkv_planes_done = false;
    for (int i = 0; i < c.num_layers; ++i) {
        if (clean) { w.tqkv = cache + i * per_layer; w.tcache = nullptr; w.tq_frames = cache_frames; }
        else { w.tqkv = nullptr; w.tcache = cache + i * per_layer; w.tshift = tshift; }
        w.skip_shadow_mlp = !c.qk_norm && i + 1 < c.num_layers;
        w.next_layer = i + 1 < c.num_layers ? &wt.layers_host[i + 1] : nullptr;
        w.stop_after_tqkv = clean && i + 1 == c.num_layers;  // nothing reads the clean pass's final hidden state
        GENIE_STUDY_LAYER(i);
        int rc = st_block(c, wt.layers_host[i], w.x, w, B, st);
        w.skip_shadow_mlp = false;
        w.next_layer = nullptr;
        w.stop_after_tqkv = false;
        w.tqkv = nullptr;
        w.tq_frames = 0;
        w.tcache = nullptr;
        w.tshift = 0;
        GENIE_TRY(rc);
    }
    return GENIE_OK;
}

// The prefix passes run on `nframes` <= T frame slots per clip: a private copy of the config with T = nframes (dense
// (B, nframes, S, *) buffers) and the positional table advanced to clip frame `frame0`.
static int prefix_view(const genie_cfg* cfg, const genie_weights* wt, int B, int frame0, int nframes, size_t cache_bytes,
                       genie_cfg& c2, genie_weights& w2, int cache_frames = 0) {
    if (cache_frames <= 0) cache_frames = nframes;
    GENIE_CHECK_ARG(nframes >= 1 && frame0 >= 0 && frame0 + nframes <= cfg->T, "prefix pass: frames [%d, %d) outside the clip (T=%d)",
                    frame0, frame0 + nframes, cfg->T);
    c2 = *cfg;
    c2.T = nframes;
    w2 = *wt;
    w2.pos_embed = wt->pos_embed + (size_t)frame0 * cfg->S * cfg->d_model;  // pos_embed_TSC[0, frame0 + i]
    GENIE_CHECK_ARG(cache_frames >= nframes && cache_frames <= cfg->T, "prefix pass: cache_frames %d outside [%d, %d]", cache_frames,
                    nframes, cfg->T);
    const size_t need = (size_t)cfg->num_layers * B * cache_frames * cfg->S * 3 * cfg->d_model * sizeof(float);
    GENIE_CHECK_ARG(cache_bytes >= need, "prefix pass: cache too small (%zu < %zu bytes)", cache_bytes, need);
    return GENIE_OK;
}

int genie_clean_pass(const genie_cfg* cfg, const genie_weights* wt, const int64_t* ids, int B, int nframes, int cache_frames,
                     float* cache, size_t cache_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && wt->layers_host && ids && cache, "clean_pass: NULL pointer");
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    genie_cfg c2;
    genie_weights w2;
    GENIE_TRY(prefix_view(cfg, wt, B, 0, nframes, cache_bytes, c2, w2, cache_frames));
    if (cache_frames != nframes && B > 1 && !(nframes >= 8 && nframes <= 16 && (cfg->head_dim == 32 || cfg->head_dim == 64))) {
        set_error("clean_pass: a strided cache (cache_frames %d != nframes %d) at B > 1 needs 8 <= nframes <= 16 and head_dim 32/64",
                  cache_frames, nframes);
        return GENIE_E_UNSUPPORTED;
    }
    Workspace w = carve(c2, B, workspace);
    w.model_T = cfg->T;
    return prefix_forward(c2, w2, ids, B, cache, true, 0, w, as_stream(stream), cache_frames);
}

int genie_masked_frames_logits(const genie_cfg* cfg, const genie_weights* wt, const int64_t* frames, int B, int frame0,
                               int nframes, const float* cache, size_t cache_bytes, float* logits, void* workspace,
                               size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && wt->layers_host && frames && cache && logits, "masked_frames_logits: NULL pointer");
    GENIE_CHECK_ARG(frame0 == 0 || frame0 == 1, "masked_frames_logits: frame0 = %d (0 or 1)", frame0);
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    genie_cfg c2;
    genie_weights w2;
    GENIE_TRY(prefix_view(cfg, wt, B, frame0, nframes, cache_bytes, c2, w2));
    Workspace w = carve(c2, B, workspace);
    w.model_T = cfg->T;
    hipStream_t st = as_stream(stream);
    GENIE_TRY(prefix_forward(c2, w2, frames, B, const_cast<float*>(cache), false, frame0, w, st));
    return readout(c2, w2, w.x, w, B, 0, nframes, GENIE_LAYOUT_TOKEN_MAJOR, logits, st);
}

int genie_frames_pass(const genie_cfg* cfg, const genie_weights* wt, const int64_t* frame_ids, int B, int t0, int nf, float* cache,
                      size_t cache_bytes, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && wt->layers_host && frame_ids && cache, "frames_pass: NULL pointer");
    GENIE_CHECK_ARG(nf >= 1 && t0 >= 0 && t0 + nf <= cfg->T, "frames_pass: frames [%d, %d) out of range", t0, t0 + nf);
    GENIE_CHECK_ARG(cache_bytes >= genie_prefix_cache_bytes(cfg, B), "frames_pass: cache too small");
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    genie_cfg c1 = *cfg;
    c1.T = nf;  // every buffer of this pass is a dense (B, nf, S, *) tensor
    // the fragment-order kernels (kernels_frame.hip) take the pass when they cover every layer; several frames per pass exist
    // only there
    // (the decode attention kernel holds 16 cache slots; the readout Linear writes 64-column tiles with no tail handling)
    bool fr = cfg->precision == GENIE_PREC_F16X3 && wt->out_frame_w16 && cfg->T <= 16 && (cfg->factored_vocab * cfg->num_factored) % 64 == 0;
    for (int i = 0; fr && i < c1.num_layers; ++i) fr = frame_path_takes(c1, wt->layers_host[i], (long)B * nf * cfg->S);
    if (nf > 1 && !fr) {
        set_error("frames_pass: %d frames per pass need the fragment-order kernels (f16x3, head_dim 64 or 32, frame_w16 "
                  "streams, B * nf * S <= 16,384 rows)", nf);
        return GENIE_E_UNSUPPORTED;
    }
    Workspace w = carve(c1, B, workspace);
    w.model_T = cfg->T;
    hipStream_t st = as_stream(stream);
    genie_weights w1 = *wt;
    w1.pos_embed = wt->pos_embed + (size_t)t0 * cfg->S * cfg->d_model;  // pos_embed_TSC[0, t0 + i]
    GENIE_TRY(launch_embed(c1, w1, frame_ids, B, w.x, st));
    if (!fr) {
        if (c1.precision == GENIE_PREC_BF16) GENIE_TRY(prepare_bf16(c1, w.x, w, B, st));
        if (c1.precision == GENIE_PREC_F16X3) GENIE_TRY(prepare_f16x3(c1, w.x, w, B, st));
    } else {
        GENIE_TRY(frame_prepare_f16x3(c1, w.x, w, B, nf, st));
    }
    const size_t per_layer = (size_t)B * cfg->T * cfg->S * 3 * cfg->d_model;
    for (int i = 0; i < c1.num_layers; ++i) {
        w.fcache = cache + i * per_layer;
        w.frame_t = t0;
        w.frame_T = cfg->T;
        w.skip_shadow_mlp = !c1.qk_norm && i + 1 < c1.num_layers;
        const int rc = fr ? st_block_frame_f16x3(c1, wt->layers_host[i], w.x, w, B, nf, logits && !w.skip_shadow_mlp, st)
                          : st_block(c1, wt->layers_host[i], w.x, w, B, st);
        w.skip_shadow_mlp = false;
        GENIE_TRY(rc);
    }
    if (!logits) return GENIE_OK;
    if (fr) return readout_frame_f16x3(c1, *wt, w, B, nf, nf - 1, logits, st);
    return readout(c1, *wt, w.x, w, B, 0, 1, GENIE_LAYOUT_TOKEN_MAJOR, logits, st);
}

int genie_frame_pass(const genie_cfg* cfg, const genie_weights* wt, const int64_t* frame_ids, int B, int t, float* cache,
                     size_t cache_bytes, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    return genie_frames_pass(cfg, wt, frame_ids, B, t, 1, cache, cache_bytes, logits, workspace, workspace_bytes, stream);
}

int genie_frame_linear(const uint16_t* a_fr, const uint16_t* w_fr, const float* bias, float* y, int M, int N, int K, int mode, void* stream) {
    GENIE_CHECK_ARG(a_fr && w_fr && y && mode >= 0 && mode <= 2, "frame_linear: bad argument");
    return launch_frame_linear(a_fr, w_fr, bias, y, M, N, K, mode, as_stream(stream));
}

// rows of S token ids: dst[b][0..S) = src[b][0..S) (clip strides in elements) or, src == NULL, the fill value
__global__ void frame_ids_kernel(const int64_t* __restrict__ src, long src_stride, int64_t* __restrict__ dst, long dst_stride, int S, int B,
                                 int64_t fill) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const long b = i / S, s = i - b * S;
    dst[b * dst_stride + s] = src ? src[b * src_stride + s] : fill;
}
static int put_frame_ids(const int64_t* src, long src_stride, int64_t* dst, long dst_stride, int S, int B, int64_t fill, hipStream_t st) {
    frame_ids_kernel<<<(unsigned)(((long)B * S + 255) / 256), 256, 0, st>>>(src, src_stride, dst, dst_stride, S, B, fill);
    GENIE_LAUNCH_CHECK("frame_ids");
    return GENIE_OK;
}

// where the loop scratch of genie_generate_cached starts: behind the workspace of its largest pass
static size_t generate_scratch_offset(const genie_cfg& c, int B, int P) {
    genie_cfg cm = c;
    cm.T = P > 2 ? P : 2;
    return carve(cm, B, nullptr).total;
}

size_t genie_generate_workspace_bytes(const genie_cfg* cfg, int B, int P) {
    if (check_cfg(cfg) != GENIE_OK || B < 1 || P < 1 || P > cfg->T) return 0;
    const size_t BS = (size_t)B * cfg->S, V = (size_t)cfg->factored_vocab * cfg->num_factored;
    size_t off = generate_scratch_offset(*cfg, B, P);
    for (size_t bytes : {BS * P * 8, BS * 2 * 8, BS * 8, BS * 8, BS * 8, BS * 4, BS, BS * V * 4}) off += align_up(bytes, 256);
    const size_t full = carve(*cfg, B, nullptr).total;   // (the passes themselves check against the model's own workspace size)
    return off > full ? off : full;
}

int genie_generate_cached(const genie_cfg* cfg, const genie_weights* wt, const int64_t* ids, int B, int P, int n_new, int steps,
                          float temperature, int unmask_mode, const float* noise, const float* uniforms, int teacher_force_time,
                          int merge_commit, int64_t* gen_out, float* logits0_out, float* cache, size_t cache_bytes, void* workspace,
                          size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    const genie_cfg& c = *cfg;
    GENIE_CHECK_ARG(wt && wt->layers_host && ids && gen_out && cache, "generate_cached: NULL pointer");
    GENIE_CHECK_ARG(B >= 1 && P >= 1 && n_new >= 1 && P + n_new <= c.T && steps >= 1,
                    "generate_cached: B=%d, %d prompt + %d new frames of at most %d, steps %d", B, P, n_new, c.T, steps);
    if (unmask_mode != GENIE_UNMASK_RANDOM && unmask_mode != GENIE_UNMASK_GREEDY) {
        set_error("Expected `unmask_mode` to be one of ['greedy', 'random']");
        return GENIE_E_UNSUPPORTED;
    }
    GENIE_CHECK_ARG(steps == 1 || unmask_mode == GENIE_UNMASK_GREEDY || noise,
                    "generate_cached: 'random' unmasking with steps > 1 needs the caller's U[0,1) draws");
    GENIE_CHECK_ARG(temperature <= 1e-8f || uniforms, "generate_cached: temperature > 0 needs uniforms");
    GENIE_CHECK_ARG(cache_bytes >= genie_prefix_cache_bytes(cfg, B), "generate_cached: cache too small");
    GENIE_TRY(check_ws(c, B, workspace, workspace_bytes));
    hipStream_t st = as_stream(stream);
    const int S = c.S, T = P + n_new;   // frames per clip in `ids` (the cache keeps the model's c.T slots per clip)
    const size_t BS = (size_t)B * S, V = (size_t)c.factored_vocab * c.num_factored;
    // scratch of the loop behind the workspace of its largest pass (the prompt's P frames; two frames for the merged passes):
    // genie_generate_workspace_bytes(cfg, B, P) is the size to allocate
    char* base = (char*)workspace;
    size_t off = generate_scratch_offset(c, B, P);
    auto take = [&](size_t bytes) { char* r = base + off; off += align_up(bytes, 256); return r; };
    int64_t* idsP = (int64_t*)take(BS * P * 8);
    int64_t* two = (int64_t*)take(BS * 2 * 8);
    int64_t* cur = (int64_t*)take(BS * 8);
    int64_t* fin = (int64_t*)take(BS * 8);
    int64_t* samples = (int64_t*)take(BS * 8);
    float* conf = (float*)take(BS * 4);
    uint8_t* unmasked = (uint8_t*)take(BS);
    float* logits = (float*)take(BS * V * 4);
    GENIE_CHECK_ARG(off <= workspace_bytes, "generate_cached: workspace too small (%zu < %zu bytes: size it with genie_generate_workspace_bytes)",
                    workspace_bytes, off);

    // ---- the prompt fills cache slots 0 .. P-1: one P-frame pass where the fragment-order kernels cover it, else the clean pass
    // with the cache's T-frame layout, else frame by frame
    for (int t = 0; t < P; ++t) GENIE_TRY(put_frame_ids(ids + (size_t)t * S, (long)T * S, idsP + (size_t)t * S, (long)P * S, S, B, 0, st));
    int rc = GENIE_E_UNSUPPORTED;
    if (P > 1) {
        rc = genie_frames_pass(cfg, wt, idsP, B, 0, P, cache, cache_bytes, nullptr, workspace, workspace_bytes, stream);
        if (rc == GENIE_E_UNSUPPORTED)
            rc = genie_clean_pass(cfg, wt, idsP, B, P, c.T, cache, cache_bytes, workspace, workspace_bytes, stream);
    }
    if (rc == GENIE_E_UNSUPPORTED) {
        for (int t = 0; t < P; ++t) {
            GENIE_TRY(put_frame_ids(ids + (size_t)t * S, (long)T * S, fin, S, S, B, 0, st));
            GENIE_TRY(genie_frames_pass(cfg, wt, fin, B, t, 1, cache, cache_bytes, nullptr, workspace, workspace_bytes, stream));
        }
    } else {
        GENIE_TRY(rc);
    }
    bool opened = false, merge = merge_commit != 0;
    for (int k = 0; k < n_new; ++k) {
        const int t = P + k;
        GENIE_TRY(put_frame_ids(nullptr, 0, cur, S, S, B, c.image_vocab_size, st));
        if (hipMemsetAsync(unmasked, 0, BS, st) != hipSuccess) { set_error("memset failed"); return GENIE_E_LAUNCH; }
        for (int step = 0; step < steps; ++step) {
            if (!(step == 0 && opened))
                GENIE_TRY(genie_frames_pass(cfg, wt, cur, B, t, 1, cache, cache_bytes, logits, workspace, workspace_bytes, stream));
            if (step == 0 && logits0_out) {   // orig_logits of the frame (st_mask_git.py:165,226): the step-0 logits, (B, n_new, S, V)
                if (hipMemcpy2DAsync(logits0_out + (size_t)k * S * V, (size_t)n_new * S * V * 4, logits, (size_t)S * V * 4, (size_t)S * V * 4,
                                     (size_t)B, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                    set_error("memcpy failed");
                    return GENIE_E_LAUNCH;
                }
            }
            const float* u = temperature > 1e-8f ? uniforms + ((size_t)k * steps + step) * c.num_factored * BS : nullptr;
            GENIE_TRY(launch_sample(c, logits, GENIE_LAYOUT_TOKEN_MAJOR, B, temperature, u, samples, conf, st));
            const bool last = step == steps - 1;
            const float* keys = last ? nullptr : (unmask_mode == GENIE_UNMASK_GREEDY ? conf : noise + ((size_t)k * (steps - 1) + step) * BS);
            GENIE_TRY(launch_mask_step(keys, last ? 0 : mask_count(step, steps, S), last, c.image_vocab_size, unmasked, samples, cur, S, B, S, st));
        }
        GENIE_TRY(put_frame_ids(cur, S, gen_out + (size_t)k * S, (long)n_new * S, S, B, 0, st));
        opened = false;
        if (t + 1 < T) {   // commit frame t: its final tokens, or the ground truth when teacher-forcing in time
            const int64_t* fsrc = teacher_force_time ? ids + (size_t)t * S : cur;
            const long fstride = teacher_force_time ? (long)T * S : S;
            if (merge) {   // ... in the pass that also carries MaskGIT step 0 of frame t + 1 (all-mask tokens)
                GENIE_TRY(put_frame_ids(fsrc, fstride, two, 2L * S, S, B, 0, st));
                GENIE_TRY(put_frame_ids(nullptr, 0, two + S, 2L * S, S, B, c.image_vocab_size, st));
                rc = genie_frames_pass(cfg, wt, two, B, t, 2, cache, cache_bytes, logits, workspace, workspace_bytes, stream);
                if (rc == GENIE_E_UNSUPPORTED) merge = false;
                else { GENIE_TRY(rc); opened = true; }
            }
            if (!opened) {
                GENIE_TRY(put_frame_ids(fsrc, fstride, fin, S, S, B, 0, st));
                GENIE_TRY(genie_frames_pass(cfg, wt, fin, B, t, 1, cache, cache_bytes, nullptr, workspace, workspace_bytes, stream));
            }
        }
    }
    return GENIE_OK;
}

int genie_pack_frame_w16(const float* src, uint16_t* dst, int N, int K, void* stream) {
    GENIE_CHECK_ARG(src && dst, "pack_frame_w16: NULL pointer");
    return launch_pack_frame_w16(src, dst, N, K, as_stream(stream));
}

int genie_factored_ce(const genie_cfg* cfg, const float* logits, int layout, const int64_t* targets,
                      const int64_t* weight_ids, int B, int t0, int t1, double* sums_out, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(logits && targets && sums_out && B >= 1, "factored_ce: bad argument");
    GENIE_CHECK_ARG(0 <= t0 && t0 <= t1 && t1 <= cfg->T, "factored_ce: bad frame range [%d,%d)", t0, t1);
    return launch_factored_ce(*cfg, logits, layout, targets, weight_ids, B, t0, t1, sums_out, as_stream(stream));
}

int genie_metric_hits(const int64_t* truth, int64_t truth_batch_stride, const int64_t* samples, int64_t samples_batch_stride,
                      int batch, int64_t n_per_batch, const double* ce3, double n_tokens, double n_frames, double n_clips,
                      double* sums6, void* stream) {
    GENIE_CHECK_ARG(truth && samples && sums6 && batch >= 0 && n_per_batch >= 0, "metric_hits: bad argument");
    return launch_count_equal(truth, (long)truth_batch_stride, samples, (long)samples_batch_stride, batch, (long)n_per_batch, ce3,
                              sums6, n_tokens, n_frames, n_clips, as_stream(stream));
}

int genie_readout_ce(const genie_cfg* cfg, const genie_weights* wt, const float* x, const int64_t* targets,
                     const int64_t* weight_ids, int B, int t0, int t1, double* sums_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(wt && x && targets && sums_out, "readout_ce: NULL pointer");
    GENIE_CHECK_ARG(0 <= t0 && t0 <= t1 && t1 <= cfg->T, "readout_ce: bad frame range [%d,%d)", t0, t1);
    GENIE_TRY(check_ws(*cfg, B, workspace, workspace_bytes));
    if (t0 == t1) return GENIE_OK;
    Workspace w = carve(*cfg, B, workspace);
    hipStream_t st = as_stream(stream);
    GENIE_TRY(readout(*cfg, *wt, x, w, B, t0, t1, GENIE_LAYOUT_TOKEN_MAJOR, w.logits, st));
    return launch_factored_ce(*cfg, w.logits, GENIE_LAYOUT_TOKEN_MAJOR, targets, weight_ids, B, t0, t1, sums_out, st);
}

int genie_sample(const genie_cfg* cfg, const float* logits, int layout, int B, float temperature,
                 const float* uniforms, int64_t* samples, float* conf, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(logits && samples && conf && B >= 1, "sample: bad argument");
    GENIE_CHECK_ARG(temperature <= 1e-8f || uniforms, "sample: temperature > 0 needs caller-supplied uniforms");
    return launch_sample(*cfg, logits, layout, B, temperature, uniforms, samples, conf, as_stream(stream));
}

int genie_mask_step(const float* keys, int n, int last_step, int64_t mask_id, uint8_t* unmasked, int64_t* samples,
                    int64_t* prompt_frame, int64_t prompt_clip_stride, int B, int S, void* stream) {
    GENIE_CHECK_ARG(unmasked && samples && prompt_frame && B >= 1 && S >= 1, "mask_step: bad argument");
    GENIE_CHECK_ARG(last_step || keys, "mask_step: keys required unless last_step");
    GENIE_CHECK_ARG(last_step || (n >= 0 && n <= S), "mask_step: n=%d out of range", n);
    return launch_mask_step(keys, n, last_step, mask_id, unmasked, samples, prompt_frame, (long)prompt_clip_stride, B,
                            S, as_stream(stream));
}

int genie_maskgit_generate(const genie_cfg* cfg, const genie_weights* wt, int64_t* prompt, int B, int out_t, int steps,
                           float temperature, int unmask_mode, const float* noise, const float* uniforms,
                           int64_t* samples_out, float* logits0_out, int layout, int32_t* status_flag,
                           void* workspace, size_t workspace_bytes, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    const genie_cfg& c = *cfg;
    GENIE_CHECK_ARG(wt && wt->layers_host && prompt && samples_out, "maskgit_generate: NULL pointer");
    if (!(out_t >= 1 && out_t < c.T)) {  // assert out_t  (st_mask_git.py:154)
        set_error("maskgit_generate requires 0 < out_t < T (got %d)", out_t);
        return GENIE_E_ASSERT;
    }
    GENIE_CHECK_ARG(steps >= 1, "maskgit_generate: steps=%d must be >= 1", steps);
    if (unmask_mode != GENIE_UNMASK_RANDOM && unmask_mode != GENIE_UNMASK_GREEDY) {
        set_error("Expected `unmask_mode` to be one of ['greedy', 'random']");
        return GENIE_E_UNSUPPORTED;
    }
    GENIE_CHECK_ARG(steps == 1 || unmask_mode == GENIE_UNMASK_GREEDY || noise,
                    "maskgit_generate: 'random' unmasking with steps > 1 needs the caller's U[0,1) draws");
    GENIE_CHECK_ARG(temperature <= 1e-8f || uniforms, "maskgit_generate: temperature > 0 needs uniforms");
    GENIE_TRY(check_ws(c, B, workspace, workspace_bytes));
    Workspace w = carve(c, B, workspace);
    hipStream_t st = as_stream(stream);
    const size_t BS = (size_t)B * c.S;
    const long V = (long)c.factored_vocab * c.num_factored;

    GENIE_TRY(launch_check_masked(prompt, B, c.T, c.S, out_t, c.image_vocab_size, status_flag, st));
    if (hipMemsetAsync(w.unmasked, 0, BS, st) != hipSuccess) { set_error("memset failed"); return GENIE_E_LAUNCH; }
    for (int step = 0; step < steps; ++step) {
        GENIE_TRY(launch_embed(c, *wt, prompt, B, w.x, st));
        GENIE_TRY(decoder(c, *wt, w.x, w, B, st));
        GENIE_TRY(readout(c, *wt, w.x, w, B, out_t, out_t + 1, GENIE_LAYOUT_TOKEN_MAJOR, w.logits, st));
        if (step == 0 && logits0_out) {  // orig_logits_CHW: step-0 logits are what is returned (:165,226)
            if (layout == GENIE_LAYOUT_TOKEN_MAJOR) {
                if (hipMemcpyAsync(logits0_out, w.logits, BS * V * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                    set_error("memcpy failed");
                    return GENIE_E_LAUNCH;
                }
            } else {
                GENIE_TRY(launch_transpose(w.logits, logits0_out, B, c.S, (int)V, st));
            }
        }
        const float* u = (temperature > 1e-8f) ? uniforms + (size_t)step * c.num_factored * BS : nullptr;
        GENIE_TRY(launch_sample(c, w.logits, GENIE_LAYOUT_TOKEN_MAJOR, B, temperature, u, w.samples, w.conf, st));
        const int last = (step == steps - 1);
        const float* keys = nullptr;
        int n = 0;
        if (!last) {
            n = mask_count(step, steps, c.S);
            keys = (unmask_mode == GENIE_UNMASK_GREEDY) ? w.conf : noise + (size_t)step * BS;
        }
        GENIE_TRY(launch_mask_step(keys, n, last, c.image_vocab_size, w.unmasked, w.samples,
                                   prompt + (size_t)out_t * c.S, (long)c.T * c.S, B, c.S, st));
    }
    if (hipMemcpyAsync(samples_out, w.samples, BS * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("memcpy failed");
        return GENIE_E_LAUNCH;
    }
    return GENIE_OK;
}

int genie_profile_enable(int class_mask) {
    if (class_mask && !g_prof_ev) {
        g_prof_ev = new hipEvent_t[2 * GENIE_PROFILE_MAX_LAUNCHES];
        for (int i = 0; i < 2 * GENIE_PROFILE_MAX_LAUNCHES; ++i) {
            if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) {
                set_error("profile: hipEventCreate failed at %d", i);
                return GENIE_E_LAUNCH;
            }
        }
    }
    g_prof_mask = class_mask;
    return GENIE_OK;
}

int genie_profile_reset(void) {
    g_prof_n = 0;
    return GENIE_OK;
}

int genie_profile_read(int kernel_class, double* out4) {
    GENIE_CHECK_ARG(out4 && kernel_class >= 0 && kernel_class < GENIE_KC_COUNT, "profile_read: bad argument");
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    for (int i = 0; i < g_prof_n; ++i) {
        if (g_prof_cls[i] != kernel_class) continue;
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) { set_error("profile: sync failed"); return GENIE_E_LAUNCH; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) {
            set_error("profile: elapsed failed");
            return GENIE_E_LAUNCH;
        }
        out4[0] += 1.0;
        out4[1] += ms;
        out4[2] += g_prof_flops[i];
        out4[3] += g_prof_bytes[i];
    }
    return GENIE_OK;
}

int genie_profile_kernels(int kernel_class, char* buf, size_t buf_bytes) {
    GENIE_CHECK_ARG(buf && buf_bytes >= 64 && kernel_class >= 0 && kernel_class < GENIE_KC_COUNT, "profile_kernels: bad argument");
    // distinct kernel names of the class (string literals: compared by content), launches / ms / flops each
    constexpr int MAXK = 32;
    const char* names[MAXK];
    double n[MAXK], ms[MAXK], fl[MAXK];
    int nk = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        if (g_prof_cls[i] != kernel_class) continue;
        int k = 0;
        while (k < nk && strcmp(names[k], g_prof_name[i]) != 0) ++k;
        if (k == nk) {
            if (nk == MAXK) continue;
            names[nk] = g_prof_name[i]; n[nk] = ms[nk] = fl[nk] = 0.0; ++nk;
        }
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) { set_error("profile: sync failed"); return GENIE_E_LAUNCH; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) { set_error("profile: elapsed failed"); return GENIE_E_LAUNCH; }
        n[k] += 1.0; ms[k] += t; fl[k] += g_prof_flops[i];
    }
    size_t off = 0;
    buf[0] = 0;
    for (int k = 0; k < nk; ++k) {
        const int w = snprintf(buf + off, buf_bytes - off, "%s\t%.0f\t%.6f\t%.6e\n", names[k], n[k], ms[k], fl[k]);
        if (w < 0 || (size_t)w >= buf_bytes - off) break;
        off += (size_t)w;
    }
    return GENIE_OK;
}

int genie_pack_temporal_fused_bf16(const float* qkv_w, const float* proj_w, uint16_t* dst, void* stream) {
    GENIE_CHECK_ARG(qkv_w && proj_w && dst, "pack_temporal_fused: NULL pointer");
    return launch_pack_temporal_fused(qkv_w, proj_w, dst, as_stream(stream));
}
int genie_pack_temporal_qkv_f16x3(const float* qkv_w, uint16_t* dst, void* stream) {
    GENIE_CHECK_ARG(qkv_w && dst, "pack_temporal_qkv_f16x3: NULL pointer");
    return launch_pack_temporal_qkv_f16x3(qkv_w, dst, as_stream(stream));
}
int genie_temporal_prefix_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, float* x, uint16_t* kv, int B, int nframes, int mode,
                                     int shift, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && x && kv && B >= 1 && nframes >= 1, "temporal_prefix_fused: bad argument");
    genie_cfg c2 = *cfg;
    c2.T = nframes;   // the pass's frames; cfg->T is the model's
    return launch_temporal_prefix_fused_bf16(c2, *aw, x, kv, B, mode, shift, cfg->T, as_stream(stream));
}
int genie_temporal_qkv_attn_f16x3(const genie_cfg* cfg, const genie_attn_weights* aw, const float* x, uint16_t* a16, int64_t plane_elems,
                                  float* kv, int B, int nframes, int mode, int shift, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && x && a16 && B >= 1 && nframes >= 1, "temporal_qkv_attn_f16x3: bad argument");
    genie_cfg c2 = *cfg;
    c2.T = nframes;
    return launch_temporal_qkv_attn_f16x3(c2, *aw, x, a16, (long)plane_elems, kv, B, mode, shift, cfg->T, as_stream(stream));
}
int genie_temporal_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, const uint16_t* x16, float* x, int B, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && x && B >= 1, "temporal_fused: bad argument");   // x16 == NULL: operands rounded from x itself
    return launch_temporal_fused_bf16(*cfg, *aw, x16, x, B, as_stream(stream));
}
int genie_mlp_fused_bf16(const genie_cfg* cfg, const genie_layer_weights* lw, float* x, uint16_t* x16_out, int64_t rows,
                         const float* next_norm_w, const float* next_norm_b, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(lw && x && rows >= 0, "mlp_fused: bad argument");
    return launch_mlp_fused_bf16(*cfg, *lw, x, x16_out, (long)rows, as_stream(stream), next_norm_w, next_norm_b);
}
int genie_mlp_fused_qkv_bf16(const genie_cfg* cfg, const genie_layer_weights* lw, const genie_layer_weights* next, float* x,
                             uint16_t* planes, int64_t rows, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(lw && next && x && planes && rows >= 0, "mlp_fused_qkv: bad argument");
    if (!next->spatial.fused_w16 || !(next->spatial.w16_wide & GENIE_FUSED_QKV_STREAM) || !next->norm1_w || !next->norm1_b)
        return GENIE_E_UNSUPPORTED;
    return launch_mlp_fused_bf16(*cfg, *lw, x, nullptr, (long)rows, as_stream(stream), next->norm1_w, next->norm1_b,
                                 next->spatial.fused_w16 + GENIE_SPATIAL_PROJ_FUSED_ELEMS, planes);
}
int genie_pack_spatial_proj_fused_bf16(const float* proj_w, uint16_t* dst, void* stream) {
    GENIE_CHECK_ARG(proj_w && dst, "pack_spatial_proj_fused: NULL pointer");
    return launch_pack_spatial_proj(proj_w, dst, as_stream(stream));
}
int genie_pack_spatial_qkv_fused_bf16(const float* qkv_w, uint16_t* dst, void* stream) {
    GENIE_CHECK_ARG(qkv_w && dst, "pack_spatial_qkv_fused: NULL pointer");
    return launch_pack_spatial_qkv(qkv_w, dst, as_stream(stream));
}
int genie_spatial_attn_proj_fused_bf16(const genie_cfg* cfg, const genie_attn_weights* aw, const uint16_t* qkv_planes, float* x,
                                       uint16_t* x16, int64_t n_seq, void* stream) {
    GENIE_TRY(check_cfg(cfg));
    GENIE_CHECK_ARG(aw && qkv_planes && x && n_seq >= 1, "spatial_attn_proj_fused: bad argument");   // x16 == NULL: no bf16 shadow written
    return launch_spatial_attn_proj_bf16(*cfg, *aw, qkv_planes, x, x16, (long)n_seq, as_stream(stream));
}
int genie_pack_mlp_fused_bf16(const float* fc1_w, const float* fc2_w, uint16_t* dst, void* stream) {
    GENIE_CHECK_ARG(fc1_w && fc2_w && dst, "pack_mlp_fused: NULL pointer");
    return launch_pack_mlp_fused(fc1_w, fc2_w, dst, as_stream(stream));
}

int genie_study_build(void) { return kStudyBuild ? 1 : 0; }

int genie_bits_from_tokens(const int64_t* ids, float* z, int n, int hw, int bits, void* stream) {
    GENIE_CHECK_ARG(ids && z && n >= 0 && hw >= 1 && bits >= 1 && bits <= 62, "bits_from_tokens: bad argument");
    return launch_bits(ids, z, n, hw, bits, as_stream(stream));
}

int genie_rescale_u8_bf16(const uint16_t* x, uint8_t* out, size_t n, void* stream) {
    GENIE_CHECK_ARG(x && out, "rescale_u8: NULL pointer");
    return launch_rescale_u8(x, 1, out, n, as_stream(stream));
}
int genie_rescale_u8_f32(const float* x, uint8_t* out, size_t n, void* stream) {
    GENIE_CHECK_ARG(x && out, "rescale_u8: NULL pointer");
    return launch_rescale_u8(x, 0, out, n, as_stream(stream));
}
int genie_tokens_from_bits(const float* h, int64_t* ids, int n, int hw, int bits, void* stream) {
    GENIE_CHECK_ARG(h && ids && n >= 0 && hw >= 1 && bits >= 1 && bits <= 62, "tokens_from_bits: bad argument");
    return launch_tokens_from_bits(h, ids, n, hw, bits, as_stream(stream));
}

int genie_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int taps, void* stream) {
    GENIE_CHECK_ARG(w && out && Cout >= 1 && Cin >= 1 && taps >= 1, "pack_conv_weight: bad argument");
    return launch_pack_conv_weight(w, out, Cout, Cin, taps, as_stream(stream));
}
int genie_conv3x3_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, const uint16_t* residual, uint16_t* y,
                       const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, int depth_to_space, void* stream) {
    GENIE_CHECK_ARG(x && w_packed && y && zero_page && n >= 0 && H >= 1 && W >= 1, "conv3x3: bad argument");
    return launch_conv3x3_igemm(x, w_packed, bias, residual, y, zero_page, n, H, W, Cin, Cout, depth_to_space,
                                as_stream(stream));
}
int genie_conv3x3_s2_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, uint16_t* y,
                          const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, void* stream) {
    GENIE_CHECK_ARG(x && w_packed && y && zero_page && n >= 0 && H >= 1 && W >= 1, "conv3x3_s2: bad argument");
    return launch_conv3x3_igemm(x, w_packed, bias, nullptr, y, zero_page, n, H, W, Cin, Cout, 0, as_stream(stream), 2);
}
int genie_frames_to_nhwc_bf16(const uint8_t* frames, uint16_t* x, int n, int HW, int cin, int cpad, void* stream) {
    GENIE_CHECK_ARG(frames && x && n >= 0 && HW >= 1 && cin >= 1 && cpad >= cin, "frames_to_nhwc: bad argument");
    return launch_frames_to_nhwc(frames, x, n, HW, cin, cpad, as_stream(stream));
}
int genie_tokens_from_code_nhwc_bf16(const uint16_t* h, int64_t* ids, int64_t n_pix, int bits, int cpad, void* stream) {
    GENIE_CHECK_ARG(h && ids && n_pix >= 0 && bits >= 1 && bits <= 62 && cpad >= bits, "tokens_from_code: bad argument");
    return launch_tokens_from_nhwc(h, ids, (long)n_pix, bits, cpad, as_stream(stream));
}
int genie_conv1x1_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, uint16_t* y, int n_pix, int Cin,
                       int Cout, void* stream) {
    GENIE_CHECK_ARG(x && w_packed && y && n_pix >= 0, "conv1x1: bad argument");
    return launch_gemm_bf16_out16(x, w_packed, bias, y, n_pix, Cout, Cin, as_stream(stream));
}
int genie_conv_direct_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, void* y, int n, int H, int W,
                           int Cin, int Cout, int out_mode, void* stream) {
    GENIE_CHECK_ARG(x && w_packed && y && n >= 0 && (out_mode == 0 || out_mode == 1), "conv_direct: bad argument");
    return launch_conv_direct(x, w_packed, bias, y, n, H, W, Cin, Cout, out_mode, as_stream(stream));
}
size_t genie_group_norm_scratch_floats(int n, int HW, int groups) {
    if (n <= 0 || HW <= 0 || groups <= 0) return 0;
    return gn_scratch_floats(n, HW, groups);
}
int genie_group_norm_swish_bf16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* stats_ws, int n,
                                int HW, int C, int groups, float eps, int apply_swish, void* stream) {
    GENIE_CHECK_ARG(x && gamma && beta && y && stats_ws && n >= 0, "group_norm_swish: bad argument");
    if (n == 0) return GENIE_OK;
    return launch_gn_swish(x, gamma, beta, y, stats_ws, n, HW, C, groups, eps, apply_swish, as_stream(stream));
}
size_t genie_conv_gn_part_floats(int n, int H, int W, int Cout) {
    if (n <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    return conv_gn_part_floats(n, H, W, Cout);
}
int genie_conv3x3_gn_bf16(const uint16_t* x, const uint16_t* w_packed, const float* bias, const uint16_t* residual, uint16_t* y,
                          const uint16_t* zero_page, int n, int H, int W, int Cin, int Cout, int depth_to_space, int stride,
                          float* gn_part, int groups, void* stream) {
    GENIE_CHECK_ARG(x && w_packed && y && zero_page && gn_part && n >= 0 && H >= 1 && W >= 1, "conv3x3_gn: bad argument");
    return launch_conv3x3_igemm(x, w_packed, bias, residual, y, zero_page, n, H, W, Cin, Cout, depth_to_space,
                                as_stream(stream), stride, gn_part, groups);
}
int genie_group_norm_swish_fused_bf16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y,
                                      const float* gn_part, float* stats_ws, int n, int H, int W, int Cout, int depth_to_space,
                                      int groups, float eps, int apply_swish, void* stream) {
    GENIE_CHECK_ARG(x && gamma && beta && y && gn_part && stats_ws && n >= 0 && groups >= 1, "group_norm_swish_fused: bad argument");
    if (n == 0) return GENIE_OK;
    return launch_gn_swish_tiles(x, gamma, beta, y, gn_part, stats_ws, n, H, W, Cout, depth_to_space, groups, eps, apply_swish,
                                 as_stream(stream));
}
int genie_bits_from_tokens_nhwc_bf16(const int64_t* ids, uint16_t* z, int64_t n_pix, int bits, int cpad, void* stream) {
    GENIE_CHECK_ARG(ids && z && n_pix >= 0 && bits >= 1 && bits <= 62 && cpad >= bits, "bits_nhwc: bad argument");
    return launch_bits_nhwc(ids, z, (long)n_pix, bits, cpad, as_stream(stream));
}
int genie_rescale_u8_nhwc_bf16(const uint16_t* x, uint8_t* out, int n, int HW, int cpad, int cout, void* stream) {
    GENIE_CHECK_ARG(x && out && n >= 0 && HW >= 1 && cout >= 1 && cpad >= cout, "rescale_u8_nhwc: bad argument");
    return launch_rescale_nhwc_u8(x, out, n, HW, cpad, cout, as_stream(stream));
}

}  // extern "C"
