// C ABI of the training step (include/genie_hip.h, "training" section): forward with saved activations, masked
// factored CE, backward layer by layer (so the caller can overlap the gradient all-reduce of finished layers with
// the backward of earlier ones), AdamW.  Two variants: GENIE_PREC_EXACT (this first part: f32 storage, every contraction on
// the f32 matrix instruction) and the 16-bit matrix-core variant for GENIE_PREC_BF16 / _F16X3 further down.
//
// HBM layout of the saved activations, exact variant (floats; M = B*T*S tokens, token-major rows):
//   per layer l at l*per_layer:  x0 (M,d) layer input | u1 (M,d) norm1(x0) | qkv_s (M,3d) | ao_s (M,d) spatial attention
//   output before proj | x1 (M,d) | qkv_t (M,3d) | ao_t (M,d) | x2 (M,d) | u2 (M,d) norm2(x2) | z (M,hid) fc1 pre-activation
//   | h (M,hid) gelu(z);  after the layers: xL (M,d) | logits (M,V) (replaced in place by d loss / d logits)
// 21*d floats per token and layer: 5.6 GB per clip for the C138 shape -- sized for 288 GB, nothing is recomputed
// except the attention probabilities (rebuilt inside the fused attention backward kernels; for S != 256 the spatial scores are
// materialised per layer in the workspace, never saved).
#include "kernels.hpp"

namespace genie {

struct TrainActs {
    size_t per_layer, o_x0, o_u1, o_qkvs, o_aos, o_x1, o_qkvt, o_aot, o_x2, o_u2, o_z, o_h, o_xL, o_logits, total;
};
static TrainActs train_acts(const genie_cfg& c, int B) {
    const size_t M = (size_t)B * c.T * c.S, d = c.d_model, hid = c.hidden;
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    TrainActs a;
    size_t o = 0;
    a.o_x0 = o; o += M * d;
    a.o_u1 = o; o += M * d;
    a.o_qkvs = o; o += M * 3 * d;
    a.o_aos = o; o += M * d;
    a.o_x1 = o; o += M * d;
    a.o_qkvt = o; o += M * 3 * d;
    a.o_aot = o; o += M * d;
    a.o_x2 = o; o += M * d;
    a.o_u2 = o; o += M * d;
    a.o_z = o; o += M * hid;
    a.o_h = o; o += M * hid;
    a.per_layer = o;
    a.o_xL = a.per_layer * c.num_layers;
    a.o_logits = a.o_xL + M * d;
    a.total = a.o_logits + M * V;
    return a;
}

struct TrainWs {
    float *dx, *d1, *g, *p, *dp, *slabs, *lnpart, *colpart;
    float* qkn;  // (M, 2d) normalised q | k of the attention being differentiated (qk_norm only)
    double* dscratch;  // 1024 doubles (sumsq partials)
    size_t slab_floats, total;
};
static TrainWs train_ws(const genie_cfg& c, int B, void* base) {
    const size_t M = (size_t)B * c.T * c.S, d = c.d_model;
    const size_t wide = (size_t)(3 * d > (size_t)c.hidden ? 3 * d : c.hidden);
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    const size_t scores = M * c.num_heads * c.S;
    size_t maxw = (size_t)c.hidden * d;
    if (3 * d * d > maxw) maxw = 3 * d * d;
    if (V * d > maxw) maxw = V * d;
    size_t maxn = wide > V ? wide : V;
    TrainWs w;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t at = o; o += (floats * 4 + 255) / 256 * 256; return at; };
    const size_t o_dx = take(M * d), o_d1 = take(M * d), o_g = take(M * wide), o_p = take(scores), o_dp = take(scores);
    w.slab_floats = 64 * maxw;
    const size_t o_sl = take(w.slab_floats), o_ln = take(ln_bwd_scratch_floats((int)d));
    const size_t cp_rows = M / 64 > (size_t)COLSUM_CHUNKS ? M / 64 : (size_t)COLSUM_CHUNKS;
    const size_t o_cp = take(cp_rows * maxn), o_ds = take(2 * 1024);
    const size_t o_qkn = take(c.qk_norm ? M * 2 * d : 0);
    w.total = o;
    char* b = (char*)base;
    w.dx = (float*)(b + o_dx); w.d1 = (float*)(b + o_d1); w.g = (float*)(b + o_g); w.p = (float*)(b + o_p);
    w.dp = (float*)(b + o_dp); w.slabs = (float*)(b + o_sl); w.lnpart = (float*)(b + o_ln);
    w.colpart = (float*)(b + o_cp); w.dscratch = (double*)(b + o_ds);
    w.qkn = (float*)(b + o_qkn);
    return w;
}

static int train_check(const genie_cfg* c, int B) {
    GENIE_CHECK_ARG(c != nullptr && B > 0, "training: cfg is NULL or B <= 0");
    GENIE_TRY(genie_check_config(c));
    if (c->precision != GENIE_PREC_EXACT)  // 16-bit GEMM operands: 64x64 transposition tiles, K-steps of 64
        GENIE_CHECK_SHAPE(c->d_model % 64 == 0 && c->hidden % 64 == 0 && (c->T * c->S) % 64 == 0 &&
                              (c->factored_vocab * c->num_factored) % 64 == 0,
                          "training step (16-bit): d_model, hidden, T*S and the vocabulary rows must be multiples of 64");
    GENIE_CHECK_SHAPE(c->S % 16 == 0 && c->head_dim % 16 == 0 && c->d_model % 16 == 0 && c->hidden % 16 == 0 && c->T <= 16,
                      "training step: S, head_dim, d_model, hidden must be multiples of 16 and T <= 16");
    return GENIE_OK;
}

// y = x . W^T + b (+ R) on the general GEMM
static int lin(const float* x, long ldx, const float* W, const float* b, const float* R, float* y, long ldy, int M, int N,
               int K, float alpha, hipStream_t st) {
    return launch_gemm_f32_gen(false, false, x, ldx, 0, 0, W, K, 0, 0, b, R, y, ldy, 0, 0, M, N, K, 1, 1, 1, 0, alpha, st);
}
// dx = alpha * dy . W (+ R), W (N,K) row-major read k-major
static int dgrad(const float* dy, const float* W, const float* R, float* dx, int M, int N, int K, float alpha,
                 hipStream_t st) {
    return launch_gemm_f32_gen(false, true, dy, N, 0, 0, W, K, 0, 0, nullptr, R, dx, K, 0, 0, M, K, N, 1, 1, 1, 0, alpha,
                               st);
}

static int spatial_attn_fwd(const genie_cfg& c, const genie_attn_weights& aw, const float* qkv, float* ao, int B,
                            hipStream_t st) {
    const float* nw = c.qk_norm ? aw.norm_w : nullptr;
    const float* nb = c.qk_norm ? aw.norm_b : nullptr;
    int rc = launch_attn_spatial_f32_mfma(qkv, ao, c.S, (long)B * c.T, c.d_model, c.num_heads, c.head_dim, c.attn_scale,
                                          nw, nb, st);
    if (rc == GENIE_E_UNSUPPORTED)
        rc = launch_attn_generic(qkv, ao, c.S, (long)B * c.T, 1, c.S, 0, 1, c.d_model, c.num_heads, c.head_dim,
                                 c.attn_scale, 0, nw, nb, st);
    return rc;
}
static int temporal_attn_fwd(const genie_cfg& c, const genie_attn_weights& aw, const float* qkv, float* ao, int B,
                             hipStream_t st) {
    const float* nw = c.qk_norm ? aw.norm_w : nullptr;
    const float* nb = c.qk_norm ? aw.norm_b : nullptr;
    int rc = launch_attn_temporal_f32_mfma(qkv, ao, B, c.T, c.S, c.d_model, c.num_heads, c.head_dim, c.attn_scale, nw, nb,
                                           st);
    if (rc == GENIE_E_UNSUPPORTED)
        rc = launch_attn_generic(qkv, ao, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, c.d_model, c.num_heads,
                                 c.head_dim, c.attn_scale, 1, nw, nb, st);
    return rc;
}
// where the backward reads q and k: the saved qkv itself, or (qk_norm) their LayerNorm'd copies in w.qkn
struct QkSrc { const float* p; long ld; };
static int qk_source(const genie_cfg& c, const genie_attn_weights& aw, const float* qkv, TrainWs& w, int B, QkSrc* out,
                     hipStream_t st) {
    if (!c.qk_norm) { *out = QkSrc{qkv, 3L * c.d_model}; return GENIE_OK; }
    GENIE_TRY(launch_qk_norm_fwd(qkv, w.qkn, aw.norm_w, aw.norm_b, (long)B * c.T * c.S, c.num_heads, c.head_dim, c.d_model,
                                 st));
    *out = QkSrc{w.qkn, 2L * c.d_model};
    return GENIE_OK;
}
// d/d(normalised q,k) -> d/d(raw q,k) in place, and the shared affine's gradient
static int qk_norm_backward(const genie_cfg& c, const genie_attn_weights& aw, const genie_attn_weights& g, const float* qkv,
                            float* dqkv, TrainWs& w, int B, float beta, hipStream_t st) {
    if (!c.qk_norm) return GENIE_OK;
    return launch_qk_norm_bwd(qkv, dqkv, aw.norm_w, (float*)g.norm_w, (float*)g.norm_b, (long)B * c.T * c.S, c.num_heads,
                              c.head_dim, c.d_model, beta, w.lnpart, st);
}

// spatial attention backward through materialised scores: qkv (M,3d), dao (M,d) -> dqkv (M,3d)
static int spatial_attn_bwd(const genie_cfg& c, const float* qkv, QkSrc qk, const float* dao, float* dqkv, TrainWs& w,
                            int B, hipStream_t st) {
    const int S = c.S, d = c.d_model, H = c.num_heads, Dh = c.head_dim, BT = B * c.T;
    const long q3 = (long)S * 3 * d, ss = (long)S * S, hss = (long)H * ss, sd = (long)S * d;
    const long qld = qk.ld, qs = (long)S * qk.ld;
    const float sc = c.attn_scale;
    if (c.precision == GENIE_PREC_BF16) {   // bf16 training: the backward products on the bf16 matrix cores as well
        const int rc = launch_attn_spatial_bwd_bf16(qkv, qk.p, qk.ld, dao, dqkv, w.p, BT, S, d, H, Dh, sc, st);
        if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    {   // production geometry: fused kernel, no S x S traffic
        const int rc = launch_attn_spatial_bwd_fused(qkv, qk.p, qk.ld, dao, dqkv, BT, S, d, H, Dh, sc, st);
        if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    // P = softmax(scale Q K^T)
    GENIE_TRY(launch_gemm_f32_gen(false, false, qk.p, qld, qs, Dh, qk.p + d, qld, qs, Dh, nullptr, nullptr, w.p, S, hss,
                                  ss, S, S, Dh, BT, H, 1, 0, sc, st));
    GENIE_TRY(launch_softmax_rows(w.p, (long)BT * H * S, S, st));
    // dV = P^T dO
    GENIE_TRY(launch_gemm_f32_gen(true, true, w.p, S, hss, ss, dao, d, sd, Dh, nullptr, nullptr, dqkv + 2 * d, 3 * d, q3,
                                  Dh, S, Dh, S, BT, H, 1, 0, 1.0f, st));
    // dP = dO V^T, dS = P (dP - rowsum(P dP))
    GENIE_TRY(launch_gemm_f32_gen(false, false, dao, d, sd, Dh, qkv + 2 * d, 3 * d, q3, Dh, nullptr, nullptr, w.dp, S, hss,
                                  ss, S, S, Dh, BT, H, 1, 0, 1.0f, st));
    GENIE_TRY(launch_softmax_bwd_rows(w.p, w.dp, (long)BT * H * S, S, st));
    // dQ = scale dS K,  dK = scale dS^T Q
    GENIE_TRY(launch_gemm_f32_gen(false, true, w.dp, S, hss, ss, qk.p + d, qld, qs, Dh, nullptr, nullptr, dqkv, 3 * d, q3,
                                  Dh, S, Dh, S, BT, H, 1, 0, sc, st));
    GENIE_TRY(launch_gemm_f32_gen(true, true, w.dp, S, hss, ss, qk.p, qld, qs, Dh, nullptr, nullptr, dqkv + d, 3 * d, q3,
                                  Dh, S, Dh, S, BT, H, 1, 0, sc, st));
    return GENIE_OK;
}


// ================================================================================================
// 16-bit matrix-core variant (GENIE_PREC_BF16 / GENIE_PREC_F16X3): every Linear product -- forward, dgrad, wgrad --
// runs on the NT 16-bit GEMM of kernels_bf16.hip; LayerNorm, softmax, both attention cores (forward and backward; bf16:
// the spatial backward's products on the bf16 matrix cores, kernels_attn_bwd16.hip),
// GELU, the residual stream, CE and all gradient reductions stay f32 exactly as in the exact variant.
// Saved per layer (bytes/token: 52 d f32 + 18 d NPL 16-bit): f32 x0, qkv_s, x1, qkv_t, x2, z;  16-bit GEMM operands
// u1 = norm1(x0), ao_s, x1, ao_t, u2 = norm2(x2), h = gelu(z).
// ================================================================================================
struct TrainActs16 {
    size_t per_layer;                            // bytes
    size_t x0, qkvs, x1, qkvt, x2, z;            // byte offsets inside a layer (f32)
    size_t u1, aos, x1h, aot, u2, h;             // byte offsets inside a layer (16-bit, NPL planes)
    size_t xL, xL16, logits, total;              // byte offsets in the buffer
};
static TrainActs16 train_acts16(const genie_cfg& c, int B, int npl) {
    const size_t M = (size_t)B * c.T * c.S, d = c.d_model, hid = c.hidden;
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    TrainActs16 a;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    a.x0 = take(M * d * 4); a.qkvs = take(M * 3 * d * 4); a.x1 = take(M * d * 4); a.qkvt = take(M * 3 * d * 4);
    a.x2 = take(M * d * 4); a.z = take(M * hid * 4);
    a.u1 = take(M * d * 2 * npl); a.aos = take(M * d * 2 * npl); a.x1h = take(M * d * 2 * npl);
    a.aot = take(M * d * 2 * npl); a.u2 = take(M * d * 2 * npl); a.h = take(M * hid * 2 * npl);
    a.per_layer = o;
    o = a.per_layer * c.num_layers;
    a.xL = take(M * d * 4); a.xL16 = take(M * d * 2 * npl); a.logits = take(M * V * 4);
    a.total = o;
    return a;
}
struct TrainWs16 {
    uint16_t *dy16, *dy16T, *xT16;
    size_t total;
};
// appended behind the exact workspace
static TrainWs16 train_ws16(const genie_cfg& c, int B, int npl, void* base, size_t exact_total) {
    const size_t M = (size_t)B * c.T * c.S, d = c.d_model;
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    size_t wide = (size_t)(3 * d > (size_t)c.hidden ? 3 * d : c.hidden);
    if (V > wide) wide = V;
    size_t o = exact_total;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    const size_t o1 = take(M * wide * 2 * npl), o2 = take(M * wide * 2 * npl), o3 = take(M * wide * 2 * npl);
    TrainWs16 w;
    w.total = o;
    char* b = (char*)base;
    w.dy16 = (uint16_t*)(b + o1); w.dy16T = (uint16_t*)(b + o2); w.xT16 = (uint16_t*)(b + o3);
    return w;
}
// bf16: the weight gradient runs on the TN kernel (kernels_gemm_tn.hip) from row-major operands when the shapes allow, and then
// neither the gradient nor the saved activation needs a transposed copy (GENIE_WGRAD_TN=0: the transposed-copy path, for A/B)
static bool use_tn(int npl, int M, int N, int K) {
    static const int tn = study_env("GENIE_WGRAD_TN", 1);
    return tn && npl == 1 && M % 64 == 0 && N % 256 == 0 && K % 128 == 0;
}
// 16-bit copies of a gradient matrix (row-major, and transposed unless its weight gradient takes the TN kernel: Kw = the K of
// that weight gradient) and, in the same pass, its column sums (= the bias gradient)
static int cast_t_bias(int npl, float* in, int cols, const float* z, TrainWs16& h, int M, float* dbias, float beta,
                       TrainWs& w, hipStream_t st, int Kw) {
    if (use_tn(npl, M, cols, Kw))
        GENIE_TRY(launch_cast_rows16(in, cols, z, h.dy16, M, cols, st, dbias ? w.colpart : nullptr));
    else
        GENIE_TRY(launch_cast_transpose16(npl, in, cols, z, h.dy16, h.dy16T, M, cols, st, dbias ? w.colpart : nullptr));
    return dbias ? launch_slab_reduce(w.colpart, M / 64, (size_t)cols, dbias, beta, st) : GENIE_OK;
}
static inline int npl_of(const genie_cfg& c) { return c.precision == GENIE_PREC_BF16 ? 1 : 2; }
static inline long PL(int npl, size_t n) { return npl == 2 ? (long)n : 0; }

// y(f32) / y16 = x16 . W16^T (+b) (+R)
static int lin16(int npl, const uint16_t* x16, size_t nx, const uint16_t* W16, const float* b, const float* R, float* y,
                 uint16_t* y16, int M, int N, int K, float alpha, hipStream_t st) {
    int flags = 0;
    if (y) flags |= G16X_OUTF32;
    if (y16) flags |= G16X_OUT16;
    if (R) flags |= G16X_ACCUM;
    return launch_gemm16_ex(npl, x16, K, PL(npl, nx), W16, K, PL(npl, (size_t)N * K), b, (R && R != y) ? R : nullptr, y, y16,
                            PL(npl, (size_t)M * N), N, M, N, K, flags, alpha, st, 1, 0, 0, 0);
}
// dW[N,K] (beta*dW +)= alpha * dY^T . X from the TRANSPOSED 16-bit copies dYT (N, Mtok), XT (K, Mtok)
static int wgrad16(int npl, const uint16_t* dYT, const uint16_t* XT, float* dW, int Mtok, int N, int K, float alpha,
                   float beta, float* slabs, size_t slab_floats, hipStream_t st) {
    const int tiles = ((N + 255) / 256) * ((K + 127) / 128);
    int ns = 1;
    while (ns < 64 && tiles * ns < 256 && Mtok % (64 * ns * 2) == 0 && (size_t)(ns * 2) * N * K <= slab_floats) ns *= 2;
    const long pa = PL(npl, (size_t)N * Mtok), pw = PL(npl, (size_t)K * Mtok);
    if (ns == 1)
        return launch_gemm16_ex(npl, dYT, Mtok, pa, XT, Mtok, pw, nullptr, nullptr, dW, nullptr, 0, K, N, K, Mtok,
                                G16X_OUTF32 | (beta != 0.f ? G16X_ACCUM : 0), alpha, st, 1, 0, 0, 0);
    const int kc = Mtok / ns;
    GENIE_TRY(launch_gemm16_ex(npl, dYT, Mtok, pa, XT, Mtok, pw, nullptr, nullptr, slabs, nullptr, 0, K, N, K, kc,
                               G16X_OUTF32, alpha, st, ns, kc, kc, (long)N * K));
    return launch_slab_reduce(slabs, ns, (size_t)N * K, dW, beta, st);
}

// weight gradient from the gradient copies in `h` and the saved row-major 16-bit activation X16 (Mtok, K)
static int wgrad_any(int npl, TrainWs16& h, const uint16_t* X16, float* dW, int Mtok, int N, int K, float alpha, float beta,
                     TrainWs& w, hipStream_t st) {
    if (use_tn(npl, Mtok, N, K)) {
        const int rc = launch_wgrad16_tn(h.dy16, N, X16, K, dW, Mtok, N, K, alpha, beta, w.slabs, w.slab_floats, st);
        GENIE_CHECK_SHAPE(rc != GENIE_E_UNSUPPORTED, "wgrad: TN kernel refused N=%d K=%d after its operand was prepared", N, K);
        return rc;
    }
    GENIE_TRY(launch_transpose16(npl, X16, h.xT16, Mtok, K, st));
    return wgrad16(npl, h.dy16T, h.xT16, dW, Mtok, N, K, alpha, beta, w.slabs, w.slab_floats, st);
}

static int attn_fwd16(const genie_cfg& c, const genie_attn_weights& aw, const float* qkv, bool temporal, uint16_t* out16,
                      float* tmp, int B, int npl, hipStream_t st) {
    const float* nw = c.qk_norm ? aw.norm_w : nullptr;
    const float* nb = c.qk_norm ? aw.norm_b : nullptr;
    const size_t pd = (size_t)B * c.T * c.S * c.d_model;
    int rc;
    if (!temporal) {
        rc = launch_attn_spatial_split(qkv, nullptr, c.S, (long)B * c.T, c.d_model, c.num_heads, c.head_dim, c.attn_scale,
                                       nw, nb, st, out16, PL(npl, pd));
        if (rc == GENIE_E_UNSUPPORTED) {
            GENIE_TRY(launch_attn_generic(qkv, tmp, c.S, (long)B * c.T, 1, c.S, 0, 1, c.d_model, c.num_heads, c.head_dim,
                                          c.attn_scale, 0, nw, nb, st));
            rc = launch_cast16(npl, tmp, out16, pd, st);
        }
    } else {
        rc = launch_attn_temporal_f32_mfma(qkv, nullptr, B, c.T, c.S, c.d_model, c.num_heads, c.head_dim, c.attn_scale, nw,
                                           nb, st, out16, PL(npl, pd));
        if (rc == GENIE_E_UNSUPPORTED) {
            GENIE_TRY(launch_attn_generic(qkv, tmp, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, c.d_model,
                                          c.num_heads, c.head_dim, c.attn_scale, 1, nw, nb, st));
            rc = launch_cast16(npl, tmp, out16, pd, st);
        }
    }
    return rc;
}

static int need16(const genie_weights* wt, const genie_cfg& c, const char* what) {
    GENIE_CHECK_ARG(wt && wt->out_w16, "%s: 16-bit weight copies missing (genie_train_pack_weights)", what);
    for (int l = 0; l < c.num_layers; ++l) {
        const genie_layer_weights& lw = wt->layers_host[l];
        GENIE_CHECK_ARG(lw.spatial.qkv_w16 && lw.spatial.proj_w16 && lw.temporal.qkv_w16 && lw.temporal.proj_w16 &&
                            lw.fc1_w16 && lw.fc2_w16,
                        "%s: 16-bit weight copies missing in layer %d (genie_train_pack_weights)", what, l);
    }
    return GENIE_OK;
}

static int train_forward16(const genie_cfg& c, const genie_weights* wt, const int64_t* input_ids, const int64_t* labels,
                           int B, char* acts, size_t acts_bytes, double* sums, hipStream_t st) {
    const int npl = npl_of(c);
    GENIE_TRY(need16(wt, c, "genie_train_forward"));
    const TrainActs16 a = train_acts16(c, B, npl);
    GENIE_CHECK_ARG(acts_bytes >= a.total, "genie_train_forward: activation buffer too small: %zu < %zu", acts_bytes, a.total);
    const int d = c.d_model, hid = c.hidden, M = B * c.T * c.S, V = c.factored_vocab * c.num_factored;
    GENIE_CHECK_SHAPE(V >= d, "training step (16-bit): vocabulary rows %d < d_model %d", V, d);
    const size_t pd = (size_t)M * d, ph = (size_t)M * hid;
    float* tmp = (float*)(acts + a.logits);  // free until the readout
    auto F = [&](int l, size_t off) { return (float*)(acts + a.per_layer * l + off); };
    auto H16 = [&](int l, size_t off) { return (uint16_t*)(acts + a.per_layer * l + off); };
    GENIE_TRY(launch_embed(c, *wt, input_ids, B, F(0, a.x0), st));
    if (c.qk_norm) GENIE_TRY(launch_cast16(npl, F(0, a.x0), H16(0, a.u1), pd, st));
    for (int l = 0; l < c.num_layers; ++l) {
        const genie_layer_weights& lw = wt->layers_host[l];
        const bool last = l + 1 == c.num_layers;
        float* xnext = last ? (float*)(acts + a.xL) : F(l + 1, a.x0);
        uint16_t* xnext16 = last ? (uint16_t*)(acts + a.xL16) : H16(l + 1, a.u1);
        if (!c.qk_norm) {
            if (npl == 1) GENIE_TRY(launch_layer_norm_bf16(F(l, a.x0), lw.norm1_w, lw.norm1_b, H16(l, a.u1), M, d, 1e-5f, st));
            else GENIE_TRY(launch_layer_norm_split(F(l, a.x0), lw.norm1_w, lw.norm1_b, H16(l, a.u1), pd, M, d, 1e-5f, st));
        }
        GENIE_TRY(lin16(npl, H16(l, a.u1), pd, lw.spatial.qkv_w16, c.qkv_bias ? lw.spatial.qkv_b : nullptr, nullptr,
                        F(l, a.qkvs), nullptr, M, 3 * d, d, 1.0f, st));
        GENIE_TRY(attn_fwd16(c, lw.spatial, F(l, a.qkvs), false, H16(l, a.aos), tmp, B, npl, st));
        GENIE_TRY(lin16(npl, H16(l, a.aos), pd, lw.spatial.proj_w16, c.proj_bias ? lw.spatial.proj_b : nullptr, F(l, a.x0),
                        F(l, a.x1), H16(l, a.x1h), M, d, d, 1.0f, st));
        GENIE_TRY(lin16(npl, H16(l, a.x1h), pd, lw.temporal.qkv_w16, c.qkv_bias ? lw.temporal.qkv_b : nullptr, nullptr,
                        F(l, a.qkvt), nullptr, M, 3 * d, d, 1.0f, st));
        GENIE_TRY(attn_fwd16(c, lw.temporal, F(l, a.qkvt), true, H16(l, a.aot), tmp, B, npl, st));
        GENIE_TRY(lin16(npl, H16(l, a.aot), pd, lw.temporal.proj_w16, c.proj_bias ? lw.temporal.proj_b : nullptr,
                        F(l, a.x1), F(l, a.x2), c.qk_norm ? H16(l, a.u2) : nullptr, M, d, d, 1.0f, st));
        if (!c.qk_norm) {
            if (npl == 1) GENIE_TRY(launch_layer_norm_bf16(F(l, a.x2), lw.norm2_w, lw.norm2_b, H16(l, a.u2), M, d, 1e-5f, st));
            else GENIE_TRY(launch_layer_norm_split(F(l, a.x2), lw.norm2_w, lw.norm2_b, H16(l, a.u2), pd, M, d, 1e-5f, st));
        }
        // fc1: the f32 pre-activation is kept for gelu'; the 16-bit operand of fc2 gets gelu applied in the epilogue
        GENIE_TRY(launch_gemm16_ex(npl, H16(l, a.u2), d, PL(npl, pd), lw.fc1_w16, d, PL(npl, (size_t)hid * d),
                                   c.mlp_bias ? lw.fc1_b : nullptr, nullptr, F(l, a.z), H16(l, a.h), PL(npl, ph), hid, M, hid,
                                   d, G16X_OUTF32 | G16X_OUT16 | G16X_GELU16, 1.0f, st, 1, 0, 0, 0));
        GENIE_TRY(lin16(npl, H16(l, a.h), ph, lw.fc2_w16, c.mlp_bias ? lw.fc2_b : nullptr, F(l, a.x2), xnext,
                        c.qk_norm ? xnext16 : nullptr, M, d, hid, 1.0f, st));
    }
    if (!c.qk_norm) GENIE_TRY(launch_cast16(npl, (float*)(acts + a.xL), (uint16_t*)(acts + a.xL16), pd, st));
    GENIE_TRY(lin16(npl, (uint16_t*)(acts + a.xL16), pd, wt->out_w16, wt->out_b, nullptr, (float*)(acts + a.logits), nullptr,
                    M, V, d, c.readout_mult, st));
    if (hipMemsetAsync(sums, 0, 3 * sizeof(double), st) != hipSuccess) {
        set_error("genie_train_forward: hipMemsetAsync failed");
        return GENIE_E_LAUNCH;
    }
    return launch_ce_fwd_bwd(c, (float*)(acts + a.logits), input_ids, labels, B, sums, st);
}

static int train_backward_head16(const genie_cfg& c, const genie_weights* wt, const genie_weights* wT,
                                 const genie_weights* grads, int B, char* acts, TrainWs& w, TrainWs16& h, float beta,
                                 hipStream_t st) {
    const int npl = npl_of(c);
    GENIE_TRY(need16(wT, c, "genie_train_backward_head (transposed copies)"));
    const TrainActs16 a = train_acts16(c, B, npl);
    const int d = c.d_model, M = B * c.T * c.S, V = c.factored_vocab * c.num_factored;
    float* dl = (float*)(acts + a.logits);
    GENIE_TRY(cast_t_bias(npl, dl, V, nullptr, h, M, (float*)grads->out_b, beta, w, st, d));
    GENIE_TRY(wgrad_any(npl, h, (const uint16_t*)(acts + a.xL16), (float*)grads->out_w, M, V, d, c.readout_mult, beta, w, st));
    return lin16(npl, h.dy16, (size_t)M * V, wT->out_w16, nullptr, nullptr, w.dx, nullptr, M, d, V, c.readout_mult, st);
}

static int train_backward_layer16(const genie_cfg& c, const genie_weights* wt, const genie_weights* wT,
                                  const genie_weights* grads, int layer, int B, char* acts, TrainWs& w, TrainWs16& h,
                                  float beta, hipStream_t st) {
    const int npl = npl_of(c);
    const TrainActs16 a = train_acts16(c, B, npl);
    const int d = c.d_model, hid = c.hidden, M = B * c.T * c.S;
    const size_t pd = (size_t)M * d, ph = (size_t)M * hid, p3 = (size_t)M * 3 * d;
    const genie_layer_weights& lw = wt->layers_host[layer];
    const genie_layer_weights& lt = wT->layers_host[layer];
    const genie_layer_weights& g = grads->layers_host[layer];
    char* L = acts + a.per_layer * layer;
    auto F = [&](size_t off) { return (float*)(L + off); };
    auto H16 = [&](size_t off) { return (const uint16_t*)(L + off); };
    float* dx = w.dx;
    QkSrc qk;

    // ---- MLP
    GENIE_TRY(cast_t_bias(npl, dx, d, nullptr, h, M, c.mlp_bias ? (float*)g.fc2_b : nullptr, beta, w, st, hid));
    GENIE_TRY(wgrad_any(npl, h, H16(a.h), (float*)g.fc2_w, M, d, hid, 1.0f, beta, w, st));
    GENIE_TRY(lin16(npl, h.dy16, pd, lt.fc2_w16, nullptr, nullptr, w.g, nullptr, M, hid, d, 1.0f, st));         // dh
    GENIE_TRY(cast_t_bias(npl, w.g, hid, F(a.z), h, M, c.mlp_bias ? (float*)g.fc1_b : nullptr, beta, w, st, d));  // dz
    GENIE_TRY(wgrad_any(npl, h, H16(a.u2), (float*)g.fc1_w, M, hid, d, 1.0f, beta, w, st));
    if (c.qk_norm) {
        GENIE_TRY(lin16(npl, h.dy16, ph, lt.fc1_w16, nullptr, dx, dx, nullptr, M, d, hid, 1.0f, st));
    } else {
        GENIE_TRY(lin16(npl, h.dy16, ph, lt.fc1_w16, nullptr, nullptr, w.d1, nullptr, M, d, hid, 1.0f, st));
        GENIE_TRY(launch_ln_bwd(F(a.x2), lw.norm2_w, w.d1, dx, (float*)g.norm2_w, (float*)g.norm2_b, M, d, 1e-5f, beta,
                                w.lnpart, st));
    }

    // ---- temporal
    GENIE_TRY(cast_t_bias(npl, dx, d, nullptr, h, M, c.proj_bias ? (float*)g.temporal.proj_b : nullptr, beta, w, st, d));
    GENIE_TRY(wgrad_any(npl, h, H16(a.aot), (float*)g.temporal.proj_w, M, d, d, 1.0f, beta, w, st));
    GENIE_TRY(lin16(npl, h.dy16, pd, lt.temporal.proj_w16, nullptr, nullptr, w.d1, nullptr, M, d, d, 1.0f, st));
    GENIE_TRY(qk_source(c, lw.temporal, F(a.qkvt), w, B, &qk, st));
    GENIE_TRY(launch_attn_temporal_bwd(F(a.qkvt), qk.p, qk.ld, w.d1, w.g, B, c.T, c.S, d, c.num_heads, c.head_dim,
                                       c.attn_scale, st));
    GENIE_TRY(qk_norm_backward(c, lw.temporal, g.temporal, F(a.qkvt), w.g, w, B, beta, st));
    GENIE_TRY(cast_t_bias(npl, w.g, 3 * d, nullptr, h, M, c.qkv_bias ? (float*)g.temporal.qkv_b : nullptr, beta, w, st, d));
    GENIE_TRY(wgrad_any(npl, h, H16(a.x1h), (float*)g.temporal.qkv_w, M, 3 * d, d, 1.0f, beta, w, st));
    GENIE_TRY(lin16(npl, h.dy16, p3, lt.temporal.qkv_w16, nullptr, dx, dx, nullptr, M, d, 3 * d, 1.0f, st));

    // ---- spatial
    GENIE_TRY(cast_t_bias(npl, dx, d, nullptr, h, M, c.proj_bias ? (float*)g.spatial.proj_b : nullptr, beta, w, st, d));
    GENIE_TRY(wgrad_any(npl, h, H16(a.aos), (float*)g.spatial.proj_w, M, d, d, 1.0f, beta, w, st));
    GENIE_TRY(lin16(npl, h.dy16, pd, lt.spatial.proj_w16, nullptr, nullptr, w.d1, nullptr, M, d, d, 1.0f, st));
    GENIE_TRY(qk_source(c, lw.spatial, F(a.qkvs), w, B, &qk, st));
    GENIE_TRY(spatial_attn_bwd(c, F(a.qkvs), qk, w.d1, w.g, w, B, st));
    GENIE_TRY(qk_norm_backward(c, lw.spatial, g.spatial, F(a.qkvs), w.g, w, B, beta, st));
    GENIE_TRY(cast_t_bias(npl, w.g, 3 * d, nullptr, h, M, c.qkv_bias ? (float*)g.spatial.qkv_b : nullptr, beta, w, st, d));
    GENIE_TRY(wgrad_any(npl, h, H16(a.u1), (float*)g.spatial.qkv_w, M, 3 * d, d, 1.0f, beta, w, st));
    if (c.qk_norm) return lin16(npl, h.dy16, p3, lt.spatial.qkv_w16, nullptr, dx, dx, nullptr, M, d, 3 * d, 1.0f, st);
    GENIE_TRY(lin16(npl, h.dy16, p3, lt.spatial.qkv_w16, nullptr, nullptr, w.d1, nullptr, M, d, 3 * d, 1.0f, st));
    return launch_ln_bwd(F(a.x0), lw.norm1_w, w.d1, dx, (float*)g.norm1_w, (float*)g.norm1_b, M, d, 1e-5f, beta, w.lnpart, st);
}

}  // namespace genie

using namespace genie;

extern "C" {

size_t genie_train_activation_bytes(const genie_cfg* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    if (cfg->precision != GENIE_PREC_EXACT) return train_acts16(*cfg, B, npl_of(*cfg)).total;
    return train_acts(*cfg, B).total * sizeof(float);
}
size_t genie_train_workspace_bytes(const genie_cfg* cfg, int B) {
    if (!cfg || B <= 0) return 0;
    const size_t exact = train_ws(*cfg, B, nullptr).total;
    if (cfg->precision != GENIE_PREC_EXACT) return train_ws16(*cfg, B, npl_of(*cfg), nullptr, exact).total;
    return exact;
}

int genie_train_pack_weights(const genie_cfg* cfg, const genie_weights* w, const genie_weights* w16,
                             const genie_weights* w16T, void* stream) {
    GENIE_CHECK_ARG(cfg && w && w16 && w16T, "genie_train_pack_weights: NULL argument");
    GENIE_CHECK_ARG(cfg->precision != GENIE_PREC_EXACT, "genie_train_pack_weights: exact precision has no 16-bit copies");
    const genie_cfg& c = *cfg;
    const int npl = npl_of(c), d = c.d_model, hid = c.hidden, V = c.factored_vocab * c.num_factored;
    hipStream_t st = (hipStream_t)stream;
    auto pack = [&](const float* src, const uint16_t* rowm, const uint16_t* tr, int out, int in) -> int {
        GENIE_CHECK_ARG(src && rowm && tr, "genie_train_pack_weights: missing pointer");
        GENIE_TRY(launch_cast16(npl, src, (uint16_t*)rowm, (size_t)out * in, st));
        return launch_cast_transpose16(npl, (float*)src, in, nullptr, nullptr, (uint16_t*)tr, out, in, st);
    };
    for (int l = 0; l < c.num_layers; ++l) {
        const genie_layer_weights &a = w->layers_host[l], &b = w16->layers_host[l], &t = w16T->layers_host[l];
        GENIE_TRY(pack(a.spatial.qkv_w, b.spatial.qkv_w16, t.spatial.qkv_w16, 3 * d, d));
        GENIE_TRY(pack(a.spatial.proj_w, b.spatial.proj_w16, t.spatial.proj_w16, d, d));
        GENIE_TRY(pack(a.temporal.qkv_w, b.temporal.qkv_w16, t.temporal.qkv_w16, 3 * d, d));
        GENIE_TRY(pack(a.temporal.proj_w, b.temporal.proj_w16, t.temporal.proj_w16, d, d));
        GENIE_TRY(pack(a.fc1_w, b.fc1_w16, t.fc1_w16, hid, d));
        GENIE_TRY(pack(a.fc2_w, b.fc2_w16, t.fc2_w16, d, hid));
    }
    return pack(w->out_w, w16->out_w16, w16T->out_w16, V, d);
}

int genie_train_forward(const genie_cfg* cfg, const genie_weights* wt, const int64_t* input_ids, const int64_t* labels,
                        int B, float* acts, size_t acts_bytes, double* sums, void* stream) {
    GENIE_TRY(train_check(cfg, B));
    GENIE_CHECK_ARG(wt && input_ids && labels && acts && sums, "genie_train_forward: NULL argument");
    const genie_cfg& c = *cfg;
    if (c.precision != GENIE_PREC_EXACT)
        return train_forward16(c, wt, input_ids, labels, B, (char*)acts, acts_bytes, sums, (hipStream_t)stream);
    const TrainActs a = train_acts(c, B);
    GENIE_CHECK_ARG(acts_bytes >= a.total * sizeof(float), "genie_train_forward: activation buffer too small: %zu < %zu",
                    acts_bytes, a.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const int d = c.d_model, hid = c.hidden, M = B * c.T * c.S;
    GENIE_TRY(launch_embed(c, *wt, input_ids, B, acts + a.o_x0, st));
    for (int l = 0; l < c.num_layers; ++l) {
        const genie_layer_weights& lw = wt->layers_host[l];
        float* L = acts + a.per_layer * l;
        float* xnext = (l + 1 < c.num_layers) ? acts + a.per_layer * (l + 1) + a.o_x0 : acts + a.o_xL;
        // qk_norm: norm1/norm2 are Identity (st_transformer.py:44,67) and the u1/u2 slots stay unused
        const float* u1 = c.qk_norm ? L + a.o_x0 : L + a.o_u1;
        const float* u2 = c.qk_norm ? L + a.o_x2 : L + a.o_u2;
        if (!c.qk_norm) GENIE_TRY(launch_layer_norm(L + a.o_x0, lw.norm1_w, lw.norm1_b, L + a.o_u1, M, d, 1e-5f, st));
        GENIE_TRY(lin(u1, d, lw.spatial.qkv_w, c.qkv_bias ? lw.spatial.qkv_b : nullptr, nullptr, L + a.o_qkvs,
                      3 * d, M, 3 * d, d, 1.0f, st));
        GENIE_TRY(spatial_attn_fwd(c, lw.spatial, L + a.o_qkvs, L + a.o_aos, B, st));
        GENIE_TRY(lin(L + a.o_aos, d, lw.spatial.proj_w, c.proj_bias ? lw.spatial.proj_b : nullptr, L + a.o_x0,
                      L + a.o_x1, d, M, d, d, 1.0f, st));
        GENIE_TRY(lin(L + a.o_x1, d, lw.temporal.qkv_w, c.qkv_bias ? lw.temporal.qkv_b : nullptr, nullptr, L + a.o_qkvt,
                      3 * d, M, 3 * d, d, 1.0f, st));
        GENIE_TRY(temporal_attn_fwd(c, lw.temporal, L + a.o_qkvt, L + a.o_aot, B, st));
        GENIE_TRY(lin(L + a.o_aot, d, lw.temporal.proj_w, c.proj_bias ? lw.temporal.proj_b : nullptr, L + a.o_x1,
                      L + a.o_x2, d, M, d, d, 1.0f, st));
        if (!c.qk_norm) GENIE_TRY(launch_layer_norm(L + a.o_x2, lw.norm2_w, lw.norm2_b, L + a.o_u2, M, d, 1e-5f, st));
        GENIE_TRY(lin(u2, d, lw.fc1_w, c.mlp_bias ? lw.fc1_b : nullptr, nullptr, L + a.o_z, hid, M, hid, d, 1.0f,
                      st));
        GENIE_TRY(launch_gelu_fwd(L + a.o_z, L + a.o_h, (size_t)M * hid, st));
        GENIE_TRY(lin(L + a.o_h, hid, lw.fc2_w, c.mlp_bias ? lw.fc2_b : nullptr, L + a.o_x2, xnext, d, M, d, hid, 1.0f,
                      st));
    }
    const int V = c.factored_vocab * c.num_factored;
    GENIE_TRY(lin(acts + a.o_xL, d, wt->out_w, wt->out_b, nullptr, acts + a.o_logits, V, M, V, d, c.readout_mult, st));
    if (hipMemsetAsync(sums, 0, 3 * sizeof(double), st) != hipSuccess) {
        set_error("genie_train_forward: hipMemsetAsync failed");
        return GENIE_E_LAUNCH;
    }
    return launch_ce_fwd_bwd(c, acts + a.o_logits, input_ids, labels, B, sums, st);
}

int genie_train_backward_head(const genie_cfg* cfg, const genie_weights* wt, const genie_weights* wT,
                              const genie_weights* grads, int B, float* acts, void* workspace, size_t workspace_bytes,
                              int accumulate, void* stream) {
    GENIE_TRY(train_check(cfg, B));
    GENIE_CHECK_ARG(wt && grads && acts && workspace, "genie_train_backward_head: NULL argument");
    const genie_cfg& c = *cfg;
    TrainWs w = train_ws(c, B, workspace);
    hipStream_t st = (hipStream_t)stream;
    if (c.precision != GENIE_PREC_EXACT) {
        TrainWs16 h = train_ws16(c, B, npl_of(c), workspace, w.total);
        GENIE_CHECK_ARG(workspace_bytes >= h.total, "training workspace too small: %zu < %zu", workspace_bytes, h.total);
        return train_backward_head16(c, wt, wT, grads, B, (char*)acts, w, h, accumulate ? 1.0f : 0.0f, st);
    }
    const TrainActs a = train_acts(c, B);
    GENIE_CHECK_ARG(workspace_bytes >= w.total, "training workspace too small: %zu < %zu", workspace_bytes, w.total);
    const int d = c.d_model, M = B * c.T * c.S, V = c.factored_vocab * c.num_factored;
    const float beta = accumulate ? 1.0f : 0.0f;
    const float* dl = acts + a.o_logits;
    GENIE_TRY(launch_wgrad_f32(dl, V, acts + a.o_xL, d, (float*)grads->out_w, M, V, d, c.readout_mult, beta, w.slabs,
                               w.slab_floats, st));
    GENIE_TRY(launch_colsum(dl, V, M, V, (float*)grads->out_b, beta, w.colpart, st));
    return dgrad(dl, wt->out_w, nullptr, w.dx, M, V, d, c.readout_mult, st);
}

int genie_train_backward_layer(const genie_cfg* cfg, const genie_weights* wt, const genie_weights* wT,
                               const genie_weights* grads, int layer, int B, float* acts, void* workspace,
                               size_t workspace_bytes, int accumulate, void* stream) {
    GENIE_TRY(train_check(cfg, B));
    GENIE_CHECK_ARG(wt && grads && acts && workspace, "genie_train_backward_layer: NULL argument");
    GENIE_CHECK_ARG(layer >= 0 && layer < cfg->num_layers, "genie_train_backward_layer: layer %d out of range", layer);
    const genie_cfg& c = *cfg;
    TrainWs w = train_ws(c, B, workspace);
    hipStream_t st = (hipStream_t)stream;
    if (c.precision != GENIE_PREC_EXACT) {
        GENIE_TRY(need16(wT, c, "genie_train_backward_layer (transposed copies)"));
        TrainWs16 h = train_ws16(c, B, npl_of(c), workspace, w.total);
        GENIE_CHECK_ARG(workspace_bytes >= h.total, "training workspace too small: %zu < %zu", workspace_bytes, h.total);
        return train_backward_layer16(c, wt, wT, grads, layer, B, (char*)acts, w, h, accumulate ? 1.0f : 0.0f, st);
    }
    const TrainActs a = train_acts(c, B);
    GENIE_CHECK_ARG(workspace_bytes >= w.total, "training workspace too small: %zu < %zu", workspace_bytes, w.total);
    const int d = c.d_model, hid = c.hidden, M = B * c.T * c.S;
    const float beta = accumulate ? 1.0f : 0.0f;
    const genie_layer_weights& lw = wt->layers_host[layer];
    const genie_layer_weights& g = grads->layers_host[layer];
    const float* L = acts + a.per_layer * layer;
    float* dx = w.dx;
    const float* u1 = c.qk_norm ? L + a.o_x0 : L + a.o_u1;
    const float* u2 = c.qk_norm ? L + a.o_x2 : L + a.o_u2;
    QkSrc qk;

    // ---- MLP: x3 = x2 + fc2(gelu(fc1(norm2(x2))))  (st_transformer.py:81, 16-25)
    GENIE_TRY(launch_wgrad_f32(dx, d, L + a.o_h, hid, (float*)g.fc2_w, M, d, hid, 1.0f, beta, w.slabs, w.slab_floats, st));
    if (c.mlp_bias) GENIE_TRY(launch_colsum(dx, d, M, d, (float*)g.fc2_b, beta, w.colpart, st));
    GENIE_TRY(dgrad(dx, lw.fc2_w, nullptr, w.g, M, d, hid, 1.0f, st));          // dh
    GENIE_TRY(launch_gelu_bwd(L + a.o_z, w.g, (size_t)M * hid, st));            // dz
    GENIE_TRY(launch_wgrad_f32(w.g, hid, u2, d, (float*)g.fc1_w, M, hid, d, 1.0f, beta, w.slabs, w.slab_floats, st));
    if (c.mlp_bias) GENIE_TRY(launch_colsum(w.g, hid, M, hid, (float*)g.fc1_b, beta, w.colpart, st));
    if (c.qk_norm) {
        GENIE_TRY(dgrad(w.g, lw.fc1_w, dx, dx, M, hid, d, 1.0f, st));           // Identity norm: dx += dz . W1
    } else {
        GENIE_TRY(dgrad(w.g, lw.fc1_w, nullptr, w.d1, M, hid, d, 1.0f, st));    // d norm2 output
        GENIE_TRY(launch_ln_bwd(L + a.o_x2, lw.norm2_w, w.d1, dx, (float*)g.norm2_w, (float*)g.norm2_b, M, d, 1e-5f, beta,
                                w.lnpart, st));
    }

    // ---- temporal: x2 = x1 + proj(attn(qkv(x1))), no pre-norm  (st_transformer.py:77-78)
    GENIE_TRY(launch_wgrad_f32(dx, d, L + a.o_aot, d, (float*)g.temporal.proj_w, M, d, d, 1.0f, beta, w.slabs,
                               w.slab_floats, st));
    if (c.proj_bias) GENIE_TRY(launch_colsum(dx, d, M, d, (float*)g.temporal.proj_b, beta, w.colpart, st));
    GENIE_TRY(dgrad(dx, lw.temporal.proj_w, nullptr, w.d1, M, d, d, 1.0f, st));  // d attention output
    GENIE_TRY(qk_source(c, lw.temporal, L + a.o_qkvt, w, B, &qk, st));
    GENIE_TRY(launch_attn_temporal_bwd(L + a.o_qkvt, qk.p, qk.ld, w.d1, w.g, B, c.T, c.S, d, c.num_heads, c.head_dim,
                                       c.attn_scale, st));
    GENIE_TRY(qk_norm_backward(c, lw.temporal, g.temporal, L + a.o_qkvt, w.g, w, B, beta, st));
    GENIE_TRY(launch_wgrad_f32(w.g, 3 * d, L + a.o_x1, d, (float*)g.temporal.qkv_w, M, 3 * d, d, 1.0f, beta, w.slabs,
                               w.slab_floats, st));
    if (c.qkv_bias) GENIE_TRY(launch_colsum(w.g, 3 * d, M, 3 * d, (float*)g.temporal.qkv_b, beta, w.colpart, st));
    GENIE_TRY(dgrad(w.g, lw.temporal.qkv_w, dx, dx, M, 3 * d, d, 1.0f, st));     // dx += dqkv . Wqkv

    // ---- spatial: x1 = x0 + proj(attn(qkv(norm1(x0))))  (st_transformer.py:73-74)
    GENIE_TRY(launch_wgrad_f32(dx, d, L + a.o_aos, d, (float*)g.spatial.proj_w, M, d, d, 1.0f, beta, w.slabs, w.slab_floats,
                               st));
    if (c.proj_bias) GENIE_TRY(launch_colsum(dx, d, M, d, (float*)g.spatial.proj_b, beta, w.colpart, st));
    GENIE_TRY(dgrad(dx, lw.spatial.proj_w, nullptr, w.d1, M, d, d, 1.0f, st));
    GENIE_TRY(qk_source(c, lw.spatial, L + a.o_qkvs, w, B, &qk, st));
    GENIE_TRY(spatial_attn_bwd(c, L + a.o_qkvs, qk, w.d1, w.g, w, B, st));
    GENIE_TRY(qk_norm_backward(c, lw.spatial, g.spatial, L + a.o_qkvs, w.g, w, B, beta, st));
    GENIE_TRY(launch_wgrad_f32(w.g, 3 * d, u1, d, (float*)g.spatial.qkv_w, M, 3 * d, d, 1.0f, beta, w.slabs,
                               w.slab_floats, st));
    if (c.qkv_bias) GENIE_TRY(launch_colsum(w.g, 3 * d, M, 3 * d, (float*)g.spatial.qkv_b, beta, w.colpart, st));
    if (c.qk_norm) return dgrad(w.g, lw.spatial.qkv_w, dx, dx, M, 3 * d, d, 1.0f, st);
    GENIE_TRY(dgrad(w.g, lw.spatial.qkv_w, nullptr, w.d1, M, 3 * d, d, 1.0f, st));  // d norm1 output
    return launch_ln_bwd(L + a.o_x0, lw.norm1_w, w.d1, dx, (float*)g.norm1_w, (float*)g.norm1_b, M, d, 1e-5f, beta, w.lnpart,
                         st);
}

int genie_train_backward_embed(const genie_cfg* cfg, const genie_weights* grads, const int64_t* input_ids, int B,
                               void* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    GENIE_TRY(train_check(cfg, B));
    GENIE_CHECK_ARG(grads && input_ids && workspace, "genie_train_backward_embed: NULL argument");
    TrainWs w = train_ws(*cfg, B, workspace);
    GENIE_CHECK_ARG(workspace_bytes >= w.total, "training workspace too small: %zu < %zu", workspace_bytes, w.total);
    float* tables[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < cfg->num_factored && j < 4; ++j) tables[j] = (float*)grads->embed[j];
    return launch_embed_bwd(*cfg, w.dx, input_ids, B, (float*)grads->pos_embed, (float*)grads->mask_embed, tables,
                            accumulate ? 1.0f : 0.0f, w.colpart, (hipStream_t)stream);
}

int genie_sumsq(const float* x, size_t n, double* out, double* scratch, void* stream) {
    GENIE_CHECK_ARG((x || !n) && out && scratch, "genie_sumsq: NULL argument");
    return launch_sumsq(x, n, out, scratch, (hipStream_t)stream);
}

int genie_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_mult, const double* grad_sumsq,
                     float max_grad_norm, void* stream) {
    GENIE_CHECK_ARG((params && grads && exp_avg && exp_avg_sq) || !n, "genie_adamw_step: NULL argument");
    GENIE_CHECK_ARG(step >= 1, "genie_adamw_step: step counts from 1");
    return launch_adamw(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_mult,
                        grad_sumsq, max_grad_norm, (hipStream_t)stream);
}

}  // extern "C"
