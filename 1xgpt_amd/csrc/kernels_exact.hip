// Exact-precision (f32) kernels of the GENIE path for gfx950: embedding gather, LayerNorm, the f32-MFMA
// "NT" GEMM with fused bias / erf-GELU / residual epilogue, the generic strided attention, factored CE,
// MaskGIT sampling and mask step, layout transposes.  Every kernel cites the reference op it replaces.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

// ------------------------------------------------------------------------------------------------
// a2  FactorizedEmbedding.forward + pos_embed (factorization_utils.py:29-52, st_mask_git.py:257-261)
// one thread per float4 of the output; ids are read through L1 (d/4 threads share one id).
// ------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ pos,
                             const float* __restrict__ mask_embed, const float* __restrict__ e0,
                             const float* __restrict__ e1, const float* __restrict__ e2,
                             const float* __restrict__ e3, float* __restrict__ x, long n_tok, int TS, int d,
                             int nfac, int vf, int64_t mask_id) {
    const int d4 = d >> 2;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tok * d4) return;
    long tok = idx / d4;
    int c = (int)(idx - tok * d4) << 2;
    int64_t id = ids[tok];
    float4 v;
    if (id == mask_id) {
        v = *reinterpret_cast<const float4*>(mask_embed + c);
    } else {
        const float* tabs[4] = {e0, e1, e2, e3};
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t rem = id;
        for (int j = 0; j < nfac; ++j) {  // factor j = (id // vf^j) % vf  (factorization_utils.py:67-68)
            int f = (int)(rem % vf);
            rem /= vf;
            float4 e = *reinterpret_cast<const float4*>(tabs[j] + (size_t)f * d + c);
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
    }
    float4 p = *reinterpret_cast<const float4*>(pos + (size_t)(tok % TS) * d + c);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    *reinterpret_cast<float4*>(x + (size_t)tok * d + c) = v;
}

int launch_embed(const genie_cfg& c, const genie_weights& w, const int64_t* ids, int B, float* x, hipStream_t st) {
    long n_tok = (long)B * c.T * c.S;
    long n = n_tok * (c.d_model / 4);
    int blocks = (int)((n + 255) / 256);
    embed_kernel<<<blocks, 256, 0, st>>>(ids, w.pos_embed, w.mask_embed, w.embed[0], w.embed[1], w.embed[2],
                                         w.embed[3], x, n_tok, c.T * c.S, c.d_model, c.num_factored,
                                         c.factored_vocab, (int64_t)c.image_vocab_size);
    GENIE_LAUNCH_CHECK("embed");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a4  nn.LayerNorm(C, eps): one wavefront per row, two-pass (mean, then centred variance), f32.
// OutT = float (exact) or uint16_t bf16 (operand of a bf16 GEMM).
// ------------------------------------------------------------------------------------------------
template <typename OutT, bool SPLIT = false>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, OutT* __restrict__ y, long rows,
                                                         int C, float eps, size_t plane = 0) {
    const int lane = threadIdx.x & 63;
    long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    OutT* yr = y + (size_t)row * C;
    if ((C & 3) == 0) {  // float4 path: 16 B per lane per access, 8 B / 2x8 B packed 16-bit stores
        float4 v[8];     // C <= 2048
        const int n4 = C >> 2;
        int n = 0;
        float s = 0.f;
        for (int c = lane; c < n4; c += 64) {
            v[n] = *reinterpret_cast<const float4*>(xr + 4 * c);
            s += (v[n].x + v[n].y) + (v[n].z + v[n].w);
            ++n;
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int k = 0; k < n; ++k) {
            const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        n = 0;
        for (int c = lane; c < n4; c += 64) {
            const float4 gg = *reinterpret_cast<const float4*>(g + 4 * c);
            const float4 bb = *reinterpret_cast<const float4*>(b + 4 * c);
            float4 o;
            o.x = (v[n].x - mean) * rstd * gg.x + bb.x;
            o.y = (v[n].y - mean) * rstd * gg.y + bb.y;
            o.z = (v[n].z - mean) * rstd * gg.z + bb.z;
            o.w = (v[n].w - mean) * rstd * gg.w + bb.w;
            if constexpr (SPLIT) {
                uint16_t h0, l0, h1, l1, h2, l2, h3, l3;
                split_f16(o.x, h0, l0); split_f16(o.y, h1, l1); split_f16(o.z, h2, l2); split_f16(o.w, h3, l3);
                uint2 ph, pl;
                ph.x = (uint32_t)h0 | ((uint32_t)h1 << 16); ph.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                pl.x = (uint32_t)l0 | ((uint32_t)l1 << 16); pl.y = (uint32_t)l2 | ((uint32_t)l3 << 16);
                *reinterpret_cast<uint2*>(yr + 4 * c) = ph;
                *reinterpret_cast<uint2*>(yr + plane + 4 * c) = pl;
            } else if constexpr (sizeof(OutT) == 2) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                *reinterpret_cast<uint2*>(yr + 4 * c) = pk;
            } else {
                *reinterpret_cast<float4*>(yr + 4 * c) = o;
            }
            ++n;
        }
        return;
    }
    float v[32];  // C <= 2048
    int n = 0;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { v[n] = xr[c]; s += v[n]; ++n; }
    float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int k = 0; k < n; ++k) { float t = v[k] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    n = 0;
    for (int c = lane; c < C; c += 64) {
        float o = (v[n] - mean) * rstd * g[c] + b[c];
        if constexpr (SPLIT) { uint16_t hi, lo; split_f16(o, hi, lo); yr[c] = hi; yr[plane + c] = lo; }
        else if constexpr (sizeof(OutT) == 2) yr[c] = f32_to_bf16(o);
        else yr[c] = o;
        ++n;
    }
}

// The same LayerNorm for C = 256 / 512 (the shipped widths), tuned for bandwidth: a lane owns VPL = C/64 CONTIGUOUS
// channels (32-byte loads, one 16-byte store per 16-bit plane), a wave walks 4 consecutive rows with the next row's
// loads in flight while the current one is reduced, and gamma / beta are read once per wave.
// RPW rows per wave: 4 for the chip-filling launches, 1 when the whole problem is a few thousand rows (one-frame passes: four
// times the waves, a quarter of the serial reduce chain per wave).
template <typename OutT, bool SPLIT, int VPL, int RPW = 4>
__global__ __launch_bounds__(256) void layer_norm_fast_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              const float* __restrict__ b, OutT* __restrict__ y, long rows,
                                                              float eps, size_t plane) {
    constexpr int C = 64 * VPL;
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float gv[VPL], bv[VPL], cur[VPL], nxt[VPL];
    auto load = [&](const float* p, float (&v)[VPL]) {
#pragma unroll
        for (int k = 0; k < VPL; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + lane * VPL + k);
            v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
        }
    };
    load(x + (size_t)row0 * C, cur);
    load(g, gv);
    load(b, bv);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        if (row >= rows) break;
        if (r + 1 < RPW && row + 1 < rows) load(x + (size_t)(row + 1) * C, nxt);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) s += cur[k];
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) { cur[k] -= mean; q += cur[k] * cur[k]; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float o[VPL];
#pragma unroll
        for (int k = 0; k < VPL; ++k) o[k] = cur[k] * rstd * gv[k] + bv[k];
        OutT* yr = y + (size_t)row * C + lane * VPL;
        if constexpr (SPLIT) {
            uint32_t ph[VPL / 2], pl[VPL / 2];
#pragma unroll
            for (int k = 0; k < VPL; k += 2) {
                uint16_t h0, l0, h1, l1;
                split_f16(o[k], h0, l0);
                split_f16(o[k + 1], h1, l1);
                ph[k / 2] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                pl[k / 2] = (uint32_t)l0 | ((uint32_t)l1 << 16);
            }
            if constexpr (VPL == 8) {
                *reinterpret_cast<uint4*>(yr) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                *reinterpret_cast<uint4*>(yr + plane) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            } else {
                *reinterpret_cast<uint2*>(yr) = make_uint2(ph[0], ph[1]);
                *reinterpret_cast<uint2*>(yr + plane) = make_uint2(pl[0], pl[1]);
            }
        } else if constexpr (sizeof(OutT) == 2) {
            uint32_t pk[VPL / 2];
#pragma unroll
            for (int k = 0; k < VPL; k += 2) pk[k / 2] = (uint32_t)f32_to_bf16(o[k]) | ((uint32_t)f32_to_bf16(o[k + 1]) << 16);
            if constexpr (VPL == 8) *reinterpret_cast<uint4*>(yr) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            else *reinterpret_cast<uint2*>(yr) = make_uint2(pk[0], pk[1]);
        } else {
#pragma unroll
            for (int k = 0; k < VPL; k += 4)
                *reinterpret_cast<float4*>(yr + k) = make_float4(o[k], o[k + 1], o[k + 2], o[k + 3]);
        }
        if (r + 1 < RPW) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) cur[k] = nxt[k];
        }
    }
}
template <typename OutT, bool SPLIT>
static bool launch_layer_norm_fast(const float* x, const float* g, const float* b, OutT* y, long rows, int C, float eps,
                                   size_t plane, hipStream_t st) {
    if (C != 512 && C != 256) return false;
    if (rows <= 16384) {
        const unsigned blocks1 = (unsigned)((rows + 3) / 4);
        if (C == 512) layer_norm_fast_kernel<OutT, SPLIT, 8, 1><<<blocks1, 256, 0, st>>>(x, g, b, y, rows, eps, plane);
        else layer_norm_fast_kernel<OutT, SPLIT, 4, 1><<<blocks1, 256, 0, st>>>(x, g, b, y, rows, eps, plane);
        return true;
    }
    const unsigned blocks = (unsigned)((rows + 15) / 16);
    if (C == 512) layer_norm_fast_kernel<OutT, SPLIT, 8><<<blocks, 256, 0, st>>>(x, g, b, y, rows, eps, plane);
    else layer_norm_fast_kernel<OutT, SPLIT, 4><<<blocks, 256, 0, st>>>(x, g, b, y, rows, eps, plane);
    return true;
}

int launch_layer_norm(const float* x, const float* g, const float* b, float* y, long rows, int C, float eps,
                      hipStream_t st) {
    GENIE_CHECK_SHAPE(C <= 2048, "layer_norm: C=%d > 2048", C);
    int blocks = (int)((rows + 3) / 4);
    ProfScope prof(GENIE_KC_LAYERNORM, 8.0 * rows * C, 8.0 * rows * C, st);
    if (!launch_layer_norm_fast<float, false>(x, g, b, y, rows, C, eps, 0, st))
        layer_norm_kernel<float><<<blocks, 256, 0, st>>>(x, g, b, y, rows, C, eps);
    GENIE_LAUNCH_CHECK("layer_norm");
    return GENIE_OK;
}
int launch_layer_norm_bf16(const float* x, const float* g, const float* b, uint16_t* y, long rows, int C, float eps,
                           hipStream_t st) {
    GENIE_CHECK_SHAPE(C <= 2048, "layer_norm: C=%d > 2048", C);
    int blocks = (int)((rows + 3) / 4);
    ProfScope prof(GENIE_KC_LAYERNORM, 8.0 * rows * C, 6.0 * rows * C, st);
    if (!launch_layer_norm_fast<uint16_t, false>(x, g, b, y, rows, C, eps, 0, st))
        layer_norm_kernel<uint16_t><<<blocks, 256, 0, st>>>(x, g, b, y, rows, C, eps);
    GENIE_LAUNCH_CHECK("layer_norm_bf16");
    return GENIE_OK;
}

int launch_layer_norm_split(const float* x, const float* g, const float* b, uint16_t* y, size_t plane, long rows, int C,
                            float eps, hipStream_t st) {
    GENIE_CHECK_SHAPE(C <= 2048, "layer_norm: C=%d > 2048", C);
    int blocks = (int)((rows + 3) / 4);
    ProfScope prof(GENIE_KC_LAYERNORM, 8.0 * rows * C, 8.0 * rows * C, st);
    if (!launch_layer_norm_fast<uint16_t, true>(x, g, b, y, rows, C, eps, plane, st))
        layer_norm_kernel<uint16_t, true><<<blocks, 256, 0, st>>>(x, g, b, y, rows, C, eps, plane);
    GENIE_LAUNCH_CHECK("layer_norm_split");
    return GENIE_OK;
}

__global__ void split_f16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t plane, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint16_t hi, lo; split_f16(src[i], hi, lo); dst[i] = hi; dst[plane + i] = lo; }
}
int launch_split_f16(const float* src, uint16_t* dst, size_t plane, size_t n, hipStream_t st) {
    if (!n) return GENIE_OK;
    split_f16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, plane, n);
    GENIE_LAUNCH_CHECK("split_f16");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// nn.Linear on the f32 matrix cores:  C[M,N] (+)= epilogue( alpha * A[M,K] . W[N,K]^T + bias )
//
// v_mfma_f32_32x32x2_f32 is an exact-f32 fmaf chain at 157 TF peak (1/16 of bf16 MFMA), so this GEMM
// is issue-bound on the matrix pipe, not on LDS/HBM: 128x128 block tile, 4 waves (2x2), each wave
// 64x64 = 2x2 MFMA tiles (64 accumulator registers), BK = 16, LDS double-buffered, global->register
// prefetch of the next K-tile while the current one is multiplied.
//
// Operand fetch trick: the MFMA wants lane (r = lane&31, h = lane>>5) to supply A[r][k0+h].  Since the
// K-sum is order-free, lane (r,h) reads ONE float4 = A[r][8*kk + 4h .. +3] and MFMA j (0..3) consumes
// component j from both halves, i.e. the k-pair {8kk+j, 8kk+4+j}: 1 ds_read_b128 feeds 4 MFMAs.
// LDS rows are padded to BK+4 floats (80 B): the 16-lane groups of ds_read_b128 then hit 16 distinct
// 16-byte slots -> conflict-free.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16, GEMM_LDS_LD = GEMM_BK + 4;

__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ A, long lda, long strideA,
                                                          const float* __restrict__ W, long ldw, long strideW,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          long ldc, long strideC, int M, int N, int K, int flags,
                                                          float alpha) {
    __shared__ __attribute__((aligned(16))) float sA[2][GEMM_BM * GEMM_LDS_LD];
    __shared__ __attribute__((aligned(16))) float sB[2][GEMM_BN * GEMM_LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    // N-tiles vary fastest so that blocks sharing an A row-panel are dispatched together (L2 reuse).
    const int n_tiles = (N + GEMM_BN - 1) / GEMM_BN;
    const int m0 = (blockIdx.x / n_tiles) * GEMM_BM, n0 = (blockIdx.x % n_tiles) * GEMM_BN;
    A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;
    C += (size_t)blockIdx.y * strideC;

    // global->LDS staging: thread t moves float4 (row = t/4 [+64], col4 = t%4) of each operand tile
    const int lrow = tid >> 2, lcol = (tid & 3) << 2;
    const float* gA[2];
    const float* gB[2];
    bool vA[2], vB[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int ra = m0 + lrow + p * 64, rb = n0 + lrow + p * 64;
        vA[p] = ra < M;
        vB[p] = rb < N;
        gA[p] = A + (size_t)(vA[p] ? ra : 0) * lda + lcol;
        gB[p] = W + (size_t)(vB[p] ? rb : 0) * ldw + lcol;
    }
    float4 ra4[2], rb4[2];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ra4[p] = vA[p] ? *reinterpret_cast<const float4*>(gA[p] + k0) : z4;
            rb4[p] = vB[p] ? *reinterpret_cast<const float4*>(gB[p] + k0) : z4;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<float4*>(&sA[buf][(lrow + p * 64) * GEMM_LDS_LD + lcol]) = ra4[p];
            *reinterpret_cast<float4*>(&sB[buf][(lrow + p * 64) * GEMM_LDS_LD + lcol]) = rb4[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / GEMM_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * GEMM_BK);
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 8; ++kk) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&sA[buf][(wm * 64 + i * 32 + r) * GEMM_LDS_LD + kk * 8 + 4 * h]);
                b[i] = *reinterpret_cast<const float4*>(&sB[buf][(wn * 64 + i * 32 + r) * GEMM_LDS_LD + kk * 8 + 4 * h]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    const bool do_gelu = flags & GEMM_GELU, do_acc = flags & GEMM_ACCUM, bias_m = flags & GEMM_BIAS_ALONG_M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
            if (col >= N) continue;
            const float bcol = (bias && !bias_m) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = acc[i][j][e] * alpha + ((bias && bias_m) ? bias[row] : bcol);
                if (do_gelu) v = gelu_erf(v);
                float* p = C + (size_t)row * ldc + col;
                if (do_acc) v += *p;
                *p = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// The same nn.Linear, LDS-DMA fed (round 3).  gemm_f32_nt_kernel above stages its operands through registers, and the compiler
// keeps those registers in scratch (runtime-indexed arrays) behind an s_waitcnt vmcnt(0) at the top of every K-step: the global
// load latency of every K-tile is exposed and the matrix pipe is 54 % busy (PMC, 75-90 TFLOP/s of the 157.3 TFLOP/s f32 MFMA
// peak).  Here every operand byte goes HBM/L2 -> LDS by buffer_load ... lds (no registers, no ds_write), one K-tile ahead of the
// multiplies:
//   * 128x128 tile, 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles (64 accumulator registers); BK = 32 floats
//     = 128-byte rows, two stages of 32 KB -> two workgroups per CU (one computes while the other sits at its barrier);
//   * the 16-byte slot of a row is XOR-swizzled with (row / 2) % 8 on the SOURCE address and on the fragment read (the LDS image of
//     a DMA piece is lane-linear): conflict-free ds_read_b128;
//   * one barrier per K-tile of 64 matrix instructions (4,096 cycles) per wave: wait for tile t (issued a whole K-tile ago) ->
//     barrier (also: everyone is done reading the other buffer) -> issue tile t+1 into it -> multiply tile t.
// Rows beyond M / N are clamped on the load side (the epilogue never stores them).  Same fmaf-chain arithmetic; the k-order of the
// sum differs from the kernel above (BK = 32, pairs {8kk+j, 8kk+4+j}), i.e. results agree to f32 rounding, not bitwise.
// ------------------------------------------------------------------------------------------------
typedef float gd_f4 __attribute__((ext_vector_type(4)));

// BK = 32: 128-byte rows, 64 KB of LDS per workgroup (2 per CU);  BK = 16: 64-byte rows, 32 KB (4 per CU: more waves to cover
// each other's barrier waits, finer rounds)
template <int BK, bool GELU, bool ACC, bool BIASM>
__global__ __launch_bounds__(256, BK == 32 ? 2 : 4) void gemm_f32_dma_kernel(const float* __restrict__ A, long lda, long strideA,
                                                               const float* __restrict__ W, long ldw, long strideW,
                                                               const float* __restrict__ bias, float* __restrict__ C,
                                                               long ldc, long strideC, int M, int N, int K, int flags,
                                                               float alpha) {
    constexpr int ROWB = BK * 4, SLOTS = ROWB / 16, RPB = 256 / ROWB, RPP = 1024 / ROWB, NP = 128 / RPP / 4, KK = BK / 8;
    constexpr int OPB = 128 * ROWB, GD_STAGE = 2 * OPB;   // bytes of one operand tile / of one stage (A tile then B tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char gd_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order (workgroups are dealt to the 8 XCDs round-robin): the 8 row tiles of a group go to the 8 XCDs and each
    // XCD walks its row tile's column tiles one after the other, so an A panel is fetched into ONE L2, not into N / 128 of them
    const int n_tiles = (N + 127) / 128, m_tiles = (M + 127) / 128;
    int m_tile, n_tile;
    {
        const int bid = blockIdx.x, full = (m_tiles / 8) * 8 * n_tiles;
        if (bid < full) {
            const int grp = bid / (8 * n_tiles), rem = bid - grp * 8 * n_tiles;
            m_tile = grp * 8 + (rem & 7);
            n_tile = rem >> 3;
        } else {
            const int rem = bid - full;
            m_tile = (m_tiles / 8) * 8 + rem / n_tiles;
            n_tile = rem % n_tiles;
        }
    }
    const int m0 = m_tile * 128, n0 = n_tile * 128;
    A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;
    C += (size_t)blockIdx.y * strideC;

    // ---- LDS-DMA: a piece = 8 rows x 128 bytes (one wave instruction); wave w moves pieces w, w+4, ... of each operand tile
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, -1, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, -1, 0x00020000);
    unsigned voA[4], voB[4];   // (NP <= 4 entries used)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (wid + 4 * j) * RPP + lane / SLOTS;              // row of the tile this lane fills
        const int ls = (lane % SLOTS) ^ ((row / RPB) % SLOTS);           // logical 16-byte slot landing in physical slot lane % SLOTS
        const int ra = m0 + row < M ? row : M - 1 - m0, rb = n0 + row < N ? row : N - 1 - n0;
        voA[j] = (unsigned)((size_t)ra * lda * 4 + ls * 16);
        voB[j] = (unsigned)((size_t)rb * ldw * 4 + ls * 16);
    }
    // (buffer indices are compile-time constants everywhere below: with a run-time buffer the compiler cannot tell the LDS
    // destination of the DMA in flight from the fragment reads and puts an s_waitcnt vmcnt(0) in front of them)
    auto stage = [&](auto bufc, int kt) {
        constexpr int BUF = decltype(bufc)::value;
        unsigned char* base = gd_smem + BUF * GD_STAGE;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + (wid + 4 * j) * 1024), 16,
                                                     voA[j], kt * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + OPB + (wid + 4 * j) * 1024),
                                                     16, voB[j], kt * ROWB, 0, 0);
        }
    };
    // fragment read offsets: lane (r, h) reads slot 2 kk + h of its row, swizzled; rows wm*64 + i*32 + r / wn*64 + j*32 + r
    unsigned offA[4], offB[4];   // (KK <= 4 entries used)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const unsigned o = (unsigned)(r * ROWB + (((2 * kk + h) ^ ((r / RPB) % SLOTS)) << 4));
        offA[kk] = o + wm * 64 * ROWB;
        offB[kk] = o + OPB + wn * 64 * ROWB;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    const int nk = K / BK;
    auto ktile = [&](auto bufc, auto nextc, int kt) {
        constexpr int BUF = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt (the only loads in flight) have landed
        __builtin_amdgcn_s_barrier();                        // ... everyone's have, and everyone is done with the other buffer
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) stage(nextc, kt + 1);
        const unsigned char* base = gd_smem + BUF * GD_STAGE;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            // (ext_vector loads, NOT HIP's float4 struct: with the struct's TBAA the compiler's waitcnt pass puts an
            // s_waitcnt vmcnt(0) in front of the first ds_read after every DMA issue -- the whole pipeline drained per K-tile)
            gd_f4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const gd_f4*>(base + offA[kk] + i * 32 * ROWB);
                b[i] = *reinterpret_cast<const gd_f4*>(base + offB[kk] + i * 32 * ROWB);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    stage(B0{}, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        ktile(B0{}, B1{}, kt);
        ktile(B1{}, B0{}, kt + 1);
    }
    if (kt < nk) ktile(B0{}, B1{}, kt);

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5): a store instruction covers
    // two rows x 128 contiguous bytes.  Per 32x32 tile, three straight sections: every load (the 16 residual values -- C may be the
    // residual buffer itself -- and, for a bias along M, the 16 row biases), then the arithmetic, then the 16 stores back to back.
    // Written element by element (load, add, store) the compiler drains vmcnt before every element, and vmcnt counts the previous
    // element's STORE as well: 64 serial store round trips per wave.  The 16 rows of a tile are distinct, so reading first is safe
    // in place.
    // Tile t + 1's residual values are requested BEFORE tile t's stores (vmcnt retires in order: a load issued behind a store
    // waits for the store's round trip); whole tiles inside the matrix (all but the last row / column of workgroups) take a path
    // without per-element bounds masks, which the compiler schedules as straight-line code with counted waits.
    // GELU, the residual accumulate and the bias orientation are compile-time (the erf code of 16 elements beside 32 live residual
    // values spilled; a run-time bias branch joining the straight-line path made the compiler drain vmcnt in every tile).
    constexpr bool do_gelu = GELU, do_acc = ACC;
    constexpr bool bias_m = BIASM;   // (bias along M: the operand-swapped readout; never with GELU / accumulate)
    // Addresses: one buffer descriptor at the tile's origin, one 32-bit lane offset, the row of an element as a SCALAR offset
    // (uniform: (32 i + ro) * ldc) -- sixteen 64-bit lane addresses per tile were what spilled under the 128-register budget.
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (size_t)m0 * ldc + n0), 0, -1, 0x00020000);
    const int lane_off = (int)(((size_t)(wm * 64 + 4 * h) * ldc + wn * 64 + r) * 4);
    const int ldc4 = (int)(ldc * 4);
    auto epilogue = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        auto rbase_of = [&](int t) { return m0 + wm * 64 + (t >> 1) * 32 + 4 * h; };
        auto col_of = [&](int t) { return n0 + wn * 64 + (t & 1) * 32 + r; };
        auto soff = [&](int t, int ro) { return ((t >> 1) * 32 + ro) * ldc4 + (t & 1) * 128; };
        auto load_res = [&](int t, float* dst) {
            const int rbase = rbase_of(t), col = col_of(t);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                dst[e] = (FULL || (col < N && rbase + ro < M))
                             ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, lane_off, soff(t, ro), 0))
                             : 0.f;
            }
        };
        float bcols[2];   // the lane's two column biases, ahead of everything: a load issued later sits behind the stores
#pragma unroll
        for (int j = 0; j < 2; ++j) bcols[j] = (bias && !bias_m && (FULL || col_of(j) < N)) ? bias[col_of(j)] : 0.f;
        float res[2][16];
        if (do_acc) load_res(0, res[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rbase = rbase_of(t), col = col_of(t);
            if (do_acc && FULL && t + 1 < 4) load_res(t + 1, res[(t + 1) & 1]);
            if (do_acc && !FULL && t > 0) load_res(t, res[t & 1]);   // (edge tiles: no look-ahead, fewer live registers)
            float add[16], v[16];
            const float bcol = bcols[t & 1];
            if (bias_m) {   // (the operand-swapped readout only)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = (e & 3) + 8 * (e >> 2);
                    add[e] = (FULL || rbase + ro < M) ? bias[rbase + ro] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) add[e] = bcol;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[t >> 1][t & 1][e] * alpha + add[e];
            if (do_gelu) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = gelu_erf(v[e]);
            }
            if (do_acc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += res[t & 1][e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (FULL || (col < N && rbase + ro < M))
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rsC, lane_off, soff(t, ro), 0);
            }
        }
    };
    if (m0 + 128 <= M && n0 + 128 <= N) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

int launch_gemm_f32(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, const float* bias,
                    float* C, long ldc, long strideC, int M, int N, int K, int batch, int flags, float alpha,
                    hipStream_t st) {
    GENIE_CHECK_SHAPE(K % GEMM_BK == 0 && K > 0, "gemm: K=%d must be a positive multiple of %d", K, GEMM_BK);
    GENIE_CHECK_SHAPE((lda % 4 == 0) && (ldw % 4 == 0), "gemm: leading dims must be multiples of 4 floats");
    if (M <= 0 || N <= 0 || batch <= 0) return GENIE_OK;
    int mt = (M + GEMM_BM - 1) / GEMM_BM, nt = (N + GEMM_BN - 1) / GEMM_BN;
    dim3 grid(mt * nt, batch);
    const double mnk = (double)M * N * batch;
    // the LDS-DMA form needs 128-byte K-tiles, 16-byte aligned rows and 32-bit byte offsets inside a tile's descriptor
    // K-tile of the LDS-DMA kernel: 16 floats (4 workgroups per CU; 113-131 TFLOP/s at K = 512 against 106-120 for 32) unless the
    // contraction is long (K >= 2048: 130 vs 124); study build: GENIE_GEMM_F32_DMA = 16 / 32 forces one, 0 = the old kernel
    static const int dma_env = study_env("GENIE_GEMM_F32_DMA", -1);
    const int dma = dma_env >= 0 ? dma_env : (K >= 2048 && K % 32 == 0 ? 32 : 16);
    const bool bias_along_m = (flags & GEMM_BIAS_ALONG_M) && bias;
    // (GELU with accumulate, or either with a bias along M: no such caller; the register-staged kernel takes them)
    const bool epi_ok = !(bias_along_m && (flags & (GEMM_GELU | GEMM_ACCUM))) && !((flags & GEMM_GELU) && (flags & GEMM_ACCUM));
    if (dma && epi_ok && K % dma == 0 && 128.0 * 4.0 * (lda > ldw ? lda : ldw) + 4.0 * K < 4.0e9 && 129.0 * 4.0 * ldc < 2.0e9 &&
        (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (strideA % 4 == 0) && (strideW % 4 == 0)) {
        ProfScope prof(GENIE_KC_GEMM, 2.0 * mnk * K,
                       4.0 * ((double)M * K * batch + (double)N * K + mnk * ((flags & GEMM_ACCUM) ? 2 : 1)), st,
                       "gemm_f32_dma_kernel (128x128 tile, LDS-DMA, v_mfma_f32_32x32x2_f32)");
        auto go = [&](auto bkc, auto geluc, auto accc, auto bmc) {
            constexpr int BKc = decltype(bkc)::value;
            constexpr bool G = decltype(geluc)::value, AC = decltype(accc)::value, BM = decltype(bmc)::value;
            constexpr int lds = 2 * 2 * 128 * BKc * 4;
            auto kern = gemm_f32_dma_kernel<BKc, G, AC, BM>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            kern<<<grid, 256, lds, st>>>(A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, flags, alpha);
        };
        auto go_bk = [&](auto bkc) {
            const bool g = flags & GEMM_GELU, ac = flags & GEMM_ACCUM;
            std::false_type F; std::true_type T;
            if (bias_along_m) go(bkc, F, F, T);
            else if (g) go(bkc, T, F, F);
            else if (ac) go(bkc, F, T, F);
            else go(bkc, F, F, F);
        };
        if (dma == 32) go_bk(std::integral_constant<int, 32>{});
        else go_bk(std::integral_constant<int, 16>{});
        GENIE_LAUNCH_CHECK("gemm_f32_dma");
        return GENIE_OK;
    }
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mnk * K,
                   4.0 * ((double)M * K * batch + (double)N * K + mnk * ((flags & GEMM_ACCUM) ? 2 : 1)), st,
                   "gemm_f32_nt_kernel (128x128x16 tile, v_mfma_f32_32x32x2_f32)");
    gemm_f32_nt_kernel<<<grid, 256, 0, st>>>(A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, flags,
                                             alpha);
    GENIE_LAUNCH_CHECK("gemm_f32");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a6-a8  Attention core on a packed qkv buffer (attention.py:38-59), generic in sequence geometry:
//   row(seq, pos) = (seq / inner) * outer_stride + (seq % inner) * inner_stride + pos * pos_stride
//   spatial : inner = 1, outer_stride = S,   pos_stride = 1, N = S, non-causal
//   temporal: inner = S, outer_stride = T*S, inner_stride = 1, pos_stride = S, N = T, causal --
//             the "(B S) T C" view of st_transformer.py:77 without the physical transpose.
// One thread per query row; K/V rows of the block's sequences staged in LDS (same-address reads
// broadcast); keys are consumed in chunks of 16 with one softmax rescale per chunk (N <= 16 is exactly
// the two-pass softmax).  qk-norm (one shared affine, eps 1e-5) and the q*scale of attention.py:42-48
// are applied on load.  InT/OutT = float or bf16 bits.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld_elem(const T* p) {
    if constexpr (sizeof(T) == 2) return bf16_to_f32(*p); else return *p;
}

template <int DH, typename InT, typename OutT>
__global__ void attn_generic_kernel(const InT* __restrict__ qkv, OutT* __restrict__ out, int N, int spb, long n_seq,
                                    int inner, long outer_stride, long inner_stride, long pos_stride, int d,
                                    float scale, int causal, const float* __restrict__ nw,
                                    const float* __restrict__ nb, int round_bf16) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                          // [spb*N][DH]
    float* sV = smem + (size_t)spb * N * DH;   // [spb*N][DH]
    const int head = blockIdx.y;
    const int t = threadIdx.x;
    const int sl = t / N, i = t - sl * N;
    const long seq = (long)blockIdx.x * spb + sl;
    const bool active = (sl < spb) && (seq < n_seq);
    float q[DH], o[DH];
    long row = 0;
    if (active) {
        row = (seq / inner) * outer_stride + (seq % inner) * inner_stride + (long)i * pos_stride;
        const InT* base = qkv + (size_t)row * 3 * d + head * DH;
        float kx[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { q[c] = ld_elem(base + c); kx[c] = ld_elem(base + d + c); }
        if (nw) {  // qk-norm in f32 (attention.py:42-47)
            float mq = 0.f, mk = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) { mq += q[c]; mk += kx[c]; }
            mq /= DH; mk /= DH;
            float vq = 0.f, vk = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) { float a = q[c] - mq, b = kx[c] - mk; vq += a * a; vk += b * b; }
            float rq = 1.0f / sqrtf(vq / DH + 1e-5f), rk = 1.0f / sqrtf(vk / DH + 1e-5f);
#pragma unroll
            for (int c = 0; c < DH; ++c) {
                q[c] = (q[c] - mq) * rq * nw[c] + nb[c];
                kx[c] = (kx[c] - mk) * rk * nw[c] + nb[c];
            }
        }
#pragma unroll
        for (int c = 0; c < DH; ++c) {
            float kc = kx[c], qc = q[c];
            if (round_bf16) {  // bf16 contract: operands of both matmuls are bf16, scale applied to the scores
                kc = bf16_to_f32(f32_to_bf16(kc));
                qc = bf16_to_f32(f32_to_bf16(qc));
            } else {
                qc *= scale;  // attention.py:48
            }
            q[c] = qc;
            sK[(size_t)t * DH + c] = kc;
            sV[(size_t)t * DH + c] = ld_elem(base + 2 * d + c);
            o[c] = 0.f;
        }
    }
    __syncthreads();
    if (!active) return;
    const float post = round_bf16 ? scale : 1.0f;
    const float* Ks = sK + (size_t)sl * N * DH;
    const float* Vs = sV + (size_t)sl * N * DH;
    const int jend = causal ? i + 1 : N;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < jend; j0 += 16) {
        float s[16];
        float cm = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = j0 + jj;
            float acc = -INFINITY;
            if (j < jend) {
                acc = 0.f;
                const float* kr = Ks + (size_t)j * DH;
#pragma unroll
                for (int c = 0; c < DH; ++c) acc = fmaf(q[c], kr[c], acc);
                acc *= post;
            }
            s[jj] = acc;
            cm = fmaxf(cm, acc);
        }
        const float mn = fmaxf(m, cm);
        const float alpha = __expf(m - mn);  // m = -inf on the first chunk -> 0
        l *= alpha;
#pragma unroll
        for (int c = 0; c < DH; ++c) o[c] *= alpha;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = j0 + jj;
            if (j < jend) {
                float p = expf(s[jj] - mn);
                l += p;
                if (round_bf16) p = bf16_to_f32(f32_to_bf16(p));
                const float* vr = Vs + (size_t)j * DH;
#pragma unroll
                for (int c = 0; c < DH; ++c) o[c] = fmaf(p, vr[c], o[c]);
            }
        }
        m = mn;
    }
    const float inv = 1.0f / l;
    OutT* op = out + (size_t)row * d + head * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        float v = o[c] * inv;
        if constexpr (sizeof(OutT) == 2) op[c] = f32_to_bf16(v); else op[c] = v;
    }
}

template <typename InT, typename OutT>
static int launch_attn_generic_t(const InT* qkv, OutT* out, int N, long n_seq, int inner, long outer_stride,
                                 long inner_stride, long pos_stride, int d, int H, int Dh, float scale, int causal,
                                 const float* nw, const float* nb, int round_bf16, hipStream_t st) {
    GENIE_CHECK_SHAPE(N >= 1 && N <= 1024, "attention: sequence length %d unsupported", N);
    int spb = N >= 64 ? 1 : 64 / N;
    int threads = ((spb * N + 63) / 64) * 64;
    size_t lds = (size_t)spb * N * Dh * 2 * sizeof(float);
    GENIE_CHECK_SHAPE(lds <= 160 * 1024, "attention: N=%d x Dh=%d does not fit LDS", N, Dh);
    dim3 grid((unsigned)((n_seq + spb - 1) / spb), H);
    // algorithmic work: QK^T and PV = 4*N*N*Dh flops per (sequence, head) (causal counted dense);
    // bytes: read q,k,v + write o once
    ProfScope prof(causal ? GENIE_KC_ATTN_TEMPORAL : GENIE_KC_ATTN_SPATIAL, 4.0 * N * N * Dh * H * (double)n_seq,
                   (double)n_seq * N * H * Dh * (3 * sizeof(InT) + sizeof(OutT)), st);
#define GENIE_ATTN_CASE(DHV)                                                                                   \
    case DHV:                                                                                                  \
        (void)hipFuncSetAttribute((const void*)attn_generic_kernel<DHV, InT, OutT>,                                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
        attn_generic_kernel<DHV, InT, OutT><<<grid, threads, lds, st>>>(qkv, out, N, spb, n_seq, inner,        \
                                                                         outer_stride, inner_stride,           \
                                                                         pos_stride, d, scale, causal, nw, nb, \
                                                                         round_bf16);                          \
        break;
    switch (Dh) {
        GENIE_ATTN_CASE(8)
        GENIE_ATTN_CASE(16)
        GENIE_ATTN_CASE(32)
        GENIE_ATTN_CASE(64)
        default:
            set_error("attention: head_dim %d unsupported (8/16/32/64)", Dh);
            return GENIE_E_SHAPE;
    }
#undef GENIE_ATTN_CASE
    GENIE_LAUNCH_CHECK("attn_generic");
    return GENIE_OK;
}

int launch_attn_generic(const float* qkv, float* out, int N, long n_seq, int inner, long outer_stride,
                        long inner_stride, long pos_stride, int d, int H, int Dh, float scale, int causal,
                        const float* nw, const float* nb, hipStream_t st) {
    return launch_attn_generic_t<float, float>(qkv, out, N, n_seq, inner, outer_stride, inner_stride, pos_stride, d,
                                               H, Dh, scale, causal, nw, nb, 0, st);
}
int launch_attn_generic_bf16(const uint16_t* qkv, uint16_t* out, int N, long n_seq, int inner, long outer_stride,
                             long inner_stride, long pos_stride, int d, int H, int Dh, float scale, int causal,
                             const float* nw, const float* nb, hipStream_t st) {
    return launch_attn_generic_t<uint16_t, uint16_t>(qkv, out, N, n_seq, inner, outer_stride, inner_stride,
                                                     pos_stride, d, H, Dh, scale, causal, nw, nb, 1, st);
}

// ------------------------------------------------------------------------------------------------
// Layout: token-major (rows, V) -> (V, rows) per batch ("B T (H W) C -> B C T H W", st_mask_git.py:264)
// ------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const float* ib = in + (size_t)blockIdx.z * rows * cols;
    float* ob = out + (size_t)blockIdx.z * rows * cols;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        int rr = r0 + k, cc = c0 + threadIdx.x;
        tile[k][threadIdx.x] = (rr < rows && cc < cols) ? ib[(size_t)rr * cols + cc] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        int cc = c0 + k, rr = r0 + threadIdx.x;
        if (rr < rows && cc < cols) ob[(size_t)cc * rows + rr] = tile[threadIdx.x][k];
    }
}

int launch_transpose(const float* in, float* out, int batch, int rows, int cols, hipStream_t st) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch), block(32, 8);
    transpose_kernel<<<grid, block, 0, st>>>(in, out, rows, cols);
    GENIE_LAUNCH_CHECK("transpose");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a12  factored cross-entropy + "all factors argmax-correct" (st_mask_git.py:231-253, eval_utils.py:72-77)
// token-major: one wavefront per token (V contiguous, 2 KB per factor, coalesced).
// BCTHW: one lane per token, 64 consecutive tokens per wave, loop over the vocab (coalesced across lanes).
// Partial sums go to 3 doubles with one atomicAdd per block.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_accumulate3(double ce, double hit, double cnt, double* sums) {
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ce += __shfl_xor(ce, o);
        hit += __shfl_xor(hit, o);
        cnt += __shfl_xor(cnt, o);
    }
    if (lane == 0) { red[0][wid] = ce; red[1][wid] = hit; red[2][wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
        if (c != 0.0) { atomicAdd(&sums[0], a); atomicAdd(&sums[1], b); atomicAdd(&sums[2], c); }
    }
}

// logits token-major (B, nt, S, V): token index n = (b*nt + tt)*S + s
__global__ __launch_bounds__(256) void ce_token_major_kernel(const float* __restrict__ logits,
                                                             const int64_t* __restrict__ targets,
                                                             const int64_t* __restrict__ weight_ids, long n_tok,
                                                             int nt, int S, int T, int t0, int vf, int nfac,
                                                             int64_t mask_id, double* sums) {
    const int lane = threadIdx.x & 63;
    double ce = 0, hit = 0, cnt = 0;
    // grid-stride over tokens: the three f64 atomics per block all land on the same addresses (~6 ns each at the L2), so one
    // block per four tokens made the atomics, not the 4 KB of logits per token, the kernel's time (1.2 ms for 246 k tokens)
    for (long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6); n < n_tok; n += (long)gridDim.x * 4) {
        long b = n / ((long)nt * S);
        long rem = n - b * (long)nt * S;
        long gi = (b * T + t0) * (long)S + rem;  // index into the full (B,T,S) clip
        bool counted = weight_ids ? (weight_ids[gi] == mask_id) : true;
        if (counted) {
            int64_t tgt = targets[gi];
            const float* lp = logits + (size_t)n * vf * nfac;
            float loss = 0.f;
            bool all_ok = true;
            if (vf == 512 && nfac == 2) {
                // the shipped vocabulary (2 x 512): a lane owns 8 consecutive logits of each factor -- all four 16-byte loads of the
                // token are in flight before the first use (the strided dword loop below waited on a memory round trip per 64 logits:
                // 1.2 ms for 1 GB of logits, 0.45 TB/s).  Same arithmetic: first-max-wins argmax, sum of expf(v - max), logf.
                typedef float ce4 __attribute__((ext_vector_type(4)));
                ce4 v[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int q = 0; q < 2; ++q) v[f][q] = *reinterpret_cast<const ce4*>(lp + f * 512 + lane * 8 + q * 4);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int tf = (int)(tgt % 512);
                    tgt /= 512;
                    float mx = -INFINITY;
                    int mi = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float x = v[f][q >> 2][q & 3];
                        if (x > mx) { mx = x; mi = lane * 8 + q; }
                    }
                    wave_argmax(mx, mi);
                    float se = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) se += expf(v[f][q >> 2][q & 3] - mx);
                    se = wave_sum(se);
                    // the target's logit sits in lane tf / 8, slot tf % 8
                    float tv = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) tv = (tf & 7) == q ? v[f][q >> 2][q & 3] : tv;
                    tv = __shfl(tv, tf >> 3);
                    loss += (logf(se) + mx) - tv;
                    all_ok = all_ok && (mi == tf);
                }
            } else
            for (int f = 0; f < nfac; ++f) {
                int tf = (int)(tgt % vf);
                tgt /= vf;
                float mx = -INFINITY;
                int mi = 0;
                for (int k = lane; k < vf; k += 64) {
                    float v = lp[f * vf + k];
                    if (v > mx) { mx = v; mi = k; }
                }
                wave_argmax(mx, mi);
                float se = 0.f;
                for (int k = lane; k < vf; k += 64) se += expf(lp[f * vf + k] - mx);
                se = wave_sum(se);
                loss += (logf(se) + mx) - lp[f * vf + tf];
                all_ok = all_ok && (mi == tf);
            }
            if (lane == 0) { ce += loss; hit += all_ok ? 1.0 : 0.0; cnt += 1.0; }
        }
    }
    block_accumulate3(ce, hit, cnt, sums);
}

// logits BCTHW (B, V, nt, S): element (b, v, tt, s) at ((b*V + v)*nt + tt)*S + s
__global__ __launch_bounds__(256) void ce_bcthw_kernel(const float* __restrict__ logits,
                                                       const int64_t* __restrict__ targets,
                                                       const int64_t* __restrict__ weight_ids, long n_tok, int nt,
                                                       int S, int T, int t0, int vf, int nfac, int64_t mask_id,
                                                       double* sums) {
    long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double ce = 0, hit = 0, cnt = 0;
    if (n < n_tok) {
        long b = n / ((long)nt * S);
        long rem = n - b * (long)nt * S;  // tt*S + s
        long gi = (b * T + t0) * (long)S + rem;
        bool counted = weight_ids ? (weight_ids[gi] == mask_id) : true;
        if (counted) {
            int64_t tgt = targets[gi];
            const long vstride = (long)nt * S;
            const float* lp = logits + (size_t)b * vf * nfac * vstride + rem;
            float loss = 0.f;
            bool all_ok = true;
            for (int f = 0; f < nfac; ++f) {
                int tf = (int)(tgt % vf);
                tgt /= vf;
                const float* lf = lp + (size_t)f * vf * vstride;
                float mx = -INFINITY;
                int mi = 0;
                for (int k = 0; k < vf; ++k) {
                    float v = lf[(size_t)k * vstride];
                    if (v > mx) { mx = v; mi = k; }
                }
                float se = 0.f;
                for (int k = 0; k < vf; ++k) se += expf(lf[(size_t)k * vstride] - mx);
                loss += (logf(se) + mx) - lf[(size_t)tf * vstride];
                all_ok = all_ok && (mi == tf);
            }
            ce = loss; hit = all_ok ? 1.0 : 0.0; cnt = 1.0;
        }
    }
    block_accumulate3(ce, hit, cnt, sums);
}

// sampled-token accuracy numerator of the evaluator (evaluate.py:176-179: (samples == ground truth).sum()): a (batch, n) view
// with batch stride `sa` against b (batch, n) with stride `sb`; the count is ADDED to sums6[2].  Block 0 also files the rest
// of the metric vector: sums6 = [sum CE, n CE tokens, hits, n tokens, n frames, n clips] with the CE pair taken from the
// 3-vector genie_factored_ce accumulated (ce3 = [sum CE, sum argmax-correct, n]; NULL = leave sums6[0..1] alone).
__global__ __launch_bounds__(256) void count_equal_kernel(const int64_t* __restrict__ a, long sa, const int64_t* __restrict__ b,
                                                          long sb, long n, long total, const double* __restrict__ ce3,
                                                          double* sums6, double n_tokens, double n_frames, double n_clips) {
    __shared__ int wsum[4];
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int hit = 0;
    if (i < total) {
        const long bi = i / n, r = i - bi * n;
        hit = a[bi * sa + r] == b[bi * sb + r];
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (c) atomicAdd(&sums6[2], (double)c);  // integer-valued addends: the f64 sum is exact in any order
        if (blockIdx.x == 0) {
            if (ce3) { sums6[0] = ce3[0]; sums6[1] = ce3[2]; }
            sums6[3] = n_tokens; sums6[4] = n_frames; sums6[5] = n_clips;
        }
    }
}
int launch_count_equal(const int64_t* a, long sa, const int64_t* b, long sb, int batch, long n, const double* ce3, double* sums6,
                       double n_tokens, double n_frames, double n_clips, hipStream_t st) {
    const long total = (long)batch * n;
    if (total <= 0) return GENIE_OK;
    count_equal_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, sa, b, sb, n, total, ce3, sums6, n_tokens, n_frames,
                                                                        n_clips);
    GENIE_LAUNCH_CHECK("count_equal");
    return GENIE_OK;
}

int launch_factored_ce(const genie_cfg& c, const float* logits, int layout, const int64_t* targets,
                       const int64_t* weight_ids, int B, int t0, int t1, double* sums, hipStream_t st) {
    int nt = t1 - t0;
    long n_tok = (long)B * nt * c.S;
    if (n_tok <= 0) return GENIE_OK;
    if (layout == GENIE_LAYOUT_TOKEN_MAJOR) {
        const long blocks_all = (n_tok + 3) / 4;
        ce_token_major_kernel<<<(unsigned)(blocks_all < 2048 ? blocks_all : 2048), 256, 0, st>>>(
            logits, targets, weight_ids, n_tok, nt, c.S, c.T, t0, c.factored_vocab, c.num_factored,
            (int64_t)c.image_vocab_size, sums);
    } else {
        ce_bcthw_kernel<<<(unsigned)((n_tok + 255) / 256), 256, 0, st>>>(
            logits, targets, weight_ids, n_tok, nt, c.S, c.T, t0, c.factored_vocab, c.num_factored,
            (int64_t)c.image_vocab_size, sums);
    }
    GENIE_LAUNCH_CHECK("factored_ce");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a13  MaskGIT sampling half (st_mask_git.py:171-190): per factor (most significant first) softmax,
// argmax (first max wins) or inverse-CDF sample, sample = hi*Vf + lo, conf = prod p[sample].
// One wavefront per token; strides make it layout-agnostic (vstride = 1 token-major, S for (B,V,S)).
// Lane l owns the contiguous vocab slice [l*vf/64, (l+1)*vf/64) so the CDF is a wave prefix sum.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, long tok_stride_b,
                                                     long tok_stride_s, long vstride, int B, int S, int vf,
                                                     int nfac, float temperature,
                                                     const float* __restrict__ uniforms,
                                                     int64_t* __restrict__ samples, float* __restrict__ conf) {
    const int lane = threadIdx.x & 63;
    long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= (long)B * S) return;
    long b = n / S, s = n - b * S;
    const float* lp = logits + (size_t)b * tok_stride_b + (size_t)s * tok_stride_s;
    const int per = (vf + 63) / 64;
    int64_t sample = 0;
    float cf = 1.0f;
    for (int k = 0; k < nfac; ++k) {
        const int f = nfac - 1 - k;  // flip(2): hi factor first (:179)
        const float* lf = lp + (size_t)f * vf * vstride;
        float mx = -INFINITY;
        int mi = 0;
        for (int q = 0; q < per; ++q) {
            int idx = lane * per + q;
            if (idx < vf) {
                float v = lf[(size_t)idx * vstride];
                if (v > mx) { mx = v; mi = idx; }
            }
        }
        wave_argmax(mx, mi);
        float part = 0.f;
        for (int q = 0; q < per; ++q) {
            int idx = lane * per + q;
            if (idx < vf) part += expf(lf[(size_t)idx * vstride] - mx);
        }
        const float tot = wave_sum(part);
        int pick = mi;
        float p = 1.0f / tot;  // softmax of the arg-max element: exp(0)/sum
        if (temperature > 1e-8f) {
            // Categorical(probs / T) renormalises: T only switches argmax -> sampling (:184-186)
            const float u = uniforms[((size_t)k * B + b) * S + s] * tot;
            float incl = part;  // inclusive scan of lane partials
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            float run = incl - part;
            int cntl = 0;  // entries of this lane whose inclusive cdf < u
            for (int q = 0; q < per; ++q) {
                int idx = lane * per + q;
                if (idx < vf) {
                    run += expf(lf[(size_t)idx * vstride] - mx);
                    cntl += (run < u) ? 1 : 0;
                }
            }
            int total = cntl;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
            pick = total < vf - 1 ? total : vf - 1;
            p = expf(lf[(size_t)pick * vstride] - mx) / tot;
        }
        sample = sample * vf + pick;
        cf *= p;
    }
    if (lane == 0) { samples[n] = sample; conf[n] = cf; }
}

// The same for token-major logits with vf = 64 PER: a lane's PER consecutive logits arrive as 16-byte loads and stay in registers
// for the three passes (sample_kernel issues PER 4-byte loads at a 4 PER-byte lane stride per pass: 16 lines touched per
// instruction for 2 lines of data).  Same operations in the same order: bit-identical samples and confidences.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PER>
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* __restrict__ logits, long n_tok, long V, int vf, int nfac,
                                                          float temperature, const float* __restrict__ uniforms,
                                                          int64_t* __restrict__ samples, float* __restrict__ conf) {
    static_assert(PER % 4 == 0, "16-byte loads");
    const int lane = threadIdx.x & 63;
    const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_tok) return;
    const float* lp = logits + (size_t)n * V;
    int64_t sample = 0;
    float cf = 1.0f;
    for (int k = 0; k < nfac; ++k) {
        const int f = nfac - 1 - k;  // flip(2): hi factor first (:179)
        const float* lf = lp + (size_t)f * vf;
        float v[PER];
#pragma unroll
        for (int q = 0; q < PER; q += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(lf + lane * PER + q);
            v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
        }
        float mx = -INFINITY;
        int mi = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (v[q] > mx) { mx = v[q]; mi = lane * PER + q; }
        wave_argmax(mx, mi);
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < PER; ++q) part += expf(v[q] - mx);
        const float tot = wave_sum(part);
        int pick = mi;
        float p = 1.0f / tot;
        if (temperature > 1e-8f) {
            const float u = uniforms[(size_t)k * n_tok + n] * tot;
            float incl = part;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                float t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            float run = incl - part;
            int cntl = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                run += expf(v[q] - mx);
                cntl += (run < u) ? 1 : 0;
            }
            int total = cntl;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
            pick = total < vf - 1 ? total : vf - 1;
            p = expf(lf[pick] - mx) / tot;
        }
        sample = sample * vf + pick;
        cf *= p;
    }
    if (lane == 0) { samples[n] = sample; conf[n] = cf; }
}

int launch_sample(const genie_cfg& c, const float* logits, int layout, int B, float temperature,
                  const float* uniforms, int64_t* samples, float* conf, hipStream_t st) {
    long V = (long)c.factored_vocab * c.num_factored;
    long sb, ss, vs;
    if (layout == GENIE_LAYOUT_TOKEN_MAJOR) { sb = (long)c.S * V; ss = V; vs = 1; }
    else { sb = (long)c.S * V; ss = 1; vs = c.S; }
    long n = (long)B * c.S;
    if (layout == GENIE_LAYOUT_TOKEN_MAJOR && c.factored_vocab == 512 && ((uintptr_t)logits & 15) == 0) {
        sample_rows_kernel<8><<<(unsigned)((n + 3) / 4), 256, 0, st>>>(logits, n, V, c.factored_vocab, c.num_factored, temperature,
                                                                       uniforms, samples, conf);
        GENIE_LAUNCH_CHECK("sample_rows");
        return GENIE_OK;
    }
    sample_kernel<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(logits, sb, ss, vs, B, c.S, c.factored_vocab,
                                                            c.num_factored, temperature, uniforms, samples, conf);
    GENIE_LAUNCH_CHECK("sample");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a14  MaskGIT mask half (st_mask_git.py:192-223), one block per clip, S <= 1024 keys in LDS.
// argsort is replaced by a stable rank-by-count: rank_i = #{j : key_j < key_i or (key_j == key_i and j < i)}.
// rank < n  -> re-mask (samples = MASK);  rank >= n -> unmasked = true.  Already-unmasked tokens carry
// key = +inf, so they always rank behind the n still-masked lowest keys, and keep the prompt's value.
// ------------------------------------------------------------------------------------------------
__global__ void mask_step_kernel(const float* __restrict__ keys, int n, int last_step, int64_t mask_id,
                                 uint8_t* __restrict__ unmasked, int64_t* __restrict__ samples,
                                 int64_t* __restrict__ prompt_frame, long clip_stride, int S) {
    extern __shared__ float skey[];
    const int b = blockIdx.x;
    uint8_t* um = unmasked + (size_t)b * S;
    int64_t* sm = samples + (size_t)b * S;
    int64_t* pf = prompt_frame + (size_t)b * clip_stride;
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        float k = 0.f;
        if (!last_step) k = um[i] ? INFINITY : keys[(size_t)b * S + i];
        skey[i] = k;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const bool prev = um[i] != 0;
        int64_t v = sm[i];
        if (!last_step) {
            const float ki = skey[i];
            int rank = 0;
            for (int j = 0; j < S; ++j) {
                const float kj = skey[j];
                rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
            }
            if (rank < n) v = mask_id; else um[i] = 1;
        }
        if (prev) v = pf[i];  // samples_flat[prev_unmasked] = prev_img_flat[prev_unmasked]  (:219)
        sm[i] = v;
        pf[i] = v;            // prompt_THW[:, out_t] = samples_HW  (:223)
    }
}

int launch_mask_step(const float* keys, int n, int last_step, int64_t mask_id, uint8_t* unmasked, int64_t* samples,
                     int64_t* prompt_frame, long clip_stride, int B, int S, hipStream_t st) {
    GENIE_CHECK_SHAPE(S <= 8192, "mask_step: S=%d too large", S);
    int threads = S >= 256 ? 256 : ((S + 63) / 64) * 64;
    mask_step_kernel<<<B, threads, S * sizeof(float), st>>>(keys, n, last_step, mask_id, unmasked, samples,
                                                             prompt_frame, clip_stride, S);
    GENIE_LAUNCH_CHECK("mask_step");
    return GENIE_OK;
}

// assert torch.all(prompt[:, out_t:] == mask)  (st_mask_git.py:155) -- on device, no host sync.
__global__ void check_masked_kernel(const int64_t* __restrict__ prompt, int B, int T, int S, int out_t,
                                    int64_t mask_id, int32_t* flag) {
    long per = (long)(T - out_t) * S;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * per) return;
    long b = idx / per, r = idx - b * per;
    if (prompt[(size_t)b * T * S + (size_t)out_t * S + r] != mask_id) atomicExch(flag, GENIE_E_ASSERT);
}

int launch_check_masked(const int64_t* prompt, int B, int T, int S, int out_t, int64_t mask_id, int32_t* flag,
                        hipStream_t st) {
    long n = (long)B * (T - out_t) * S;
    if (n <= 0 || !flag) return GENIE_OK;
    check_masked_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(prompt, B, T, S, out_t, mask_id, flag);
    GENIE_LAUNCH_CHECK("check_masked");
    return GENIE_OK;
}

// a18  tokens -> +-1 bit planes, LSB first (lookup_free_quantize.py:181-194 + visualize.py:115)
__global__ void bits_kernel(const int64_t* __restrict__ ids, float* __restrict__ z, long n, int hw, int bits) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * bits * hw) return;
    long img = idx / ((long)bits * hw);
    long r = idx - img * (long)bits * hw;
    int c = (int)(r / hw), p = (int)(r - (long)c * hw);
    z[idx] = ((ids[img * hw + p] >> c) & 1) ? 1.0f : -1.0f;
}

int launch_bits(const int64_t* ids, float* z, int n, int hw, int bits, hipStream_t st) {
    long tot = (long)n * bits * hw;
    if (tot <= 0) return GENIE_OK;
    bits_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(ids, z, n, hw, bits);
    GENIE_LAUNCH_CHECK("bits_from_tokens");
    return GENIE_OK;
}

__global__ void pack_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f32_to_bf16(src[i]);
}
int launch_pack_bf16(const float* src, uint16_t* dst, size_t n, hipStream_t st) {
    if (!n) return GENIE_OK;
    pack_bf16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, n);
    GENIE_LAUNCH_CHECK("pack_bf16");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a7  Spatial attention on the f32 matrix cores (exact precision), S = NKT*32 keys, one workgroup per
// (sequence, head), 8 waves, each wave owns 32 query rows.
//
//   S^T = K Q^T   "swapped" so that lane (q = lane&31, h = lane>>5) ends up holding, for ITS query, the
//                 scores of keys {32*kt + (e&3) + 8*(e>>2) + 4h}: the softmax row-reduction is in-lane
//                 plus one cross-half exchange, no LDS round trip.
//   O   = P V     P stays in those registers: register e of tile kt is exactly the A operand of
//                 v_mfma_f32_32x32x2_f32 for the key pair {k0, k0+4}, with B = V[k0 + 4h][d] read
//                 row-wise (conflict-free) from LDS.
// K and V rows (qk-normed on the way in) live in LDS with a +4 float row pad, so the float4 operand
// fetches of K hit 16 distinct 16-byte slots per ds_read_b128 lane group.
// ------------------------------------------------------------------------------------------------
template <int DH, int NKT>
__global__ __launch_bounds__(512) void attn_spatial_f32_mfma_kernel(const float* __restrict__ qkv,
                                                                    float* __restrict__ out, int d, float scale,
                                                                    const float* __restrict__ nw,
                                                                    const float* __restrict__ nb,
                                                                    uint16_t* __restrict__ out16, size_t plane) {
    constexpr int S = NKT * 32, LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;
    float* sV = smem + S * LD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * S;
    const int head = blockIdx.y;
    const float* base = qkv + (size_t)row0 * 3 * d + head * DH;

    // ---- stage K (normalised) and V: two adjacent lanes per row, DH/2 contiguous floats each
    for (int rr = tid >> 1; rr < S; rr += 256) {
        const int half = tid & 1;
        const float* kp = base + (size_t)rr * 3 * d + d + half * (DH / 2);
        const float* vp = kp + d;
        float kx[DH / 2];
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float4 t = *reinterpret_cast<const float4*>(kp + 4 * c);
            kx[4 * c] = t.x; kx[4 * c + 1] = t.y; kx[4 * c + 2] = t.z; kx[4 * c + 3] = t.w;
            *reinterpret_cast<float4*>(&sV[rr * LD + half * (DH / 2) + 4 * c]) =
                *reinterpret_cast<const float4*>(vp + 4 * c);
        }
        if (nw) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) s += kx[c];
            s += __shfl_xor(s, 1);
            const float mu = s / DH;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) { float t = kx[c] - mu; v += t * t; }
            v += __shfl_xor(v, 1);
            const float rs = 1.0f / sqrtf(v / DH + 1e-5f);
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) {
                const int cc = half * (DH / 2) + c;
                kx[c] = (kx[c] - mu) * rs * nw[cc] + nb[cc];
            }
        }
#pragma unroll
        for (int c = 0; c < DH / 8; ++c)
            *reinterpret_cast<float4*>(&sK[rr * LD + half * (DH / 2) + 4 * c]) =
                make_float4(kx[4 * c], kx[4 * c + 1], kx[4 * c + 2], kx[4 * c + 3]);
    }
    __syncthreads();

    for (int qb = wid; qb < NKT; qb += 8) {
        // ---- Q fragment of this wave's 32 queries: lane (r,h) holds Q[r][8kk + 4h + j]
        float4 qf[DH / 8];
        {
            const float* qp = base + (size_t)(qb * 32 + r) * 3 * d + 4 * h;
#pragma unroll
            for (int kk = 0; kk < DH / 8; ++kk) qf[kk] = *reinterpret_cast<const float4*>(qp + 8 * kk);
            if (nw) {
                float s = 0.f;
#pragma unroll
                for (int kk = 0; kk < DH / 8; ++kk) s += qf[kk].x + qf[kk].y + qf[kk].z + qf[kk].w;
                s += __shfl_xor(s, 32);
                const float mu = s / DH;
                float v = 0.f;
#pragma unroll
                for (int kk = 0; kk < DH / 8; ++kk) {
                    float a = qf[kk].x - mu, b = qf[kk].y - mu, c = qf[kk].z - mu, e = qf[kk].w - mu;
                    v += a * a + b * b + c * c + e * e;
                }
                v += __shfl_xor(v, 32);
                const float rs = 1.0f / sqrtf(v / DH + 1e-5f);
#pragma unroll
                for (int kk = 0; kk < DH / 8; ++kk) {
                    const int c0 = 8 * kk + 4 * h;
                    qf[kk].x = (qf[kk].x - mu) * rs * nw[c0] + nb[c0];
                    qf[kk].y = (qf[kk].y - mu) * rs * nw[c0 + 1] + nb[c0 + 1];
                    qf[kk].z = (qf[kk].z - mu) * rs * nw[c0 + 2] + nb[c0 + 2];
                    qf[kk].w = (qf[kk].w - mu) * rs * nw[c0 + 3] + nb[c0 + 3];
                }
            }
            // q *= scale (attention.py:48), and log2(e) with it: the softmax below is 2^(s - max) on v_exp_f32 (one instruction,
            // ~1 ulp) instead of the library expf's dozen -- 128 of them per lane sit between the two MFMA sections
            const float qsc = scale * 1.4426950408889634f;
#pragma unroll
            for (int kk = 0; kk < DH / 8; ++kk) {
                qf[kk].x *= qsc; qf[kk].y *= qsc; qf[kk].z *= qsc; qf[kk].w *= qsc;
            }
        }
        // ---- S^T tiles
        f32x16 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < DH / 8; ++kk) {
                const float4 a = *reinterpret_cast<const float4*>(&sK[(kt * 32 + r) * LD + 8 * kk + 4 * h]);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[kk].x, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[kk].y, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[kk].z, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[kk].w, sc[kt], 0, 0, 0);
            }
        }
        // ---- softmax over the 256 keys of query r (128 in this lane, 128 in lane r^32)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[kt][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sc[kt][e] = __builtin_amdgcn_exp2f(sc[kt][e] - mx); sum += sc[kt][e]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // ---- O = P V
        f32x16 oc[DH / 32];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oc[dt][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = sc[kt][e] * inv;
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
                    oc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, sV[key * LD + dt * 32 + r], oc[dt], 0, 0, 0);
            }
        // ---- store: C/D map row = (e&3) + 8*(e>>2) + 4h (query), col = r (feature)
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int q = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const size_t oi = (size_t)(row0 + q) * d + head * DH + dt * 32 + r;
                if (!out16) out[oi] = oc[dt][e];
                else if (plane) { uint16_t hi, lo; split_f16(oc[dt][e], hi, lo); out16[oi] = hi; out16[plane + oi] = lo; }
                else out16[oi] = f32_to_bf16(oc[dt][e]);
            }
    }
}

// Spatial attention, contiguous sequences of S rows.  Returns GENIE_E_UNSUPPORTED when the shape has no
// MFMA instantiation (the caller then uses the generic kernel).
int launch_attn_spatial_f32_mfma(const float* qkv, float* out, int S, long n_seq, int d, int H, int Dh, float scale,
                                 const float* nw, const float* nb, hipStream_t st, uint16_t* out16, size_t plane) {
    if (S != 256 || (Dh != 32 && Dh != 64)) return GENIE_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * S * (Dh + 4) * sizeof(float);
    dim3 grid((unsigned)n_seq, H);
    ProfScope prof(GENIE_KC_ATTN_SPATIAL, 4.0 * S * S * Dh * H * (double)n_seq, (double)n_seq * S * H * Dh * 16.0, st);
    if (Dh == 64) {
        (void)hipFuncSetAttribute((const void*)attn_spatial_f32_mfma_kernel<64, 8>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_spatial_f32_mfma_kernel<64, 8><<<grid, 512, lds, st>>>(qkv, out, d, scale, nw, nb, out16, plane);
    } else {
        (void)hipFuncSetAttribute((const void*)attn_spatial_f32_mfma_kernel<32, 8>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_spatial_f32_mfma_kernel<32, 8><<<grid, 512, lds, st>>>(qkv, out, d, scale, nw, nb, out16, plane);
    }
    GENIE_LAUNCH_CHECK("attn_spatial_f32_mfma");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// a8  Temporal causal attention on the f32 matrix cores, T = 16 frames, one wavefront per (b, s, head),
// operands straight from HBM into MFMA fragments -- no LDS, no "(B S) T C" transpose (st_transformer.py:77).
//
//   S^T = K Q^T : v_mfma_f32_16x16x4_f32, lane (r = lane&15, g = lane>>4) holds K[r][16g..16g+15] and
//                 Q[r][16g..16g+15] (4 float4 each, 256 B contiguous per token across the 4 lane groups);
//                 MFMA step s consumes element s of every lane, i.e. k in {s, 16+s, 32+s, 48+s}: the k-sum
//                 is order-free so 16 steps cover Dh = 64.  Result: lane (i = r, g) holds scores of query i
//                 against keys j = 4g + e -> causal mask + softmax = 4 in-lane values + 2 cross-group shuffles.
//   O = P V     : same trick on the key axis: step e feeds P[i][4g+e] and V[4g+e][16*dt + r].
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// N contiguous values of a temporal qkv row as f32: from an f32 buffer (exact, f16x3) or from a bf16 one (GENIE_PREC_BF16 keeps
// its temporal qkv / KV cache in bf16: half the bytes of these HBM-bound kernels; the arithmetic below stays f32)
template <int N>
__device__ __forceinline__ void load_vals(const float* p, float* v) {
    if constexpr (N >= 4) {
#pragma unroll
        for (int c = 0; c < N / 4; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(p + 4 * c);
            v[4 * c] = a.x; v[4 * c + 1] = a.y; v[4 * c + 2] = a.z; v[4 * c + 3] = a.w;
        }
    } else {
        const float2 a = *reinterpret_cast<const float2*>(p);
        v[0] = a.x; v[1] = a.y;
    }
}
template <int N>
__device__ __forceinline__ void load_vals(const uint16_t* p, float* v) {
    if constexpr (N >= 8) {
#pragma unroll
        for (int c = 0; c < N / 8; ++c) {
            const uint4 a = *reinterpret_cast<const uint4*>(p + 8 * c);
            const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[8 * c + 2 * j] = __uint_as_float(w[j] << 16);
                v[8 * c + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
            }
        }
    } else if constexpr (N == 4) {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xFFFF0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xFFFF0000u);
    } else {
        const uint32_t a = *reinterpret_cast<const uint32_t*>(p);
        v[0] = __uint_as_float(a << 16); v[1] = __uint_as_float(a & 0xFFFF0000u);
    }
}
__device__ __forceinline__ float load_val(const float* p) { return *p; }
__device__ __forceinline__ float load_val(const uint16_t* p) { return bf16_to_f32(*p); }

// NV (2 or 4) contiguous outputs of one row: f32, bf16 (plane == 0) or split-f16 planes
template <int NV>
__device__ __forceinline__ void store_row_chunk(float* out, uint16_t* out16, size_t plane, size_t oi, const float* v) {
    if (!out16) {
        if constexpr (NV == 4) *reinterpret_cast<float4*>(out + oi) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<float2*>(out + oi) = make_float2(v[0], v[1]);
    } else if (plane) {
        uint16_t hi[NV], lo[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) split_f16(v[c], hi[c], lo[c]);
        if constexpr (NV == 4) {
            *reinterpret_cast<uint2*>(out16 + oi) = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
            *reinterpret_cast<uint2*>(out16 + plane + oi) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
        } else {
            *reinterpret_cast<uint32_t*>(out16 + oi) = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
            *reinterpret_cast<uint32_t*>(out16 + plane + oi) = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
        }
    } else {
        if constexpr (NV == 4)
            *reinterpret_cast<uint2*>(out16 + oi) = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                                                               (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
        else
            *reinterpret_cast<uint32_t*>(out16 + oi) = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    }
}

template <int DH, typename TI>
__global__ __launch_bounds__(256) void attn_temporal_f32_mfma_kernel(const TI* __restrict__ qkv,
                                                                     float* __restrict__ out, long n_bs, int S,
                                                                     int d, int H, int T, int Tq, float scale,
                                                                     const float* __restrict__ nw,
                                                                     const float* __restrict__ nb,
                                                                     uint16_t* __restrict__ out16, size_t plane) {
    // Tq: frames per clip in the layout of `qkv` (>= T; the output is dense (B,T,S,d))
    constexpr int PER = DH / 4;  // floats per lane per row; T <= 16 frames fill a 16x16 tile (rows >= T are padding:
                                 // they repeat frame T-1 on the load side and are never stored)
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long bs = wave / H;
    const int head = (int)(wave - bs * H);
    if (bs >= n_bs) return;
    const long b = bs / S, s = bs - b * S;
    const long tok_stride = (long)S * 3 * d;                       // frame t -> t+1
    const TI* base = qkv + ((size_t)(b * Tq) * S + s) * 3 * d + head * DH;
    const TI* qp = base + (size_t)(r < T ? r : T - 1) * tok_stride + g * PER;     // row t = r
    float q[PER], k[PER];
    load_vals<PER>(qp, q);
    load_vals<PER>(qp + d, k);
    if (nw) {  // qk-norm, f32, one shared affine (attention.py:42-47); row spread over the 4 lane groups
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int c = 0; c < PER; ++c) { sq += q[c]; sk += k[c]; }
        sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
        sk += __shfl_xor(sk, 16); sk += __shfl_xor(sk, 32);
        const float mq = sq / DH, mk = sk / DH;
        float vq = 0.f, vk = 0.f;
#pragma unroll
        for (int c = 0; c < PER; ++c) { float a = q[c] - mq, bb = k[c] - mk; vq += a * a; vk += bb * bb; }
        vq += __shfl_xor(vq, 16); vq += __shfl_xor(vq, 32);
        vk += __shfl_xor(vk, 16); vk += __shfl_xor(vk, 32);
        const float rq = 1.0f / sqrtf(vq / DH + 1e-5f), rk = 1.0f / sqrtf(vk / DH + 1e-5f);
#pragma unroll
        for (int c = 0; c < PER; ++c) {
            const float w_ = nw[g * PER + c], b_ = nb[g * PER + c];
            q[c] = (q[c] - mq) * rq * w_ + b_;
            k[c] = (k[c] - mk) * rk * w_ + b_;
        }
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f};  // S^T: row = key j = 4g + e, col = query i = r
#pragma unroll
    for (int c = 0; c < PER; ++c) st = __builtin_amdgcn_mfma_f32_16x16x4f32(k[c], q[c] * scale, st, 0, 0, 0);
    // causal mask (attention.py:51-55: -finfo.max before softmax -> exact zeros) + softmax over j
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * g + e > r) st[e] = -INFINITY;
        mx = fmaxf(mx, st[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { st[e] = expf(st[e] - mx); sum += st[e]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    // O = P V.  Feature permutation: MFMA column r of "d-tile" c is feature NV*r + c, so a lane fetches NV
    // CONTIGUOUS features of V[4g+e] (16 lanes = one 4*DH-byte row segment) and later stores NV contiguous outputs.
    constexpr int NV = DH / 16;
    const TI* vp = base + 2 * d + NV * r;
    float vv[4][NV];
#pragma unroll
    for (int e = 0; e < 4; ++e) load_vals<NV>(vp + (size_t)(4 * g + e < T ? 4 * g + e : T - 1) * tok_stride, vv[e]);
    f32x4 o[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[e] * inv, vv[e][c], o[c], 0, 0, 0);
    }
    // C/D map: col = lane&15 (-> features NV*r..NV*r+NV-1 over c), row = 4*(lane>>4) + e (query frame)
    const size_t obase = ((size_t)(b * T) * S + s) * d + head * DH + NV * r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * g + e >= T) continue;
        const size_t oi = obase + (size_t)(4 * g + e) * S * d;
        float ov[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) ov[c] = o[c][e];
        store_row_chunk<NV>(out, out16, plane, oi, ov);
    }
}

// Temporal attention over 8 <= T <= 16 frames on a (B,T,S,3d) buffer; GENIE_E_UNSUPPORTED for other geometries.
// in16: `qkv` holds bf16 values (GENIE_PREC_BF16's temporal qkv); any 1 <= T <= 16 then (there is no other bf16-input kernel:
// rows >= T of the 16x16 tile are padding whatever T is).
int launch_attn_temporal_f32_mfma(const float* qkv, float* out, int B, int T, int S, int d, int H, int Dh, float scale,
                                  const float* nw, const float* nb, hipStream_t st, uint16_t* out16, size_t plane, int Tq,
                                  bool in16) {
    if (T > 16 || T < (in16 ? 1 : 8) || (Dh != 32 && Dh != 64)) return GENIE_E_UNSUPPORTED;
    if (Tq <= 0) Tq = T;
    const long n_bs = (long)B * S, waves = n_bs * H;
    ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 4.0 * T * T * Dh * (double)waves,
                   (double)waves * T * Dh * (in16 ? 8.0 : 16.0), st);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (in16) {
        const uint16_t* q16 = reinterpret_cast<const uint16_t*>(qkv);
        if (Dh == 64) attn_temporal_f32_mfma_kernel<64, uint16_t><<<blocks, 256, 0, st>>>(q16, out, n_bs, S, d, H, T, Tq, scale, nw, nb, out16, plane);
        else attn_temporal_f32_mfma_kernel<32, uint16_t><<<blocks, 256, 0, st>>>(q16, out, n_bs, S, d, H, T, Tq, scale, nw, nb, out16, plane);
    }
    else if (Dh == 64) attn_temporal_f32_mfma_kernel<64, float><<<blocks, 256, 0, st>>>(qkv, out, n_bs, S, d, H, T, Tq, scale, nw, nb, out16, plane);
    else attn_temporal_f32_mfma_kernel<32, float><<<blocks, 256, 0, st>>>(qkv, out, n_bs, S, d, H, T, Tq, scale, nw, nb, out16, plane);
    GENIE_LAUNCH_CHECK("attn_temporal_f32_mfma");
    return GENIE_OK;
}

// rescale_magvit_output (visualize.py:84-92) with the reference's bf16 intermediate roundings
__global__ void rescale_u8_bf16_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = bf16_to_f32(x[i]);
    v = bf16_to_f32(f32_to_bf16(v + 1.0f));
    v = bf16_to_f32(f32_to_bf16(v * 127.5f));
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    out[i] = (uint8_t)v;  // truncating cast
}
__global__ void rescale_u8_f32_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = (x[i] + 1.0f) * 127.5f;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    out[i] = (uint8_t)v;
}
int launch_rescale_u8(const void* x, int is_bf16, uint8_t* out, size_t n, hipStream_t st) {
    if (!n) return GENIE_OK;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (is_bf16) rescale_u8_bf16_kernel<<<blocks, 256, 0, st>>>((const uint16_t*)x, out, n);
    else rescale_u8_f32_kernel<<<blocks, 256, 0, st>>>((const float*)x, out, n);
    GENIE_LAUNCH_CHECK("rescale_u8");
    return GENIE_OK;
}

// encoder output (n, bits, hw) -> dataset-convention token ids, bit c = [h_c > 0]
__global__ void tokens_from_bits_kernel(const float* __restrict__ h, int64_t* __restrict__ ids, long n, int hw,
                                        int bits) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * hw) return;
    long img = idx / hw, p = idx - img * hw;
    int64_t id = 0;
    for (int c = 0; c < bits; ++c) id |= (int64_t)(h[(img * bits + c) * hw + p] > 0.0f) << c;
    ids[idx] = id;
}
int launch_tokens_from_bits(const float* h, int64_t* ids, int n, int hw, int bits, hipStream_t st) {
    long tot = (long)n * hw;
    if (tot <= 0) return GENIE_OK;
    tokens_from_bits_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(h, ids, n, hw, bits);
    GENIE_LAUNCH_CHECK("tokens_from_bits");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Teacher-forced prefix reuse: temporal attention of "frame t in timeline t".
// In the evaluator's timeline t the frames < t are ground truth, and because temporal attention is causal
// and everything else is per-frame, their activations are those of ONE clean pass over the ground-truth clip.
// So query frame i attends keys j < i taken from the clean pass's cached temporal qkv (`cache`, same layout
// (B,T,S,3d)) and key j = i from the current buffer `cur`; current frames never see each other.
//   scores_j<i = scale q_i . kc_j   (MFMA, as attn_temporal_f32_mfma_kernel)      diag = scale q_i . k_i
//   o_i = sum_j<i p_ij vc_j  +  p_ii v_i
// ------------------------------------------------------------------------------------------------
template <int DH, typename TI>
__global__ __launch_bounds__(256) void attn_temporal_prefix_f32_mfma_kernel(
    const TI* __restrict__ cur, const TI* __restrict__ cache, float* __restrict__ out, long n_bs, int S, int d,
    int H, int T, int sh, float scale, const float* __restrict__ nw, const float* __restrict__ nb,
    uint16_t* __restrict__ out16, size_t plane) {
    // T <= 16 frame slots in `cur` and in `cache` (tile rows / columns >= T are padding: loads repeat slot T-1, nothing is
    // stored); query slot i sees cached slots j < i + sh (sh = 0: cur and cache hold the same frames; sh = 1: cur holds
    // clip frames 1..T, cache clip frames 0..T-1)
    constexpr int PER = DH / 4;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long bs = wave / H;
    const int head = (int)(wave - bs * H);
    if (bs >= n_bs) return;
    const long b = bs / S, s = bs - b * S;
    const long tok_stride = (long)S * 3 * d;
    const size_t off0 = ((size_t)(b * T) * S + s) * 3 * d + head * DH;
    const int rr = r < T ? r : T - 1;
    const TI* qp = cur + off0 + (size_t)rr * tok_stride + g * PER;
    const TI* kcp = cache + off0 + (size_t)rr * tok_stride + d + g * PER;
    float q[PER], k[PER], kc[PER];
    load_vals<PER>(qp, q);
    load_vals<PER>(qp + d, k);
    load_vals<PER>(kcp, kc);
    if (nw) {
        float sq = 0.f, sk = 0.f, sc = 0.f;
#pragma unroll
        for (int c = 0; c < PER; ++c) { sq += q[c]; sk += k[c]; sc += kc[c]; }
        sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
        sk += __shfl_xor(sk, 16); sk += __shfl_xor(sk, 32);
        sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
        const float mq = sq / DH, mk = sk / DH, mc = sc / DH;
        float vq = 0.f, vk = 0.f, vc = 0.f;
#pragma unroll
        for (int c = 0; c < PER; ++c) {
            float a = q[c] - mq, bb = k[c] - mk, cc = kc[c] - mc;
            vq += a * a; vk += bb * bb; vc += cc * cc;
        }
        vq += __shfl_xor(vq, 16); vq += __shfl_xor(vq, 32);
        vk += __shfl_xor(vk, 16); vk += __shfl_xor(vk, 32);
        vc += __shfl_xor(vc, 16); vc += __shfl_xor(vc, 32);
        const float rq = 1.0f / sqrtf(vq / DH + 1e-5f), rk = 1.0f / sqrtf(vk / DH + 1e-5f),
                    rc = 1.0f / sqrtf(vc / DH + 1e-5f);
#pragma unroll
        for (int c = 0; c < PER; ++c) {
            const float w_ = nw[g * PER + c], b_ = nb[g * PER + c];
            q[c] = (q[c] - mq) * rq * w_ + b_;
            k[c] = (k[c] - mk) * rk * w_ + b_;
            kc[c] = (kc[c] - mc) * rc * w_ + b_;
        }
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f};  // row = cached key j = 4g + e, col = query i = r
    float dg = 0.f;                   // diagonal: q_i . k_i (same q*scale products as the MFMA path)
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        const float qs = q[c] * scale;
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[c], qs, st, 0, 0, 0);
        dg = fmaf(k[c], qs, dg);
    }
    dg += __shfl_xor(dg, 16);
    dg += __shfl_xor(dg, 32);
    float mx = dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * g + e >= r + sh || 4 * g + e >= T) st[e] = -INFINITY;  // strictly earlier clip frames only
        mx = fmaxf(mx, st[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { st[e] = expf(st[e] - mx); sum += st[e]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float pd = expf(dg - mx);
    const float inv = 1.0f / (sum + pd);
    const float pdn = pd * inv;  // weight of the query's own (current) frame, held by every lane with r == i
    constexpr int NV = DH / 16;  // feature permutation as in attn_temporal_f32_mfma_kernel
    const TI* vcp = cache + off0 + 2 * d + NV * r;
    const TI* vp = cur + off0 + 2 * d + NV * r;
    float vc[4][NV], vs[4][NV];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t fo = (size_t)(4 * g + e < T ? 4 * g + e : T - 1) * tok_stride;
        load_vals<NV>(vcp + fo, vc[e]);
        load_vals<NV>(vp + fo, vs[e]);
    }
    float pself[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pself[e] = __shfl(pdn, 4 * g + e);  // lane index 4g+e has r = 4g+e
    f32x4 o[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[e] * inv, vc[e][c], o[c], 0, 0, 0);
    }
    const size_t obase = ((size_t)(b * T) * S + s) * d + head * DH + NV * r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * g + e >= T) continue;
        const size_t oi = obase + (size_t)(4 * g + e) * S * d;
        float ov[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) ov[c] = fmaf(pself[e], vs[e][c], o[c][e]);
        store_row_chunk<NV>(out, out16, plane, oi, ov);
    }
}

// generic geometry (any T <= 64, Dh in {8,16,32,64}): one thread per (b, s, head, frame)
template <int DH>
__global__ void attn_temporal_prefix_generic_kernel(const float* __restrict__ cur, const float* __restrict__ cache,
                                                    float* __restrict__ out, long n, int T, int sh, int S, int d, int H,
                                                    float scale, const float* __restrict__ nw,
                                                    const float* __restrict__ nb) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int i = (int)(idx % T);
    long rest = idx / T;
    const int head = (int)(rest % H);
    rest /= H;
    const long s = rest % S, b = rest / S;
    const long tok_stride = (long)S * 3 * d;
    const size_t off0 = ((size_t)(b * T) * S + s) * 3 * d + head * DH;
    auto load_norm = [&](const float* p, float* v) {
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c) { v[c] = p[c]; m += v[c]; }
        if (nw) {
            m /= DH;
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) { float t = v[c] - m; var += t * t; }
            const float rs = 1.0f / sqrtf(var / DH + 1e-5f);
#pragma unroll
            for (int c = 0; c < DH; ++c) v[c] = (v[c] - m) * rs * nw[c] + nb[c];
        }
    };
    float q[DH], kk[DH], o[DH];
    load_norm(cur + off0 + (size_t)i * tok_stride, q);
#pragma unroll
    for (int c = 0; c < DH; ++c) { q[c] *= scale; o[c] = 0.f; }
    float sc[65];
    float mx = -INFINITY;
    const int nc = i + sh;  // cached slots 0..nc-1, then the query's own (current) slot
    for (int j = 0; j <= nc; ++j) {
        const float* src = (j < nc ? cache + off0 + (size_t)j * tok_stride : cur + off0 + (size_t)i * tok_stride) + d;
        load_norm(src, kk);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c) a = fmaf(q[c], kk[c], a);
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    for (int j = 0; j <= nc; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
    const float inv = 1.0f / sum;
    for (int j = 0; j <= nc; ++j) {
        const float* vsrc = (j < nc ? cache + off0 + (size_t)j * tok_stride : cur + off0 + (size_t)i * tok_stride) + 2 * d;
        const float p = sc[j] * inv;
#pragma unroll
        for (int c = 0; c < DH; ++c) o[c] = fmaf(p, vsrc[c], o[c]);
    }
    float* op = out + ((size_t)(b * T + i) * S + s) * d + head * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) op[c] = o[c];
}

// out16/plane as in launch_attn_temporal_f32_mfma; the generic fallback writes f32 `out` only
// (returns GENIE_E_UNSUPPORTED if a 16-bit output is requested for a geometry without an MFMA instantiation).
int launch_attn_temporal_prefix(const float* cur, const float* cache, float* out, int B, int T, int S, int d, int H,
                                int Dh, float scale, const float* nw, const float* nb, hipStream_t st,
                                uint16_t* out16, size_t plane, int sh, bool in16) {
    // T frame slots in both buffers; query slot i sees cached slots j < i + sh and itself (sh in {0, 1})
    // in16: both buffers hold bf16 values (GENIE_PREC_BF16); MFMA kernel only, any 1 <= T <= 16
    GENIE_CHECK_SHAPE(sh == 0 || sh == 1, "prefix attention: shift %d", sh);
    const long n_bs = (long)B * S, waves = n_bs * H;
    ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 4.0 * T * T * Dh * (double)waves,
                   (double)waves * T * Dh * (in16 ? 14.0 : 28.0), st);
    if (T <= 16 && T >= (in16 ? 1 : 8) && (Dh == 32 || Dh == 64)) {
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        if (in16) {
            const uint16_t* c16 = reinterpret_cast<const uint16_t*>(cur);
            const uint16_t* k16 = reinterpret_cast<const uint16_t*>(cache);
            if (Dh == 64)
                attn_temporal_prefix_f32_mfma_kernel<64, uint16_t><<<blocks, 256, 0, st>>>(c16, k16, out, n_bs, S, d, H, T, sh, scale,
                                                                                           nw, nb, out16, plane);
            else
                attn_temporal_prefix_f32_mfma_kernel<32, uint16_t><<<blocks, 256, 0, st>>>(c16, k16, out, n_bs, S, d, H, T, sh, scale,
                                                                                           nw, nb, out16, plane);
        } else if (Dh == 64)
            attn_temporal_prefix_f32_mfma_kernel<64, float><<<blocks, 256, 0, st>>>(cur, cache, out, n_bs, S, d, H, T, sh, scale, nw,
                                                                                    nb, out16, plane);
        else
            attn_temporal_prefix_f32_mfma_kernel<32, float><<<blocks, 256, 0, st>>>(cur, cache, out, n_bs, S, d, H, T, sh, scale, nw,
                                                                                    nb, out16, plane);
        GENIE_LAUNCH_CHECK("attn_temporal_prefix_mfma");
        return GENIE_OK;
    }
    if (out16 || in16) return GENIE_E_UNSUPPORTED;
    GENIE_CHECK_SHAPE(T <= 64, "prefix attention: T=%d > 64", T);
    const long n = waves * T;
    const unsigned blocks = (unsigned)((n + 127) / 128);
    switch (Dh) {
        case 8: attn_temporal_prefix_generic_kernel<8><<<blocks, 128, 0, st>>>(cur, cache, out, n, T, sh, S, d, H, scale, nw, nb); break;
        case 16: attn_temporal_prefix_generic_kernel<16><<<blocks, 128, 0, st>>>(cur, cache, out, n, T, sh, S, d, H, scale, nw, nb); break;
        case 32: attn_temporal_prefix_generic_kernel<32><<<blocks, 128, 0, st>>>(cur, cache, out, n, T, sh, S, d, H, scale, nw, nb); break;
        case 64: attn_temporal_prefix_generic_kernel<64><<<blocks, 128, 0, st>>>(cur, cache, out, n, T, sh, S, d, H, scale, nw, nb); break;
        default: set_error("prefix attention: head_dim %d unsupported", Dh); return GENIE_E_SHAPE;
    }
    GENIE_LAUNCH_CHECK("attn_temporal_prefix_generic");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Temporal attention of ONE query frame t against the cached qkv of frames 0..t (the frame's own qkv has just been
// written into slot t): the decode step of the temporal KV cache used by generate().  One wavefront per
// (b, s, head), lane = feature; T <= 64 scores live in registers of lane 0..t after wave reductions.
// out: dense (B, S, d) f32 / bf16 / split-f16.
// ------------------------------------------------------------------------------------------------
template <int DH, typename TI>
__global__ __launch_bounds__(256) void attn_temporal_single_kernel(const TI* __restrict__ cache,
                                                                   float* __restrict__ out, long n_items, int T, int S,
                                                                   int t, int d, int H, float scale,
                                                                   const float* __restrict__ nw,
                                                                   const float* __restrict__ nb,
                                                                   uint16_t* __restrict__ out16, size_t plane) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int head = (int)(item % H);
    const long bs = item / H;
    const long b = bs / S, s = bs - b * S;
    const bool act = lane < DH;
    const size_t tok_stride = (size_t)S * 3 * d;
    const TI* base = cache + ((size_t)(b * T) * S + s) * 3 * d + head * DH + (act ? lane : 0);
    auto norm = [&](float v) {  // qk-norm over the DH active lanes
        if (!nw) return v;
        const float mu = wave_sum(act ? v : 0.f) / DH;
        const float c = act ? v - mu : 0.f;
        const float var = wave_sum(c * c) / DH;
        return c * (1.0f / sqrtf(var + 1e-5f)) * nw[act ? lane : 0] + nb[act ? lane : 0];
    };
    if constexpr (DH >= 16) {
        if (t < 16 && !nw) {
            // The shipped window without qk-norm: lane (j = lane / 4, c = lane % 4) takes frame j's key, features [c CH, (c + 1) CH):
            // all t + 1 scores come out of ONE pass -- a CH-long dot product per lane and two cross-lane adds -- instead of t + 1
            // 64-lane reductions one after the other (the kernel was bound by that chain: 58 us at 16 clips for 200 MB of cache);
            // max / sum over the 16 frame groups are four butterfly steps each.  P.V stays lane = feature with p_j broadcast from
            // lane 4 j; every key and value load is issued before the first use.
            constexpr int CH = DH / 4;
            const int j = lane >> 2, c4 = lane & 3;
            const TI* hb = cache + ((size_t)(b * T) * S + s) * 3 * d + head * DH;
            float qv[CH], kv[CH], vj[16];
            load_vals<CH>(hb + (size_t)t * tok_stride + c4 * CH, qv);
            load_vals<CH>(hb + (size_t)(j <= t ? j : t) * tok_stride + d + c4 * CH, kv);
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)   // (frames past t re-read frame t, branch-free: their probability is 0)
                vj[jj] = load_val(hb + (size_t)(jj <= t ? jj : t) * tok_stride + 2 * d + (act ? lane : 0));
            // (ties the score arithmetic behind the issue of the value loads: one memory round trip, not two)
            asm volatile("" : "+v"(qv[0]), "+v"(kv[0]) :: "memory");
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < CH; ++e) part = fmaf(qv[e], kv[e], part);
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            const float sc = j <= t ? part * scale : -INFINITY;
            float mxs = sc;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) mxs = fmaxf(mxs, __shfl_xor(mxs, o));
            const float pe = j <= t ? expf(sc - mxs) : 0.f;
            float sm = pe;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) sm += __shfl_xor(sm, o);
            const float pn = pe * (1.0f / sm);
            float o16 = 0.f;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) o16 = fmaf(__shfl(pn, 4 * jj), vj[jj], o16);
            if (!act) return;
            const size_t oi16 = (size_t)bs * d + head * DH + lane;
            if (!out16) out[oi16] = o16;
            else if (plane) { uint16_t hi, lo; split_f16(o16, hi, lo); out16[oi16] = hi; out16[plane + oi16] = lo; }
            else out16[oi16] = f32_to_bf16(o16);
            return;
        }
    }
    float q = norm(load_val(base + (size_t)t * tok_stride)) * scale;
    if (!act) q = 0.f;
    if (t < 16) {
        // the shipped window (T = 16): every cached key / value of the (position, head) is fetched up front -- 2 (t + 1)
        // independent 4-byte-per-lane loads in flight instead of a load -> reduce -> load chain -- and the scores live in
        // registers (static indices)
        float kj[16], vj[16], sc16[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            kj[j] = 0.f; vj[j] = 0.f;
            if (j <= t) { kj[j] = load_val(base + (size_t)j * tok_stride + d); vj[j] = load_val(base + (size_t)j * tok_stride + 2 * d); }
        }
        float mx16 = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j <= t) {
                const float a = wave_sum(act ? q * norm(kj[j]) : 0.f);
                sc16[j] = a;
                mx16 = fmaxf(mx16, a);
            }
        }
        float sum16 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j <= t) { sc16[j] = expf(sc16[j] - mx16); sum16 += sc16[j]; }
        const float inv16 = 1.0f / sum16;
        float o16 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j <= t) o16 = fmaf(sc16[j] * inv16, vj[j], o16);
        if (!act) return;
        const size_t oi16 = (size_t)bs * d + head * DH + lane;
        if (!out16) out[oi16] = o16;
        else if (plane) { uint16_t hi, lo; split_f16(o16, hi, lo); out16[oi16] = hi; out16[plane + oi16] = lo; }
        else out16[oi16] = f32_to_bf16(o16);
        return;
    }
    float sc[64];
    float mx = -INFINITY;
    for (int j = 0; j <= t; ++j) {
        const float kj = norm(load_val(base + (size_t)j * tok_stride + d));
        const float a = wave_sum(act ? q * kj : 0.f);
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    for (int j = 0; j <= t; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
    const float inv = 1.0f / sum;
    float o = 0.f;
    for (int j = 0; j <= t; ++j) o = fmaf(sc[j] * inv, load_val(base + (size_t)j * tok_stride + 2 * d), o);
    if (!act) return;
    const size_t oi = (size_t)bs * d + head * DH + lane;
    if (!out16) out[oi] = o;
    else if (plane) { uint16_t hi, lo; split_f16(o, hi, lo); out16[oi] = hi; out16[plane + oi] = lo; }
    else out16[oi] = f32_to_bf16(o);
}

int launch_attn_temporal_single(const float* cache, float* out, int B, int T, int S, int t, int d, int H, int Dh,
                                float scale, const float* nw, const float* nb, hipStream_t st, uint16_t* out16,
                                size_t plane, bool in16) {
    GENIE_CHECK_SHAPE(T <= 64 && t >= 0 && t < T, "temporal_single: bad frame %d of %d", t, T);
    const long n = (long)B * S * H;
    ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 4.0 * (t + 1) * Dh * (double)n, (double)n * Dh * (in16 ? 2.0 : 4.0) * (2 * t + 4), st);
    const unsigned blocks = (unsigned)((n + 3) / 4);
    const uint16_t* c16 = reinterpret_cast<const uint16_t*>(cache);
#define SINGLE(DH_)                                                                                                          \
    case DH_:                                                                                                                \
        if (in16) attn_temporal_single_kernel<DH_, uint16_t><<<blocks, 256, 0, st>>>(c16, out, n, T, S, t, d, H, scale, nw, nb, out16, plane); \
        else attn_temporal_single_kernel<DH_, float><<<blocks, 256, 0, st>>>(cache, out, n, T, S, t, d, H, scale, nw, nb, out16, plane);     \
        break;
    switch (Dh) {
        SINGLE(8) SINGLE(16) SINGLE(32) SINGLE(64)
        default: set_error("temporal_single: head_dim %d unsupported", Dh); return GENIE_E_SHAPE;
    }
#undef SINGLE
    GENIE_LAUNCH_CHECK("attn_temporal_single");
    return GENIE_OK;
}

}  // namespace genie
