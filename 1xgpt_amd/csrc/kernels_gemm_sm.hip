// gemm16_sm_kernel: the 16-bit NT GEMM for SMALL problems (batch-1 generate: M = 256 rows per frame pass), where a launch is
// bound by the serial chain "load a K-step -> barrier -> MFMA" of a tiled kernel, not by throughput.
//
//   C[M,N] (+)= epilogue( alpha * A[M,K] . W[N,K]^T + bias )      (nn.Linear; st_transformer.py:16-25, attention.py:27-29)
//
//   * No LDS staging and no barrier in the main loop: a workgroup owns one TM x TN output tile and its NW waves split K
//     between them (split-K INSIDE the workgroup).  Each wave streams its K range of the tile's A rows and W rows straight
//     from L2 into MFMA fragments, the whole range (or 64-wide chunks, double buffered) in flight at once.
//   * Fragment loads are 64 contiguous bytes per lane: the k index inside a 64-wide block is permuted (lane half h takes
//     k = 32h .. 32h+31; values 8s .. 8s+7 feed MFMA step s = 0..3) -- a contraction does not care about the order of k as
//     long as A and W agree.  The two halves of the wave consume one whole 128-byte line of every row they touch.
//   * The NW partial tiles meet in LDS and are added in wave order (fixed order: bit-reproducible), then the usual fused
//     epilogue (bias, erf-GELU, residual, f32 and/or 16-bit operand output) runs on whole rows.
//   * LNF: the A operand is LayerNorm(x) of an f32 activation (nn.LayerNorm in front of qkv / fc1, st_transformer.py:73, 81): the
//     kernel loads A into registers anyway, so every wave reads ITS K range of the tile's 32 f32 rows, the row statistics are
//     completed across the 8 waves through LDS (mean, then the centred second moment: the two-pass form of the stand-alone
//     kernel), and the normalised values are rounded into MFMA operands in registers -- the LayerNorm launch and its operand
//     round trip disappear from the one-frame passes.
//   * NPL = 2 ("f16x3"): split operands a = hi + lo'/2048; hi.hi goes to one accumulator, hi.lo' + lo'.hi to a second one
//     that is scaled by 2^-11 at the end (no in-register weight scaling: valid for any operand magnitudes).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__device__ __forceinline__ f32x16 mma_sm(const s16x8& a, const s16x8& b, const f32x16& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
}  // namespace

template <int NPL, int TM, int TN, int NW, int NB, bool LNF = false>
__global__ __launch_bounds__(NW * 64, 1) void gemm16_sm_kernel(const uint16_t* __restrict__ A, long lda, long planeA,
                                                                const uint16_t* __restrict__ W, long ldw, long planeW,
                                                                const float* __restrict__ bias, float* __restrict__ Cf,
                                                                uint16_t* __restrict__ C16, long plane16, long ldc, int M, int N,
                                                                int K, int flags, float alpha, long strideA, long strideC,
                                                                const float* Rf, long strideW, const float* __restrict__ Xf = nullptr,
                                                                long ldx = 0, const float* __restrict__ ln_g = nullptr,
                                                                const float* __restrict__ ln_b = nullptr, float ln_eps = 0.f) {
    constexpr int MI = TM / 32, NJ = TN / 32;
    static_assert(!LNF || (MI == 1 && NB == 1), "LayerNorm-fused A operand: 32-row tiles, one 64-k block per wave");
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][TM][TN] partial tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nt = (N + TN - 1) / TN;
    const int m0 = (blockIdx.x / nt) * TM, n0 = (blockIdx.x % nt) * TN;
    if constexpr (!LNF) A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;
    const int KW = K / NW;                  // this wave's K range (launcher: KW % 64 == 0)
    const int kw0 = wid * KW;

    const uint16_t* ap[MI];
    const uint16_t* bp[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int row = m0 + 32 * i + r;
        row = row < M ? row : M - 1;        // ragged tiles: the surplus rows are computed and never stored
        ap[i] = LNF ? nullptr : A + (size_t)row * lda + kw0 + 32 * h;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int row = n0 + 32 * j + r;
        row = row < N ? row : N - 1;
        bp[j] = W + (size_t)row * ldw + kw0 + 32 * h;
    }

    f32x16 accm[MI][NJ], accc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[i][j][e] = 0.f; accc[i][j][e] = 0.f; }

    // The epilogue's residual and bias values of this thread (one float4 each when the tile has <= NT float4s: every shipped
    // shape) are requested NOW: read after the reduce they were one more exposed L2 round trip of a kernel that is one round
    // trip long.  (The residual may alias the output; this thread is the only one that writes what it reads.)
    constexpr int C4 = TN / 4;
    constexpr bool PRE = TM * C4 <= NT;
    const float* Rsrc0 = Rf ? Rf + (size_t)blockIdx.y * strideC : (Cf ? Cf + (size_t)blockIdx.y * strideC : nullptr);
    float4 pre_r = make_float4(0.f, 0.f, 0.f, 0.f), pre_b = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PRE) {
        const int rl = tid / C4, c4 = (tid % C4) * 4;
        const int row = m0 + rl, col = n0 + c4;
        if (tid < TM * C4 && row < M && col < N) {
            if (bias) pre_b = *reinterpret_cast<const float4*>(bias + col);
            if ((flags & G16X_ACCUM) && Rsrc0) pre_r = *reinterpret_cast<const float4*>(Rsrc0 + (size_t)row * ldc + col);
        }
    }

    // one block = 64 k: per (row tile, plane) a lane holds 32 values = the operands of four MFMA steps.  NB blocks are in
    // flight per wave (a register ring: a block's registers are refilled right after its MFMAs have been issued).
    struct Block {
        s16x8 a[MI][NPL][4];
        s16x8 b[NJ][NPL][4];
    };
    constexpr int REGS = (MI + NJ) * NPL * 16;
    auto load = [&](Block& c, int k) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const s16x8* src = reinterpret_cast<const s16x8*>(ap[i] + (size_t)p * planeA + k);
#pragma unroll
                for (int s = 0; s < 4; ++s) c.a[i][p][s] = src[s];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const s16x8* src = reinterpret_cast<const s16x8*>(bp[j] + (size_t)p * planeW + k);
#pragma unroll
                for (int s = 0; s < 4; ++s) c.b[j][p][s] = src[s];
            }
        }
    };
    auto compute = [&](const Block& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (NPL == 1) {
                        accm[i][j] = mma_sm<false>(c.a[i][0][s], c.b[j][0][s], accm[i][j]);
                    } else {
                        accm[i][j] = mma_sm<true>(c.a[i][0][s], c.b[j][0][s], accm[i][j]);
                        accc[i][j] = mma_sm<true>(c.a[i][0][s], c.b[j][NPL - 1][s], accc[i][j]);
                        accc[i][j] = mma_sm<true>(c.a[i][NPL - 1][s], c.b[j][0][s], accc[i][j]);
                    }
                }
    };
    static_assert(REGS * NB + MI * NJ * 16 * NPL <= (NW > 4 ? 232 : 480), "fragment ring + accumulators exceed the register file");
    Block buf[NB];
    const int nblk = KW / 64;
    if constexpr (LNF) {
        // ---- A = LayerNorm(x): lane (r, h) owns x[row r][kw0 + 32 h .. + 31] (K = NW * 64: the waves tile the whole row)
        float* stat = red + (size_t)NW * TM * TN;            // [2][NW][32] partial sums
        int row = m0 + r;
        row = row < M ? row : M - 1;
        const float* xp = Xf + (size_t)row * ldx + kw0 + 32 * h;
        float xv[32], gv[32], bv[32];
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(xp + k);
            xv[k] = t.x; xv[k + 1] = t.y; xv[k + 2] = t.z; xv[k + 3] = t.w;
        }
        // the W fragments of this wave's block go out now: they are in flight while the statistics are exchanged
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const s16x8* src = reinterpret_cast<const s16x8*>(bp[j] + (size_t)p * planeW);
#pragma unroll
                for (int s = 0; s < 4; ++s) buf[0].b[j][p][s] = src[s];
            }
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const float4 tg = *reinterpret_cast<const float4*>(ln_g + kw0 + 32 * h + k);
            const float4 tb = *reinterpret_cast<const float4*>(ln_b + kw0 + 32 * h + k);
            gv[k] = tg.x; gv[k + 1] = tg.y; gv[k + 2] = tg.z; gv[k + 3] = tg.w;
            bv[k] = tb.x; bv[k + 1] = tb.y; bv[k + 2] = tb.z; bv[k + 3] = tb.w;
        }
        float sx = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) sx += xv[k];
        sx += __shfl_xor(sx, 32);
        if (h == 0) stat[wid * 32 + r] = sx;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += stat[w * 32 + r];   // wave order: the same sum in every wave
        const float mean = tot * (1.0f / (float)(NW * 64));
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) { xv[k] -= mean; q += xv[k] * xv[k]; }
        q += __shfl_xor(q, 32);
        if (h == 0) stat[NW * 32 + wid * 32 + r] = q;
        __syncthreads();
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) qt += stat[NW * 32 + w * 32 + r];
        const float rstd = 1.0f / sqrtf(qt * (1.0f / (float)(NW * 64)) + ln_eps);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * s + e;
                const float y = xv[k] * rstd * gv[k] + bv[k];
                if constexpr (NPL == 2) {
                    uint16_t hi, lo;
                    split_f16(y, hi, lo);
                    vh[e] = (short)hi; vl[e] = (short)lo;
                } else {
                    vh[e] = (short)f32_to_bf16(y);
                }
            }
            buf[0].a[0][0][s] = vh;
            if constexpr (NPL == 2) buf[0].a[0][NPL - 1][s] = vl;
        }
        compute(buf[0]);
    } else {
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b < nblk) load(buf[b], b * 64);
    for (int c = 0; c < nblk; c += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (c + b < nblk) {
                compute(buf[b]);
                if (c + b + NB < nblk) load(buf[b], (c + b + NB) * 64);
            }
        }
    }
    }

    // ---- the NW partial tiles meet in LDS (accumulator element e of lane (r, h): row 8*(e>>2) + 4h + (e&3), column r)
    float* mine = red + (size_t)wid * TM * TN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = NPL == 2 ? accm[i][j][e] + accc[i][j][e] * (1.0f / 2048.0f) : accm[i][j][e];
                mine[(32 * i + 8 * (e >> 2) + 4 * h + (e & 3)) * TN + 32 * j + r] = v;
            }
    __syncthreads();
    if (Cf) Cf += (size_t)blockIdx.y * strideC;
    if (C16) C16 += (size_t)blockIdx.y * strideC;
    const float* Rsrc = Rf ? Rf + (size_t)blockIdx.y * strideC : Cf;
    const bool do_gelu = flags & G16X_GELU, do_acc = flags & G16X_ACCUM;
    const bool out16 = flags & G16X_OUT16, outf = flags & G16X_OUTF32;
#pragma unroll
    for (int idx4 = tid; idx4 < TM * C4; idx4 += NT) {
        const int rl = idx4 / C4, c4 = (idx4 % C4) * 4;
        float4 v = *reinterpret_cast<const float4*>(red + rl * TN + c4);
#pragma unroll
        for (int w = 1; w < NW; ++w) {  // wave order: the sum does not depend on timing
            const float4 o = *reinterpret_cast<const float4*>(red + (size_t)w * TM * TN + rl * TN + c4);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        const int row = m0 + rl, col = n0 + c4;
        if (row >= M || col >= N) continue;
        float4 bv = pre_b;
        if constexpr (!PRE) { if (bias) bv = *reinterpret_cast<const float4*>(bias + col); }
        v.x = v.x * alpha + bv.x; v.y = v.y * alpha + bv.y; v.z = v.z * alpha + bv.z; v.w = v.w * alpha + bv.w;
        if (do_gelu) {
            // (bf16 operands and no f32 output: the polynomial form of every bf16 kernel, common.hpp gelu16_2)
            const bool lowp = NPL == 1 && !outf;
            const genie_f2 g0 = lowp ? gelu16_2<true>(genie_f2{v.x, v.y}) : gelu_erf_fast2(genie_f2{v.x, v.y});
            const genie_f2 g1 = lowp ? gelu16_2<true>(genie_f2{v.z, v.w}) : gelu_erf_fast2(genie_f2{v.z, v.w});
            v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
        }
        const size_t idx = (size_t)row * ldc + col;
        if (do_acc) {
            float4 o = pre_r;
            if constexpr (!PRE) o = *reinterpret_cast<const float4*>(Rsrc + idx);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (outf) *reinterpret_cast<float4*>(Cf + idx) = v;
        if (out16) {
            if (flags & G16X_GELU16) {
                const genie_f2 g0 = gelu_erf_fast2(genie_f2{v.x, v.y}), g1 = gelu_erf_fast2(genie_f2{v.z, v.w});
                v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
            }
            if (NPL == 1 || plane16 == 0) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                *reinterpret_cast<uint2*>(C16 + idx) = pk;
            } else {
                uint32_t h01, h23, l01, l23;
                split_f16_x4(v.x, v.y, v.z, v.w, h01, h23, l01, l23);
                *reinterpret_cast<uint2*>(C16 + idx) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(C16 + (size_t)plane16 + idx) = make_uint2(l01, l23);
            }
        }
    }
}

// Returns GENIE_E_UNSUPPORTED when the problem is not "small" or does not fit the tiling; the caller (launch_gemm16 in
// kernels_bf16.hip) then takes a throughput kernel.  npl = 1: bf16 operands; 2: split-f16 planes.
int launch_gemm16_sm(int npl, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw, long planeW,
                     const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M, int N, int K,
                     int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC) {
    static const int on = study_env("GENIE_GEMM16_SM", 1);
    static const long max_out = study_env("GENIE_GEMM16_SM_MAX", (int)(1L << 20));
    if (!on || (long)M * N * batch > max_out) return GENIE_E_UNSUPPORTED;
    // long contractions (fc2, K = 2048) above 512 K outputs (2,048 rows x 512): the LDS-tiled 128x128 kernel is ahead -- every
    // workgroup here streams its operands from L2 itself, and at 32x32 tiles that is the bound (f16x3 63.4 -> 48.4 us, bf16
    // 25.4 -> 24.1; at 1,024 rows this kernel is 1.5-1.7x ahead, `profiles/r03_sm_threshold.txt`)
    if (K > 512 && (long)M * N * batch > (1L << 19)) return GENIE_E_UNSUPPORTED;
    if (N % 4 || ldc % 4 || lda % 8 || ldw % 8 || planeA % 8 || planeW % 8) return GENIE_E_UNSUPPORTED;
    if (npl == 2 && (flags & G16X_OUT16) && plane16 == 0) return GENIE_E_UNSUPPORTED;
    // 8 waves = 8 K-splits (every wave gets K/8 >= 64 k; at K = 512 the whole contraction is in flight at once); K = 256
    // (the 35M config's width) runs 4 waves of 64 k
    if (K % 512 && K != 256) return GENIE_E_UNSUPPORTED;
    const bool one = K == 512 || K == 256;   // one 64-k block per wave
    const double mn = (double)M * N * batch;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                   2.0 * npl * ((double)M * K * batch + (double)N * K) +
                       mn * ((flags & G16X_ACCUM ? 4 : 0) + (flags & G16X_OUTF32 ? 4 : 0) +
                             (flags & G16X_OUT16 ? 2 * npl : 0)),
                   st, "gemm16_sm_kernel (32x64 / 32x32 tile, in-workgroup split-K, no LDS ring)");
    // tile 32x64 unless that leaves most CUs idle (N = 512 at M = 256: 64 tiles) or the ring would not fit (K/8 > 64 with
    // split operands) -> 32x32
    const long t64 = (long)((M + 31) / 32) * ((N + 63) / 64) * batch;
    const int tn = (t64 >= 96 && !(npl == 2 && !one)) ? 64 : 32;
    const dim3 grid((unsigned)(((M + 31) / 32) * ((N + tn - 1) / tn)), (unsigned)batch);
#define SM_LAUNCH(NPL_, TM_, TN_, NW_, NB_)                                                                               \
    do {                                                                                                                  \
        const size_t lds = (size_t)NW_ * TM_ * TN_ * 4;                                                                   \
        (void)hipFuncSetAttribute((const void*)gemm16_sm_kernel<NPL_, TM_, TN_, NW_, NB_>,                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
        gemm16_sm_kernel<NPL_, TM_, TN_, NW_, NB_><<<grid, NW_ * 64, lds, st>>>(A, lda, planeA, W, ldw, planeW, bias, Cf, \
                                                                               C16, plane16, ldc, M, N, K, flags, alpha, \
                                                                               strideA, strideC, Rf, strideW);           \
    } while (0)
    // NB = blocks of 64 k in flight per wave: one when K/waves = 64, longer contractions ring two
#define SM_SHAPE(NPL_, TN_)                                                                                               \
    do {                                                                                                                  \
        if (K == 256) SM_LAUNCH(NPL_, 32, TN_, 4, 1);                                                                     \
        else if (K == 512) SM_LAUNCH(NPL_, 32, TN_, 8, 1);                                                                \
        else if constexpr (!(NPL_ == 2 && TN_ == 64)) SM_LAUNCH(NPL_, 32, TN_, 8, 2);                                     \
    } while (0)
    if (npl == 1) { if (tn == 64) SM_SHAPE(1, 64); else SM_SHAPE(1, 32); }
    else { if (tn == 64) SM_SHAPE(2, 64); else SM_SHAPE(2, 32); }
#undef SM_SHAPE
#undef SM_LAUNCH
    GENIE_LAUNCH_CHECK("gemm16_sm");
    return GENIE_OK;
}

// C = epilogue(alpha * LayerNorm(x) . W^T + bias) for small problems with K = 512 (8 waves) or 256 (4 waves): x (M, K) f32 with
// row stride ldx, LayerNorm over the whole row (eps, gamma, beta).  GENIE_E_UNSUPPORTED = take the separate LayerNorm + GEMM.
int launch_gemm16_sm_ln(int npl, const float* x, long ldx, const float* ln_g, const float* ln_b, float eps, const uint16_t* W,
                        long ldw, long planeW, const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc,
                        int M, int N, int K, int flags, float alpha, hipStream_t st) {
    static const int on = study_env("GENIE_GEMM16_SM_LN", 1);
    static const int sm_on = study_env("GENIE_GEMM16_SM", 1);
    static const long max_out = study_env("GENIE_GEMM16_SM_MAX", (int)(1L << 20));
    if (!on || !sm_on || (long)M * N > max_out || (K != 512 && K != 256)) return GENIE_E_UNSUPPORTED;
    if (N % 4 || ldc % 4 || ldx % 4 || ldw % 8 || planeW % 8) return GENIE_E_UNSUPPORTED;
    if (npl == 2 && (flags & G16X_OUT16) && plane16 == 0) return GENIE_E_UNSUPPORTED;
    const double mn = (double)M * N;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                   4.0 * (double)M * K + 2.0 * npl * (double)N * K +
                       mn * ((flags & G16X_ACCUM ? 4 : 0) + (flags & G16X_OUTF32 ? 4 : 0) + (flags & G16X_OUT16 ? 2 * npl : 0)),
                   st, "gemm16_sm_ln_kernel (LayerNorm in the fragment path, in-workgroup split-K)");
    const long t64 = (long)((M + 31) / 32) * ((N + 63) / 64);
    const int tn = t64 >= 96 ? 64 : 32;
    const dim3 grid((unsigned)(((M + 31) / 32) * ((N + tn - 1) / tn)), 1);
#define SMLN_LAUNCH(NPL_, TN_, NW_)                                                                                       \
    do {                                                                                                                  \
        const size_t lds = (size_t)NW_ * 32 * TN_ * 4 + 2 * NW_ * 32 * 4;                                                 \
        (void)hipFuncSetAttribute((const void*)gemm16_sm_kernel<NPL_, 32, TN_, NW_, 1, true>,                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
        gemm16_sm_kernel<NPL_, 32, TN_, NW_, 1, true><<<grid, NW_ * 64, lds, st>>>(                                       \
            nullptr, 0, 0, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, 0, 0, Rf, 0, x, ldx, ln_g, ln_b, \
            eps);                                                                                                         \
    } while (0)
#define SMLN_SHAPE(NPL_, TN_)                                                                                             \
    do { if (K == 512) SMLN_LAUNCH(NPL_, TN_, 8); else SMLN_LAUNCH(NPL_, TN_, 4); } while (0)
    if (npl == 1) { if (tn == 64) SMLN_SHAPE(1, 64); else SMLN_SHAPE(1, 32); }
    else { if (tn == 64) SMLN_SHAPE(2, 64); else SMLN_SHAPE(2, 32); }
#undef SMLN_SHAPE
#undef SMLN_LAUNCH
    GENIE_LAUNCH_CHECK("gemm16_sm_ln");
    return GENIE_OK;
}

}  // namespace genie
