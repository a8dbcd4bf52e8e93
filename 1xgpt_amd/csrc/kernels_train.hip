// Kernels of the training step: what autograd derives for STMaskGIT.forward
// (genie/st_mask_git.py:231-279) plus the optimizer of train.py:426-441, 628-633.  Everything here is f32 and serves all
// three precisions (LayerNorm / attention / loss / embedding backward, reductions, AdamW); the Linear products of the
// "exact" precision run on v_mfma_f32_32x32x2_f32 through ONE general GEMM that reads either operand in either orientation,
// so no activation or weight is ever physically transposed for a backward product (the 16-bit precisions take the NT GEMM
// of kernels_bf16.hip with the operand copies of kernels_train16.hip instead):
//     forward   Y  = X . W^T          A = X  [rows][k]      B = W  [cols][k]
//     dgrad     dX = dY . W           A = dY [rows][k]      B = W  [k][cols]   (TB)
//     wgrad     dW = dY^T . X         A = dY [k][rows] (TA) B = X  [k][cols]   (TB), split over the token axis
// All reductions that feed a gradient (split-K partials, LayerNorm / bias column sums, embedding scatter) are
// two-stage with a fixed summation order: a training step is bit-reproducible run to run.
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// C[b1][b2][M,N] = alpha * sum_k A(m,k) B(n,k) (+ bias[n]) (+ R[m,n])
//   TA: A(m,k) = A[k*lda + m] else A[m*lda + k];   TB: B(n,k) = W[k*ldw + n] else W[n*ldw + k]
//   blockIdx.y = b1, blockIdx.z = b2 * nsplit + split; split s contracts k in [s*K/nsplit, (s+1)*K/nsplit) and
//   writes its own slab C + s*sCsplit (no bias / residual there; splitk_reduce adds the slabs in order).
// 128x128x16 tile, 4 waves (2x2) x 64x64, double-buffered LDS.  Row-major operands use the k-permuted
// ds_read_b128 fetch of gemm_f32_nt_kernel (kernels_exact.hip); k-major operands are staged as they lie
// ([16][128] floats, coalesced float4) and fetched with one conflict-free ds_read_b32 per MFMA -- the f32 MFMA
// issues every 64 cycles per SIMD, so four narrow LDS reads per four MFMAs stay hidden.
// ------------------------------------------------------------------------------------------------
constexpr int GG_BM = 128, GG_BK = 16, GG_LD = GG_BK + 4, GG_TLD = GG_BM + 4;
constexpr int GG_TILE = GG_BM * GG_LD;  // 2560 floats >= 16 * 132

template <bool TR>
__device__ __forceinline__ void gg_stage_load(const float* __restrict__ G, long ld, int row0, int limit, int k0, int tid,
                                              float4 (&regs)[2]) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if constexpr (!TR) {
            const int row = row0 + (tid >> 2) + p * 64;
            regs[p] = row < limit ? *reinterpret_cast<const float4*>(G + (size_t)row * ld + k0 + ((tid & 3) << 2)) : z4;
        } else {
            const int kr = (tid >> 5) + p * 8, c4 = (tid & 31) << 2;
            regs[p] = (row0 + c4) < limit ? *reinterpret_cast<const float4*>(G + (size_t)(k0 + kr) * ld + row0 + c4) : z4;
        }
    }
}
template <bool TR>
__device__ __forceinline__ void gg_stage_store(float* s, int tid, const float4 (&regs)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if constexpr (!TR)
            *reinterpret_cast<float4*>(&s[((tid >> 2) + p * 64) * GG_LD + ((tid & 3) << 2)]) = regs[p];
        else
            *reinterpret_cast<float4*>(&s[((tid >> 5) + p * 8) * GG_TLD + ((tid & 31) << 2)]) = regs[p];
    }
}
// the four k-values (8kk + 4h + j, j = 0..3) of operand row `row` that MFMA j consumes
template <bool TR>
__device__ __forceinline__ float4 gg_frag(const float* s, int row, int kk, int h) {
    if constexpr (!TR) return *reinterpret_cast<const float4*>(&s[row * GG_LD + kk * 8 + 4 * h]);
    const float* p = s + (kk * 8 + 4 * h) * GG_TLD + row;
    return make_float4(p[0], p[GG_TLD], p[2 * GG_TLD], p[3 * GG_TLD]);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_gen_kernel(const float* __restrict__ A, long lda, long sA1, long sA2,
                                                           const float* __restrict__ W, long ldw, long sW1, long sW2,
                                                           const float* __restrict__ bias, const float* R,
                                                           float* C, long ldc, long sC1, long sC2, int M, int N, int K,
                                                           float alpha, int nsplit, long sCsplit) {
    __shared__ __attribute__((aligned(16))) float sA[2][GG_TILE];
    __shared__ __attribute__((aligned(16))) float sB[2][GG_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_tiles = (N + GG_BM - 1) / GG_BM;
    const int m0 = (blockIdx.x / n_tiles) * GG_BM, n0 = (blockIdx.x % n_tiles) * GG_BM;
    const int b2 = blockIdx.z / nsplit, sp = blockIdx.z - b2 * nsplit;
    A += (size_t)blockIdx.y * sA1 + (size_t)b2 * sA2;
    W += (size_t)blockIdx.y * sW1 + (size_t)b2 * sW2;
    const size_t coff = (size_t)blockIdx.y * sC1 + (size_t)b2 * sC2 + (size_t)sp * sCsplit;
    C += coff;
    if (R) R += coff;
    const int kc = K / nsplit, kbeg = sp * kc;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[2], rb[2];
    const int nk = kc / GG_BK;
    gg_stage_load<TA>(A, lda, m0, M, kbeg, tid, ra);
    gg_stage_load<TB>(W, ldw, n0, N, kbeg, tid, rb);
    gg_stage_store<TA>(sA[0], tid, ra);
    gg_stage_store<TB>(sB[0], tid, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            gg_stage_load<TA>(A, lda, m0, M, kbeg + (kt + 1) * GG_BK, tid, ra);
            gg_stage_load<TB>(W, ldw, n0, N, kbeg + (kt + 1) * GG_BK, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GG_BK / 8; ++kk) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = gg_frag<TA>(sA[buf], wm * 64 + i * 32 + r, kk, h);
                b[i] = gg_frag<TB>(sB[buf], wn * 64 + i * 32 + r, kk, h);
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
        if (kt + 1 < nk) {
            gg_stage_store<TA>(sA[buf ^ 1], tid, ra);
            gg_stage_store<TB>(sB[buf ^ 1], tid, rb);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
            if (col >= N) continue;
            const float bcol = bias ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                const size_t idx = (size_t)row * ldc + col;
                float v = acc[i][j][e] * alpha + bcol;
                if (R) v += R[idx];
                C[idx] = v;
            }
        }
}

int launch_gemm_f32_gen(bool ta, bool tb, const float* A, long lda, long sA1, long sA2, const float* W, long ldw, long sW1,
                        long sW2, const float* bias, const float* R, float* C, long ldc, long sC1, long sC2, int M, int N,
                        int K, int batch1, int batch2, int nsplit, long sCsplit, float alpha, hipStream_t st) {
    GENIE_CHECK_SHAPE(K > 0 && nsplit > 0 && K % (GG_BK * nsplit) == 0,
                      "gemm_gen: K=%d must be a positive multiple of %d", K, GG_BK * nsplit);
    GENIE_CHECK_SHAPE(lda % 4 == 0 && ldw % 4 == 0, "gemm_gen: leading dims must be multiples of 4 floats");
    GENIE_CHECK_SHAPE((!ta || M % 4 == 0) && (!tb || N % 4 == 0), "gemm_gen: k-major operands need row counts % 4 == 0");
    if (M <= 0 || N <= 0 || batch1 <= 0 || batch2 <= 0) return GENIE_OK;
    const int mt = (M + GG_BM - 1) / GG_BM, nt = (N + GG_BM - 1) / GG_BM;
    dim3 grid(mt * nt, batch1, batch2 * nsplit);
    const double mn = (double)M * N * batch1 * batch2;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                   4.0 * (((double)M * K + (double)N * K) * batch1 * batch2 + mn * nsplit * (R ? 2 : 1)), st);
#define GG_LAUNCH(TA_, TB_)                                                                                          \
    gemm_f32_gen_kernel<TA_, TB_><<<grid, 256, 0, st>>>(A, lda, sA1, sA2, W, ldw, sW1, sW2, bias, R, C, ldc, sC1, sC2, \
                                                        M, N, K, alpha, nsplit, sCsplit)
    if (!ta && !tb) GG_LAUNCH(false, false);
    else if (!ta && tb) GG_LAUNCH(false, true);
    else if (ta && tb) GG_LAUNCH(true, true);
    else GG_LAUNCH(true, false);
#undef GG_LAUNCH
    GENIE_LAUNCH_CHECK("gemm_f32_gen");
    return GENIE_OK;
}

// out[i] = beta * out[i] + sum_{s < ns} part[s*n + i].  Fixed order: thread group g (0..3) of a block sums slabs
// g, g+4, g+8, ... of its 64 columns with 4 independent chains, then the 4 group sums are added 0+1+2+3.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, int ns, size_t n,
                                                          float* __restrict__ out, float beta) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int k = g;
        for (; k + 12 < ns; k += 16) {
            s0 += part[(size_t)k * n + i];
            s1 += part[(size_t)(k + 4) * n + i];
            s2 += part[(size_t)(k + 8) * n + i];
            s3 += part[(size_t)(k + 12) * n + i];
        }
        for (; k < ns; k += 4) s0 += part[(size_t)k * n + i];
    }
    red[g][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) {
        const float s = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
        out[i] = (beta != 0.f ? beta * out[i] : 0.f) + s;
    }
}
int launch_slab_reduce(const float* part, int ns, size_t n, float* out, float beta, hipStream_t st) {
    if (!n) return GENIE_OK;
    slab_reduce_kernel<<<(unsigned)((n + 63) / 64), 256, 0, st>>>(part, ns, n, out, beta);
    GENIE_LAUNCH_CHECK("slab_reduce");
    return GENIE_OK;
}

// dW[N,K] (beta*dW +)= alpha * dY[M,N]^T . X[M,K], contraction over the M tokens split into slabs
int launch_wgrad_f32(const float* dY, long ldy, const float* X, long ldx, float* dW, int Mtok, int N, int K, float alpha,
                     float beta, float* slabs, size_t slab_floats, hipStream_t st) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    int ns = 1;
    while (ns < 64 && tiles * ns < 512 && Mtok % (GG_BK * ns * 2) == 0 && (size_t)(ns * 2) * N * K <= slab_floats) ns *= 2;
    if (ns == 1)
        return launch_gemm_f32_gen(true, true, dY, ldy, 0, 0, X, ldx, 0, 0, nullptr, beta != 0.f ? dW : nullptr, dW, K, 0, 0,
                                   N, K, Mtok, 1, 1, 1, 0, alpha, st);
    GENIE_CHECK_ARG(beta == 0.f || beta == 1.f, "wgrad: beta must be 0 or 1");
    GENIE_TRY(launch_gemm_f32_gen(true, true, dY, ldy, 0, 0, X, ldx, 0, 0, nullptr, nullptr, slabs, K, 0, 0, N, K, Mtok, 1,
                                  1, ns, (long)N * K, alpha, st));
    return launch_slab_reduce(slabs, ns, (size_t)N * K, dW, beta, st);
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): part[chunk][N] = sum over the chunk's rows, then slab_reduce
// ------------------------------------------------------------------------------------------------
__global__ void colsum_partial_kernel(const float* __restrict__ Y, long ld, long rows, int N, float* __restrict__ part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += Y[(size_t)r * ld + c];
    part[(size_t)blockIdx.y * N + c] = s;
}
// the same restricted to the rows whose id equals `key` (gradient of the mask-token embedding)
__global__ void colsum_where_partial_kernel(const float* __restrict__ Y, long ld, long rows, int N,
                                            const int64_t* __restrict__ ids, int64_t key, float* __restrict__ part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    for (long r = r0; r < r1; ++r)
        if (ids[r] == key) s += Y[(size_t)r * ld + c];
    part[(size_t)blockIdx.y * N + c] = s;
}
int launch_colsum_where(const float* Y, long ld, long rows, int N, const int64_t* ids, int64_t key, float* out, float beta,
                        float* part, hipStream_t st) {
    if (rows <= 0 || N <= 0) return GENIE_OK;
    colsum_where_partial_kernel<<<dim3((N + 255) / 256, COLSUM_CHUNKS), 256, 0, st>>>(Y, ld, rows, N, ids, key, part);
    GENIE_LAUNCH_CHECK("colsum_where");
    return launch_slab_reduce(part, COLSUM_CHUNKS, (size_t)N, out, beta, st);
}
int launch_colsum(const float* Y, long ld, long rows, int N, float* out, float beta, float* part, hipStream_t st) {
    if (rows <= 0 || N <= 0) return GENIE_OK;
    colsum_partial_kernel<<<dim3((N + 255) / 256, COLSUM_CHUNKS), 256, 0, st>>>(Y, ld, rows, N, part);
    GENIE_LAUNCH_CHECK("colsum");
    return launch_slab_reduce(part, COLSUM_CHUNKS, (size_t)N, out, beta, st);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (nn.LayerNorm(C, eps), biased variance): one wave per row, statistics recomputed from x.
//   dxout[row] += (gy - mean(gy) - xhat * mean(gy * xhat)) * rstd,  gy = dy * gamma
//   part[block][0][C] = sum_rows dy * xhat, part[block][1][C] = sum_rows dy   (then slab_reduce -> dgamma, dbeta)
// ------------------------------------------------------------------------------------------------
constexpr int LNB_MAXI = 16;  // C <= 1024
constexpr int LNB_BLOCKS = 512;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ dxout,
                                                     float* __restrict__ part, long rows, int C, float eps) {
    extern __shared__ float red[];  // [4][2][C]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float dg[LNB_MAXI], db[LNB_MAXI], gm[LNB_MAXI];
#pragma unroll
    for (int i = 0; i < LNB_MAXI; ++i) {
        dg[i] = 0.f; db[i] = 0.f;
        const int c = lane + 64 * i;
        gm[i] = c < C ? gamma[c] : 0.f;
    }
    const float invC = 1.0f / (float)C;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + (size_t)row * C;
        const float* dr = dy + (size_t)row * C;
        float xv[LNB_MAXI], dv[LNB_MAXI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXI; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < C ? xr[c] : 0.f;
            dv[i] = c < C ? dr[c] : 0.f;
            s += xv[i];
        }
        const float mean = wave_sum(s) * invC;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXI; ++i) {
            const int c = lane + 64 * i;
            const float xc = c < C ? xv[i] - mean : 0.f;
            xv[i] = xc;
            v += xc * xc;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) * invC + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXI; ++i) {
            xv[i] *= rstd;  // xhat
            const float gy = dv[i] * gm[i];
            s1 += gy;
            s2 += gy * xv[i];
            dg[i] += dv[i] * xv[i];
            db[i] += dv[i];
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
        float* o = dxout + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < LNB_MAXI; ++i) {
            const int c = lane + 64 * i;
            if (c < C) o[c] += (dv[i] * gm[i] - s1 - xv[i] * s2) * rstd;
        }
    }
#pragma unroll
    for (int i = 0; i < LNB_MAXI; ++i) {
        const int c = lane + 64 * i;
        if (c < C) { red[(wid * 2 + 0) * C + c] = dg[i]; red[(wid * 2 + 1) * C + c] = db[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, cc = c - which * C;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * C + cc];
        part[(size_t)blockIdx.x * 2 * C + c] = s;  // [block][which][C]
    }
}
// The same for C = 256 / 512 with the access pattern of layer_norm_fast_kernel: a lane owns VPL = C/64 contiguous
// channels (float4 loads / read-modify-write of dxout instead of 4-byte strided ones), the next row's x and dy are in
// flight while the current row is reduced.
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_fast_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ dy, float* __restrict__ dxout,
                                                          float* __restrict__ part, long rows, float eps) {
    constexpr int C = 64 * VPL;
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float dg[VPL], db[VPL], gm[VPL], xv[VPL], dv[VPL], xn[VPL], dn[VPL];
    auto load = [&](const float* p, float (&v)[VPL]) {
#pragma unroll
        for (int k = 0; k < VPL; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + lane * VPL + k);
            v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
        }
    };
    load(gamma, gm);
#pragma unroll
    for (int k = 0; k < VPL; ++k) { dg[k] = 0.f; db[k] = 0.f; }
    const long stride = (long)gridDim.x * 4;
    long row = (long)blockIdx.x * 4 + wid;
    if (row < rows) { load(x + (size_t)row * C, xv); load(dy + (size_t)row * C, dv); }
    for (; row < rows; row += stride) {
        const long nrow = row + stride;
        if (nrow < rows) { load(x + (size_t)nrow * C, xn); load(dy + (size_t)nrow * C, dn); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) s += xv[k];
        const float mean = wave_sum(s) * (1.0f / C);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) { xv[k] -= mean; v += xv[k] * xv[k]; }
        const float rstd = 1.0f / sqrtf(wave_sum(v) * (1.0f / C) + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            xv[k] *= rstd;  // xhat
            const float gy = dv[k] * gm[k];
            s1 += gy;
            s2 += gy * xv[k];
            dg[k] += dv[k] * xv[k];
            db[k] += dv[k];
        }
        s1 = wave_sum(s1) * (1.0f / C);
        s2 = wave_sum(s2) * (1.0f / C);
        float* o = dxout + (size_t)row * C + lane * VPL;
#pragma unroll
        for (int k = 0; k < VPL; k += 4) {
            float4 t = *reinterpret_cast<const float4*>(o + k);
            t.x += (dv[k] * gm[k] - s1 - xv[k] * s2) * rstd;
            t.y += (dv[k + 1] * gm[k + 1] - s1 - xv[k + 1] * s2) * rstd;
            t.z += (dv[k + 2] * gm[k + 2] - s1 - xv[k + 2] * s2) * rstd;
            t.w += (dv[k + 3] * gm[k + 3] - s1 - xv[k + 3] * s2) * rstd;
            *reinterpret_cast<float4*>(o + k) = t;
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) { xv[k] = xn[k]; dv[k] = dn[k]; }
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k) { red[wid][0][lane * VPL + k] = dg[k]; red[wid][1][lane * VPL + k] = db[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, cc = c - which * C;
        part[(size_t)blockIdx.x * 2 * C + c] = ((red[0][which][cc] + red[1][which][cc]) + red[2][which][cc]) + red[3][which][cc];
    }
}

// part slabs are [block][2][C]: dgamma = reduce of the first C, dbeta of the second C
int launch_ln_bwd(const float* x, const float* gamma, const float* dy, float* dxout, float* dgamma, float* dbeta,
                  long rows, int C, float eps, float beta, float* part, hipStream_t st) {
    GENIE_CHECK_SHAPE(C <= 64 * LNB_MAXI, "ln_bwd: C=%d > %d", C, 64 * LNB_MAXI);
    if (rows <= 0) return GENIE_OK;
    ProfScope prof(GENIE_KC_LAYERNORM, 12.0 * rows * C, 16.0 * rows * C, st);
    if (C == 512) ln_bwd_fast_kernel<8><<<LNB_BLOCKS, 256, 0, st>>>(x, gamma, dy, dxout, part, rows, eps);
    else if (C == 256) ln_bwd_fast_kernel<4><<<LNB_BLOCKS, 256, 0, st>>>(x, gamma, dy, dxout, part, rows, eps);
    else ln_bwd_kernel<<<LNB_BLOCKS, 256, (size_t)8 * C * sizeof(float), st>>>(x, gamma, dy, dxout, part, rows, C, eps);
    GENIE_LAUNCH_CHECK("ln_bwd");
    // the two halves of each slab are contiguous ([2][C]), so one reduce of 2C columns serves both when dbeta follows
    // dgamma in memory; they need not, so reduce separately through strided views
    float* tmp = part + (size_t)LNB_BLOCKS * 2 * C;  // [2][C] reduced
    GENIE_TRY(launch_slab_reduce(part, LNB_BLOCKS, (size_t)2 * C, tmp, 0.f, st));
    GENIE_TRY(launch_slab_reduce(tmp, 1, (size_t)C, dgamma, beta, st));
    return launch_slab_reduce(tmp + C, 1, (size_t)C, dbeta, beta, st);
}
size_t ln_bwd_scratch_floats(int C) { return (size_t)(LNB_BLOCKS + 1) * 2 * C; }

// ------------------------------------------------------------------------------------------------
// erf-GELU forward (saving the pre-activation) and backward: dz = dh * (Phi(z) + z * phi(z))
// ------------------------------------------------------------------------------------------------
__global__ void gelu_fwd_kernel(const float4* __restrict__ z, float4* __restrict__ h, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = z[i];
    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
    h[i] = v;
}
__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * z * z) * 0.39894228040143267794f;
    return cdf + z * pdf;
}
__global__ void gelu_bwd_kernel(const float4* __restrict__ z, float4* __restrict__ g, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 zz = z[i];
    float4 v = g[i];
    v.x *= gelu_grad(zz.x); v.y *= gelu_grad(zz.y); v.z *= gelu_grad(zz.z); v.w *= gelu_grad(zz.w);
    g[i] = v;
}
int launch_gelu_fwd(const float* z, float* h, size_t n, hipStream_t st) {
    GENIE_CHECK_SHAPE(n % 4 == 0, "gelu: n %% 4");
    if (!n) return GENIE_OK;
    gelu_fwd_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>((const float4*)z, (float4*)h, n / 4);
    GENIE_LAUNCH_CHECK("gelu_fwd");
    return GENIE_OK;
}
int launch_gelu_bwd(const float* z, float* g, size_t n, hipStream_t st) {
    GENIE_CHECK_SHAPE(n % 4 == 0, "gelu: n %% 4");
    if (!n) return GENIE_OK;
    gelu_bwd_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>((const float4*)z, (float4*)g, n / 4);
    GENIE_LAUNCH_CHECK("gelu_bwd");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// row softmax (in place) and its backward dS = P * (dP - sum_j P dP) (in place on dP); one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ P, long rows, int N) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = P + (size_t)row * N;
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, p[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < N; j += 64) { const float e = expf(p[j] - m); p[j] = e; s += e; }
    const float inv = 1.0f / wave_sum(s);
    for (int j = lane; j < N; j += 64) p[j] *= inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                               long rows, int N) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + (size_t)row * N;
    float* g = dP + (size_t)row * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += p[j] * g[j];
    s = wave_sum(s);
    for (int j = lane; j < N; j += 64) g[j] = p[j] * (g[j] - s);
}
int launch_softmax_rows(float* P, long rows, int N, hipStream_t st) {
    if (rows <= 0) return GENIE_OK;
    softmax_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(P, rows, N);
    GENIE_LAUNCH_CHECK("softmax_rows");
    return GENIE_OK;
}
int launch_softmax_bwd_rows(const float* P, float* dP, long rows, int N, hipStream_t st) {
    if (rows <= 0) return GENIE_OK;
    softmax_bwd_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(P, dP, rows, N);
    GENIE_LAUNCH_CHECK("softmax_bwd_rows");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Temporal (causal, N = T <= 16 frames) attention backward on the frame-strided rows of (B,T,S,3d) -- the
// "(B S) T C" view of st_transformer.py:77 without a transpose.  One thread per (sequence, head, frame i);
// a block holds GPB (sequence, head) groups of 16 threads; q (pre-scaled), k, v, dO rows sit in LDS.
//   p_ij = softmax_j(scale q_i.k_j), j <= i;  dp_ij = dO_i.v_j;  ds_ij = p_ij (dp_ij - sum_j p_ij dp_ij)
//   dq_i = scale sum_j ds_ij k_j;  dk_j = scale sum_{i>=j} ds_ij q_i;  dv_j = sum_{i>=j} p_ij dO_i
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(64) void attn_temporal_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ qk,
                                                               long qk_ld, const float* __restrict__ dO,
                                                               float* __restrict__ dqkv, int B, int T, int S, int d, int H,
                                                               float scale) {
    constexpr int GPB = 4, NMAX = 16, DHP = DH + 4;  // padded rows: threads of a group read different rows, same column
    extern __shared__ float sm[];
    const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
    float* sq = sm + (size_t)g * (4 * NMAX * DHP + 2 * NMAX * NMAX);
    float* sk = sq + NMAX * DHP;
    float* sv = sk + NMAX * DHP;
    float* sd = sv + NMAX * DHP;
    float* sp = sd + NMAX * DHP;  // [i][j]
    float* sds = sp + NMAX * NMAX;
    const long grp = (long)blockIdx.x * GPB + g;  // (b, s, head)
    const long n_grp = (long)B * S * H;
    const bool active = grp < n_grp && i < T;
    long row = 0;
    int head = 0;
    if (active) {
        head = (int)(grp % H);
        const long bs = grp / H;
        const long b = bs / S, s = bs - b * S;
        row = (b * T + i) * (long)S + s;
        const float* base = qkv + (size_t)row * 3 * d + head * DH;
        const float* qb = qk + (size_t)row * qk_ld + head * DH;  // q at column 0, k at column d of the (normalised) source
        const float* dob = dO + (size_t)row * d + head * DH;
#pragma unroll 4
        for (int c = 0; c < DH; c += 4) {
            float4 q4 = *reinterpret_cast<const float4*>(qb + c);
            q4.x *= scale; q4.y *= scale; q4.z *= scale; q4.w *= scale;
            *reinterpret_cast<float4*>(sq + i * DHP + c) = q4;
            *reinterpret_cast<float4*>(sk + i * DHP + c) = *reinterpret_cast<const float4*>(qb + d + c);
            *reinterpret_cast<float4*>(sv + i * DHP + c) = *reinterpret_cast<const float4*>(base + 2 * d + c);
            *reinterpret_cast<float4*>(sd + i * DHP + c) = *reinterpret_cast<const float4*>(dob + c);
        }
    }
    __syncthreads();
    float acc[DH];
    if (active) {
        float p[NMAX], dp[NMAX];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            float s = 0.f, t = 0.f;
            if (j <= i) {
#pragma unroll 8
                for (int c = 0; c < DH; ++c) {
                    s = fmaf(sq[i * DHP + c], sk[j * DHP + c], s);
                    t = fmaf(sd[i * DHP + c], sv[j * DHP + c], t);
                }
                m = fmaxf(m, s);
            }
            p[j] = s; dp[j] = t;
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            p[j] = j <= i ? expf(p[j] - m) : 0.f;
            l += p[j];
        }
        const float inv = 1.0f / l;
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) { p[j] *= inv; dsum += p[j] * dp[j]; }
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            const float ds = p[j] * (dp[j] - dsum);  // p[j] = 0 beyond the causal bound
            sp[i * NMAX + j] = p[j];
            sds[i * NMAX + j] = ds;
            if (j <= i) {
#pragma unroll 8
                for (int c = 0; c < DH; ++c) acc[c] = fmaf(ds, sk[j * DHP + c], acc[c]);
            }
        }
        float* o = dqkv + (size_t)row * 3 * d + head * DH;
#pragma unroll 4
        for (int c = 0; c < DH; c += 4)
            *reinterpret_cast<float4*>(o + c) =
                make_float4(acc[c] * scale, acc[c + 1] * scale, acc[c + 2] * scale, acc[c + 3] * scale);
    }
    __syncthreads();
    if (active) {
        const int j = i;
        float* o = dqkv + (size_t)row * 3 * d + head * DH;
        // dk_j: sq holds scale*q, so sum_i ds_ij * sq_i is already the scaled product
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] = 0.f;
        for (int ii = j; ii < T; ++ii) {
            const float ds = sds[ii * NMAX + j];
#pragma unroll 8
            for (int c = 0; c < DH; ++c) acc[c] = fmaf(ds, sq[ii * DHP + c], acc[c]);
        }
#pragma unroll 4
        for (int c = 0; c < DH; c += 4)
            *reinterpret_cast<float4*>(o + d + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] = 0.f;
        for (int ii = j; ii < T; ++ii) {
            const float pp = sp[ii * NMAX + j];
#pragma unroll 8
            for (int c = 0; c < DH; ++c) acc[c] = fmaf(pp, sd[ii * DHP + c], acc[c]);
        }
#pragma unroll 4
        for (int c = 0; c < DH; c += 4)
            *reinterpret_cast<float4*>(o + 2 * d + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    }
}
// ------------------------------------------------------------------------------------------------
// The same backward for the production geometry (T = 16 frames, head_dim 32 / 64) on v_mfma_f32_16x16x4_f32: one
// WAVE per (b, s, head), no LDS.  The 16x16 MFMA D tile puts D[4g+e][r] in lane (r = lane&15, g = lane>>4), so a
// 16x16 score matrix is available in two register layouts, chosen by operand order:
//     T-layout  mfma(k, q)  -> lane (r,g) holds X[i = r][j = 4g+e]   (row i across the 4 lane groups; the forward's)
//     N-layout  mfma(q, k)  -> lane (r,g) holds X[i = 4g+e][j = r]
// and a tile held in T-layout is directly the A operand (row r, contraction index 4g+e) of a product contracted over
// j, one held in N-layout of a product contracted over i.  So:  P, dP, dS are computed in BOTH layouts (4 x DH/4
// MFMAs); dQ = scale dS K uses the T copy, dK = scale dS^T Q and dV = P^T dO the N copy; row statistics (max, 1/sum,
// D_i = sum_j P dP) are reduced once in the T layout and fetched by lane index for the N layout.
// Feature permutation of the forward kernel: MFMA output column r of chunk c is feature NV*r + c (NV = DH/16), so
// every global access is NV contiguous floats per lane.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DH>
__global__ __launch_bounds__(256) void attn_temporal_bwd_mfma_kernel(const float* __restrict__ qkv,
                                                                     const float* __restrict__ qk, long qk_ld,
                                                                     const float* __restrict__ dO,
                                                                     float* __restrict__ dqkv, long n_bs, int S, int d,
                                                                     int H, float scale) {
    constexpr int T = 16, PER = DH / 4, NV = DH / 16;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long bs = wave / H;
    const int head = (int)(wave - bs * H);
    if (bs >= n_bs) return;
    const long b = bs / S, s = bs - b * S;
    const size_t row0 = (size_t)(b * T) * S + s;  // token row of frame 0; frame t is row0 + t*S
    const float* qb = qk + row0 * qk_ld + head * DH;          // q at column 0, k at column d
    const float* vb = qkv + row0 * 3 * d + 2 * d + head * DH;
    const float* ob = dO + row0 * d + head * DH;
    const long qs = (long)S * qk_ld, vs = (long)S * 3 * d, os = (long)S * d;
    // ---- row operands: lane (r,g) holds features g*PER .. of row r
    float q[PER], k[PER], v[PER], go[PER];
#pragma unroll
    for (int c = 0; c < PER; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(qb + (size_t)r * qs + g * PER + c);
        const float4 bb = *reinterpret_cast<const float4*>(qb + (size_t)r * qs + d + g * PER + c);
        const float4 cc = *reinterpret_cast<const float4*>(vb + (size_t)r * vs + g * PER + c);
        const float4 dd = *reinterpret_cast<const float4*>(ob + (size_t)r * os + g * PER + c);
        q[c] = a.x * scale; q[c + 1] = a.y * scale; q[c + 2] = a.z * scale; q[c + 3] = a.w * scale;
        k[c] = bb.x; k[c + 1] = bb.y; k[c + 2] = bb.z; k[c + 3] = bb.w;
        v[c] = cc.x; v[c + 1] = cc.y; v[c + 2] = cc.z; v[c + 3] = cc.w;
        go[c] = dd.x; go[c + 1] = dd.y; go[c + 2] = dd.z; go[c + 3] = dd.w;
    }
    f32x4 sT = {0.f, 0.f, 0.f, 0.f}, pT = {0.f, 0.f, 0.f, 0.f}, sN = {0.f, 0.f, 0.f, 0.f}, pN = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < PER; ++c) {
        sT = __builtin_amdgcn_mfma_f32_16x16x4f32(k[c], q[c], sT, 0, 0, 0);   // S[i=r][j=4g+e]
        pT = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c], go[c], pT, 0, 0, 0);  // dP[i=r][j=4g+e]
        sN = __builtin_amdgcn_mfma_f32_16x16x4f32(q[c], k[c], sN, 0, 0, 0);   // S[i=4g+e][j=r]
        pN = __builtin_amdgcn_mfma_f32_16x16x4f32(go[c], v[c], pN, 0, 0, 0);  // dP[i=4g+e][j=r]
    }
    // ---- softmax statistics of row i = r in the T layout (causal: j <= i)
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * g + e > r) sT[e] = -INFINITY;
        mx = fmaxf(mx, sT[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sT[e] = expf(sT[e] - mx); sum += sT[e]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    float dsum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sT[e] *= inv; dsum += sT[e] * pT[e]; }
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    f32x4 dsT, dsN, prN;
#pragma unroll
    for (int e = 0; e < 4; ++e) dsT[e] = sT[e] * (pT[e] - dsum);  // masked entries: P = 0
    // ---- the same in the N layout: statistics of row 4g+e come from lane 4g+e (any group holds them)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        const float mi = __shfl(mx, i), ii = __shfl(inv, i), di = __shfl(dsum, i);
        const float p = r <= i ? expf(sN[e] - mi) * ii : 0.f;
        prN[e] = p;
        dsN[e] = p * (pN[e] - di);
    }
    // ---- dQ = scale dS K,  dK = scale dS^T Q,  dV = P^T dO: B operands are rows 4g+e, features NV*r .. NV*r+NV-1
    float kk[4][NV], qq[4][NV], dd[4][NV];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t t = (size_t)(4 * g + e);
        if constexpr (NV == 4) {
            const float4 a = *reinterpret_cast<const float4*>(qb + t * qs + d + NV * r);
            const float4 bb = *reinterpret_cast<const float4*>(qb + t * qs + NV * r);
            const float4 cc = *reinterpret_cast<const float4*>(ob + t * os + NV * r);
            kk[e][0] = a.x; kk[e][1] = a.y; kk[e][2] = a.z; kk[e][3] = a.w;
            qq[e][0] = bb.x; qq[e][1] = bb.y; qq[e][2] = bb.z; qq[e][3] = bb.w;
            dd[e][0] = cc.x; dd[e][1] = cc.y; dd[e][2] = cc.z; dd[e][3] = cc.w;
        } else {
            const float2 a = *reinterpret_cast<const float2*>(qb + t * qs + d + NV * r);
            const float2 bb = *reinterpret_cast<const float2*>(qb + t * qs + NV * r);
            const float2 cc = *reinterpret_cast<const float2*>(ob + t * os + NV * r);
            kk[e][0] = a.x; kk[e][1] = a.y;
            qq[e][0] = bb.x; qq[e][1] = bb.y;
            dd[e][0] = cc.x; dd[e][1] = cc.y;
        }
    }
    f32x4 dq[NV], dk[NV], dv[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        dk[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dq[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsT[e], kk[e][c], dq[c], 0, 0, 0);
            dk[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(dsN[e], qq[e][c], dk[c], 0, 0, 0);
            dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(prN[e], dd[e][c], dv[c], 0, 0, 0);
        }
    }
    // D map: lane (r,g) holds rows 4g+e, features NV*r + c
    float* outb = dqkv + row0 * 3 * d + head * DH + NV * r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float* o = outb + (size_t)(4 * g + e) * vs;
        if constexpr (NV == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(dq[0][e] * scale, dq[1][e] * scale, dq[2][e] * scale, dq[3][e] * scale);
            *reinterpret_cast<float4*>(o + d) = make_float4(dk[0][e] * scale, dk[1][e] * scale, dk[2][e] * scale, dk[3][e] * scale);
            *reinterpret_cast<float4*>(o + 2 * d) = make_float4(dv[0][e], dv[1][e], dv[2][e], dv[3][e]);
        } else {
            *reinterpret_cast<float2*>(o) = make_float2(dq[0][e] * scale, dq[1][e] * scale);
            *reinterpret_cast<float2*>(o + d) = make_float2(dk[0][e] * scale, dk[1][e] * scale);
            *reinterpret_cast<float2*>(o + 2 * d) = make_float2(dv[0][e], dv[1][e]);
        }
    }
}

int launch_attn_temporal_bwd(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, int B, int T,
                             int S, int d, int H, int Dh, float scale, hipStream_t st) {
    GENIE_CHECK_SHAPE(T <= 16, "temporal attention backward: T=%d > 16", T);
    const long n_grp = (long)B * S * H;
    if (n_grp <= 0) return GENIE_OK;
    if (T == 16 && (Dh == 64 || Dh == 32)) {
        ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 10.0 * n_grp * T * T * Dh, 4.0 * n_grp * T * Dh * 7, st);
        const unsigned wblocks = (unsigned)((n_grp + 3) / 4);
        if (Dh == 64)
            attn_temporal_bwd_mfma_kernel<64><<<wblocks, 256, 0, st>>>(qkv, qk, qk_ld, dO, dqkv, (long)B * S, S, d, H, scale);
        else
            attn_temporal_bwd_mfma_kernel<32><<<wblocks, 256, 0, st>>>(qkv, qk, qk_ld, dO, dqkv, (long)B * S, S, d, H, scale);
        GENIE_LAUNCH_CHECK("attn_temporal_bwd_mfma");
        return GENIE_OK;
    }
    const unsigned blocks = (unsigned)((n_grp + 3) / 4);
    ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 10.0 * n_grp * T * T * Dh, 4.0 * n_grp * T * Dh * 7, st);
#define TB_LAUNCH(DH_)                                                                                       \
    {                                                                                                        \
        const size_t lds = (size_t)4 * (4 * 16 * (DH_ + 4) + 2 * 16 * 16) * sizeof(float);                        \
        (void)hipFuncSetAttribute((const void*)attn_temporal_bwd_kernel<DH_>,                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        attn_temporal_bwd_kernel<DH_><<<blocks, 64, lds, st>>>(qkv, qk, qk_ld, dO, dqkv, B, T, S, d, H, scale); \
    }
    if (Dh == 64) TB_LAUNCH(64)
    else if (Dh == 32) TB_LAUNCH(32)
    else if (Dh == 128) TB_LAUNCH(128)
    else {
        set_error("temporal attention backward: head_dim %d not in {32, 64, 128}", Dh);
        return GENIE_E_UNSUPPORTED;
    }
#undef TB_LAUNCH
    GENIE_LAUNCH_CHECK("attn_temporal_bwd");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// Spatial attention backward, fused (S = 256 tokens of one frame, head_dim 32 / 64, non-causal): one workgroup of 4
// waves per (b, t, head); K and V of the head stay in LDS (2 x 68 KB at head_dim 64, rows padded to DH+4 floats so
// the k-permuted ds_read_b128 fragments are conflict-free), queries are streamed in blocks of 32 rows.  Wave w owns
// keys [64w, 64w+64).  Per query block, on v_mfma_f32_32x32x2_f32:
//   [two passes over the query blocks -- pass A: T layout, statistics and dQ; pass B: N layout, dK and dV -- so that only
//   one layout's tiles are live at a time (the single-pass form needed all 512 registers and spilled)]
//   S = scale Q K^T and dP = dO V^T in BOTH register layouts (the same operand fragments, swapped): T-layout = one
//     query row per lane (softmax statistics: in-lane + one cross-half shuffle + a 3-value exchange between the 4 waves,
//     combined like an online softmax), N-layout = one key column per lane;
//   dS = P (dP - D), D_i = sum_j P dP;   dQ_i += dS K (A = dS in T-layout, partial over the wave's 64 keys, reduced over
//     the 4 waves through LDS in a FIXED order);   dK += dS^T Q_i and dV += P^T dO_i (A = N-layout tiles), accumulated in
//     registers over the 8 query blocks and written once.
// Nothing of size S x S ever touches HBM: 7 d floats per token in, 3 d out.  ~450 VGPRs at one wave per SIMD.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rowmap32(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_spatial_bwd_fused_kernel(const float* __restrict__ qkv,
                                                                        const float* __restrict__ qk, long qk_ld,
                                                                        const float* __restrict__ dO,
                                                                        float* __restrict__ dqkv, int d, int H,
                                                                        float scale) {
    constexpr int S = 256, LD = DH + 4, IB = 32, NF = DH / 32, NKK = DH / 8, NPF = IB * DH / 4 / 256;
    static_assert(NPF >= 1, "head_dim >= 32");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sK = sm;
    float* sV = sK + S * LD;
    float* sQ = sV + S * LD;
    float* sdO = sQ + IB * LD;
    float* sPM = sdO + IB * LD;  // per-wave partial statistics [4][32]
    float* sPL = sPM + 128;
    float* sPD = sPL + 128;
    float* stM = sPD + 128;      // final row max (log2 units) / 1/sum / D for all 256 query rows (pass A -> pass B)
    float* stI = stM + S;
    float* stD = stI + S;
    float* red = sQ;             // dQ reduction buffer [4][16][DH], aliases sQ | sdO
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long bt = blockIdx.x / H;
    const int head = (int)(blockIdx.x - bt * H);
    const size_t row0 = (size_t)bt * S;
    const float* qb = qk + row0 * qk_ld + head * DH;
    const float* kb = qb + d;
    const float* vb = qkv + row0 * 3 * d + 2 * d + head * DH;
    const float* ob = dO + row0 * d + head * DH;
    float* outb = dqkv + row0 * 3 * d + head * DH;
    // scores are carried as s*log2(e) so that every exponential is one v_exp_f32 (the probabilities are the same)
    const float scale_l2 = scale * 1.4426950408889634f;

    for (int idx = tid; idx < S * DH / 4; idx += 256) {
        const int row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
        *reinterpret_cast<float4*>(&sK[row * LD + c4]) = *reinterpret_cast<const float4*>(kb + (size_t)row * qk_ld + c4);
        *reinterpret_cast<float4*>(&sV[row * LD + c4]) = *reinterpret_cast<const float4*>(vb + (size_t)row * 3 * d + c4);
    }
    float4 pq[NPF], pd[NPF];
    auto fetch = [&](int ib) {
#pragma unroll
        for (int p = 0; p < NPF; ++p) {
            const int idx = tid + 256 * p, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            pq[p] = *reinterpret_cast<const float4*>(qb + (size_t)(ib * IB + row) * qk_ld + c4);
            pd[p] = *reinterpret_cast<const float4*>(ob + (size_t)(ib * IB + row) * d + c4);
        }
    };
    auto publish = [&]() {  // prefetched Q_i, dO_i -> LDS
#pragma unroll
        for (int p = 0; p < NPF; ++p) {
            const int idx = tid + 256 * p, row = idx / (DH / 4), c4 = (idx % (DH / 4)) * 4;
            *reinterpret_cast<float4*>(&sQ[row * LD + c4]) = pq[p];
            *reinterpret_cast<float4*>(&sdO[row * LD + c4]) = pd[p];
        }
    };

    // ================= pass A: T layout (one query row per lane): row statistics and dQ =================
    fetch(0);
    for (int ib = 0; ib < S / IB; ++ib) {
        publish();
        __syncthreads();  // Q_i, dO_i (and K, V on the first pass) are in LDS
        fetch(ib + 1 < S / IB ? ib + 1 : 0);  // next block; after the last one: block 0 again for pass B
        f32x16 sT[2], pT[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sT[jt][e] = 0.f; pT[jt][e] = 0.f; }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int jrow = w * 64 + jt * 32 + c;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const float4 kf = *reinterpret_cast<const float4*>(&sK[jrow * LD + kk * 8 + 4 * h]);
                const float4 vf = *reinterpret_cast<const float4*>(&sV[jrow * LD + kk * 8 + 4 * h]);
                const float4 qf = *reinterpret_cast<const float4*>(&sQ[c * LD + kk * 8 + 4 * h]);
                const float4 of = *reinterpret_cast<const float4*>(&sdO[c * LD + kk * 8 + 4 * h]);
#define SP_STEP(X)                                                              \
    sT[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.X, qf.X, sT[jt], 0, 0, 0); \
    pT[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.X, of.X, pT[jt], 0, 0, 0);
                SP_STEP(x) SP_STEP(y) SP_STEP(z) SP_STEP(w)
#undef SP_STEP
            }
        }
        // lane (c, h) holds S[i = c][j = 64w + 32jt + rowmap(e, h)]: statistics of row c over this wave's keys
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sT[jt][e] *= scale_l2; m = fmaxf(m, sT[jt][e]); }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f, ds = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(sT[jt][e] - m);
                sT[jt][e] = p;
                l += p;
                ds += p * pT[jt][e];
            }
        l += __shfl_xor(l, 32);
        ds += __shfl_xor(ds, 32);
        if (h == 0) { sPM[w * 32 + c] = m; sPL[w * 32 + c] = l; sPD[w * 32 + c] = ds; }
        __syncthreads();
        float gm = sPM[c];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) gm = fmaxf(gm, sPM[ww * 32 + c]);
        float gl = 0.f, gd = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float f = __builtin_amdgcn_exp2f(sPM[ww * 32 + c] - gm);
            gl += sPL[ww * 32 + c] * f;
            gd += sPD[ww * 32 + c] * f;
        }
        const float inv = 1.0f / gl, Di = gd * inv;
        if (w == 0 && h == 0) { stM[ib * IB + c] = gm; stI[ib * IB + c] = inv; stD[ib * IB + c] = Di; }
        const float corr = __builtin_amdgcn_exp2f(m - gm) * inv;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) sT[jt][e] = sT[jt][e] * corr * (pT[jt][e] - Di);  // dS
        // dQ partial over this wave's 64 keys: A = dS (row c, contraction index = key rowmap(e, h))
        f32x16 dq[NF];
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[ft][e] = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float* kr = &sK[(w * 64 + jt * 32 + rowmap32(e, h)) * LD + c];
#pragma unroll
                for (int ft = 0; ft < NF; ++ft)
                    dq[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[jt][e], kr[32 * ft], dq[ft], 0, 0, 0);
            }
        __syncthreads();  // every wave is done with sQ / sdO and the partial statistics
#pragma unroll
        for (int round = 0; round < 2; ++round) {
#pragma unroll
            for (int ft = 0; ft < NF; ++ft)
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = 8 * round + e8;
                    red[(w * 16 + rowmap32(e, h) - 16 * round) * DH + 32 * ft + c] = dq[ft][e];
                }
            __syncthreads();
            if (tid < 16 * DH / 4) {
                const int row = tid / (DH / 4), f4 = (tid % (DH / 4)) * 4;
                float4 a = *reinterpret_cast<const float4*>(&red[row * DH + f4]);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) {
                    const float4 b = *reinterpret_cast<const float4*>(&red[(ww * 16 + row) * DH + f4]);
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
                a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
                *reinterpret_cast<float4*>(outb + (size_t)(ib * IB + 16 * round + row) * 3 * d + f4) = a;
            }
            __syncthreads();
        }
    }

    // ================= pass B: N layout (one key column per lane): dK and dV =================
    f32x16 dk[2][NF], dv[2][NF];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[jt][ft][e] = 0.f; dv[jt][ft][e] = 0.f; }
    for (int ib = 0; ib < S / IB; ++ib) {
        publish();
        __syncthreads();
        if (ib + 1 < S / IB) fetch(ib + 1);
        f32x16 sN[2], pN[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { sN[jt][e] = 0.f; pN[jt][e] = 0.f; }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int jrow = w * 64 + jt * 32 + c;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const float4 kf = *reinterpret_cast<const float4*>(&sK[jrow * LD + kk * 8 + 4 * h]);
                const float4 vf = *reinterpret_cast<const float4*>(&sV[jrow * LD + kk * 8 + 4 * h]);
                const float4 qf = *reinterpret_cast<const float4*>(&sQ[c * LD + kk * 8 + 4 * h]);
                const float4 of = *reinterpret_cast<const float4*>(&sdO[c * LD + kk * 8 + 4 * h]);
#define SP_STEP(X)                                                              \
    sN[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf.X, kf.X, sN[jt], 0, 0, 0); \
    pN[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(of.X, vf.X, pN[jt], 0, 0, 0);
                SP_STEP(x) SP_STEP(y) SP_STEP(z) SP_STEP(w)
#undef SP_STEP
            }
        }
        // lane (c, h) holds X[i = rowmap(e, h)][j = 64w + 32jt + c]; the statistics of row i were published by pass A
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = ib * IB + rowmap32(e, h);
            const float mi = stM[i], ii = stI[i], di = stD[i];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sN[jt][e], scale_l2, -mi)) * ii;
                sN[jt][e] = p;                        // P
                pN[jt][e] = p * (pN[jt][e] - di);     // dS
            }
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = rowmap32(e, h);
                const float* qr = &sQ[rr * LD + c];
                const float* orow = &sdO[rr * LD + c];
#pragma unroll
                for (int ft = 0; ft < NF; ++ft) {
                    dk[jt][ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(pN[jt][e], qr[32 * ft], dk[jt][ft], 0, 0, 0);
                    dv[jt][ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(sN[jt][e], orow[32 * ft], dv[jt][ft], 0, 0, 0);
                }
            }
        __syncthreads();  // sQ / sdO are rewritten by the next block
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* o = outb + (size_t)(w * 64 + jt * 32 + rowmap32(e, h)) * 3 * d + 32 * ft + c;
                o[d] = dk[jt][ft][e] * scale;
                o[2 * d] = dv[jt][ft][e];
            }
}

// GENIE_E_UNSUPPORTED for other geometries (the caller then takes the materialised-scores path)
int launch_attn_spatial_bwd_fused(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, long n_bt, int S,
                                  int d, int H, int Dh, float scale, hipStream_t st) {
    if (S != 256 || (Dh != 64 && Dh != 32)) return GENIE_E_UNSUPPORTED;
    if (n_bt <= 0) return GENIE_OK;
    const size_t lds = (size_t)((2 * 256 + 2 * 32) * (Dh + 4) + 12 * 32 + 3 * 256) * sizeof(float);
    ProfScope prof(GENIE_KC_ATTN_SPATIAL, 14.0 * S * S * Dh * (double)n_bt * H, 4.0 * 10 * S * Dh * (double)n_bt * H, st);
    if (Dh == 64) {
        (void)hipFuncSetAttribute((const void*)attn_spatial_bwd_fused_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attn_spatial_bwd_fused_kernel<64><<<(unsigned)(n_bt * H), 256, lds, st>>>(qkv, qk, qk_ld, dO, dqkv, d, H, scale);
    } else {
        (void)hipFuncSetAttribute((const void*)attn_spatial_bwd_fused_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attn_spatial_bwd_fused_kernel<32><<<(unsigned)(n_bt * H), 256, lds, st>>>(qkv, qk, qk_ld, dO, dqkv, d, H, scale);
    }
    GENIE_LAUNCH_CHECK("attn_spatial_bwd_fused");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// qk-norm (attention.py:42-47): LayerNorm over head_dim of every q and k head row, ONE affine shared by q, k and all
// heads.  Forward writes the normalised operands qkn (M, 2d) = [LN(q) | LN(k)] for the score GEMMs of the backward;
// backward turns d/d(normalised) into d/d(raw) in place on the q,k columns of dqkv and emits the affine's gradient
// partials.  DH/4 lanes per (row, q|k, head) item, float4 each.
// ------------------------------------------------------------------------------------------------
template <int LPI>
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
    for (int o = LPI / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int DH>
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ qkn,
                                                          const float* __restrict__ nw, const float* __restrict__ nb,
                                                          long n_items, int H, int d) {
    constexpr int LPI = DH / 4, IPB = 256 / LPI;
    const long item = (long)blockIdx.x * IPB + threadIdx.x / LPI;
    const int sub = threadIdx.x % LPI;
    if (item >= n_items) return;
    const long row = item / (2 * H);
    const int rem = (int)(item - row * 2 * H), which = rem / H, head = rem - which * H;
    const float4 v = *reinterpret_cast<const float4*>(qkv + (size_t)row * 3 * d + which * d + head * DH + sub * 4);
    const float mean = sub_sum<LPI>(v.x + v.y + v.z + v.w) * (1.0f / DH);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
    const float rstd = 1.0f / sqrtf(sub_sum<LPI>(a * a + b * b + c * c + e * e) * (1.0f / DH) + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(nw + sub * 4), bb = *reinterpret_cast<const float4*>(nb + sub * 4);
    *reinterpret_cast<float4*>(qkn + (size_t)row * 2 * d + which * d + head * DH + sub * 4) =
        make_float4(a * rstd * g.x + bb.x, b * rstd * g.y + bb.y, c * rstd * g.z + bb.z, e * rstd * g.w + bb.w);
}
constexpr int QKN_BLOCKS = 512;
template <int DH>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const float* __restrict__ qkv, float* __restrict__ dqkv,
                                                          const float* __restrict__ nw, float* __restrict__ part,
                                                          long n_items, int H, int d) {
    constexpr int LPI = DH / 4, IPB = 256 / LPI;
    __shared__ float red[256][8];
    const int sub = threadIdx.x % LPI;
    const float4 g = *reinterpret_cast<const float4*>(nw + sub * 4);
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    for (long item = (long)blockIdx.x * IPB + threadIdx.x / LPI; item < n_items; item += (long)gridDim.x * IPB) {
        const long row = item / (2 * H);
        const int rem = (int)(item - row * 2 * H), which = rem / H, head = rem - which * H;
        const size_t off = (size_t)row * 3 * d + which * d + head * DH + sub * 4;
        const float4 v = *reinterpret_cast<const float4*>(qkv + off);
        const float4 dy = *reinterpret_cast<const float4*>(dqkv + off);
        const float mean = sub_sum<LPI>(v.x + v.y + v.z + v.w) * (1.0f / DH);
        float xh[4] = {v.x - mean, v.y - mean, v.z - mean, v.w - mean};
        const float rstd =
            1.0f / sqrtf(sub_sum<LPI>(xh[0] * xh[0] + xh[1] * xh[1] + xh[2] * xh[2] + xh[3] * xh[3]) * (1.0f / DH) + 1e-5f);
        const float dyv[4] = {dy.x, dy.y, dy.z, dy.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float gy[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xh[k] *= rstd;
            gy[k] = dyv[k] * gv[k];
            s1 += gy[k];
            s2 += gy[k] * xh[k];
            dg[k] += dyv[k] * xh[k];
            db[k] += dyv[k];
        }
        s1 = sub_sum<LPI>(s1) * (1.0f / DH);
        s2 = sub_sum<LPI>(s2) * (1.0f / DH);
        *reinterpret_cast<float4*>(dqkv + off) = make_float4((gy[0] - s1 - xh[0] * s2) * rstd, (gy[1] - s1 - xh[1] * s2) * rstd,
                                                            (gy[2] - s1 - xh[2] * s2) * rstd, (gy[3] - s1 - xh[3] * s2) * rstd);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[threadIdx.x][k] = dg[k]; red[threadIdx.x][4 + k] = db[k]; }
    __syncthreads();
    if (threadIdx.x < 2 * DH) {  // part[block][which][DH]: channel c is held by the threads with sub == c/4
        const int which = threadIdx.x / DH, c = threadIdx.x - which * DH;
        float s = 0.f;
        for (int t = c / 4; t < 256; t += LPI) s += red[t][which * 4 + (c & 3)];
        part[(size_t)blockIdx.x * 2 * DH + threadIdx.x] = s;
    }
}
int launch_qk_norm_fwd(const float* qkv, float* qkn, const float* nw, const float* nb, long M, int H, int Dh, int d,
                       hipStream_t st) {
    const long n_items = M * 2 * H;
    if (n_items <= 0) return GENIE_OK;
    const unsigned blocks = (unsigned)((n_items + (1024 / Dh) - 1) / (1024 / Dh));
    if (Dh == 64) qk_norm_fwd_kernel<64><<<blocks, 256, 0, st>>>(qkv, qkn, nw, nb, n_items, H, d);
    else if (Dh == 32) qk_norm_fwd_kernel<32><<<blocks, 256, 0, st>>>(qkv, qkn, nw, nb, n_items, H, d);
    else if (Dh == 128) qk_norm_fwd_kernel<128><<<blocks, 256, 0, st>>>(qkv, qkn, nw, nb, n_items, H, d);
    else { set_error("qk-norm: head_dim %d not in {32, 64, 128}", Dh); return GENIE_E_UNSUPPORTED; }
    GENIE_LAUNCH_CHECK("qk_norm_fwd");
    return GENIE_OK;
}
// in place on the q,k columns of dqkv; dnw/dnb (Dh) = beta * old + sums over rows, q and k, and heads
int launch_qk_norm_bwd(const float* qkv, float* dqkv, const float* nw, float* dnw, float* dnb, long M, int H, int Dh,
                       int d, float beta, float* part, hipStream_t st) {
    const long n_items = M * 2 * H;
    if (n_items <= 0) return GENIE_OK;
    if (Dh == 64) qk_norm_bwd_kernel<64><<<QKN_BLOCKS, 256, 0, st>>>(qkv, dqkv, nw, part, n_items, H, d);
    else if (Dh == 32) qk_norm_bwd_kernel<32><<<QKN_BLOCKS, 256, 0, st>>>(qkv, dqkv, nw, part, n_items, H, d);
    else if (Dh == 128) qk_norm_bwd_kernel<128><<<QKN_BLOCKS, 256, 0, st>>>(qkv, dqkv, nw, part, n_items, H, d);
    else { set_error("qk-norm: head_dim %d not in {32, 64, 128}", Dh); return GENIE_E_UNSUPPORTED; }
    GENIE_LAUNCH_CHECK("qk_norm_bwd");
    float* tmp = part + (size_t)QKN_BLOCKS * 2 * Dh;
    GENIE_TRY(launch_slab_reduce(part, QKN_BLOCKS, (size_t)2 * Dh, tmp, 0.f, st));
    GENIE_TRY(launch_slab_reduce(tmp, 1, (size_t)Dh, dnw, beta, st));
    return launch_slab_reduce(tmp + Dh, 1, (size_t)Dh, dnb, beta, st);
}

// ------------------------------------------------------------------------------------------------
// masked factored cross-entropy, forward and backward in one pass (st_mask_git.py:231-253, 276):
//   counted(token) = frame >= 1 and input id == MASK;  n = number of counted tokens (count_masked_kernel)
//   sums += [sum ce, sum all-factors-correct, (n is written by the count kernel)]
//   logits row (V = nfac*vf) is REPLACED by d loss / d logits = counted ? (softmax_f - onehot_f) / n : 0
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void count_masked_kernel(const int64_t* __restrict__ ids, long B, int T, int S,
                                                            int64_t mask_id, double* __restrict__ sums) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    const long n = B * T * (long)S;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const int t = (int)((i / S) % T);
        c += (t >= 1 && ids[i] == mask_id) ? 1ull : 0ull;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < 16; ++w) s += red[w];
        sums[2] = (double)s;
    }
}
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(float* __restrict__ logits, const int64_t* __restrict__ ids,
                                                         const int64_t* __restrict__ labels, long n_tok, int T, int S,
                                                         int vf, int nfac, int64_t mask_id, double* __restrict__ sums) {
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long n = (long)blockIdx.x * 4 + wid;
    double ce = 0, hit = 0;
    if (n < n_tok) {
        float* lp = logits + (size_t)n * vf * nfac;
        const int t = (int)((n / S) % T);
        const bool counted = t >= 1 && ids[n] == mask_id;
        if (!counted) {
            for (int k = lane; k < vf * nfac; k += 64) lp[k] = 0.f;
        } else {
            const float wgt = (float)(1.0 / sums[2]);
            int64_t tgt = labels[n];
            float loss = 0.f;
            bool all_ok = true;
            for (int f = 0; f < nfac; ++f) {
                const int tf = (int)(tgt % vf);
                tgt /= vf;
                float* lf = lp + f * vf;
                float mx = -INFINITY;
                int mi = 0;
                for (int k = lane; k < vf; k += 64) {
                    const float v = lf[k];
                    if (v > mx) { mx = v; mi = k; }
                }
                wave_argmax(mx, mi);
                float se = 0.f;
                for (int k = lane; k < vf; k += 64) se += expf(lf[k] - mx);
                se = wave_sum(se);
                loss += logf(se) + mx - lf[tf];
                all_ok = all_ok && (mi == tf);
                const float inv = 1.0f / se;
                for (int k = lane; k < vf; k += 64) {
                    const float p = expf(lf[k] - mx) * inv;
                    lf[k] = (p - (k == tf ? 1.0f : 0.0f)) * wgt;
                }
            }
            if (lane == 0) { ce = loss; hit = all_ok ? 1.0 : 0.0; }
        }
    }
    if (lane == 0) { red[0][wid] = ce; red[1][wid] = hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int w = 0; w < 4; ++w) { a += red[0][w]; b += red[1][w]; }
        if (a != 0.0 || b != 0.0) { atomicAdd(&sums[0], a); atomicAdd(&sums[1], b); }
    }
}
int launch_ce_fwd_bwd(const genie_cfg& c, float* logits, const int64_t* ids, const int64_t* labels, int B, double* sums,
                      hipStream_t st) {
    const long n_tok = (long)B * c.T * c.S;
    if (n_tok <= 0) return GENIE_OK;
    count_masked_kernel<<<1, 1024, 0, st>>>(ids, B, c.T, c.S, (int64_t)c.image_vocab_size, sums);
    GENIE_LAUNCH_CHECK("count_masked");
    ce_fwd_bwd_kernel<<<(unsigned)((n_tok + 3) / 4), 256, 0, st>>>(logits, ids, labels, n_tok, c.T, c.S, c.factored_vocab,
                                                                  c.num_factored, (int64_t)c.image_vocab_size, sums);
    GENIE_LAUNCH_CHECK("ce_fwd_bwd");
    return GENIE_OK;
}

// ------------------------------------------------------------------------------------------------
// embedding backward (factorization_utils.py:29-52, st_mask_git.py:257-261), scatter-free and ordered:
//   dpos[t,s,:]  = sum_b dx[b,t,s,:]
//   row v of table f (block f*vf + v): sum over tokens n (ascending) with factor_f(id_n) == v and id_n != MASK
//   mask row: a predicated column sum over the tokens with id_n == MASK (they are a large fraction of a training
//   batch, so they get the chunked two-stage reduction instead of one block)
// ------------------------------------------------------------------------------------------------
struct EmbedTables { float* p[4]; };
__global__ void embed_bwd_pos_kernel(const float* __restrict__ dx, float* __restrict__ dpos, int B, size_t per_clip,
                                     float beta) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_clip) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[(size_t)b * per_clip + i];
    dpos[i] = (beta != 0.f ? beta * dpos[i] : 0.f) + s;
}
__global__ __launch_bounds__(256) void embed_bwd_tables_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                               long n_tok, int d, int vf, int nfac, int64_t mask_id,
                                                               EmbedTables tables, float* __restrict__ dmask,
                                                               float beta) {
    // One block per table row (and one for the mask row).  The block sweeps the ids 1024 at a time: every thread tests
    // 4 tokens, wave ballots give an ORDERED match list (token index ascending), and the (rare: 1/vf) matching rows
    // of dx are summed channel-parallel in that order -- no atomics, no sort, bit-reproducible.
    __shared__ unsigned long long masks[4][4];  // [k][wave]: tokens base + k*256 + wave*64 + bit
    const int blk = blockIdx.x;
    constexpr bool is_mask_row = false;
    const int f = blk / vf, v = blk - f * vf;
    int64_t div = 1;
    for (int k = 0; k < f; ++k) div *= vf;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // channels threadIdx.x + 256*k, d <= 1024
    for (long base = 0; base < n_tok; base += 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long i = base + k * 256 + threadIdx.x;
            bool m = false;
            if (i < n_tok) {
                const int64_t id = ids[i];
                m = is_mask_row ? (id == mask_id) : (id != mask_id && (int)((id / div) % vf) == v);
            }
            const unsigned long long bal = __ballot(m);
            if (lane == 0) masks[k][wid] = bal;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned long long bits = masks[k][w];
                while (bits) {
                    const int bit = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    const float* r = dx + (size_t)(base + k * 256 + w * 64 + bit) * d;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int c = threadIdx.x + 256 * c4;
                        if (c < d) acc[c4] += r[c];
                    }
                }
            }
        __syncthreads();
    }
    float* out = is_mask_row ? dmask : tables.p[f] + (size_t)v * d;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < d) out[c] = (beta != 0.f ? beta * out[c] : 0.f) + acc[k];
    }
}
int launch_embed_bwd(const genie_cfg& c, const float* dx, const int64_t* ids, int B, float* dpos, float* dmask,
                     float* const* tables_host, float beta, float* colpart, hipStream_t st) {
    EmbedTables tables_dev;
    for (int j = 0; j < 4; ++j) tables_dev.p[j] = j < c.num_factored ? tables_host[j] : nullptr;
    GENIE_CHECK_SHAPE(c.d_model <= 1024, "embed backward: d_model > 1024");
    const size_t per_clip = (size_t)c.T * c.S * c.d_model;
    embed_bwd_pos_kernel<<<(unsigned)((per_clip + 255) / 256), 256, 0, st>>>(dx, dpos, B, per_clip, beta);
    GENIE_LAUNCH_CHECK("embed_bwd_pos");
    embed_bwd_tables_kernel<<<c.num_factored * c.factored_vocab, 256, 0, st>>>(
        dx, ids, (long)B * c.T * c.S, c.d_model, c.factored_vocab, c.num_factored, (int64_t)c.image_vocab_size, tables_dev,
        dmask, beta);
    GENIE_LAUNCH_CHECK("embed_bwd_tables");
    return launch_colsum_where(dx, c.d_model, (long)B * c.T * c.S, c.d_model, ids, (int64_t)c.image_vocab_size, dmask, beta,
                               colpart, st);
}

// ------------------------------------------------------------------------------------------------
// optimizer: sum of squares (two-stage, f64) and torch.optim.AdamW's update with clip_grad_norm_ folded in
// ------------------------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = x[i];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += part[i];
        *out += s;
    }
}
int launch_sumsq(const float* x, size_t n, double* out, double* scratch, hipStream_t st) {
    if (!n) return GENIE_OK;
    sumsq_partial_kernel<<<SUMSQ_BLOCKS, 256, 0, st>>>(x, n, scratch);
    GENIE_LAUNCH_CHECK("sumsq");
    sumsq_final_kernel<<<1, 64, 0, st>>>(scratch, SUMSQ_BLOCKS, out);
    GENIE_LAUNCH_CHECK("sumsq_final");
    return GENIE_OK;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps, float decay,
                             float step_size, float inv_sqrt_bc2, float grad_mult, const double* __restrict__ sumsq,
                             float max_norm) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float coef = grad_mult;
    if (sumsq && max_norm > 0.f) {  // clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6))
        const float tn = (float)sqrt(*sumsq) * grad_mult;
        coef *= fminf(1.0f, max_norm / (tn + 1e-6f));
    }
    const float gi = g[i] * coef;
    float pi = p[i] * decay;  // decoupled weight decay: p *= 1 - lr*wd
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
}
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_mult, const double* sumsq, float max_norm, hipStream_t st) {
    if (!n) return GENIE_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    adamw_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, g, m, v, n, lr, beta1, beta2, eps,
                                                             (float)(1.0 - (double)lr * weight_decay), (float)(lr / bc1),
                                                             (float)(1.0 / sqrt(bc2)), grad_mult, sumsq, max_norm);
    GENIE_LAUNCH_CHECK("adamw");
    return GENIE_OK;
}

}  // namespace genie
