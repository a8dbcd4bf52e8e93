// 16-bit matrix-core path of the GENIE forward for gfx950.
//
// gemm16_nt_kernel<NPL, BK>:  C[M,N] (+)= epilogue( alpha * A[M,K] . W[N,K]^T + bias ) on v_mfma_f32_32x32x16
//   NPL = 1  operands are bf16 (one product)                     -> GENIE_PREC_BF16, the throughput mode
//   NPL = 2  operands are f16 split pairs  a = hi + lo * 2^-11   -> 3 products hi.hi + (hi.lo + lo.hi) 2^-11,
//            22-bit effective mantissa: f32-class results at 1/3 of the f16 MFMA rate (5x the f32 MFMA peak)
// 128x128 block tile, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles, K-tile BK, LDS double-buffered and
// filled by global_load_lds (16 B per lane, HBM -> LDS without a register round trip).  The LDS image of
// a tile is lane-linear (that is what global_load_lds writes), so bank conflicts of the 16-byte fragment
// reads are removed by an XOR swizzle applied on the SOURCE address (which 16-byte slot of its row a lane
// fetches) and mirrored on the fragment read:  phys_slot = slot ^ ((row / rows_per_256B) % slots_per_row).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { G16_GELU = 1, G16_ACCUM = 2, G16_OUT16 = 4, G16_OUTF32 = 8,
       G16_GELU16 = 16, /* erf-GELU on the 16-bit output only: Cf keeps the pre-activation (training forward) */
       G16_NT = 32,     /* non-temporal output stores: the output is larger than the on-die caches (launcher) */
       G16_WIDEW = 1024 /* launcher only (never reaches a kernel): the f16x3 weight's hi plane reaches |w| >= 32 -> not the
                           2^11-scaling single-accumulator kernel (the `w16_wide` flags of genie_hip.h) */ };

constexpr float SPLIT_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void store_u2(uint16_t* p, uint2 v, bool nt) {
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    if (nt) {
        u2v t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<u2v*>(p));
    } else {
        *reinterpret_cast<uint2*>(p) = v;
    }
}

template <int NPL>
__device__ __forceinline__ void store16(uint16_t* base, size_t plane_stride, size_t idx, float v) {
    if constexpr (NPL == 1) {
        base[idx] = f32_to_bf16(v);
    } else {
        uint16_t hi, lo;
        split_f16(v, hi, lo);
        base[idx] = hi;
        base[plane_stride + idx] = lo;
    }
}

template <int NPL, int BK>
__global__ __launch_bounds__(256, 2) void gemm16_nt_kernel(const uint16_t* __restrict__ A, long lda, long planeA,
                                                        const uint16_t* __restrict__ W, long ldw, long planeW,
                                                        const float* __restrict__ bias, float* __restrict__ Cf,
                                                        uint16_t* __restrict__ C16, long plane16, long ldc, int M,
                                                        int N, int K, int flags, float alpha, long strideA,
                                                        long strideC, const float* Rf, long strideW) {
    constexpr int BM = 128, BN = 128;
    constexpr int ROWB = BK * 2;              // bytes per tile row
    constexpr int SPR = ROWB / 16;            // 16-byte slots per row
    constexpr int RPB = 256 / ROWB;           // rows per 256-byte LDS bank row
    constexpr int TILE_B = BM * ROWB;         // bytes of one operand plane tile
    constexpr int CHUNK_ROWS = 1024 / ROWB;   // rows moved by one wave-wide global_load_lds
    constexpr int NCHUNK = BM / CHUNK_ROWS;   // chunks per plane tile (split over the 4 waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: [buf][operand A/W][plane][TILE_B]
    auto tile_ptr = [&](int buf, int op, int pl) { return smem + ((buf * 2 + op) * NPL + pl) * TILE_B; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int n_tiles = (N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_tiles) * BM, n0 = (blockIdx.x % n_tiles) * BN;
    A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;
    if (Cf) Cf += (size_t)blockIdx.y * strideC;
    if (C16) C16 += (size_t)blockIdx.y * strideC;
    // residual source of G16_ACCUM: Cf itself (in place) unless the caller keeps the input (training: Rf != Cf)
    const float* Rsrc = Rf ? Rf + (size_t)blockIdx.y * strideC : Cf;

    // ---- staging addresses: wave w moves chunks w*NCHUNK/4 .. of every plane tile
    constexpr int CPW = NCHUNK / 4;  // chunks per wave per plane tile
    const int c_row = lane / SPR, c_phys = lane % SPR;
    const uint16_t* gsrcA[CPW];
    const uint16_t* gsrcW[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int row_local = (wid * CPW + c) * CHUNK_ROWS + c_row;
        const int slot = c_phys ^ ((row_local / RPB) % SPR);
        int ra = m0 + row_local; ra = ra < M ? ra : M - 1;   // clamp: rows past the edge are never stored
        int rw = n0 + row_local; rw = rw < N ? rw : N - 1;
        gsrcA[c] = A + (size_t)ra * lda + slot * 8;
        gsrcW[c] = W + (size_t)rw * ldw + slot * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int off = (wid * CPW + c) * 1024;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(gsrcA[c] + (size_t)pl * planeA + k0),
                    (__attribute__((address_space(3))) void*)(tile_ptr(buf, 0, pl) + off), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(gsrcW[c] + (size_t)pl * planeW + k0),
                    (__attribute__((address_space(3))) void*)(tile_ptr(buf, 1, pl) + off), 16, 0, 0);
            }
    };

    f32x16 acc[2][2], corr[NPL == 2 ? 2 : 1][NPL == 2 ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (NPL == 2) corr[i][j][e] = 0.f;
            }

    // fragment read offsets (bytes) within a plane tile for k-step kk: lane (r,h) reads slot 2*kk + h of its row
    int fragA[2], fragB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fragA[i] = (wm * 64 + i * 32 + r);
        fragB[i] = (wn * 64 + i * 32 + r);
    }
    auto frag_off = [&](int row_local, int kk) {
        const int slot = 2 * kk + h;
        return row_local * ROWB + ((slot ^ ((row_local / RPB) % SPR)) << 4);
    };

    const int nk = K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if constexpr (NPL == 1) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const bf16x8*>(tile_ptr(buf, 0, 0) + frag_off(fragA[i], kk));
                    b[i] = *reinterpret_cast<const bf16x8*>(tile_ptr(buf, 1, 0) + frag_off(fragB[i], kk));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(tile_ptr(buf, 0, 0) + frag_off(fragA[i], kk));
                    al[i] = *reinterpret_cast<const f16x8*>(tile_ptr(buf, 0, 1) + frag_off(fragA[i], kk));
                    bh[i] = *reinterpret_cast<const f16x8*>(tile_ptr(buf, 1, 0) + frag_off(fragB[i], kk));
                    bl[i] = *reinterpret_cast<const f16x8*>(tile_ptr(buf, 1, 1) + frag_off(fragB[i], kk));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], corr[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], corr[i][j], 0, 0, 0);
            }
        }
        __syncthreads();  // drains this wave's global_load_lds (vmcnt) and orders the buffer swap
    }

    // epilogue through LDS (see gemm16_v2_kernel): 4 waves x 16 KB = the 64 KB ring
    float* ct = reinterpret_cast<float*>(smem) + wid * (64 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e];
                if constexpr (NPL == 2) v += corr[i][j][e] * SPLIT_INV;
                ct[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r] = v;
            }
    const bool do_gelu = flags & G16_GELU, do_acc = flags & G16_ACCUM;
    const bool out16 = flags & G16_OUT16, outf = flags & G16_OUTF32;
    const bool vec = ((N | (int)ldc) & 3) == 0;
    const int c4 = (lane & 15) << 2;
    const int col = n0 + wn * 64 + c4;
    typedef float ef4 __attribute__((ext_vector_type(4)));
    const bool fast = vec && col + 3 < N;
    ef4 bv4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && fast) bv4 = *reinterpret_cast<const ef4*>(bias + col);   // read once, not per row
    auto row_of = [&](int it) { return m0 + wm * 64 + it * 4 + (lane >> 4); };
    auto value_of = [&](int it) {
        const ef4 cv = *reinterpret_cast<const ef4*>(ct + (it * 4 + (lane >> 4)) * 64 + c4);
        float4 v = make_float4(cv.x * alpha + bv4.x, cv.y * alpha + bv4.y, cv.z * alpha + bv4.z, cv.w * alpha + bv4.w);
        if (do_gelu) {
            // (bf16 operands and no f32 output: the value leaves only as bf16 -> the polynomial form, as in every bf16 kernel)
            if (NPL == 1 && !outf) {
                const genie_f2 g0 = gelu16_2<true>(genie_f2{v.x, v.y}), g1 = gelu16_2<true>(genie_f2{v.z, v.w});
                v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
            } else {
                v.x = gelu_erf_fast(v.x); v.y = gelu_erf_fast(v.y); v.z = gelu_erf_fast(v.z); v.w = gelu_erf_fast(v.w);
            }
        }
        return v;
    };
    auto store_row = [&](int row, float4 v) {
        const size_t idx = (size_t)row * ldc + col;
        if (outf) {
            if (flags & G16_NT) {
                ef4 t = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(t, reinterpret_cast<ef4*>(Cf + idx));
            } else {
                *reinterpret_cast<float4*>(Cf + idx) = v;
            }
        }
        if (out16) {
            if (flags & G16_GELU16) { v.x = gelu_erf_fast(v.x); v.y = gelu_erf_fast(v.y); v.z = gelu_erf_fast(v.z); v.w = gelu_erf_fast(v.w); }
            store16<NPL>(C16, (size_t)plane16, idx, v.x); store16<NPL>(C16, (size_t)plane16, idx + 1, v.y);
            store16<NPL>(C16, (size_t)plane16, idx + 2, v.z); store16<NPL>(C16, (size_t)plane16, idx + 3, v.w);
        }
    };
    if (fast && do_acc) {
        // residual accumulate: groups of four wave-instructions in three straight sections, the next group's residual rows
        // requested before this group's stores (see gemm16_v2_kernel's epilogue)
        ef4 rsd[2][4];
        auto load_group = [&](int g, ef4* dst) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = row_of(g * 4 + u);
                dst[u] = *reinterpret_cast<const ef4*>(Rsrc + (size_t)(row < M ? row : M - 1) * ldc + col);
            }
        };
        load_group(0, rsd[0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) load_group(g + 1, rsd[(g + 1) & 1]);
            float4 vals[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 v = value_of(g * 4 + u);
                const ef4 o = rsd[g & 1][u];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                vals[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = row_of(g * 4 + u);
                if (row < M) store_row(row, vals[u]);
            }
        }
    } else if (fast) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = row_of(it);
            const float4 v = value_of(it);
            if (row < M) store_row(row, v);
        }
    } else {   // ragged N / unaligned rows: element by element
        for (int it = 0; it < 16; ++it) {
            const int rl = it * 4 + (lane >> 4);
            const int row = m0 + wm * 64 + rl;
            if (row >= M) continue;
            const ef4 cv = *reinterpret_cast<const ef4*>(ct + rl * 64 + c4);
            const float vv[4] = {cv.x, cv.y, cv.z, cv.w};
            for (int c = 0; c < 4; ++c) {
                if (col + c >= N) break;
                float v = vv[c] * alpha + (bias ? bias[col + c] : 0.f);
                if (do_gelu) v = (NPL == 1 && !outf) ? gelu16_1<true>(v) : gelu_erf_fast(v);
                const size_t idx = (size_t)row * ldc + col + c;
                if (do_acc) v += Rsrc[idx];
                if (outf) Cf[idx] = v;
                if (out16) store16<NPL>(C16, (size_t)plane16, idx, (flags & G16_GELU16) ? gelu_erf_fast(v) : v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm16_v2_kernel: the same contraction, restructured for the model's shapes (M = 4096*B rows, K = 512..2048):
//   * 256x128 block tile, 8 waves (4x2), wave tile 64x64: 1.5x the operand reuse of the 128x128 tile
//   * 3-stage LDS ring of 48 KB stages (A 256 rows + W 128 rows, x NPL planes), filled by global_load_lds;
//     the wave waits with a COUNTED vmcnt (the next stage stays in flight across the barrier) and a raw
//     s_barrier -- the 2-stage / vmcnt(0)-per-iteration structure above is load-latency-bound at K = 512
//   * XCD-aware block order: the 8 m-tiles of a group go to the 8 XCDs and each XCD walks the n-tiles of ITS
//     m-tile, so the big operand (A) is fetched into one L2 only; W (<= 2 MB) is resident in every L2.
// ------------------------------------------------------------------------------------------------
template <int NPL, int BK, int NWN, int BN_ = 128>
__global__ __launch_bounds__(256 * NWN, 1) void gemm16_v2_kernel(const uint16_t* __restrict__ A, long lda, long planeA,
                                                           const uint16_t* __restrict__ W, long ldw, long planeW,
                                                           const float* __restrict__ bias, float* __restrict__ Cf,
                                                           uint16_t* __restrict__ C16, long plane16, long ldc, int M,
                                                           int N, int K, int flags, float alpha, long strideA,
                                                           long strideC, const float* Rf, long strideW) {
    constexpr int BM = 256, BN = BN_, NST = 3;
    constexpr int ROWB = BK * 2, SPR = ROWB / 16, RPB = 256 / ROWB;
    constexpr int A_TILE = BM * ROWB, W_TILE = BN * ROWB;
    constexpr int STAGE_B = NPL * (A_TILE + W_TILE);       // 48 KB (256x128 tiles) or 32 KB (bf16 256x256, BK = 32)
    constexpr int NW = 4 * NWN;                             // waves per block: 4 (M) x NWN (N)
    constexpr int NJ = BN / NWN / 32;                       // 32-wide MFMA tiles per wave along N
    constexpr int WN_COLS = BN / NWN;                       // columns per wave
    constexpr int NCH = STAGE_B / 1024;                     // 1 KB chunks per stage (48)
    constexpr int CPW = NCH / NW;                           // per wave (6 or 3)
    // epilogue transposition: the ring holds EROUNDS-th of every wave's 64 x WN_COLS accumulator tile at a time
    constexpr int EROUNDS = (NW * 64 * WN_COLS * 4 + NST * STAGE_B - 1) / (NST * STAGE_B) <= 1 ? 1 : 4;
    constexpr int RR = 64 / EROUNDS;                        // rows of the wave tile per round
    static_assert(NST * STAGE_B <= 144 * 1024 && CPW * NW == NCH && NW * RR * WN_COLS * 4 <= NST * STAGE_B,
                  "stage / epilogue geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / NWN, wn = wid % NWN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order (see header)
    const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    int bid = blockIdx.x, m_tile, n_tile;
    const int full = (mt / 8) * 8 * nt;
    if (bid < full) {
        const int grp = bid / (8 * nt), rem = bid - grp * 8 * nt;
        m_tile = grp * 8 + (rem & 7);
        n_tile = rem >> 3;
    } else {
        const int rem = bid - full;
        m_tile = (mt / 8) * 8 + rem / nt;
        n_tile = rem % nt;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;
    if (Cf) Cf += (size_t)blockIdx.y * strideC;
    if (C16) C16 += (size_t)blockIdx.y * strideC;
    // residual source of G16_ACCUM: Cf itself (in place) unless the caller keeps the input (training: Rf != Cf)
    const float* Rsrc = Rf ? Rf + (size_t)blockIdx.y * strideC : Cf;

    // ---- staging: chunk c = wid + 8*i of the stage image [A pl0 .. | W pl0 ..]
    const uint16_t* gsrc[CPW];
    int ldsoff[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = wid + NW * i;
        int off = c * 1024;
        const bool isA = off < NPL * A_TILE;
        if (!isA) off -= NPL * A_TILE;
        const int tile_b = isA ? A_TILE : W_TILE;
        const int pl = off / tile_b;
        const int in_tile = off - pl * tile_b;
        const int row_local = in_tile / ROWB + lane / SPR;
        const int slot = (lane % SPR) ^ ((row_local / RPB) % SPR);
        if (isA) {
            int ra = m0 + row_local; ra = ra < M ? ra : M - 1;
            gsrc[i] = A + (size_t)pl * planeA + (size_t)ra * lda + slot * 8;
        } else {
            int rw = n0 + row_local; rw = rw < N ? rw : N - 1;
            gsrc[i] = W + (size_t)pl * planeW + (size_t)rw * ldw + slot * 8;
        }
        ldsoff[i] = c * 1024;
    }
    auto stage = [&](int st, int k0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(smem + st * STAGE_B + ldsoff[i]),
                                             16, 0, 0);
    };

    f32x16 acc[2][NJ], corr[NPL == 2 ? 2 : 1][NPL == 2 ? NJ : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if constexpr (NPL == 2) corr[i][j][e] = 0.f;
            }
    int rowA[2], rowB[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) rowA[i] = wm * 64 + i * 32 + r;
#pragma unroll
    for (int j = 0; j < NJ; ++j) rowB[j] = wn * WN_COLS + j * 32 + r;
    auto frag_off = [&](int row_local, int kk) {
        const int slot = 2 * kk + h;
        return row_local * ROWB + ((slot ^ ((row_local / RPB) % SPR)) << 4);
    };

    const int nk = K / BK;
    constexpr int KK = BK / 16;
    // one piece of a stage's loads (CPW pieces in total) -- issued between MFMA groups so that the VMEM issue cost
    // (~100 cycles per global_load_lds) overlaps matrix-pipe time instead of sitting in front of it
    auto stage_piece = [&](int st, int k0, int i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + k0),
                                         (__attribute__((address_space(3))) void*)(smem + st * STAGE_B + ldsoff[i]), 16, 0,
                                         0);
    };
    stage(0, 0);
    if (nk > 1) stage(1, BK);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt must have landed: at most the CPW loads of stage kt+1 may still be in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's share of stage kt is visible; stage kt-1's buffer is free
        const bool more = kt + 2 < nk;
        const int nst = (kt + 2) % NST, nk0 = (kt + 2) * BK;
        const unsigned char* sa = smem + (kt % NST) * STAGE_B;
        const unsigned char* sw = sa + NPL * A_TILE;
        if constexpr (NPL == 1) {
            bf16x8 a[2][2], b[2][NJ];  // [kk parity]: fragments of step kk+1 are fetched while step kk multiplies
#pragma unroll
            for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(sa + frag_off(rowA[i], 0));
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[0][j] = *reinterpret_cast<const bf16x8*>(sw + frag_off(rowB[j], 0));
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < KK) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        a[nxt][i] = *reinterpret_cast<const bf16x8*>(sa + frag_off(rowA[i], kk + 1));
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        b[nxt][j] = *reinterpret_cast<const bf16x8*>(sw + frag_off(rowB[j], kk + 1));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int i = kk * CPW / KK; i < (kk + 1) * CPW / KK; ++i) stage_piece(nst, nk0, i);
                }
            }
        } else {
            f16x8 ah[2][2], al[2][2], bh[2][NJ], bl[2][NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[0][i] = *reinterpret_cast<const f16x8*>(sa + frag_off(rowA[i], 0));
                al[0][i] = *reinterpret_cast<const f16x8*>(sa + A_TILE + frag_off(rowA[i], 0));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[0][j] = *reinterpret_cast<const f16x8*>(sw + frag_off(rowB[j], 0));
                bl[0][j] = *reinterpret_cast<const f16x8*>(sw + W_TILE + frag_off(rowB[j], 0));
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < KK) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[nxt][i] = *reinterpret_cast<const f16x8*>(sa + frag_off(rowA[i], kk + 1));
                        al[nxt][i] = *reinterpret_cast<const f16x8*>(sa + A_TILE + frag_off(rowA[i], kk + 1));
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        bh[nxt][j] = *reinterpret_cast<const f16x8*>(sw + frag_off(rowB[j], kk + 1));
                        bl[nxt][j] = *reinterpret_cast<const f16x8*>(sw + W_TILE + frag_off(rowB[j], kk + 1));
                    }
                }
                // three sweeps over the tiles so that no accumulator is touched by two consecutive MFMAs
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int i = (2 * kk) * CPW / (2 * KK); i < (2 * kk + 1) * CPW / (2 * KK); ++i) stage_piece(nst, nk0, i);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur][j], corr[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        corr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur][j], corr[i][j], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int i = (2 * kk + 1) * CPW / (2 * KK); i < (2 * kk + 2) * CPW / (2 * KK); ++i)
                        stage_piece(nst, nk0, i);
                }
            }
        }
    }

    // ---- epilogue.  The accumulators hold one COLUMN per lane (C/D map of the 32x32 MFMA), which would make
    // every global access a 4-byte-per-lane row fragment (store-issue-bound: measured ~2x the K=512 main loop).
    // Each wave transposes its 64x64 tile through its own 16 KB of the (now free) LDS ring and then works on
    // whole rows: 16 lanes x float4 = one 256-byte row segment per quarter-wave for the residual read, the f32
    // store and the 16-bit operand store.
    __syncthreads();  // every wave is done with the stage buffers (and has drained its loads)
    float* ct = reinterpret_cast<float*>(smem) + wid * (RR * WN_COLS);
    const bool do_gelu = flags & G16_GELU, do_acc = flags & G16_ACCUM;
    const bool out16 = flags & G16_OUT16, outf = flags & G16_OUTF32;
    constexpr int LPR = WN_COLS / 4;  // lanes per row (float4 each)
    constexpr int RPI = 64 / LPR;     // rows per wave-instruction
    const int c4 = (lane % LPR) << 2;
    const int col = n0 + wn * WN_COLS + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && col < N) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
    for (int q = 0; q < EROUNDS; ++q) {
    // round q: rows [q*RR, (q+1)*RR) of the wave tile = MFMA tile i, e>>2 groups as below
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rowt = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // row inside the 64-row wave tile
                if (EROUNDS > 1 && (i * 32 + 8 * (e >> 2)) / RR != q) continue;  // compile-time: (e&3) + 4h < 8 <= RR
                float v = acc[i][j][e];
                if constexpr (NPL == 2) v += corr[i][j][e] * SPLIT_INV;
                ct[(rowt - q * RR) * WN_COLS + j * 32 + r] = v;
            }
    __builtin_amdgcn_wave_barrier();
    typedef float ef4 __attribute__((ext_vector_type(4)));
    constexpr int NIT = RR / RPI;
    auto row_of = [&](int it) { return m0 + wm * 64 + q * RR + it * RPI + lane / LPR; };
    auto value_of = [&](int it) {   // alpha * tile + bias (+ GELU) of wave-instruction `it`
        const ef4 cv = *reinterpret_cast<const ef4*>(ct + (it * RPI + lane / LPR) * WN_COLS + c4);
        float4 v = make_float4(cv.x * alpha + bv.x, cv.y * alpha + bv.y, cv.z * alpha + bv.z, cv.w * alpha + bv.w);
        if (do_gelu) {
            // (bf16 operands and no f32 output: the value leaves only as bf16 -> the polynomial form, as in every bf16 kernel)
            if (NPL == 1 && !outf) {
                const genie_f2 g0 = gelu16_2<true>(genie_f2{v.x, v.y}), g1 = gelu16_2<true>(genie_f2{v.z, v.w});
                v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
            } else {
                v.x = gelu_erf_fast(v.x); v.y = gelu_erf_fast(v.y); v.z = gelu_erf_fast(v.z); v.w = gelu_erf_fast(v.w);
            }
        }
        return v;
    };
    auto store_row = [&](int row, float4 v) {
        const size_t idx = (size_t)row * ldc + col;
        if (outf) {
            if (flags & G16_NT) {
                typedef float nt4 __attribute__((ext_vector_type(4)));
                nt4 t = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(t, reinterpret_cast<nt4*>(Cf + idx));
            } else {
                *reinterpret_cast<float4*>(Cf + idx) = v;
            }
        }
        if (out16) {
            if (flags & G16_GELU16) { v.x = gelu_erf_fast(v.x); v.y = gelu_erf_fast(v.y); v.z = gelu_erf_fast(v.z); v.w = gelu_erf_fast(v.w); }
            if constexpr (NPL == 1) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                store_u2(C16 + idx, pk, flags & G16_NT);
            } else {
                uint16_t h0, l0, h1, l1, h2, l2, h3, l3;
                split_f16(v.x, h0, l0); split_f16(v.y, h1, l1); split_f16(v.z, h2, l2); split_f16(v.w, h3, l3);
                uint2 ph, pl;
                ph.x = (uint32_t)h0 | ((uint32_t)h1 << 16); ph.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                pl.x = (uint32_t)l0 | ((uint32_t)l1 << 16); pl.y = (uint32_t)l2 | ((uint32_t)l3 << 16);
                store_u2(C16 + idx, ph, flags & G16_NT);
                store_u2(C16 + (size_t)plane16 + idx, pl, flags & G16_NT);
            }
        }
    };
    if (do_acc) {
        // Residual accumulate: rows in groups of EG wave-instructions, each group in three straight sections -- request the NEXT
        // group's residual rows, finish this group's values in registers, store them.  The residual may alias the output, so a
        // read written after a store stays behind it, and vmcnt retires in order: row by row (read, add, store) every row
        // exposed a store + load round trip (proj at 4,096 rows: 23.2 -> 20.9 us).  The rows of a tile are distinct, so reading
        // ahead is safe in place; rows past M read row M - 1 (never stored).
        constexpr int EG = NIT % 4 == 0 ? 4 : (NIT % 2 == 0 ? 2 : 1);
        ef4 rsd[2][EG];
        const int colc = col < N ? col : 0;
        auto load_group = [&](int g, ef4* dst) {
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                const int row = row_of(g * EG + u);
                dst[u] = *reinterpret_cast<const ef4*>(Rsrc + (size_t)(row < M ? row : M - 1) * ldc + colc);
            }
        };
        load_group(0, rsd[0]);
#pragma unroll
        for (int g = 0; g < NIT / EG; ++g) {
            if (g + 1 < NIT / EG) load_group(g + 1, rsd[(g + 1) & 1]);
            float4 vals[EG];
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                float4 v = value_of(g * EG + u);
                const ef4 o = rsd[g & 1][u];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                vals[u] = v;
            }
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                const int row = row_of(g * EG + u);
                if (row < M && col < N) store_row(row, vals[u]);
            }
        }
    } else {
#pragma unroll 4
        for (int it = 0; it < NIT; ++it) {
            const int row = row_of(it);
            const float4 v = value_of(it);
            if (row < M && col < N) store_row(row, v);
        }
    }
    __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next round
    }
}

// Weight tensors whose packed hi plane reaches |w| >= 32 (the `w16_wide` flags of the weight structs, genie_hip.h): the
// single-accumulator split GEMM (gemm16_pp) multiplies the hi plane by 2^11 in f16, exact below 32 only, so the layer drivers pass
// weights_on_w = false for these and they run on the two-accumulator kernels (gemm16_v2 / gemm16_nt / gemm16_sm: hi.hi and the
// cross terms in separate accumulators, no operand scaling; limit = the f16 range).

#ifdef GENIE_STUDY
int g_study_gemm_class = 0, g_study_layer = 0;
int study_terms() {
    // re-read on every launch (study build only): tools/precision_study.py sweeps the settings inside one process
    auto env = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    const int all = env("GENIE_F16_TERMS", 3);
    const int mask = env("GENIE_F16_TERMS2_CLASSES", 0);
    const int l0 = env("GENIE_F16_TERMS2_LAYER_LO", 0), l1 = env("GENIE_F16_TERMS2_LAYER_HI", 1 << 30);
    if (all != 3) return all;
    return (((mask >> g_study_gemm_class) & 1) && g_study_layer >= l0 && g_study_layer < l1) ? 2 : 3;
}
#endif

// A, W: 16-bit operands (NPL planes each, plane strides in elements); Cf f32 (ACCUM / OUTF32), C16 16-bit out.
template <int NPL>
static int launch_gemm16(const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw, long planeW,
                         const float* bias, float* Cf, uint16_t* C16, long plane16, long ldc, int M, int N, int K,
                         int flags, float alpha, hipStream_t st, int batch = 1, long strideA = 0, long strideC = 0,
                         const float* Rf = nullptr, long strideW = 0, bool weights_on_w = true, bool allow_sm = true) {
    constexpr int BK = NPL == 1 ? 64 : 32;
    GENIE_CHECK_SHAPE(K % BK == 0 && K > 0, "gemm16: K=%d must be a positive multiple of %d", K, BK);
    GENIE_CHECK_SHAPE(lda % 8 == 0 && ldw % 8 == 0, "gemm16: leading dims must be multiples of 8 elements");
    if (M <= 0 || N <= 0) return GENIE_OK;
    if (flags & G16_WIDEW) { weights_on_w = false; flags &= ~G16_WIDEW; }
    const double mn = (double)M * N * batch;
    {   // outputs that cannot stay in the 256 MB Infinity Cache anyway are stored non-temporally: +10..18 % on the
        // K = 512 GEMMs at >= 8 clips (they no longer evict the A panel / weights they share the L2 with)
        static const int nt_mode = study_env("GENIE_GEMM16_NT", -1);
        const double out_bytes = mn * ((flags & G16_OUTF32 ? 4 : 0) + (flags & G16_OUT16 ? 2 * NPL : 0));
        if (nt_mode == 1 || (nt_mode < 0 && out_bytes >= 192e6)) flags |= G16_NT;
    }
    {   // the 256x256 phase-scheduled kernel (kernels_gemm_pp.hip) takes every problem that fills the chip with its tiles.
        // Its split-f16 form multiplies the W operand's hi plane by 2^11 in registers, so W must be a weight matrix
        // (|w| < 32): the training step's activation-by-activation products keep the two-accumulator kernel below.
        static const int pp = study_env("GENIE_GEMM16_PP", 1);
#ifdef GENIE_STUDY
        const int terms = study_terms();
#else
        constexpr int terms = 3;
#endif
        if (pp && (NPL == 1 || weights_on_w)) {
            const int npl = (NPL == 2 && terms == 1) ? 1 : NPL;
            const int rc = launch_gemm16_pp(npl, terms, NPL == 2, A, lda, planeA, W, ldw, planeW, bias, Rf, Cf, C16, plane16,
                                            ldc, M, N, K, flags, alpha, st, batch, strideA, strideW, strideC);
            if (rc != GENIE_E_UNSUPPORTED) return rc;
        }
    }
    if (allow_sm) {   // latency-bound problems (batch-1 generate: 256 rows per frame pass): split-K inside the workgroup, no LDS ring
        const int rc = launch_gemm16_sm(NPL, A, lda, planeA, W, ldw, planeW, bias, Rf, Cf, C16, plane16, ldc, M, N, K, flags, alpha,
                                        st, batch, strideA, strideW, strideC);
        if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    static const int force_v1 = study_env("GENIE_GEMM16_V1", 0);
    // small problems (batch-1 generate: M = 4096 or 256 rows): 256x128 tiles would leave most of the 256 CUs idle, the
    // 128x128 kernel below makes 2x the workgroups (and runs two of them per CU)
    const long tiles_v2 = (long)((M + 255) / 256) * ((N + 127) / 128) * batch;
    const bool small = tiles_v2 < 160;
    if (!force_v1 && !small && M >= 256 && N % 4 == 0 && ldc % 4 == 0) {
        const int mt2 = (M + 255) / 256, nt2 = (N + 127) / 128;
        const size_t lds2 = 3 * 48 * 1024;
        ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                       2.0 * NPL * ((double)M * K * batch + (double)N * K) +
                           mn * ((flags & G16_ACCUM ? 4 : 0) + (flags & G16_OUTF32 ? 4 : 0) +
                                 (flags & G16_OUT16 ? 2 * NPL : 0)),
                       st, NPL == 2 ? "gemm16_v2_kernel<2,...> (256x128 tile, f16x3 two-accumulator form)" : "gemm16_v2_kernel<1,...> (256x128 / 256x256 tile, bf16)");
        static const int nwn = study_env("GENIE_GEMM16_NWN", 4);
        static const int big = study_env("GENIE_GEMM16_BN256", 1);
        const long tiles_256 = (long)mt2 * ((N + 255) / 256) * batch;
        if constexpr (NPL == 1) {
            if (big && N % 4 == 0 && tiles_256 >= 224) {
                // bf16 only (one accumulator set): 256x256 tiles, BK = 32, 3 x 32 KB stages -- a third of the LDS-DMA
                // bytes and two thirds of the fragment reads per MFMA of the 256x128 tile, 16 MFMAs per barrier
                const int nt4 = (N + 255) / 256;
                const size_t lds4 = 3 * 32 * 1024;
                (void)hipFuncSetAttribute((const void*)gemm16_v2_kernel<1, 32, 4, 256>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                gemm16_v2_kernel<1, 32, 4, 256><<<dim3(mt2 * nt4, batch), 1024, lds4, st>>>(
                    A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf,
                    strideW);
                GENIE_LAUNCH_CHECK("gemm16_v2_256");
                return GENIE_OK;
            }
        }
        if (nwn == 4) {
            (void)hipFuncSetAttribute((const void*)gemm16_v2_kernel<NPL, BK, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            gemm16_v2_kernel<NPL, BK, 4><<<dim3(mt2 * nt2, batch), 1024, lds2, st>>>(
                A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf,
                strideW);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm16_v2_kernel<NPL, BK, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            gemm16_v2_kernel<NPL, BK, 2><<<dim3(mt2 * nt2, batch), 512, lds2, st>>>(
                A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf,
                strideW);
        }
        GENIE_LAUNCH_CHECK("gemm16_v2");
        return GENIE_OK;
    }
    const int mt = (M + 127) / 128, nt = (N + 127) / 128;
    const size_t lds = (size_t)2 * 2 * NPL * 128 * BK * 2;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                   2.0 * NPL * ((double)M * K * batch + (double)N * K) +
                       mn * ((flags & G16_ACCUM ? 4 : 0) + (flags & G16_OUTF32 ? 4 : 0) + (flags & G16_OUT16 ? 2 * NPL : 0)),
                   st, NPL == 2 ? "gemm16_nt_kernel<2,...> (128x128 tile, f16x3)" : "gemm16_nt_kernel<1,...> (128x128 tile, bf16)");
    // Problems this small (batch-1 generate: a GEMM is 21..36 us) are bound by the serial chain of K-steps -- each one a
    // barrier plus an L2 round trip -- not by throughput, and fewer workgroups than CUs are resident anyway: double the
    // K-step (128 KB of LDS, one workgroup per CU) to halve the chain.
    static const int wide_k = study_env("GENIE_GEMM16_WIDEK", 1);
    if (wide_k && mt * nt * batch <= 256 && K % (2 * BK) == 0) {
        (void)hipFuncSetAttribute((const void*)gemm16_nt_kernel<NPL, 2 * BK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(2 * lds));
        gemm16_nt_kernel<NPL, 2 * BK><<<dim3(mt * nt, batch), 256, 2 * lds, st>>>(
            A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf, strideW);
        GENIE_LAUNCH_CHECK("gemm16_widek");
        return GENIE_OK;
    }
    (void)hipFuncSetAttribute((const void*)gemm16_nt_kernel<NPL, BK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    gemm16_nt_kernel<NPL, BK><<<dim3(mt * nt, batch), 256, lds, st>>>(A, lda, planeA, W, ldw, planeW, bias, Cf, C16,
                                                                      plane16, ldc, M, N, K, flags, alpha, strideA,
                                                                      strideC, Rf, strideW);
    GENIE_LAUNCH_CHECK("gemm16");
    return GENIE_OK;
}

// Non-template entry for the training step (kernels_train16.hip): npl = 1 bf16, 2 = f16 split planes.
//   Rf: residual source when it must differ from Cf (G16_ACCUM);  batch > 1 strides A, W and C (split-K slabs:
//   strideA = strideW = K elements of one slab, strideC = M*N)
int launch_gemm16_ex(int npl, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw, long planeW,
                     const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M, int N,
                     int K, int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC) {
    if (npl == 1)
        return launch_gemm16<1>(A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, st,
                                batch, strideA, strideC, Rf, strideW);
    return launch_gemm16<2>(A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, st, batch,
                            strideA, strideC, Rf, strideW, /*weights_on_w=*/false);
}

// ---- fc2 of the one-frame passes at 2,048-4,096 rows (generate at 8-16 clips): N = d = 512 gives 64-128 tiles of 128x128 for a
// K = 2,048 contraction -- half of the CUs idle and a 64-step K chain.  Split K in two over the GEMM's batch index (256 workgroups,
// 32 steps each) into two f32 slabs in the (idle) logits scratch, then x += bias + slab0 + slab1 in that fixed order
// (profiles/r03_fc2_splitk_ab.txt).  Only in the one-frame passes of generate (w.frame_t >= 0) and only when the 16-bit shadow of x is
// not wanted (every layer but the last of a LayerNorm model): full forwards keep the single fused K chain at every batch size.
__global__ __launch_bounds__(256) void splitk2_residual_kernel(float* __restrict__ x, const float* __restrict__ s0,
                                                               const float* __restrict__ s1, const float* __restrict__ bias,
                                                               size_t n4, int N) {
    typedef float sk4 __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    sk4 v = reinterpret_cast<const sk4*>(x)[i];
    const sk4 a = reinterpret_cast<const sk4*>(s0)[i], b = reinterpret_cast<const sk4*>(s1)[i];
    sk4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const sk4*>(bias + (i * 4) % (size_t)N);
    v = v + ((a + bv) + b);
    reinterpret_cast<sk4*>(x)[i] = v;
}
// returns GENIE_E_UNSUPPORTED when the shape is not in that range (the caller then runs the fused-epilogue GEMM)
template <int NPL>
static int fc2_splitk2(const genie_cfg& c, const uint16_t* h16, long plane_h, const uint16_t* w16, long plane_w, const float* bias,
                       float* x, Workspace& w, int M, hipStream_t st, bool wide = false) {
    static const int on = study_env("GENIE_FC2_SPLITK", 1);
    const int d = c.d_model, K = c.hidden;
    const long tiles = (long)((M + 127) / 128) * ((d + 127) / 128);
    const size_t V = (size_t)c.factored_vocab * c.num_factored;
    if (!on || w.frame_t < 0 /* one-frame passes only: a clip's fc2 sum order must not depend on the batch size of a full forward */ ||
        !w.skip_shadow_mlp || !w.logits || K < 2048 || K % 256 || d % 4 || tiles > 128 || (long)M * d <= (1L << 19) ||
        V < 2 * (size_t)d || wide)
        return GENIE_E_UNSUPPORTED;
    float* slabs = w.logits;
    GENIE_TRY(launch_gemm16<NPL>(h16, K, plane_h, w16, K, plane_w, nullptr, slabs, nullptr, 0, d, M, d, K / 2, G16_OUTF32, 1.0f, st,
                                 2, (long)(K / 2), (long)M * d, nullptr, (long)(K / 2), true, false));
    const size_t n4 = (size_t)M * d / 4;
    splitk2_residual_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(x, slabs, slabs + (size_t)M * d, bias, n4, d);
    GENIE_LAUNCH_CHECK("splitk2_residual");
    return GENIE_OK;
}

// ---- elementwise helpers -----------------------------------------------------------------------
__global__ void cast16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f32_to_bf16(src[i]);
}


// Spatial attention on the fused operand path: the QKV GEMM writes [Q * scale * log2e | K | V^T] as 16-bit planes in the
// attention kernel's own layout (kernels_gemm_pp.hip, G16X_QKV) and kernels_attn_dma.hip streams them through LDS -- no f32
// qkv round trip, no operand split / transpose inside the attention kernel.  Covers the shipped geometry (S = 256, head_dim
// 32 / 64, d % 256 == 0, LayerNorm or qk-norm blocks, chip-filling batches); anything else returns GENIE_E_UNSUPPORTED and the caller
// runs the f32-qkv path below.
// x / x16 / proj_done (bf16 only): when given and the geometry allows, the attention AND the out-projection + residual run as
// kernels_fused.hip's spatial_attn_proj kernel (x, x16 updated, *proj_done = true: the caller skips its proj GEMM)
static int spatial_attention_fused(int npl, const genie_cfg& c, const genie_layer_weights& lw, const uint16_t* u, size_t planeA,
                                   size_t planeW, Workspace& w, int B, uint16_t* out16, size_t out_plane, hipStream_t st,
                                   float* x = nullptr, uint16_t* x16 = nullptr, bool* proj_done = nullptr, bool shadow16 = true) {
    static const int on = study_env("GENIE_ATTN_DMA", 1);
    const int d = c.d_model;
#ifdef GENIE_STUDY
    if (npl == 2 && study_terms() != 3) return GENIE_E_UNSUPPORTED;
#endif
    if (npl == 2 && (lw.spatial.w16_wide & GENIE_WIDE_QKV)) return GENIE_E_UNSUPPORTED;  // |w| >= 32: two-accumulator GEMM + f32-qkv attention
    // (qk_norm: the per-head LayerNorm of q and k is the QKV GEMM's epilogue -- G16X_QKNORM -- so the planes hold normalised, scaled
    // operands and the attention kernels below are the same in both variants)
    if (!on || c.S != 256 || (c.head_dim != 64 && c.head_dim != 32) || d % 256 || d != c.num_heads * c.head_dim ||
        (c.qk_norm && !(lw.spatial.norm_w && lw.spatial.norm_b)))
        return GENIE_E_UNSUPPORTED;
    const long n_seq = (long)B * c.T;
    const int M = (int)(n_seq * c.S);
    uint16_t* qkv16 = (uint16_t*)w.big;  // 3 * npl planes of M * d 16-bit values <= the (M, 3d) f32 buffer
    const int rc = launch_gemm16_pp(npl, 3, npl == 2, u, d, (long)planeA, lw.spatial.qkv_w16, d, (long)planeW,
                                    c.qkv_bias ? lw.spatial.qkv_b : nullptr, nullptr, nullptr, qkv16, (long)M * d, d, M, 3 * d, d,
                                    G16X_OUT16 | G16X_QKV | (c.qk_norm ? G16X_QKNORM : 0), 1.0f, st, 1, 0, 0, 0,
                                    c.attn_scale * 1.4426950408889634f, c.head_dim, c.qk_norm ? lw.spatial.norm_w : nullptr,
                                    c.qk_norm ? lw.spatial.norm_b : nullptr);
    if (rc != GENIE_OK) return rc;
    if (npl == 1 && x && x16 && proj_done) {   // shipped geometry, bf16: attention over all heads + out-projection + residual in one kernel
        const int rf = launch_spatial_attn_proj_bf16(c, lw.spatial, qkv16, x, shadow16 ? x16 : nullptr, n_seq, st);
        if (rf == GENIE_OK) { *proj_done = true; return GENIE_OK; }
        if (rf != GENIE_E_UNSUPPORTED) return rf;
    }
    return launch_attn_spatial_dma(npl, qkv16, n_seq, d, c.num_heads, c.head_dim, out16, out_plane, st);
}

// bf16 precision contract (mirrored by oracle.genie_oracle.BF16_MFMA):
//   * every nn.Linear operand is bf16 (weights packed once; activations rounded by their producer)
//   * accumulation, bias, GELU, LayerNorm and the residual stream are f32
//   * qkv leaves its GEMM as f32 and the attention core runs on the f32-MFMA kernels of the exact path (its
//     output is rounded to bf16 for the out-projection) -- attention is ~6 % of the FLOPs.
int st_block_bf16(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st) {
    const int d = c.d_model, M = B * c.T * c.S;
    uint16_t* x16 = (uint16_t*)w.xn;              // bf16 shadow of the residual stream
    uint16_t* xn16 = x16 + (size_t)M * d;         // LayerNorm output, then attention output
    float* qkv = (float*)w.big;                   // f32 qkv
    uint16_t* big16 = (uint16_t*)w.big;           // later: bf16 MLP hidden
    GENIE_CHECK_ARG(lw.spatial.qkv_w16 && lw.spatial.proj_w16 && lw.temporal.qkv_w16 && lw.temporal.proj_w16 &&
                        lw.fc1_w16 && lw.fc2_w16,
                    "bf16 precision needs packed bf16 weights (genie_pack_bf16)");
    const float* nws = c.qk_norm ? lw.spatial.norm_w : nullptr;
    const float* nbs = c.qk_norm ? lw.spatial.norm_b : nullptr;
    const float* nwt = c.qk_norm ? lw.temporal.norm_w : nullptr;
    const float* nbt = c.qk_norm ? lw.temporal.norm_b : nullptr;
    // Will the temporal sub-block run as the fused kernel?  Then it rounds its operands from the f32 rows itself and the spatial kernel in
    // front need not write the bf16 shadow of x (134 MB per layer at 64 clips it would write and the temporal kernel read).
    static const int no_shadow_env = study_env("GENIE_T_FROM_F32", 1);   // (a study-build knob: the shipping library reads no environment)
    const bool fused_t = w.frame_t < 0 && !w.tqkv && !w.tcache && !w.stop_after_tqkv && temporal_qkv16(c, w.model_T) &&
                         temporal_fused_takes(c, lw.temporal, B);
    // ... or, in the prefix-cache passes, as the fused kernel that keeps the K / V fragment images in the cache slice (kernels_fused_prefix.hip)
    const bool frag_t = w.frame_t < 0 && (w.tqkv || w.tcache) && (!w.tqkv || w.tq_frames <= c.T) && temporal_qkv16(c, w.model_T) &&
                        temporal_prefix_fused_takes(c, lw.temporal, B, w.model_T);
    const bool shadow16 = !((fused_t || frag_t) && no_shadow_env);
    // spatial
    const uint16_t* u = x16;
    int rc = GENIE_E_UNSUPPORTED;
    bool qkv_done = false, proj_done = false;
    if (w.qkv_planes_done) {   // the previous block's fused MLP kernel left this block's operand planes in `big`
        w.qkv_planes_done = false;
        rc = launch_spatial_attn_proj_bf16(c, lw.spatial, (const uint16_t*)w.big, x, shadow16 ? x16 : nullptr, (long)B * c.T, st);
        if (rc == GENIE_OK) proj_done = true;
        else if (rc == GENIE_E_UNSUPPORTED)   // (fewer sequences than the fused kernel takes: the stand-alone attention kernel reads the same planes, proj GEMM below)
            rc = launch_attn_spatial_dma(1, (uint16_t*)w.big, (long)B * c.T, d, c.num_heads, c.head_dim, xn16, 0, st);
        GENIE_TRY(rc);
        qkv_done = true;
    }
    if (!qkv_done && !c.qk_norm) {  // one-frame passes: LayerNorm inside the small GEMM's fragment path (no LayerNorm launch)
        const int r2 = launch_gemm16_sm_ln(1, x, d, lw.norm1_w, lw.norm1_b, 1e-5f, lw.spatial.qkv_w16, d, 0,
                                           c.qkv_bias ? lw.spatial.qkv_b : nullptr, nullptr, qkv, nullptr, 0, 3 * d, M, 3 * d, d,
                                           G16_OUTF32, 1.0f, st);
        if (r2 == GENIE_OK) qkv_done = true;
        else if (r2 != GENIE_E_UNSUPPORTED) return r2;
    }
    if (!qkv_done) {
        if (!c.qk_norm) {
            if (!w.ln1_done)   // (else: the previous block's fused MLP kernel wrote norm1(x) into xn16)
                GENIE_TRY(launch_layer_norm_bf16(x, lw.norm1_w, lw.norm1_b, xn16, M, d, 1e-5f, st));
            u = xn16;
        }
        rc = spatial_attention_fused(1, c, lw, u, 0, 0, w, B, xn16, 0, st, x, x16, &proj_done, shadow16);
    }
    w.ln1_done = false;
    if (rc == GENIE_E_UNSUPPORTED) {
    if (!qkv_done)
    GENIE_TRY(launch_gemm16<1>(u, d, 0, lw.spatial.qkv_w16, d, 0, c.qkv_bias ? lw.spatial.qkv_b : nullptr, qkv, nullptr,
                               0, 3 * d, M, 3 * d, d, G16_OUTF32, 1.0f, st));
    rc = launch_attn_spatial_split(qkv, nullptr, c.S, (long)B * c.T, d, c.num_heads, c.head_dim, c.attn_scale, nws,
                                   nbs, st, xn16, 0);
    if (rc == GENIE_E_UNSUPPORTED) {
        GENIE_TRY(launch_attn_generic(qkv, w.logits, c.S, (long)B * c.T, 1, c.S, 0, 1, d, c.num_heads, c.head_dim,
                                      c.attn_scale, 0, nws, nbs, st));
        rc = launch_pack_bf16(w.logits, xn16, (size_t)M * d, st);
    }
    }
    GENIE_TRY(rc);
    if (!proj_done)
    GENIE_TRY(launch_gemm16<1>(xn16, d, 0, lw.spatial.proj_w16, d, 0, c.proj_bias ? lw.spatial.proj_b : nullptr, x, x16,
                               0, d, M, d, d, G16_ACCUM | G16_OUTF32 | G16_OUT16, 1.0f, st));
    // temporal (no pre-norm): operand = bf16 shadow of x.  t16: the temporal qkv (and the KV cache slices) hold bf16 -- the qkv
    // GEMM stores 2 bytes per value instead of 4 and the attention kernels (HBM-bound) read half the bytes; softmax and both
    // products stay f32 inside them
    const bool t16 = temporal_qkv16(c, w.model_T);
    const int oflag = t16 ? G16_OUT16 : G16_OUTF32;
    float* tq = w.tqkv ? w.tqkv : qkv;
    uint16_t* tq16 = reinterpret_cast<uint16_t*>(tq);
    bool temporal_done = false;
    if (w.frame_t < 0 && !w.tqkv && !w.tcache && !w.stop_after_tqkv && t16) {
        // plain full-clip forward of the shipped geometry: qkv + attention + proj + residual in ONE kernel, the qkv never
        // leaves the registers (kernels_fused.hip); same rounding points as the launches below
        // (the bf16 shadow of x exists unless the fused spatial kernel ran and was told not to write it)
        rc = launch_temporal_fused_bf16(c, lw.temporal, (proj_done && !shadow16) ? nullptr : x16, x, B, st);
        if (rc == GENIE_OK) temporal_done = true;
        else if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    if (frag_t) {   // prefix-cache passes of the shipped geometry: one kernel, the cache slice holds K / V fragment images
        GENIE_TRY(launch_temporal_prefix_fused_bf16(c, lw.temporal, x, reinterpret_cast<uint16_t*>(w.tqkv ? w.tqkv : const_cast<float*>(w.tcache)),
                                                    B, w.tqkv ? 1 : 2, w.tshift, w.model_T, st));
        if (w.stop_after_tqkv) return GENIE_OK;
        temporal_done = true;
    }
    if (!temporal_done) {
    if (w.frame_t >= 0) {  // single-frame decode: qkv -> cache slot frame_t, attend slots 0..frame_t
        float* slot = w.fcache + (size_t)w.frame_t * c.S * 3 * d;
        uint16_t* slot16 = reinterpret_cast<uint16_t*>(w.fcache) + (size_t)w.frame_t * c.S * 3 * d;
        GENIE_TRY(launch_gemm16<1>(x16, d, 0, lw.temporal.qkv_w16, d, 0, c.qkv_bias ? lw.temporal.qkv_b : nullptr, t16 ? nullptr : slot,
                                   t16 ? slot16 : nullptr, 0, 3 * d, c.S, 3 * d, d, oflag, 1.0f, st, B, (long)c.S * d,
                                   (long)w.frame_T * c.S * 3 * d));
        rc = launch_attn_temporal_single(w.fcache, nullptr, B, w.frame_T, c.S, w.frame_t, d, c.num_heads, c.head_dim,
                                         c.attn_scale, nwt, nbt, st, xn16, 0, t16);
    } else {
    const int Tq = (w.tqkv && w.tq_frames > c.T) ? w.tq_frames : c.T;  // frames per clip in tq's layout
    if (Tq != c.T && B > 1)  // a short clean pass into a longer cache: one GEMM batch entry per clip
        GENIE_TRY(launch_gemm16<1>(x16, d, 0, lw.temporal.qkv_w16, d, 0, c.qkv_bias ? lw.temporal.qkv_b : nullptr, t16 ? nullptr : tq,
                                   t16 ? tq16 : nullptr, 0, 3 * d, c.T * c.S, 3 * d, d, oflag, 1.0f, st, B, (long)c.T * c.S * d,
                                   (long)Tq * c.S * 3 * d));
    else
    GENIE_TRY(launch_gemm16<1>(x16, d, 0, lw.temporal.qkv_w16, d, 0, c.qkv_bias ? lw.temporal.qkv_b : nullptr, t16 ? nullptr : tq,
                               t16 ? tq16 : nullptr, 0, 3 * d, M, 3 * d, d, oflag, 1.0f, st));
    if (w.stop_after_tqkv) return GENIE_OK;
    if (w.tcache) {
        rc = launch_attn_temporal_prefix(tq, w.tcache, nullptr, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale,
                                         nwt, nbt, st, xn16, 0, w.tshift, t16);
        if (rc == GENIE_E_UNSUPPORTED && !t16) {
            GENIE_TRY(launch_attn_temporal_prefix(tq, w.tcache, w.logits, B, c.T, c.S, d, c.num_heads, c.head_dim,
                                                  c.attn_scale, nwt, nbt, st, nullptr, 0, w.tshift));
            rc = launch_pack_bf16(w.logits, xn16, (size_t)M * d, st);
        }
    } else {
        rc = launch_attn_temporal_f32_mfma(tq, nullptr, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale, nwt, nbt,
                                           st, xn16, 0, Tq, t16);
        if (rc == GENIE_E_UNSUPPORTED && !t16) {
            GENIE_CHECK_ARG(Tq == c.T || B == 1, "strided temporal qkv needs the MFMA temporal kernel (8 <= frames <= 16)");
            GENIE_TRY(launch_attn_generic(tq, w.logits, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, d, c.num_heads,
                                          c.head_dim, c.attn_scale, 1, nwt, nbt, st));
            rc = launch_pack_bf16(w.logits, xn16, (size_t)M * d, st);
        }
    }
    }
    GENIE_TRY(rc);
    // the 16-bit shadow of x is only read by a Linear that has no LayerNorm in front: temporal qkv always, fc1 and the next
    // block's spatial qkv only in the qk-norm variant, the readout after the last block
    GENIE_TRY(launch_gemm16<1>(xn16, d, 0, lw.temporal.proj_w16, d, 0, c.proj_bias ? lw.temporal.proj_b : nullptr, x,
                               x16, 0, d, M, d, d, G16_ACCUM | G16_OUTF32 | (c.qk_norm ? G16_OUT16 : 0), 1.0f, st));
    }
    // MLP
    if (w.frame_t < 0) {   // LayerNorm + fc1 + GELU + fc2 + residual in one kernel for the shipped geometry (kernels_fused.hip)
        const genie_layer_weights* nx = w.next_layer;
#ifdef GENIE_VAR_NO_LNOUT
        constexpr int lnout = 0;
#else
        constexpr int lnout = 1;
#endif
        if (lnout && nx && nx->norm1_w && nx->norm1_b && w.skip_shadow_mlp) {
            rc = GENIE_E_UNSUPPORTED;
            if (nx->spatial.fused_w16 && (nx->spatial.w16_wide & GENIE_FUSED_QKV_STREAM)) {
                // ... and the next block's spatial qkv Linear too: its operand planes (in `big`, where its qkv GEMM would put them)
                rc = launch_mlp_fused_bf16(c, lw, x, nullptr, (long)M, st, nx->norm1_w, nx->norm1_b,
                                           nx->spatial.fused_w16 + GENIE_SPATIAL_PROJ_FUSED_ELEMS, (uint16_t*)w.big);
                if (rc == GENIE_OK) w.qkv_planes_done = true;
            }
            if (rc == GENIE_E_UNSUPPORTED) {
                rc = launch_mlp_fused_bf16(c, lw, x, xn16, (long)M, st, nx->norm1_w, nx->norm1_b);
                if (rc == GENIE_OK) w.ln1_done = true;
            }
        } else {
            rc = launch_mlp_fused_bf16(c, lw, x, w.skip_shadow_mlp ? nullptr : x16, (long)M, st);
        }
        if (rc != GENIE_E_UNSUPPORTED) return rc;
    }
    u = x16;
    bool fc1_done = false;
    if (!c.qk_norm) {
        const int r2 = launch_gemm16_sm_ln(1, x, d, lw.norm2_w, lw.norm2_b, 1e-5f, lw.fc1_w16, d, 0, c.mlp_bias ? lw.fc1_b : nullptr,
                                           nullptr, nullptr, big16, 0, c.hidden, M, c.hidden, d, G16_GELU | G16_OUT16, 1.0f, st);
        if (r2 == GENIE_OK) fc1_done = true;
        else if (r2 != GENIE_E_UNSUPPORTED) return r2;
    }
    if (!fc1_done) {
    if (!c.qk_norm) {
        GENIE_TRY(launch_layer_norm_bf16(x, lw.norm2_w, lw.norm2_b, xn16, M, d, 1e-5f, st));
        u = xn16;
    }
    GENIE_TRY(launch_gemm16<1>(u, d, 0, lw.fc1_w16, d, 0, c.mlp_bias ? lw.fc1_b : nullptr, nullptr, big16, 0, c.hidden,
                               M, c.hidden, d, G16_GELU | G16_OUT16, 1.0f, st));
    }
    {
        const int rs = fc2_splitk2<1>(c, big16, 0, lw.fc2_w16, 0, c.mlp_bias ? lw.fc2_b : nullptr, x, w, M, st);
        if (rs != GENIE_E_UNSUPPORTED) return rs;
    }
    GENIE_TRY(launch_gemm16<1>(big16, c.hidden, 0, lw.fc2_w16, c.hidden, 0, c.mlp_bias ? lw.fc2_b : nullptr, x, x16, 0,
                               d, M, d, c.hidden, G16_ACCUM | G16_OUTF32 | (w.skip_shadow_mlp ? 0 : G16_OUT16), 1.0f, st));
    return GENIE_OK;
}

// The bf16 shadow of x must exist before the first layer when the block has no pre-norm (qk_norm configs).
int prepare_bf16(const genie_cfg& c, const float* x, Workspace& w, int B, hipStream_t st) {
    // LayerNorm blocks read the f32 x (norm1) first and their spatial out-projection writes the shadow before anything reads it
    if (!c.qk_norm) return GENIE_OK;
    const size_t n = (size_t)B * c.T * c.S * c.d_model;
    cast16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, (uint16_t*)w.xn, n);
    GENIE_LAUNCH_CHECK("cast16");
    return GENIE_OK;
}

// out_x_proj on frames [t0,t1) from the bf16 shadow; logits f32.  BCTHW goes through the token-major scratch.
int readout_bf16(const genie_cfg& c, const genie_weights& wt, const float* x, Workspace& w, int B, int t0, int t1,
                 int layout, float* logits, hipStream_t st) {
    GENIE_CHECK_ARG(wt.out_w16, "bf16 precision needs packed bf16 weights (genie_pack_bf16)");
    const int d = c.d_model, nt = t1 - t0, V = c.factored_vocab * c.num_factored;
    const long rows = (long)nt * c.S;
    GENIE_CHECK_ARG((const void*)x == (const void*)w.x, "bf16 readout reads the workspace's own hidden state");
    const uint16_t* x16 = (const uint16_t*)w.xn;
    float* dst = (layout == GENIE_LAYOUT_TOKEN_MAJOR) ? logits : w.logits;
    GENIE_TRY(launch_gemm16<1>(x16 + (size_t)t0 * c.S * d, d, 0, wt.out_w16, d, 0, wt.out_b, dst, nullptr, 0, V,
                               (int)rows, V, d, G16_OUTF32, c.readout_mult, st, B, (long)c.T * c.S * d, rows * V));
    if (layout != GENIE_LAYOUT_TOKEN_MAJOR) GENIE_TRY(launch_transpose(w.logits, logits, B, (int)rows, V, st));
    return GENIE_OK;
}

// ---- GENIE_PREC_F16X3 ---------------------------------------------------------------------------
// Every Linear runs on the f16 matrix cores with split operands (3 MFMAs per K-step); everything else is
// the exact path: f32 qkv -> f32-MFMA attention kernels -> outputs re-split for the next Linear.
// Buffers: w.xn  = split planes of the residual stream x (hi | lo), M*d each
//          w.aux = split planes of the LayerNorm output, then of the attention output
//          w.big = qkv as f32 (M*3d), later the split planes of the MLP hidden (M*hidden each)
int st_block_f16x3(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, hipStream_t st) {
    const int d = c.d_model, M = B * c.T * c.S, hid = c.hidden;
    const size_t pd = (size_t)M * d, ph = (size_t)M * hid;
    uint16_t* xs = (uint16_t*)w.xn;
    uint16_t* as = (uint16_t*)w.aux;
    float* qkv = (float*)w.big;
    uint16_t* hs = (uint16_t*)w.big;
    GENIE_CHECK_ARG(lw.spatial.qkv_w16 && lw.spatial.proj_w16 && lw.temporal.qkv_w16 && lw.temporal.proj_w16 &&
                        lw.fc1_w16 && lw.fc2_w16,
                    "f16x3 precision needs split-f16 weights (genie_pack_split_f16)");
    const size_t pw_qkv = (size_t)3 * d * d, pw_proj = (size_t)d * d, pw_fc = (size_t)hid * d;
    // |w| >= 32 range flags of the packed tensors (genie_hip.h): those Linears stay off the 2^11-scaling kernel
    const int wsq = (lw.spatial.w16_wide & GENIE_WIDE_QKV) ? G16_WIDEW : 0, wsp = (lw.spatial.w16_wide & GENIE_WIDE_PROJ) ? G16_WIDEW : 0;
    const int wtq = (lw.temporal.w16_wide & GENIE_WIDE_QKV) ? G16_WIDEW : 0, wtp = (lw.temporal.w16_wide & GENIE_WIDE_PROJ) ? G16_WIDEW : 0;
    const int wf1 = (lw.w16_wide & GENIE_WIDE_FC1) ? G16_WIDEW : 0, wf2 = (lw.w16_wide & GENIE_WIDE_FC2) ? G16_WIDEW : 0;
    const float* nws = c.qk_norm ? lw.spatial.norm_w : nullptr;
    const float* nbs = c.qk_norm ? lw.spatial.norm_b : nullptr;
    const float* nwt = c.qk_norm ? lw.temporal.norm_w : nullptr;
    const float* nbt = c.qk_norm ? lw.temporal.norm_b : nullptr;
    // ---- spatial
    GENIE_STUDY_CLASS(0);
    const uint16_t* u = xs;
    int rc = GENIE_E_UNSUPPORTED;
    bool qkv_done = false;
    if (!c.qk_norm) {  // one-frame passes: LayerNorm inside the small GEMM's fragment path (no LayerNorm launch)
        const int r2 = launch_gemm16_sm_ln(2, x, d, lw.norm1_w, lw.norm1_b, 1e-5f, lw.spatial.qkv_w16, d, pw_qkv,
                                           c.qkv_bias ? lw.spatial.qkv_b : nullptr, nullptr, qkv, nullptr, 0, 3 * d, M, 3 * d, d,
                                           G16_OUTF32, 1.0f, st);
        if (r2 == GENIE_OK) qkv_done = true;
        else if (r2 != GENIE_E_UNSUPPORTED) return r2;
    }
    if (!qkv_done) {
        if (!c.qk_norm) {
            GENIE_TRY(launch_layer_norm_split(x, lw.norm1_w, lw.norm1_b, as, pd, M, d, 1e-5f, st));
            u = as;
        }
        rc = spatial_attention_fused(2, c, lw, u, pd, pw_qkv, w, B, as, pd, st);
    }
    if (rc == GENIE_E_UNSUPPORTED) {
    if (!qkv_done)
    GENIE_TRY(launch_gemm16<2>(u, d, pd, lw.spatial.qkv_w16, d, pw_qkv, c.qkv_bias ? lw.spatial.qkv_b : nullptr, qkv,
                               nullptr, 0, 3 * d, M, 3 * d, d, G16_OUTF32 | wsq, 1.0f, st));
    rc = launch_attn_spatial_split(qkv, nullptr, c.S, (long)B * c.T, d, c.num_heads, c.head_dim, c.attn_scale, nws,
                                   nbs, st, as, pd);
    if (rc == GENIE_E_UNSUPPORTED) {  // generic kernel writes f32 into x-sized scratch (logits region), then split
        float* tmp = w.logits;
        GENIE_TRY(launch_attn_generic(qkv, tmp, c.S, (long)B * c.T, 1, c.S, 0, 1, d, c.num_heads, c.head_dim,
                                      c.attn_scale, 0, nws, nbs, st));
        rc = launch_split_f16(tmp, as, pd, pd, st);
    }
    }
    GENIE_TRY(rc);
    GENIE_STUDY_CLASS(2);
    // the shipped geometry: temporal qkv Linear + attention as one kernel on the f32 rows of x (kernels_fused_f16x3.hip) -- the spatial
    // out-projection then need not write the split planes of x; in the prefix-cache passes the cache slice holds that kernel's k, v accumulators
    const bool fused_tq = w.frame_t < 0 && (!w.tqkv || w.tq_frames <= c.T) &&
                          temporal_qkv_attn_f16x3_takes(c, lw.temporal, B, w.model_T, w.tqkv || w.tcache);
    GENIE_TRY(launch_gemm16<2>(as, d, pd, lw.spatial.proj_w16, d, pw_proj, c.proj_bias ? lw.spatial.proj_b : nullptr, x,
                               xs, pd, d, M, d, d, G16_ACCUM | G16_OUTF32 | (fused_tq ? 0 : G16_OUT16) | wsp, 1.0f, st));
    // ---- temporal
    GENIE_STUDY_CLASS(1);
    float* tq = w.tqkv ? w.tqkv : qkv;
    if (fused_tq) {
        GENIE_TRY(launch_temporal_qkv_attn_f16x3(c, lw.temporal, x, as, (long)pd, w.tqkv ? w.tqkv : const_cast<float*>(w.tcache), B,
                                                 w.tqkv ? 1 : (w.tcache ? 2 : 0), w.tshift, w.model_T, st));
        if (w.stop_after_tqkv) return GENIE_OK;
        rc = GENIE_OK;
    } else if (w.frame_t >= 0) {  // single-frame decode: qkv -> cache slot frame_t, attend slots 0..frame_t
        float* slot = w.fcache + (size_t)w.frame_t * c.S * 3 * d;
        GENIE_TRY(launch_gemm16<2>(xs, d, pd, lw.temporal.qkv_w16, d, pw_qkv, c.qkv_bias ? lw.temporal.qkv_b : nullptr,
                                   slot, nullptr, 0, 3 * d, c.S, 3 * d, d, G16_OUTF32 | wtq, 1.0f, st, B, (long)c.S * d,
                                   (long)w.frame_T * c.S * 3 * d));
        rc = launch_attn_temporal_single(w.fcache, nullptr, B, w.frame_T, c.S, w.frame_t, d, c.num_heads, c.head_dim,
                                         c.attn_scale, nwt, nbt, st, as, pd);
    } else {
    const int Tq = (w.tqkv && w.tq_frames > c.T) ? w.tq_frames : c.T;  // frames per clip in tq's layout
    if (Tq != c.T && B > 1)  // a short clean pass into a longer cache: one GEMM batch entry per clip
        GENIE_TRY(launch_gemm16<2>(xs, d, pd, lw.temporal.qkv_w16, d, pw_qkv, c.qkv_bias ? lw.temporal.qkv_b : nullptr, tq,
                                   nullptr, 0, 3 * d, c.T * c.S, 3 * d, d, G16_OUTF32 | wtq, 1.0f, st, B, (long)c.T * c.S * d,
                                   (long)Tq * c.S * 3 * d));
    else
    GENIE_TRY(launch_gemm16<2>(xs, d, pd, lw.temporal.qkv_w16, d, pw_qkv, c.qkv_bias ? lw.temporal.qkv_b : nullptr, tq,
                               nullptr, 0, 3 * d, M, 3 * d, d, G16_OUTF32 | wtq, 1.0f, st));
    if (w.stop_after_tqkv) return GENIE_OK;
    if (w.tcache) {
        rc = launch_attn_temporal_prefix(tq, w.tcache, nullptr, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale,
                                         nwt, nbt, st, as, pd, w.tshift);
        if (rc == GENIE_E_UNSUPPORTED) {
            GENIE_TRY(launch_attn_temporal_prefix(tq, w.tcache, w.logits, B, c.T, c.S, d, c.num_heads, c.head_dim,
                                                  c.attn_scale, nwt, nbt, st, nullptr, 0, w.tshift));
            rc = launch_split_f16(w.logits, as, pd, pd, st);
        }
    } else {
        rc = launch_attn_temporal_f32_mfma(tq, nullptr, B, c.T, c.S, d, c.num_heads, c.head_dim, c.attn_scale, nwt, nbt,
                                           st, as, pd, Tq);
        if (rc == GENIE_E_UNSUPPORTED) {
            GENIE_CHECK_ARG(Tq == c.T || B == 1, "strided temporal qkv needs the MFMA temporal kernel (8 <= frames <= 16)");
            float* tmp = w.logits;
            GENIE_TRY(launch_attn_generic(tq, tmp, c.T, (long)B * c.S, c.S, (long)c.T * c.S, 1, c.S, d, c.num_heads,
                                          c.head_dim, c.attn_scale, 1, nwt, nbt, st));
            rc = launch_split_f16(tmp, as, pd, pd, st);
        }
    }
    }
    GENIE_TRY(rc);
    GENIE_STUDY_CLASS(3);
    GENIE_TRY(launch_gemm16<2>(as, d, pd, lw.temporal.proj_w16, d, pw_proj, c.proj_bias ? lw.temporal.proj_b : nullptr,
                               x, xs, pd, d, M, d, d, G16_ACCUM | G16_OUTF32 | (c.qk_norm ? G16_OUT16 : 0) | wtp, 1.0f, st));
    // ---- MLP
    GENIE_STUDY_CLASS(4);
    u = xs;
    bool fc1_done = false;
    if (!c.qk_norm) {
        const int r2 = launch_gemm16_sm_ln(2, x, d, lw.norm2_w, lw.norm2_b, 1e-5f, lw.fc1_w16, d, pw_fc, c.mlp_bias ? lw.fc1_b : nullptr,
                                           nullptr, nullptr, hs, ph, hid, M, hid, d, G16_GELU | G16_OUT16, 1.0f, st);
        if (r2 == GENIE_OK) fc1_done = true;
        else if (r2 != GENIE_E_UNSUPPORTED) return r2;
    }
    if (!fc1_done) {
    if (!c.qk_norm) {
        GENIE_TRY(launch_layer_norm_split(x, lw.norm2_w, lw.norm2_b, as, pd, M, d, 1e-5f, st));
        u = as;
    }
    GENIE_TRY(launch_gemm16<2>(u, d, pd, lw.fc1_w16, d, pw_fc, c.mlp_bias ? lw.fc1_b : nullptr, nullptr, hs, ph, hid, M,
                               hid, d, G16_GELU | G16_OUT16 | wf1, 1.0f, st));
    }
    GENIE_STUDY_CLASS(5);
    {
        const int rs = fc2_splitk2<2>(c, hs, (long)ph, lw.fc2_w16, (long)pw_fc, c.mlp_bias ? lw.fc2_b : nullptr, x, w, M, st, wf2 != 0);
        if (rs != GENIE_E_UNSUPPORTED) return rs;
    }
    GENIE_TRY(launch_gemm16<2>(hs, hid, ph, lw.fc2_w16, hid, pw_fc, c.mlp_bias ? lw.fc2_b : nullptr, x, xs, pd, d, M, d,
                               hid, G16_ACCUM | G16_OUTF32 | (w.skip_shadow_mlp ? 0 : G16_OUT16) | wf2, 1.0f, st));
    return GENIE_OK;
}

int prepare_f16x3(const genie_cfg& c, const float* x, Workspace& w, int B, hipStream_t st) {
    if (!c.qk_norm) return GENIE_OK;   // (as prepare_bf16: nothing reads the split planes of x before the first out-projection rewrites them)
    const size_t n = (size_t)B * c.T * c.S * c.d_model;
    return launch_split_f16(x, (uint16_t*)w.xn, n, n, st);
}

int readout_f16x3(const genie_cfg& c, const genie_weights& wt, const float* x, Workspace& w, int B, int t0, int t1,
                  int layout, float* logits, hipStream_t st) {
    GENIE_CHECK_ARG(wt.out_w16, "f16x3 precision needs split-f16 weights (genie_pack_split_f16)");
    const int d = c.d_model, nt = t1 - t0, V = c.factored_vocab * c.num_factored;
    const long rows = (long)nt * c.S;
    GENIE_CHECK_ARG((const void*)x == (const void*)w.x, "f16x3 readout reads the workspace's own hidden state");
    const size_t pd = (size_t)B * c.T * c.S * d;
    const uint16_t* xs = (const uint16_t*)w.xn;
    float* dst = (layout == GENIE_LAYOUT_TOKEN_MAJOR) ? logits : w.logits;
    GENIE_STUDY_CLASS(6);
    GENIE_TRY(launch_gemm16<2>(xs + (size_t)t0 * c.S * d, d, pd, wt.out_w16, d, (size_t)V * d, wt.out_b, dst, nullptr, 0,
                               V, (int)rows, V, d, G16_OUTF32 | (wt.out_w16_wide ? G16_WIDEW : 0), c.readout_mult, st, B, (long)c.T * c.S * d, rows * V));
    if (layout != GENIE_LAYOUT_TOKEN_MAJOR) GENIE_TRY(launch_transpose(w.logits, logits, B, (int)rows, V, st));
    return GENIE_OK;
}

int launch_pack_split(const float* src, uint16_t* dst, size_t n, hipStream_t st) { return launch_split_f16(src, dst, n, n, st); }

int launch_linear_lowp(int precision, const uint16_t* x16, const uint16_t* W16, const float* b, float* y, int M, int N,
                       int K, int gelu, int accumulate, hipStream_t st) {
    const int flags = G16_OUTF32 | (gelu ? G16_GELU : 0) | (accumulate ? G16_ACCUM : 0);
#ifdef GENIE_STUDY
    {   // leading-dimension study (tools/bench_gemm.py --pad-a / --pad-c allocate the padded buffers): does a power-of-two
        // row stride of the activation planes / of the output cost memory-channel conflicts?
        const int pa = study_env("GENIE_STUDY_LDA_PAD", 0), pc = study_env("GENIE_STUDY_LDC_PAD", 0);
        if (pa || pc) {
            const long lda = K + pa, ldc = N + pc;
            if (precision == GENIE_PREC_BF16)
                return launch_gemm16<1>(x16, lda, 0, W16, K, 0, b, y, nullptr, 0, ldc, M, N, K, flags, 1.0f, st);
            return launch_gemm16<2>(x16, lda, (size_t)M * lda, W16, K, (size_t)N * K, b, y, nullptr, 0, ldc, M, N, K, flags, 1.0f, st);
        }
    }
#endif
    if (precision == GENIE_PREC_BF16)
        return launch_gemm16<1>(x16, K, 0, W16, K, 0, b, y, nullptr, 0, N, M, N, K, flags, 1.0f, st);
    if (precision == GENIE_PREC_F16X3)
        return launch_gemm16<2>(x16, K, (size_t)M * K, W16, K, (size_t)N * K, b, y, nullptr, 0, N, M, N, K, flags, 1.0f,
                                st);
    set_error("linear_lowp: precision %d has no 16-bit GEMM", precision);
    return GENIE_E_UNSUPPORTED;
}

// bf16 x bf16 -> bf16 Linear (1x1 convolution on NHWC activations)
int launch_gemm_bf16_out16(const uint16_t* A16, const uint16_t* W16, const float* bias, uint16_t* C16, int M, int N, int K,
                           hipStream_t st) {
    // (no split-K kernel here: the MAGVIT2 stacks promise the same bytes for an image whatever else is in the batch, so the
    // accumulation order must not depend on the problem size)
    return launch_gemm16<1>(A16, K, 0, W16, K, 0, bias, nullptr, C16, 0, N, M, N, K, G16_OUT16, 1.0f, st, 1, 0, 0, nullptr, 0, true,
                            /*allow_sm=*/false);
}

}  // namespace genie
