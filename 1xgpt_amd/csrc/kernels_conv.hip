// MAGVIT2 decoder convolutions for gfx950: activations NHWC bf16, f32 accumulate.
//
//   conv3x3_igemm_kernel   3x3 / pad 1 / stride 1 as an implicit GEMM on v_mfma_f32_32x32x16_bf16:
//                          rows = output pixels (n*H*W), cols = C_out, K = 9*C_in ordered tap-major, so one
//                          64-wide K stage never straddles a tap (C_in % 64 == 0).  Same machinery as gemm16_v2:
//                          256x128 tile, 8 waves, 3-stage global_load_lds ring, counted vmcnt, XOR-swizzled LDS
//                          image (swizzle on the source address).  The A operand is gathered on the fly: lane ->
//                          (pixel, tap) -> neighbour pixel's channel slice; out-of-image taps read a zero page.
//                          Epilogue: + bias, + residual (ResBlock skip), bf16 NHWC store, optionally through the DCR
//                          depth-to-space permutation of the Upsampler (improved_model.py:185-237).
//   gn_stats / gn_swish    GroupNorm(32, eps 1e-6) statistics (f32 sums via atomics) and the fused normalise + affine +
//                          x*sigmoid(x) pass that writes the next conv's bf16 operand (improved_model.py:24-51).
//   conv_direct_kernel     small direct convolution for the two edge layers (C_in = 18 -> 512 and 128 -> 3).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { CONV_D2S = 1 };

// BK = 64, OCC = 1: three 48 KB stages, one workgroup per CU.  BK = 32, OCC = 2: three 24 KB stages and <= 128 registers, two
// workgroups per CU -- one's barrier / DMA waits are covered by the other's MFMAs.
template <int BK, int OCC>
__global__ __launch_bounds__(512, OCC) void conv3x3_igemm_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                              const float* __restrict__ bias,
                                                              const uint16_t* __restrict__ residual,
                                                              uint16_t* __restrict__ Y, const uint16_t* __restrict__ zero,
                                                              int n_img, int H, int Wd, int Cin, int Cout, int flags,
                                                              int stride, float* __restrict__ gn_part, int gn_cpg) {
    constexpr int BM = 256, BN = 128, NST = 3;
    constexpr int ROWB = BK * 2;         // bytes of one LDS row
    constexpr int SPR = ROWB / 16;       // 16-byte slots per row
    constexpr int RPC = 1024 / ROWB;     // rows per 1 KB DMA chunk
    constexpr int RPB = 2;
    constexpr int A_TILE = BM * ROWB, W_TILE = BN * ROWB, STAGE_B = A_TILE + W_TILE;  // 48 KB / 24 KB
    constexpr int ACH = A_TILE / 1024 / 8, WCH = W_TILE / 1024 / 8;  // 1 KB chunks per wave per stage: 4 + 2 / 2 + 1
    constexpr int CPW = ACH + WCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    // H, Wd: OUTPUT height/width; the input image is (H*stride, Wd*stride) (stride 2 = the encoder's downsample conv)
    const int Hin = H * stride, Win = Wd * stride;
    const int HW = H * Wd;
    const int M = n_img * HW;            // launcher: pixel counts (output and input) < 2^31
    const int K = 9 * Cin;
    const int mt = (M + BM - 1) / BM, nt = (Cout + BN - 1) / BN;
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
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;
    // pixel index -> (image, y, x): shifts when the image sides are powers of two (every MAGVIT2 level), 32-bit divisions
    // otherwise (the 64-bit divisions this replaces were a fifth of a 256^2 tile's time)
    const bool pow2 = !(Wd & (Wd - 1)) && !(HW & (HW - 1));
    const int sh_w = 31 - __builtin_clz(Wd), sh_hw = 31 - __builtin_clz(HW);
    auto pix_of = [&](int p, int& img, int& y, int& x) {
        if (pow2) {
            img = p >> sh_hw;
            const int in_img = p & (HW - 1);
            y = in_img >> sh_w;
            x = in_img & (Wd - 1);
        } else {
            img = (int)((unsigned)p / (unsigned)HW);
            const int in_img = p - img * HW;
            y = (int)((unsigned)in_img / (unsigned)Wd);
            x = in_img - y * Wd;
        }
    };

    // ---- staging descriptors.  A chunks c = wid + 8*i (i < ACH): rows c*RPC .. c*RPC+RPC-1; W chunks: wid + 8*j (j < WCH)
    int a_y[ACH], a_x[ACH], a_slot[ACH];
    int a_pix[ACH];  // INPUT pixel index under the centre tap of the lane's output pixel, -1 if the row is past M
    const int c_row = lane / SPR, c_phys = lane % SPR;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int row_local = (wid + 8 * i) * RPC + c_row;
        a_slot[i] = c_phys ^ ((row_local / RPB) % SPR);
        const int p = m0 + row_local;
        if (p >= M) { a_pix[i] = -1; a_y[i] = 0; a_x[i] = 0; continue; }
        int img, y, x;
        pix_of(p, img, y, x);
        a_y[i] = y * stride;   // input coordinates of the centre tap
        a_x[i] = x * stride;
        a_pix[i] = (img * Hin + a_y[i]) * Win + a_x[i];
    }
    const uint16_t* w_src[WCH];
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
        const int row_local = (wid + 8 * j) * RPC + c_row;
        const int slot = c_phys ^ ((row_local / RPB) % SPR);
        int rw = n0 + row_local;
        rw = rw < Cout ? rw : Cout - 1;
        w_src[j] = Wt + (size_t)rw * K + slot * 8;
    }
    auto stage = [&](int st, int k0) {
        const int tap = k0 / Cin, kc = k0 - tap * Cin;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        unsigned char* base = smem + st * STAGE_B;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = a_pix[i] >= 0 && yy >= 0 && yy < Hin && xx >= 0 && xx < Win;
            const uint16_t* src = ok ? X + (size_t)(a_pix[i] + dy * Win + dx) * Cin + kc + a_slot[i] * 8 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + (wid + 8 * i) * 1024), 16, 0,
                                             0);
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(w_src[j] + k0),
                (__attribute__((address_space(3))) void*)(base + A_TILE + (wid + 8 * j) * 1024), 16, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int rowA[2], rowB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { rowA[i] = wm * 64 + i * 32 + r; rowB[i] = wn * 64 + i * 32 + r; }
    auto frag_off = [&](int row_local, int kk) {
        const int slot = 2 * kk + h;
        return row_local * ROWB + ((slot ^ ((row_local / RPB) % SPR)) << 4);
    };
    const int nk = K / BK;
    stage(0, 0);
    if (nk > 1) stage(1, BK);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk && !(flags & 512)) stage((kt + 2) % NST, (kt + 2) * BK);
        if (flags & 1024) continue;
        const unsigned char* sa = smem + (kt % NST) * STAGE_B;
        const unsigned char* sw = sa + A_TILE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(sa + frag_off(rowA[i], kk));
                b[i] = *reinterpret_cast<const bf16x8*>(sw + frag_off(rowB[i], kk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue through LDS: rows = pixels, 8 consecutive output channels per lane (16-byte stores: 8 lanes cover the wave's
    // 64 columns, 8 rows per instruction); each wave transposes its 64x64 tile in two 32-row halves through an 8 KB slice
    // (64 KB for the workgroup: fits the BK = 32 form's 72 KB together with the GroupNorm partials)
    __syncthreads();
    if ((flags & 256) && acc[0][0][0] != 12345.f) return;  // study knobs (GENIE_CONV_ABL): 256 no epilogue, 512 no DMA, 1024 no MFMA
    float* ct = reinterpret_cast<float*>(smem) + wid * (32 * 64);
    const int c8 = (lane & 7) << 3;
    const int col = n0 + wn * 64 + c8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (bias && col < Cout) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    const bool d2s = flags & CONV_D2S;
    const int Cq = Cout >> 2;  // channels after depth-to-space (launcher: Cq % 8 == 0, so a lane's 8 channels share a sub-pixel)
    const int d2_grp = d2s ? col / Cq : 0, d2_c = d2s ? col - d2_grp * Cq : 0;
    // GroupNorm partials of this lane's channels 0-3 / 4-7 (each inside one group: cpg % 4 == 0) over its 8 rows
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) ct[((e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r] = acc[i][j][e];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3);
            const int p = m0 + wm * 64 + i * 32 + rl;
            const float4 v0 = *reinterpret_cast<const float4*>(ct + rl * 64 + c8);
            const float4 v1 = *reinterpret_cast<const float4*>(ct + rl * 64 + c8 + 4);
            if (p >= M || col >= Cout) continue;
            float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3],
                          v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
            size_t oidx;
            if (!d2s) {
                oidx = (size_t)p * Cout + col;
            } else {  // DCR: conv channel (i*2 + j)*Cq + c -> pixel (2y+i, 2x+j), channel c
                int img, y, x;
                pix_of(p, img, y, x);
                oidx = (((size_t)img * (2 * H) + (2 * y + (d2_grp >> 1))) * (2 * Wd) + (2 * x + (d2_grp & 1))) * Cq + d2_c;
            }
            if (residual) {
                const uint4 rr = *reinterpret_cast<const uint4*>(residual + oidx);
                const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf16_to_f32((uint16_t)(rw[e] & 0xFFFF));
                    v[2 * e + 1] += bf16_to_f32((uint16_t)(rw[e] >> 16));
                }
            }
            uint4 pk;
            pk.x = f32x2_to_bf16x2(v[0], v[1]);
            pk.y = f32x2_to_bf16x2(v[2], v[3]);
            pk.z = f32x2_to_bf16x2(v[4], v[5]);
            pk.w = f32x2_to_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(Y + oidx) = pk;
            if (gn_part) {  // statistics of the STORED (bf16-rounded) tensor, like a separate pass over Y would see
                const uint32_t pw[4] = {pk.x, pk.y, pk.z, pk.w};
                float q[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[2 * e] = __uint_as_float(pw[e] << 16);
                    q[2 * e + 1] = __uint_as_float(pw[e] & 0xFFFF0000u);
                }
                gs0 += (q[0] + q[1]) + (q[2] + q[3]);
                gq0 += (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]);
                gs1 += (q[4] + q[5]) + (q[6] + q[7]);
                gq1 += (q[4] * q[4] + q[5] * q[5]) + (q[6] * q[6] + q[7] * q[7]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the second half
    }
    if (gn_part) {
        // ---- fixed-order reduction of the tile's GroupNorm partials: every lane parks its two (sum, sumsq) pairs behind the
        // transpose slices, then one thread per (local group, moment) adds its contributors in (wave row, row group, channel
        // quad) order.  The tile's 128 columns hold 128 / cpg whole groups (launcher: 128 % cpg == 0; with depth-to-space the
        // 128 columns lie inside one sub-pixel block, so col - n0 indexes the GroupNorm channel relative to the tile as well).
        float* ps = reinterpret_cast<float*>(smem + 8 * 32 * 64 * 4);  // [8 waves][64 lanes][4]
        *reinterpret_cast<float4*>(ps + (wid * 64 + lane) * 4) = make_float4(gs0, gq0, gs1, gq1);
        __syncthreads();
        const int ngrp = BN / gn_cpg;
        if (tid < ngrp * 2) {
            const int g = tid >> 1, which = tid & 1;
            const int c_lo = g * gn_cpg;                 // first tile column of the group
            const int wn_g = c_lo >> 6;                  // the wave column that owns it
            const int q_lo = (c_lo & 63) >> 2, q_n = gn_cpg >> 2;  // channel quads [q_lo, q_lo + q_n): lane & 7 = q >> 1, pair q & 1
            float a = 0.f;
            for (int wmi = 0; wmi < 4; ++wmi)
                for (int rg = 0; rg < 8; ++rg)
                    for (int q = q_lo; q < q_lo + q_n; ++q)
                        a += ps[(((wmi * 2 + wn_g) * 64) + rg * 8 + (q >> 1)) * 4 + (q & 1) * 2 + which];
            gn_part[((size_t)m_tile * nt + n_tile) * 64 + tid] = a;
        }
    }
}

// ---- stride-1 form with the three horizontal taps served from ONE LDS slab.
// A tile of 256 consecutive pixels needs, for a vertical tap dy and a 64-channel chunk, the pixels -1 .. 256 of the row(s) dy
// away: the taps dx = -1, 0, +1 are the same slab read one LDS row up or down.  So the K loop runs (dy, chunk) slabs of 33 KB
// (256 pixels + one halo chunk) with three weight stages each, instead of nine 32 KB A tiles per chunk: the LDS-DMA bytes of
// a 128-column tile drop from 864 KB to 486 KB per 256 x 128 x 1152 tile (the DMA path alone took 0.4 of the kernel's time,
// tools/gpu_conv_abl.sh).  Rows whose horizontal neighbour lies outside the image get their fragment zeroed in registers.
//   slab rows: L = 0: pixel -1, L = 1: pixel 256, L = 8 + q: pixel q of the tile; XOR swizzle keyed on L.
//   BN = 128: 4 x 2 waves of 64 x 64;  BN = 32: 8 x 1 waves of 32 x 32 for the narrow head (conv_out, 3 -> 8 padded columns).
typedef float cv_f4 __attribute__((ext_vector_type(4)));

template <int BN>
__global__ __launch_bounds__(512, 1) void conv3x3_slab_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                              const float* __restrict__ bias,
                                                              const uint16_t* __restrict__ residual, uint16_t* __restrict__ Y,
                                                              const uint16_t* __restrict__ zero, int n_img, int H, int Wd, int Cin,
                                                              int Cout, int flags, float* __restrict__ gn_part, int gn_cpg) {
    constexpr int BM = 256, BK = 64, ROWB = 128, SPR = 8, RPB = 2;
    constexpr int TI = BN == 128 ? 2 : 1, TJ = BN == 128 ? 2 : 1;  // 32x32 MFMA tiles per wave
    constexpr int WCOLS = TJ * 32;                                 // columns of a wave's tile
    constexpr int SLAB_B = (8 + BM) * ROWB;                        // 33 KB
    constexpr int WROWS = BN == 128 ? 128 : 64;                    // BN = 32: 64 rows staged so every wave issues one chunk
    constexpr int W_TILE = WROWS * ROWB, WCH = W_TILE / 1024 / 8;  // 2 / 1 chunks per wave per stage
    constexpr int CT_OFF = 2 * SLAB_B + 3 * W_TILE;                // epilogue scratch behind the rings: 8 waves x 16 rows
    constexpr int PS_OFF = CT_OFF + 8 * 16 * WCOLS * 4;            // GroupNorm partials [8 waves][64 lanes][4]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_off = BN == 128 ? (wid >> 1) * 64 : wid * 32, wn_off = BN == 128 ? (wid & 1) * 64 : 0;
    const int r = lane & 31, h = lane >> 5;
    const int HW = H * Wd;
    const int M = n_img * HW;
    const int K = 9 * Cin;
    const int mt = (M + BM - 1) / BM, nt = (Cout + BN - 1) / BN;
    const int ntiles = mt * nt;
    const int full = (mt / 8) * 8 * nt;
    auto tile_of = [&](int bid, int& m_tile, int& n_tile) {  // 8 consecutive row tiles (one per XCD) share a column tile
        if (bid < full) {
            const int grp = bid / (8 * nt), rem = bid - grp * 8 * nt;
            m_tile = grp * 8 + (rem & 7);
            n_tile = rem >> 3;
        } else {
            const int rem = bid - full;
            m_tile = (mt / 8) * 8 + rem / nt;
            n_tile = rem % nt;
        }
    };
    const bool pow2 = !(Wd & (Wd - 1)) && !(HW & (HW - 1));
    const int sh_w = 31 - __builtin_clz(Wd), sh_hw = 31 - __builtin_clz(HW);
    auto pix_of = [&](int p, int& img, int& y, int& x) {
        if (pow2) {
            img = p >> sh_hw;
            const int in_img = p & (HW - 1);
            y = in_img >> sh_w;
            x = in_img & (Wd - 1);
        } else {
            img = (int)((unsigned)p / (unsigned)HW);
            const int in_img = p - img * HW;
            y = (int)((unsigned)in_img / (unsigned)Wd);
            x = in_img - y * Wd;
        }
    };

    // ---- staging descriptors of a tile.  Main slab chunks c = wid + 8*i (i < 4): pixels c*8 .. c*8+7 at L = 8 + pixel; wave 0
    // also loads the halo chunk (L = 0: pixel -1, L = 1: pixel 256, L = 2..7 zero).  pix = -1: no source (zero page).
    const int c_row = lane >> 3, c_phys = lane & 7;
    int a_pix[4], a_y[4], a_slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_slot[i] = c_phys ^ (((8 + (wid + 8 * i) * 8 + c_row) / RPB) % SPR);
    int h_pix = -1, h_y = 0;
    const int h_slot = c_phys ^ ((c_row / RPB) % SPR);
    const uint16_t* w_src[WCH];
    auto setup_loads = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = m0 + (wid + 8 * i) * 8 + c_row;
            a_pix[i] = -1; a_y[i] = 0;
            if (p < M) {
                int img, x;
                pix_of(p, img, a_y[i], x);
                a_pix[i] = p;
            }
        }
        h_pix = -1; h_y = 0;
        if (wid == 0 && c_row < 2) {
            const int p = c_row == 0 ? m0 - 1 : m0 + BM;
            if (p >= 0 && p < M) {
                int img, x;
                pix_of(p, img, h_y, x);
                h_pix = p;
            }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int row_local = (wid + 8 * j) * 8 + c_row;
            const int slot = c_phys ^ ((row_local / RPB) % SPR);
            int rw = n0 + row_local;
            rw = rw < Cout ? rw : Cout - 1;
            w_src[j] = Wt + (size_t)rw * K + slot * 8;
        }
    };
    const int nchunk = Cin / BK, nslab = 3 * nchunk, nk = 3 * nslab;
    unsigned char* const wbase = smem + 2 * SLAB_B;
    auto load_slab = [&](int a) {
        const int dyi = a / nchunk, kc = (a - dyi * nchunk) * BK, dy = dyi - 1;
        unsigned char* base = smem + (a & 1) * SLAB_B;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = a_y[i] + dy;
            const bool ok = a_pix[i] >= 0 && yy >= 0 && yy < H;
            const uint16_t* src = ok ? X + (size_t)(a_pix[i] + dy * Wd) * Cin + kc + a_slot[i] * 8 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + (1 + wid + 8 * i) * 1024), 16, 0,
                                             0);
        }
        if (wid == 0) {
            const int yy = h_y + dy;
            const bool ok = h_pix >= 0 && yy >= 0 && yy < H;
            const uint16_t* src = ok ? X + (size_t)(h_pix + dy * Wd) * Cin + kc + h_slot * 8 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)base, 16, 0, 0);
        }
    };
    auto load_w = [&](int s) {
        const int a = s / 3, d = s - 3 * a;
        const int dyi = a / nchunk, kc = (a - dyi * nchunk) * BK;
        const int k0 = (dyi * 3 + d) * Cin + kc;
        unsigned char* base = wbase + d * W_TILE;
#pragma unroll
        for (int j = 0; j < WCH; ++j)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(w_src[j] + k0),
                (__attribute__((address_space(3))) void*)(base + (wid + 8 * j) * 1024), 16, 0, 0);
    };

    // fragment addresses (the same for every tile): the lane's pixel rows under the three horizontal taps
    int aoff[TI][3], akey[TI][3];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int q = wm_off + i * 32 + r + d - 1;
            const int L = q < 0 ? 0 : (q >= BM ? 1 : q + 8);
            aoff[i][d] = L * ROWB;
            akey[i][d] = (L / RPB) % SPR;
        }
    int boff[TJ], bkey[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int rowB = wn_off + j * 32 + r;
        boff[j] = rowB * ROWB;
        bkey[j] = (rowB / RPB) % SPR;
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool d2s = flags & CONV_D2S;
    const int Cq = Cout >> 2;
    constexpr int LPR = WCOLS / 8, RPI = 64 / LPR;  // epilogue: lanes per row, rows per iteration
    const int c8 = (lane % LPR) << 3;
    float* const ct = reinterpret_cast<float*>(smem + CT_OFF) + wid * (16 * WCOLS);

    // ---- persistent: workgroup b walks tiles b, b + grid, ...; the next tile's first slab and weight stages are requested
    // before the epilogue, whose stores then drain under the next tile's main loop
    int tile = blockIdx.x, m_tile, n_tile;
    tile_of(tile, m_tile, n_tile);
    setup_loads(m_tile * BM, n_tile * BN);
    load_slab(0);
    load_w(0);
    if (nk > 1) load_w(1);
    for (;;) {
        const int m0 = m_tile * BM, n0 = n_tile * BN;
        const int cur_m_tile = m_tile, cur_n_tile = n_tile;
        bool okL[TI], okR[TI];  // is the horizontal neighbour inside the image?
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            int img, y, x;
            pix_of(m0 + wm_off + i * 32 + r, img, y, x);
            okL[i] = x > 0;
            okR[i] = x < Wd - 1;
        }
        f32x16 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // The tile's bias is requested HERE and consumed (as far as the compiler can tell) right behind the first stage's vmcnt(0),
        // where everything is drained anyway: an ordinary load whose first use sits in the epilogue makes the compiler put an
        // s_waitcnt vmcnt(0) there -- and drain the next tile's first slab and weight stages, requested just before the epilogue.
        const int col = n0 + wn_off + c8;
        cv_f4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = bq0;
        if (bias && col < Cout) {
            bq0 = *reinterpret_cast<const cv_f4*>(bias + col);
            bq1 = *reinterpret_cast<const cv_f4*>(bias + col + 4);
        }
        // ResBlock skip: ALL residual rows this lane will add are requested together, right after the main loop, and are complete at
        // the __syncthreads() that ends the tile (it drains vmcnt anyway): ONE memory round trip per tile, and no ordinary load is
        // outstanding in the epilogue.  Requested inside the store loop -- as before -- each one was an exposed round trip behind the
        // next tile's 40-50 KB prefetch, one per 8 / 16 rows, and its vmcnt(0) also waited for the previous rows' stores.  (Requested
        // inside the main loop's last stage instead, the ordinary loads made the compiler wait vmcnt(0) all over the K loop.)
        constexpr int NRES = TI * 2 * (16 / RPI);
        typedef unsigned int cv_u4 __attribute__((ext_vector_type(4)));
        cv_u4 resv[NRES];
        auto out_index = [&](int p) -> size_t {
            if (!d2s) return (size_t)p * Cout + col;
            int img, y, x;   // DCR: conv channel (i*2 + j)*Cq + c -> pixel (2y+i, 2x+j), channel c
            pix_of(p, img, y, x);
            const int d2_grp = col / Cq, d2_c = col - d2_grp * Cq;
            return (((size_t)img * (2 * H) + (2 * y + (d2_grp >> 1))) * (2 * Wd) + (2 * x + (d2_grp & 1))) * Cq + d2_c;
        };
        for (int a = 0; a < nslab; ++a) {
            const unsigned char* sa = smem + (a & 1) * SLAB_B;
            const bool more = a + 1 < nslab;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int s = 3 * a + d;
                // outstanding DMAs younger than W(s): W(s+1) and, behind stage d = 0, the next slab (4 chunks, 5 for wave 0);
                // the first stage of a tile also waits for the previous tile's epilogue stores
                if (s + 1 >= nk || s == 0) {            // everything, including the bias at the top of a tile
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (s == 0) asm volatile("" : "+v"(bq0), "+v"(bq1));   // the bias has landed (see above)
                } else if (d == 1 && more) {
                    if (wid == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCH + 5) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCH + 4) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCH) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (!(flags & 512)) {
                    if (d == 0 && more) load_slab(a + 1);
                    if (s + 2 < nk) load_w(s + 2);
                }
                if (flags & 1024) continue;
                const unsigned char* sw = wbase + d * W_TILE;
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk) {
                    const int slot = 2 * kk + h;
                    bf16x8 av[TI], bv[TJ];
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        av[i] = *reinterpret_cast<const bf16x8*>(sa + aoff[i][d] + ((slot ^ akey[i][d]) << 4));
                        if (d == 0) av[i] = okL[i] ? av[i] : zero8;
                        if (d == 2) av[i] = okR[i] ? av[i] : zero8;
                    }
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        bv[j] = *reinterpret_cast<const bf16x8*>(sw + boff[j] + ((slot ^ bkey[j]) << 4));
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if (residual) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh)
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int p = m0 + wm_off + i * 32 + eh * 16 + it * RPI + lane / LPR;
                        cv_u4 z = {0u, 0u, 0u, 0u};
                        resv[(i * 2 + eh) * (16 / RPI) + it] =
                            (p < M && col < Cout) ? *reinterpret_cast<const cv_u4*>(residual + out_index(p)) : z;
                    }
        }
        __syncthreads();  // every wave is done with this tile's slabs and weight stages
        if (has_next) {
            tile_of(next, m_tile, n_tile);
            setup_loads(m_tile * BM, n_tile * BN);
            load_slab(0);
            load_w(0);
            if (nk > 1) load_w(1);
        }
        // ---- epilogue through LDS: each wave transposes its tile 16 rows at a time through a 4 KB slice; 8 consecutive output
        // channels per lane, 16-byte stores
        if (!((flags & 256) && acc[0][0][0] != 12345.f)) {
            const float bvv[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
            float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            ct[((e & 3) + 8 * (e >> 2) + 4 * h) * WCOLS + j * 32 + r] = acc[i][j][eh * 8 + e];
#pragma unroll
                    for (int it = 0; it < 16 / RPI; ++it) {
                        const int rl = it * RPI + lane / LPR;
                        const int p = m0 + wm_off + i * 32 + eh * 16 + rl;
                        // (ext_vector LDS reads: HIP's float4 struct would make the compiler wait vmcnt(0) here, for the prefetch)
                        const cv_f4 v0 = *reinterpret_cast<const cv_f4*>(ct + rl * WCOLS + c8);
                        const cv_f4 v1 = *reinterpret_cast<const cv_f4*>(ct + rl * WCOLS + c8 + 4);
                        if (p >= M || col >= Cout) continue;
                        float v[8] = {v0.x + bvv[0], v0.y + bvv[1], v0.z + bvv[2], v0.w + bvv[3],
                                      v1.x + bvv[4], v1.y + bvv[5], v1.z + bvv[6], v1.w + bvv[7]};
                        const size_t oidx = out_index(p);
                        if (residual) {
                            const cv_u4 rr = resv[(i * 2 + eh) * (16 / RPI) + it];
                            const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += bf16_to_f32((uint16_t)(rw[e] & 0xFFFF));
                                v[2 * e + 1] += bf16_to_f32((uint16_t)(rw[e] >> 16));
                            }
                        }
                        uint4 pk;
                        pk.x = f32x2_to_bf16x2(v[0], v[1]);
                        pk.y = f32x2_to_bf16x2(v[2], v[3]);
                        pk.z = f32x2_to_bf16x2(v[4], v[5]);
                        pk.w = f32x2_to_bf16x2(v[6], v[7]);
                        *reinterpret_cast<uint4*>(Y + oidx) = pk;
                        if (BN == 128 && gn_part) {  // statistics of the STORED (bf16-rounded) tensor
                            const uint32_t pw[4] = {pk.x, pk.y, pk.z, pk.w};
                            float q[8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                q[2 * e] = __uint_as_float(pw[e] << 16);
                                q[2 * e + 1] = __uint_as_float(pw[e] & 0xFFFF0000u);
                            }
                            gs0 += (q[0] + q[1]) + (q[2] + q[3]);
                            gq0 += (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]);
                            gs1 += (q[4] + q[5]) + (q[6] + q[7]);
                            gq1 += (q[4] * q[4] + q[5] * q[5]) + (q[6] * q[6] + q[7] * q[7]);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next 16 rows
                }
            if (BN == 128 && gn_part) {  // same fixed-order reduction and partial layout as conv3x3_igemm_kernel
                float* ps = reinterpret_cast<float*>(smem + PS_OFF);
                *reinterpret_cast<cv_f4*>(ps + (wid * 64 + lane) * 4) = cv_f4{gs0, gq0, gs1, gq1};
                // (not __syncthreads(): with LDS-DMA in flight it carries an s_waitcnt vmcnt(0) -- the next tile's prefetch AND this
                // tile's output stores would be drained here instead of under the next main loop)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int ngrp = BN / gn_cpg;
                if (tid < ngrp * 2) {
                    const int g = tid >> 1, which = tid & 1;
                    const int c_lo = g * gn_cpg;
                    const int wn_g = c_lo >> 6;
                    const int q_lo = (c_lo & 63) >> 2, q_n = gn_cpg >> 2;
                    float a = 0.f;
                    for (int wmi = 0; wmi < 4; ++wmi)
                        for (int rg = 0; rg < 8; ++rg)
                            for (int q = q_lo; q < q_lo + q_n; ++q)
                                a += ps[(((wmi * 2 + wn_g) * 64) + rg * 8 + (q >> 1)) * 4 + (q & 1) * 2 + which];
                    gn_part[((size_t)cur_m_tile * nt + cur_n_tile) * 64 + tid] = a;
                }
            }
        }
        if (!has_next) break;
        tile = next;
    }
}

// per-tile (sum, sumsq) partials written by conv3x3_igemm_kernel -> (mean, rstd) per (image, group).  One wave per (image, group):
// lane l adds tiles l, l + 64, ... (f64), then a butterfly over the 64 lanes -- a fixed order, so the statistics are
// bit-reproducible; the 256 tiles x 4 sub-pixel blocks of a 256^2 layer were a 1,024-long dependent chain per thread before.
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                                int n_img, int groups, int tiles_per_img, int nt, int cpg, int Cq,
                                                                int d2s, float cnt, float eps) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_img * groups) return;
    const int img = i / groups, g = i - img * groups;
    double s = 0.0, q = 0.0;
    const int nsub = d2s ? 4 : 1;
    for (int f = lane; f < nsub * tiles_per_img; f += 64) {
        const int sub = f / tiles_per_img, j = f - sub * tiles_per_img;
        const int colb = sub * Cq + g * cpg;  // conv output column of the group's first channel
        const int n_tile = colb >> 7, lg = (colb & 127) / cpg;
        const float* pp = part + ((size_t)(img * tiles_per_img + j) * nt + n_tile) * 64 + lg * 2;
        s += (double)pp[0];
        q += (double)pp[1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if (lane) return;
    const double mean = s / cnt;
    const double var = fmax(q / cnt - mean * mean, 0.0);
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// gn_part != NULL: also write the GroupNorm(gn_groups) partial sums of the output (per 256x128 tile: [64] floats =
// (sum, sumsq) per local group), to be finalised by launch_gn_swish_tiles.  Needs whole images per 256-pixel tile row block
// (H*W % 256 == 0), whole groups per 128-column tile and C_out % 128 == 0; GENIE_E_UNSUPPORTED otherwise (nothing launched).
int launch_conv3x3_igemm(const uint16_t* X, const uint16_t* Wt, const float* bias, const uint16_t* residual, uint16_t* Y,
                         const uint16_t* zero_page, int n_img, int H, int Wd, int Cin, int Cout, int d2s, hipStream_t st,
                         int stride, float* gn_part, int gn_groups) {
    GENIE_CHECK_SHAPE(stride == 1 || (stride == 2 && !d2s), "conv3x3_igemm: stride %d unsupported", stride);
    GENIE_CHECK_SHAPE(Cin % 64 == 0 && Cout % 8 == 0, "conv3x3_igemm: C_in %% 64 and C_out %% 8 required (got %d, %d)", Cin,
                      Cout);
    GENIE_CHECK_SHAPE(!d2s || (Cout % 32 == 0), "conv3x3_igemm: depth-to-space needs C_out %% 32 == 0");
    const long M = (long)n_img * H * Wd;
    if (M <= 0) return GENIE_OK;
    GENIE_CHECK_SHAPE(M * stride * stride < (1L << 31) - 4096, "conv3x3_igemm: %ld pixels exceed the 32-bit pixel index", M);
    const int mt = (int)((M + 255) / 256), nt = (Cout + 127) / 128;
    int gn_cpg = 0;
    if (gn_part) {
        const int Cgn = d2s ? Cout / 4 : Cout;
        if (gn_groups <= 0 || Cgn % gn_groups) return GENIE_E_UNSUPPORTED;
        gn_cpg = Cgn / gn_groups;
        if (((long)H * Wd) % 256 || Cout % 128 || gn_cpg % 4 || 128 % gn_cpg || 128 / gn_cpg > 32 || (d2s && Cgn % 128))
            return GENIE_E_UNSUPPORTED;
    }
    ProfScope prof(GENIE_KC_OTHER, 2.0 * M * Cout * 9.0 * Cin, 2.0 * (M * (double)Cin + M * (double)Cout + 9.0 * Cin * Cout),
                   st);
    static const int bk = study_env("GENIE_CONV_BK", 64);
    static const int abl = study_env("GENIE_CONV_ABL", 0);
    static const int slab = study_env("GENIE_CONV_SLAB", 1);
    if (slab && stride == 1) {
        const int fl = (d2s ? CONV_D2S : 0) | abl;
        static const bool persist = study_env("GENIE_CONV_PERSIST", 1) != 0;  // 0: one workgroup per tile (A/B runs)
        const int n_cu = persist ? device_cu_count() : 1 << 30;               // (per device: common.hpp)
        if (Cout <= 32 && !gn_part) {
            const size_t lds = 2 * 33 * 1024 + 3 * 8 * 1024 + 8 * 16 * 32 * 4;
            const int tiles = mt * ((Cout + 31) / 32);
            (void)hipFuncSetAttribute((const void*)conv3x3_slab_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            conv3x3_slab_kernel<32><<<tiles < n_cu ? tiles : n_cu, 512, lds, st>>>(X, Wt, bias, residual, Y, zero_page, n_img, H, Wd,
                                                                                  Cin, Cout, fl, nullptr, 0);
        } else {
            const size_t lds = 2 * 33 * 1024 + 3 * 16 * 1024 + 8 * 16 * 64 * 4 + 8 * 64 * 4 * 4;
            const int tiles = mt * nt;
            (void)hipFuncSetAttribute((const void*)conv3x3_slab_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            conv3x3_slab_kernel<128><<<tiles < n_cu ? tiles : n_cu, 512, lds, st>>>(X, Wt, bias, residual, Y, zero_page, n_img, H, Wd,
                                                                                   Cin, Cout, fl, gn_part, gn_cpg);
        }
        GENIE_LAUNCH_CHECK("conv3x3_slab");
        return GENIE_OK;
    }
    if (bk == 32) {
        const size_t lds = 3 * 24 * 1024;
        (void)hipFuncSetAttribute((const void*)conv3x3_igemm_kernel<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv3x3_igemm_kernel<32, 2><<<mt * nt, 512, lds, st>>>(X, Wt, bias, residual, Y, zero_page, n_img, H, Wd, Cin, Cout,
                                                               (d2s ? CONV_D2S : 0) | abl, stride, gn_part, gn_cpg);
    } else {
        const size_t lds = 3 * 48 * 1024;
        (void)hipFuncSetAttribute((const void*)conv3x3_igemm_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv3x3_igemm_kernel<64, 1><<<mt * nt, 512, lds, st>>>(X, Wt, bias, residual, Y, zero_page, n_img, H, Wd, Cin, Cout,
                                                               (d2s ? CONV_D2S : 0) | abl, stride, gn_part, gn_cpg);
    }
    GENIE_LAUNCH_CHECK("conv3x3_igemm");
    return GENIE_OK;
}

// ---- GroupNorm(32) statistics, order-fixed (bit-reproducible: no atomics anywhere).  block = 256 threads over a slab of
// pixels of one image; thread t owns a fixed 8-channel slice, so its group(s) are fixed.  Per block: every thread parks its
// (sum, sumsq) partials in LDS and 2*groups reducer threads add the contributors of their (group, moment) in thread order;
// the block's partials go to part[(img, block)][groups][2] and gn_finalize adds the blocks of an image in block order (f64).
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ X, float* __restrict__ part, int HW, int C,
                                                       int groups, int pix_per_block) {
    __shared__ float ps[256][4];  // per thread: s0, q0 (first half of its chunk), s1, q1 (second half)
    const int img = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, HW);
    const int cpg = C / groups;
    const int c8 = C >> 3;                  // 16-byte chunks per pixel; 256 % c8 == 0 (checked by the launcher)
    const int chunk = threadIdx.x % c8;     // fixed channel slice of this thread -> fixed group(s)
    const int prow = threadIdx.x / c8, pstep = blockDim.x / c8;
    const uint16_t* base = X + ((size_t)img * HW) * C + chunk * 8;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;  // first / second half of the chunk (cpg == 4: two groups)
    for (int p = p0 + prow; p < p1; p += pstep) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)p * C);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float a = bf16_to_f32((uint16_t)(w[k] & 0xFFFF)), b = bf16_to_f32((uint16_t)(w[k] >> 16));
            s0 += a + b; q0 += a * a + b * b;
            const float c = bf16_to_f32((uint16_t)(w[2 + k] & 0xFFFF)), d = bf16_to_f32((uint16_t)(w[2 + k] >> 16));
            s1 += c + d; q1 += c * c + d * d;
        }
    }
    ps[threadIdx.x][0] = s0; ps[threadIdx.x][1] = q0; ps[threadIdx.x][2] = s1; ps[threadIdx.x][3] = q1;
    __syncthreads();
    if ((int)threadIdx.x < groups * 2) {
        const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
        float acc = 0.f;
        for (int t = 0; t < 256; ++t) {
            const int ch = t % c8;
            if ((ch * 8) / cpg == g) acc += ps[t][which];
            if ((ch * 8 + 4) / cpg == g) acc += ps[t][2 + which];
        }
        part[(((size_t)img * gridDim.x + blockIdx.x) * groups) * 2 + threadIdx.x] = acc;
    }
}

// per-block (sum, sumsq) partials -> (mean, rstd), one thread per (image, group), blocks added in order
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int n, int groups, int nblk,
                                   float cnt, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int img = i / groups, g = i - img * groups;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const float* p = part + (((size_t)img * nblk + b) * groups + g) * 2;
        s += (double)p[0];
        q += (double)p[1];
    }
    const double mean = s / cnt;
    const double var = fmax(q / cnt - mean * mean, 0.0);
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = swish(gn(x)) as bf16 NHWC (apply_swish = 0: GroupNorm only); stats hold (mean, rstd)
__global__ void gn_swish_kernel(const uint16_t* __restrict__ X, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, uint16_t* __restrict__ Y,
                                long n_chunks, int HW, int C, int groups, int apply_swish) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_chunks) return;
    const int c8 = C >> 3, cpg = C / groups;
    const int chunk = (int)(idx % c8);
    const long pix = idx / c8;
    const int img = (int)(pix / HW);
    const uint4 v = *reinterpret_cast<const uint4*>(X + idx * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const int c0 = chunk * 8;
    const float* st0 = stats + ((size_t)img * groups + c0 / cpg) * 2;
    const float* st1 = stats + ((size_t)img * groups + (c0 + 4) / cpg) * 2;
    const float m0 = st0[0], r0 = st0[1], m1 = st1[0], r1 = st1[1];
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0), gb = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 ba = *reinterpret_cast<const float4*>(beta + c0), bb = *reinterpret_cast<const float4*>(beta + c0 + 4);
    const float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    const float be[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float r2[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int j = 2 * k + hh;
            const float x = bf16_to_f32((uint16_t)(hh ? (w[k] >> 16) : (w[k] & 0xFFFF)));
            float y = (x - (j < 4 ? m0 : m1)) * (j < 4 ? r0 : r1) * gg[j] + be[j];
            if (apply_swish) y = y / (1.0f + __expf(-y));
            r2[hh] = y;
        }
        o[k] = (uint32_t)f32_to_bf16(r2[0]) | ((uint32_t)f32_to_bf16(r2[1]) << 16);
    }
    *reinterpret_cast<uint4*>(Y + idx * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// The same pass with the per-channel and per-group constants hoisted: a block covers a run of pixels of ONE image, thread t owns
// the fixed 8-channel slice t % (C/8) and walks the pixels t / (C/8), + 256/(C/8), ...  (the kernel above spends most of its
// instructions on 64-bit index divisions and a full-precision division per element: 3.8 TB/s; this one is a plain stream).
__global__ __launch_bounds__(256) void gn_swish_rows_kernel(const uint16_t* __restrict__ X, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            uint16_t* __restrict__ Y, int HW, int C, int groups, int ppb,
                                                            int apply_swish) {
    const int c8 = C >> 3, cpg = C / groups;
    const int chunk = threadIdx.x % c8, prow = threadIdx.x / c8, pstep = 256 / c8;
    const int img = blockIdx.y;
    const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, HW);
    const int c0 = chunk * 8;
    const float* st0 = stats + ((size_t)img * groups + c0 / cpg) * 2;
    const float* st1 = stats + ((size_t)img * groups + (c0 + 4) / cpg) * 2;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0), gb = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 ba = *reinterpret_cast<const float4*>(beta + c0), bb = *reinterpret_cast<const float4*>(beta + c0 + 4);
    // y = (x - m) * r * g + b  ==  x * sc + sh is NOT used: the product order of the kernel above is kept
    const float mean[2] = {st0[0], st1[0]}, rstd[2] = {st0[1], st1[1]};
    const float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    const float be[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
    const size_t base = ((size_t)img * HW) * C + c0;
#pragma unroll 2
    for (int p = p0 + prow; p < p1; p += pstep) {
        const uint4 v = *reinterpret_cast<const uint4*>(X + base + (size_t)p * C);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int j = 2 * k + hh;
                const float x = __uint_as_float(hh ? (w[k] & 0xFFFF0000u) : (w[k] << 16));
                float y = (x - mean[j >> 2]) * rstd[j >> 2] * gg[j] + be[j];
                if (apply_swish) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                r2[hh] = y;
            }
            o[k] = f32x2_to_bf16x2(r2[0], r2[1]);
        }
        *reinterpret_cast<uint4*>(Y + base + (size_t)p * C) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
static int launch_gn_apply(const uint16_t* X, const float* stats, const float* gamma, const float* beta, uint16_t* Y, int n_img,
                           int HW, int C, int groups, int apply_swish, hipStream_t st) {
    const int c8 = C / 8;
    static const int rows = study_env("GENIE_GN_ROWS", 1);
    if (rows && c8 <= 256 && 256 % c8 == 0) {
        const int ppb = 8 * (256 / c8);
        gn_swish_rows_kernel<<<dim3((HW + ppb - 1) / ppb, n_img), 256, 0, st>>>(X, stats, gamma, beta, Y, HW, C, groups, ppb,
                                                                               apply_swish);
    } else {
        const long n_chunks = (long)n_img * HW * c8;
        gn_swish_kernel<<<(unsigned)((n_chunks + 255) / 256), 256, 0, st>>>(X, stats, gamma, beta, Y, n_chunks, HW, C, groups,
                                                                            apply_swish);
    }
    GENIE_LAUNCH_CHECK("gn_swish");
    return GENIE_OK;
}

// floats of scratch genie_group_norm_swish_bf16 needs: final (mean, rstd) + the per-block partials
size_t gn_scratch_floats(int n_img, int HW, int groups) {
    const int ppb = HW >= 16384 ? 512 : (HW >= 1024 ? 128 : (HW >= 64 ? 64 : HW));
    const int nblk = (HW + ppb - 1) / ppb;
    return (size_t)n_img * groups * 2 * (1 + (size_t)nblk);
}

int launch_gn_swish(const uint16_t* X, const float* gamma, const float* beta, uint16_t* Y, float* stats, int n_img, int HW,
                    int C, int groups, float eps, int apply_swish, hipStream_t st) {
    GENIE_CHECK_SHAPE(C % groups == 0 && (C / groups) % 4 == 0 && C % 8 == 0 && groups <= 64 && 256 % (C / 8) == 0,
                      "group_norm: C=%d groups=%d unsupported", C, groups);
    const int ppb = HW >= 16384 ? 512 : (HW >= 1024 ? 128 : (HW >= 64 ? 64 : HW));
    const int nblk = (HW + ppb - 1) / ppb;
    float* part = stats + (size_t)n_img * groups * 2;  // scratch layout: [n][groups][2] (mean, rstd) | [n][nblk][groups][2]
    dim3 grid(nblk, n_img);
    gn_stats_kernel<<<grid, 256, 0, st>>>(X, part, HW, C, groups, ppb);
    GENIE_LAUNCH_CHECK("gn_stats");
    gn_finalize_kernel<<<(n_img * groups + 255) / 256, 256, 0, st>>>(part, stats, n_img * groups, groups, nblk,
                                                                      (float)HW * (C / groups), eps);
    GENIE_LAUNCH_CHECK("gn_finalize");
    return launch_gn_apply(X, stats, gamma, beta, Y, n_img, HW, C, groups, apply_swish, st);
}

// floats of `gn_part` a conv launch with fused statistics writes: 64 per 256x128 tile
size_t conv_gn_part_floats(int n_img, int H, int Wd, int Cout) {
    const long M = (long)n_img * H * Wd;
    return (size_t)((M + 255) / 256) * (size_t)((Cout + 127) / 128) * 64;
}

// GroupNorm + swish of the tensor a conv with fused statistics produced: X is that conv's output ((n, H, W, Cout), or
// (n, 2H, 2W, Cout/4) with depth-to-space), `part` its partials.  stats: n * groups * 2 floats of scratch.
int launch_gn_swish_tiles(const uint16_t* X, const float* gamma, const float* beta, uint16_t* Y, const float* part, float* stats,
                          int n_img, int H, int Wd, int Cout, int d2s, int groups, float eps, int apply_swish, hipStream_t st) {
    const int C = d2s ? Cout / 4 : Cout;
    const int HW = (d2s ? 4 : 1) * H * Wd;
    GENIE_CHECK_SHAPE(C % groups == 0 && (C / groups) % 4 == 0 && C % 8 == 0 && ((long)H * Wd) % 256 == 0 && Cout % 128 == 0,
                      "group_norm (fused statistics): geometry unsupported (C=%d, H*W=%d)", C, H * Wd);
    const int cpg = C / groups;
    gn_finalize_tiles_kernel<<<(n_img * groups + 3) / 4, 256, 0, st>>>(part, stats, n_img, groups, (H * Wd) / 256, Cout / 128, cpg,
                                                                           Cout / 4, d2s, (float)HW * cpg, eps);
    GENIE_LAUNCH_CHECK("gn_finalize_tiles");
    return launch_gn_apply(X, stats, gamma, beta, Y, n_img, HW, C, groups, apply_swish, st);
}

// ---- direct 3x3 / pad 1 convolution for the edge layers: X NHWC bf16, Wt (Cout, 3, 3, Cin) bf16, f32 accumulate.
// out_mode 0: Y NHWC bf16;  1: NCHW f32 (decoder conv_out feeding the u8 rescale)
__global__ void conv_direct_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                   const float* __restrict__ bias, void* __restrict__ Yv, int n_img, int H, int Wd, int Cin,
                                   int Cout, int out_mode) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_img * H * Wd * Cout;
    if (idx >= total) return;
    int co;
    long p;
    if (out_mode == 0) { co = (int)(idx % Cout); p = idx / Cout; }            // channel fastest
    else { p = idx % ((long)n_img * H * Wd); co = (int)(idx / ((long)n_img * H * Wd)); }  // pixel fastest
    const long img = p / ((long)H * Wd);
    const long in_img = p - img * (long)H * Wd;
    const int y = (int)(in_img / Wd), x = (int)(in_img % Wd);
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= Wd) continue;
        const uint16_t* xp = X + (((size_t)img * H + yy) * Wd + xx) * Cin;
        const uint16_t* wp = Wt + ((size_t)co * 9 + t) * Cin;
        for (int c = 0; c < Cin; ++c) acc = fmaf(bf16_to_f32(xp[c]), bf16_to_f32(wp[c]), acc);
    }
    if (out_mode == 0) reinterpret_cast<uint16_t*>(Yv)[(size_t)p * Cout + co] = f32_to_bf16(acc);
    else reinterpret_cast<float*>(Yv)[((size_t)img * Cout + co) * H * Wd + in_img] = acc;
}

int launch_conv_direct(const uint16_t* X, const uint16_t* Wt, const float* bias, void* Y, int n_img, int H, int Wd, int Cin,
                       int Cout, int out_mode, hipStream_t st) {
    const long total = (long)n_img * H * Wd * Cout;
    if (total <= 0) return GENIE_OK;
    ProfScope prof(GENIE_KC_OTHER, 2.0 * total * 9.0 * Cin, 2.0 * total, st);
    conv_direct_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(X, Wt, bias, Y, n_img, H, Wd, Cin, Cout, out_mode);
    GENIE_LAUNCH_CHECK("conv_direct");
    return GENIE_OK;
}

// tokens (n, hw) int64 -> +-1 bit planes as NHWC bf16 (n, hw, bits): the decoder's input operand (a18 + layout)
__global__ void bits_nhwc_kernel(const int64_t* __restrict__ ids, uint16_t* __restrict__ z, long n_pix, int bits, int cpad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix * cpad) return;
    const long p = idx / cpad;
    const int c = (int)(idx - p * cpad);
    // +1.0 / -1.0 in bf16; channels >= bits are zero padding (so that C_in %% 64 == 0 for the implicit GEMM)
    z[idx] = c >= bits ? (uint16_t)0 : (((ids[p] >> c) & 1) ? (uint16_t)0x3F80 : (uint16_t)0xBF80);
}
int launch_bits_nhwc(const int64_t* ids, uint16_t* z, long n_pix, int bits, int cpad, hipStream_t st) {
    if (n_pix <= 0) return GENIE_OK;
    bits_nhwc_kernel<<<(unsigned)((n_pix * cpad + 255) / 256), 256, 0, st>>>(ids, z, n_pix, bits, cpad);
    GENIE_LAUNCH_CHECK("bits_nhwc");
    return GENIE_OK;
}

// (C_out, C_in, 3, 3) f32 -> (C_out, 3, 3, C_in) bf16 (tap-major K for the implicit GEMM); 1x1: (C_out, C_in) cast
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int taps) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)Cout * Cin * taps;
    if (idx >= total) return;
    const int c = (int)(idx % Cin);
    const int t = (int)((idx / Cin) % taps);
    const int co = (int)(idx / ((long)Cin * taps));
    out[idx] = f32_to_bf16(w[((size_t)co * Cin + c) * taps + t]);
}
int launch_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int taps, hipStream_t st) {
    const long total = (long)Cout * Cin * taps;
    if (total <= 0) return GENIE_OK;
    pack_conv_weight_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(w, out, Cout, Cin, taps);
    GENIE_LAUNCH_CHECK("pack_conv_weight");
    return GENIE_OK;
}

// decoder tail: conv_out result as NHWC bf16 with cpad channels -> (n, c_out, H*W) uint8 through the reference's
// bf16 rescale (visualize.py:84-92): u8 = trunc(clamp(bf16(bf16(x + 1) * 127.5), 0, 255))
__global__ void rescale_nhwc_to_nchw_u8_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ out, long n_pix_total,
                                               int HW, int cpad, int cout) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix_total * cout) return;
    const int c = (int)(idx / n_pix_total % cout);
    (void)c;
    const long p = idx % n_pix_total;
    const int ch = (int)(idx / n_pix_total);
    const long img = p / HW, in_img = p - img * HW;
    float v = bf16_to_f32(x[(size_t)p * cpad + ch]);
    v = bf16_to_f32(f32_to_bf16(v + 1.0f));
    v = bf16_to_f32(f32_to_bf16(v * 127.5f));
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    out[((size_t)img * cout + ch) * HW + in_img] = (uint8_t)v;
}
int launch_rescale_nhwc_u8(const uint16_t* x, uint8_t* out, long n_img, int HW, int cpad, int cout, hipStream_t st) {
    const long total = n_img * HW * cout;
    if (total <= 0) return GENIE_OK;
    rescale_nhwc_to_nchw_u8_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, out, n_img * HW, HW, cpad, cout);
    GENIE_LAUNCH_CHECK("rescale_nhwc_u8");
    return GENIE_OK;
}

// encoder head: (n, c_in, H*W) uint8 frames -> (n, H*W, cpad) bf16 with x/127.5 - 1 (channels >= c_in zero)
__global__ void frames_to_nhwc_kernel(const uint8_t* __restrict__ f, uint16_t* __restrict__ x, long n_pix_total, int HW, int cin,
                                      int cpad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pix_total * cpad) return;
    const long p = idx / cpad;
    const int c = (int)(idx - p * cpad);
    if (c >= cin) { x[idx] = 0; return; }
    const long img = p / HW, in_img = p - img * HW;
    x[idx] = f32_to_bf16((float)f[((size_t)img * cin + c) * HW + in_img] / 127.5f - 1.0f);
}
int launch_frames_to_nhwc(const uint8_t* f, uint16_t* x, long n_img, int HW, int cin, int cpad, hipStream_t st) {
    const long total = n_img * HW * cpad;
    if (total <= 0) return GENIE_OK;
    frames_to_nhwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(f, x, n_img * HW, HW, cin, cpad);
    GENIE_LAUNCH_CHECK("frames_to_nhwc");
    return GENIE_OK;
}
// encoder tail: code (n_pix, cpad) bf16 -> dataset-convention ids: bit c = [h_c > 0]
__global__ void tokens_from_nhwc_kernel(const uint16_t* __restrict__ h, int64_t* __restrict__ ids, long n_pix, int bits, int cpad) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    int64_t id = 0;
    for (int c = 0; c < bits; ++c) id |= (int64_t)(bf16_to_f32(h[(size_t)p * cpad + c]) > 0.0f) << c;
    ids[p] = id;
}
int launch_tokens_from_nhwc(const uint16_t* h, int64_t* ids, long n_pix, int bits, int cpad, hipStream_t st) {
    if (n_pix <= 0) return GENIE_OK;
    tokens_from_nhwc_kernel<<<(unsigned)((n_pix + 255) / 256), 256, 0, st>>>(h, ids, n_pix, bits, cpad);
    GENIE_LAUNCH_CHECK("tokens_from_nhwc");
    return GENIE_OK;
}

}  // namespace genie
