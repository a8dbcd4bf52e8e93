// The fused temporal sub-block of the shipped geometry (d = 256, 8 heads of 32, GENIE_PREC_BF16) for the PREFIX-CACHE passes of the
// evaluator (evaluate.py:107-116 scores "frame t in timeline t": one clean pass over the ground-truth frames, then MaskGIT passes
// whose frame t attends the clean keys of the frames before it and its own):
//
//   MODE 1  clean pass     x += proj_t( causal-attention( qkv_t( bf16(x) ) ) )   and the head's K / V operand fragments are DUMPED
//                          from the registers into the layer's slice of the cache
//   MODE 2  masked passes  x += proj_t( attention( q_i ; cached k_j, v_j for j < i + shift, own k_i, v_i ) ): the cached fragments
//                          come back lane-linear, one head ahead of their use
//
// replacing qkv GEMM + attn_temporal_(prefix_)f32_mfma + proj GEMM (st_transformer.py:77-78, attention.py:36-61).  Structure, weight
// stream (genie_pack_temporal_fused_bf16), ring protocol and rounding points are those of temporal_fused_bf16_kernel
// (kernels_fused.hip, "lane = token": a wave owns two spatial positions x 16 frame slots); what differs:
//   * the pass has nf <= 16 frames per clip (the evaluator: 15): frame slots >= nf are phantoms -- their loads repeat frame nf - 1,
//     nothing of them is stored, and causal / prefix masking keeps their keys away from every real query;
//   * the cache is not the (B, frames, S, 3d) qkv of the unfused passes but the register images themselves: per (clip, position, head)
//     2 KB = K' as the A operand of S^T = K' Q'^T (lane: frame l & 15, 8 features; 16 B per lane) followed by V as the A operand of
//     O^T = V^T P for the two 16-feature tiles (lane: feature l & 15, frames 4 (l >> 4) .. + 3; 8 B per lane).  A dump is three
//     lane-linear stores, a read three lane-linear loads -- whole 128-byte lines both ways, no transposition anywhere.  (The row
//     layout would need V transposed: 2-byte gathers at frame stride.)
//   * MODE 2's scores are two products per head and position: cached keys (masked to j < i + shift, j < nf) and the pass's own keys
//     (diagonal only); one softmax over both; O^T = Vc^T Pc + V^T Pdiag.
// vmcnt bookkeeping: the ring's acquire waits "all but the 8 youngest" vector-memory operations (two stages of four LDS-DMA loads).
// MODE 2's cached-fragment loads are issued between two acquires and retire in order with the ring's loads, so the acquires of the
// next three intervals allow for them (8 + 6); they are inline asm (the compiler neither moves them nor waits for them) and their
// first use is more than three acquires later.  MODE 1's dump stores only ever make a wait stricter.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int PF_STAGE = 16384;  // bytes of one stage = 16 fragments of 1 KB
constexpr int PF_NS = 4;         // ring slots
constexpr int PF_RING = PF_NS * PF_STAGE;

__device__ __forceinline__ void pf_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void pf_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// n is a compile-time constant at every call site once the head loops are unrolled
__device__ __forceinline__ void pf_wait_vm_n(int n) {
    if (n >= 14) pf_wait_vm<14>();
    else pf_wait_vm<8>();
}
__device__ __forceinline__ void pf_wave_lds_fence() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ f32x4 pf_mma32(const s16x8& a, const s16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 pf_mma16k(const s16x4& a, const s16x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ s16x8 pf_pack8(const f32x4& lo, const f32x4& hi) {
    u32x4 p;
    p.x = f32x2_to_bf16x2(lo.x, lo.y);
    p.y = f32x2_to_bf16x2(lo.z, lo.w);
    p.z = f32x2_to_bf16x2(hi.x, hi.y);
    p.w = f32x2_to_bf16x2(hi.z, hi.w);
    return __builtin_bit_cast(s16x8, p);
}
__device__ __forceinline__ s16x4 pf_pack4(const f32x4& v) {
    u32x2 p;
    p.x = f32x2_to_bf16x2(v.x, v.y);
    p.y = f32x2_to_bf16x2(v.z, v.w);
    return __builtin_bit_cast(s16x4, p);
}
// cached fragments: wave-uniform base in SGPRs, lane offset in one VGPR
template <int OFF>
__device__ __forceinline__ void pf_ld16(s16x8& dst, unsigned voff, const uint16_t* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void pf_ld8(s16x4& dst, unsigned voff, const uint16_t* base) {
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}

}  // namespace

// x: (B, nf, S, 256) f32, updated in place.  kv: this layer's fragment cache, [(b S + s) 8 + head][1024] bf16 values.
// A block = 8 consecutive spatial positions of one clip x 16 frame slots; wave w owns positions 2 w, 2 w + 1.
template <bool QKV_BIAS, int MODE>
__global__ __launch_bounds__(256, 2) void temporal_prefix_fused_bf16_kernel(float* __restrict__ x, const uint16_t* __restrict__ wstream,
                                                                            const float* __restrict__ qkv_b,
                                                                            const float* __restrict__ proj_b, uint16_t* __restrict__ kv,
                                                                            int n_blocks, int S, int nf, int sh, float scale_log2e) {
    constexpr int D = 256, NH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    float* sbias = reinterpret_cast<float*>(smem + PF_RING);   // proj 256 floats, then qkv 768 (see temporal_fused_bf16_kernel)
    for (int i = tid; i < 256 + (QKV_BIAS ? 768 : 0); i += 256) sbias[i] = i < 256 ? (proj_b ? proj_b[i] : 0.f) : qkv_b[i - 256];
    __syncthreads();

    auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, 32 * PF_STAGE, 0x00020000);
    const unsigned voff = (unsigned)lane * 16;
    int n_issue = 0;  // stages issued so far (stream position = n & 31, slot = n & 3)
    auto issue_stage = [&]() {
        const int soff = (n_issue & 31) * PF_STAGE + wid * 4096;
        unsigned char* dst = smem + (n_issue & (PF_NS - 1)) * PF_STAGE + wid * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff,
                                                     soff + j * 1024, 0, 0);
        ++n_issue;
    };
    int n_use = 0;
    // other vector-memory operations issued since the last three acquires (younger than the loads of the stage being acquired)
    int ex0 = 0, ex1 = 0, ex2 = 0;
    auto acquire = [&]() -> const unsigned char* {
        pf_wait_vm_n(8 + ex0 + ex1 + ex2);
        pf_barrier();
        issue_stage();
        __builtin_amdgcn_sched_barrier(0);
        ex0 = ex1; ex1 = ex2; ex2 = 0;
        const unsigned char* p = smem + (n_use & (PF_NS - 1)) * PF_STAGE + lane * 16;
        ++n_use;
        return p;
    };
    auto frag = [&](const unsigned char* stage, int f) { return *reinterpret_cast<const s16x8*>(stage + f * 1024); };

    issue_stage();
    issue_stage();
    issue_stage();

    const int bps = S / 8;  // blocks per clip
    // row-major side (operand rows in, residual update): lane -> frame slots tt and 8 + tt, columns 4 (lane & 7) .. + 3 of a 32-column slab
    int tt = lane >> 3, cc = (lane & 7) * 4;
    asm volatile("" : "+v"(tt), "+v"(cc));
    const int fa = tt < nf ? tt : nf - 1, fb = 8 + tt < nf ? 8 + tt : nf - 1;
    const bool st_a = tt < nf, st_b = 8 + tt < nf;
    float* tile = reinterpret_cast<float*>(smem + PF_RING + 4096 + wid * 2304);

    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int b = blk / bps, s0 = (blk - b * bps) * 8 + 2 * wid;
        float* xa = x + (((size_t)b * nf + fa) * S + s0) * D + cc;
        float* xb = x + (((size_t)b * nf + fb) * S + s0) * D + cc;
        const uint16_t* kvw = kv + ((size_t)b * S + s0) * (NH * 1024);   // this wave's two positions: 16 KB each, wave-uniform

        s16x8 kc[2][2];      // [head parity][group]   cached K' fragments (MODE 2)
        s16x4 vc[2][2][2];   // [head parity][group][feature tile]
        if constexpr (MODE == 2) {
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const uint16_t* p = kvw + grp * (NH * 1024);
                pf_ld16<0>(kc[0][grp], voff, p);
                pf_ld8<1024>(vc[0][grp][0], voff >> 1, p);
                pf_ld8<1536>(vc[0][grp][1], voff >> 1, p);
            }
        }

        s16x8 xf[2][8];
        {
            f32x4 raw[16][2];
            auto load_slab = [&](int i) {   // slab i = (grp = i >> 3, K-step i & 7: columns 32 (i & 7) ..)
                raw[i][0] = *reinterpret_cast<const f32x4*>(xa + (i >> 3) * D + 32 * (i & 7));
                raw[i][1] = *reinterpret_cast<const f32x4*>(xb + (i >> 3) * D + 32 * (i & 7));
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) load_slab(i);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 4 < 16) load_slab(i + 4);
                *reinterpret_cast<f32x4*>(tile + tt * 36 + cc) = raw[i][0];
                *reinterpret_cast<f32x4*>(tile + (8 + tt) * 36 + cc) = raw[i][1];
                pf_wave_lds_fence();
                xf[i >> 3][i & 7] = pf_pack8(*reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g), *reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g + 4));
                pf_wave_lds_fence();
            }
            pf_wait_vm<0>();   // (also the ring's run-ahead stages and MODE 2's first cached fragments)
            ex0 = ex1 = ex2 = 0;
        }

        s16x8 oall[2][NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            s16x8 qb[2], kb[2];
            s16x4 vb[2][2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const unsigned char* stg = acquire();
                if constexpr (MODE == 2) {
                    if (part == 0 && h + 1 < NH) {   // the next head's cached fragments: used more than three acquires from here
#pragma unroll
                        for (int grp = 0; grp < 2; ++grp) {
                            const uint16_t* p = kvw + grp * (NH * 1024) + (h + 1) * 1024;
                            pf_ld16<0>(kc[(h + 1) & 1][grp], voff, p);
                            pf_ld8<1024>(vc[(h + 1) & 1][grp][0], voff >> 1, p);
                            pf_ld8<1536>(vc[(h + 1) & 1][grp][1], voff >> 1, p);
                        }
                        ex2 += 6;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                f32x4 acc[2][2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (QKV_BIAS) {
                        const float* bp = sbias + 256 + part * D + h * 32 + ft * 16;
                        if (part < 2) b0 = *reinterpret_cast<const f32x4*>(bp + 4 * g);  // lane holds features 4 g .. 4 g + 3 of the tile
                        else b0 = f32x4{bp[r], bp[r], bp[r], bp[r]};                     // lane holds feature r
                    }
                    acc[0][ft] = b0;
                    acc[1][ft] = b0;
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        const s16x8 wf = frag(stg, 2 * ks + ft);
#pragma unroll
                        for (int grp = 0; grp < 2; ++grp)
                            acc[grp][ft] = part < 2 ? pf_mma32(wf, xf[grp][ks], acc[grp][ft]) : pf_mma32(xf[grp][ks], wf, acc[grp][ft]);
                    }
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    if (part == 0) qb[grp] = pf_pack8(acc[grp][0], acc[grp][1]);
                    else if (part == 1) kb[grp] = pf_pack8(acc[grp][0], acc[grp][1]);
                    else { vb[grp][0] = pf_pack4(acc[grp][0]); vb[grp][1] = pf_pack4(acc[grp][1]); }
                }
                if constexpr (MODE == 1) {   // dump the register images (lane-linear: whole lines)
                    if (part == 1) {
#pragma unroll
                        for (int grp = 0; grp < 2; ++grp)
                            *reinterpret_cast<s16x8*>(const_cast<uint16_t*>(kvw) + grp * (NH * 1024) + h * 1024 + lane * 8) = kb[grp];
                    } else if (part == 2) {
#pragma unroll
                        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                            for (int ft = 0; ft < 2; ++ft)
                                *reinterpret_cast<s16x4*>(const_cast<uint16_t*>(kvw) + grp * (NH * 1024) + h * 1024 + 512 + ft * 256 + lane * 4) = vb[grp][ft];
                    }
                }
            }
            if constexpr (MODE == 2) {
                // the cached fragments of this head landed at least one acquire ago (header); from here on they are ordinary values
                asm volatile("" : "+v"(kc[h & 1][0]), "+v"(kc[h & 1][1]), "+v"(vc[h & 1][0][0]), "+v"(vc[h & 1][0][1]),
                             "+v"(vc[h & 1][1][0]), "+v"(vc[h & 1][1][1]));
            }
            // attention over the frame slots of each group (attention.py:48-58): lane = query frame r, keys 4 g + e
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                f32x4 st = pf_mma32(MODE == 2 ? kc[h & 1][grp] : kb[grp], qb[grp], f32x4{0.f, 0.f, 0.f, 0.f});
                f32x4 so = {0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE == 2) so = pf_mma32(kb[grp], qb[grp], f32x4{0.f, 0.f, 0.f, 0.f});   // own keys: the diagonal is used
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * g + e;
                    if constexpr (MODE == 2) {
                        if (j >= r + sh || j >= nf) st[e] = -INFINITY;   // cached clip frames strictly before the query's
                        if (j != r) so[e] = -INFINITY;
                        mx = fmaxf(mx, so[e]);
                    } else {
                        if (j > r) st[e] = -INFINITY;
                    }
                    mx = fmaxf(mx, st[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mxs = mx * scale_log2e;
                float sum = 0.f;
                f32x4 p, plo, po, polo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, -mxs));
                    sum += p[e];
                    if constexpr (MODE == 2) {
                        po[e] = __builtin_amdgcn_exp2f(fmaf(so[e], scale_log2e, -mxs));
                        sum += po[e];
                    }
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = __builtin_amdgcn_rcpf(sum);
                const s16x4 phi = pf_pack4(p);
#pragma unroll
                for (int e = 0; e < 4; ++e) plo[e] = p[e] - bf16_to_f32((uint16_t)phi[e]);
                const s16x4 plo16 = pf_pack4(plo);
                s16x4 pohi, polo16;
                if constexpr (MODE == 2) {
                    pohi = pf_pack4(po);
#pragma unroll
                    for (int e = 0; e < 4; ++e) polo[e] = po[e] - bf16_to_f32((uint16_t)pohi[e]);
                    polo16 = pf_pack4(polo);
                }
                f32x4 o[2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    const s16x4 va = MODE == 2 ? vc[h & 1][grp][ft] : vb[grp][ft];
                    o[ft] = pf_mma16k(va, phi, f32x4{0.f, 0.f, 0.f, 0.f});
                    o[ft] = pf_mma16k(va, plo16, o[ft]);
                    if constexpr (MODE == 2) {
                        o[ft] = pf_mma16k(vb[grp][ft], pohi, o[ft]);
                        o[ft] = pf_mma16k(vb[grp][ft], polo16, o[ft]);
                    }
                    o[ft] *= inv;
                }
                oall[grp][h] = pf_pack8(o[0], o[1]);
            }
        }

        // out-projection, swapped: D[out column][token]; lane = frame r holds columns 16 ct + 4 g .. + 3
        f32x4 out[2][16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            out[0][ct] = *reinterpret_cast<const f32x4*>(sbias + ct * 16 + 4 * g);
            out[1][ct] = out[0][ct];
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const unsigned char* stg = acquire();
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const s16x8 wf = frag(stg, ct);
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) out[grp][ct] = pf_mma32(wf, oall[grp][h], out[grp][ct]);
            }
        }
        // residual update in place through the wave's LDS tile: row-major, every request moves whole 128-byte lines
        {
            f32x4 rs[2][2];
            auto load_slab = [&](int i, f32x4* dst) {   // slab i = (grp = i >> 3, columns 32 (i & 7) ..)
                dst[0] = *reinterpret_cast<const f32x4*>(xa + (i >> 3) * D + 32 * (i & 7));
                dst[1] = *reinterpret_cast<const f32x4*>(xb + (i >> 3) * D + 32 * (i & 7));
            };
            load_slab(0, rs[0]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int grp = i >> 3, sl = i & 7;
                if (i + 1 < 16) load_slab(i + 1, rs[(i + 1) & 1]);
                *reinterpret_cast<f32x4*>(tile + r * 36 + 4 * g) = out[grp][2 * sl];
                *reinterpret_cast<f32x4*>(tile + r * 36 + 16 + 4 * g) = out[grp][2 * sl + 1];
                pf_wave_lds_fence();
                const f32x4 va = *reinterpret_cast<const f32x4*>(tile + tt * 36 + cc);
                const f32x4 vb2 = *reinterpret_cast<const f32x4*>(tile + (8 + tt) * 36 + cc);
                if (st_a) *reinterpret_cast<f32x4*>(xa + grp * D + 32 * sl) = rs[i & 1][0] + va;
                if (st_b) *reinterpret_cast<f32x4*>(xb + grp * D + 32 * sl) = rs[i & 1][1] + vb2;
                pf_wave_lds_fence();
            }
        }
    }
    pf_wait_vm<0>();  // the ring's run-ahead stages must not outlive the workgroup's LDS allocation
}

#ifndef GENIE_VAR_TP_MIN_CLIPS
#define GENIE_VAR_TP_MIN_CLIPS 2
#endif
// ONE predicate for the clean pass (which then writes fragment images instead of qkv rows into the cache) and the masked passes
// that read them: the model's geometry, a pass of fewer frames than the model's T (a cache that genie_frame_pass could continue
// always has T frame slots, so it never takes this form; and at least 8 frames: the fragment images of 16 frame slots must fit the
// layer's slice of the cache), the same B.
bool temporal_prefix_fused_takes(const genie_cfg& c, const genie_attn_weights& aw, int B, int model_T) {
#ifdef GENIE_VAR_TP_OFF   // (A/B variant: the prefix-cache passes on the unfused launches)
    return false;
#endif
    return aw.fused_w16 && c.precision == GENIE_PREC_BF16 && c.d_model == 256 && c.num_heads == 8 && c.head_dim == 32 && c.T >= 8 &&
           c.T <= 16 && c.T < model_T && model_T <= 16 && c.S % 8 == 0 && !c.qk_norm && (long)B * c.S >= GENIE_VAR_TP_MIN_CLIPS * 256;
}

// mode 1: clean pass (kv written); mode 2: masked pass (kv read, query slot i sees cached slots j < i + shift and itself)
int launch_temporal_prefix_fused_bf16(const genie_cfg& c, const genie_attn_weights& aw, float* x, uint16_t* kv, int B, int mode,
                                      int shift, int model_T, hipStream_t st) {
    if (!temporal_prefix_fused_takes(c, aw, B, model_T)) return GENIE_E_UNSUPPORTED;
    GENIE_CHECK_ARG(x && kv && (mode == 1 || mode == 2) && (shift == 0 || shift == 1), "temporal_prefix_fused: bad argument");
    const int n_blocks = B * c.S / 8;
    static const int cus = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int grid = n_blocks < 2 * cus ? n_blocks : 2 * cus;
    const double M = (double)B * c.T * c.S;
    ProfScope prof(GENIE_KC_FUSED, M * (2.0 * 256 * 1024 + 4.0 * 16 * 256 * (mode == 2 ? 2 : 1)), M * (2048.0 + 1024.0), st,
                   mode == 1 ? "temporal_prefix_fused_bf16_kernel<1> (clean pass: qkv + causal attention + proj + residual, K/V fragments out)"
                             : "temporal_prefix_fused_bf16_kernel<2> (masked pass: qkv + attention over cached fragments + proj + residual)");
    const size_t lds = PF_RING + 4096 + 4 * 2304;   // ring, biases, one 16 x 36-float tile per wave
    const float sl2e = c.attn_scale * 1.4426950408889634f;
    const float* pb = c.proj_bias ? aw.proj_b : nullptr;
    const bool qb = c.qkv_bias && aw.qkv_b;
#define TP_LAUNCH(QB_, MODE_)                                                                                                              \
    do {                                                                                                                                   \
        static PerDevice<bool> once;                                                                                                       \
        if (once.needs()) {                                                                                                                \
            (void)hipFuncSetAttribute((const void*)temporal_prefix_fused_bf16_kernel<QB_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            once.set(true);                                                                                                                \
        }                                                                                                                                  \
        temporal_prefix_fused_bf16_kernel<QB_, MODE_><<<grid, 256, lds, st>>>(x, aw.fused_w16, QB_ ? aw.qkv_b : nullptr, pb, kv, n_blocks, \
                                                                               c.S, c.T, shift, sl2e);                                     \
    } while (0)
    if (mode == 1) { if (qb) TP_LAUNCH(true, 1); else TP_LAUNCH(false, 1); }
    else { if (qb) TP_LAUNCH(true, 2); else TP_LAUNCH(false, 2); }
#undef TP_LAUNCH
    GENIE_LAUNCH_CHECK("temporal_prefix_fused_bf16");
    return GENIE_OK;
}

}  // namespace genie
