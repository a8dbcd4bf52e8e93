// The temporal qkv Linear + temporal attention of the shipped geometry (d = 256, 8 heads of 32) in GENIE_PREC_F16X3 as ONE kernel:
//   a = attention_T( qkv_t(x) )     (st_transformer.py:77-78, attention.py:36-58; the out-projection + residual stay the proj GEMM)
// replacing the qkv GEMM (3 KB of f32 qkv per token written) and attn_temporal_(prefix_)f32_mfma (read back, plus 2 KB of cached
// k, v per token in the prefix-cache passes): the qkv never leaves the registers.  Split operands double every fragment, so the whole
// sub-block does not fit the registers -- this half does, with ONE 16-token group per wave.
//
//   * "lane = token" as temporal_fused_bf16_kernel: a wave owns one spatial position's 16 frame slots; its operand fragments
//     (lane: token l & 15, k-group l >> 4, 8 consecutive k; hi and lo' planes of the split, made here from the f32 rows of x) are the B
//     operand of the swapped products D[feature][token] = W . X^T (q, k) and the A operand of the plain one (v).
//   * every Linear product is three v_mfma_f32_16x16x32_f16: acc_m += Whi . xhi, acc_c += Whi . xlo' + Wlo' . xhi, value =
//     acc_m + acc_c / 2048 (+ bias) -- the f16x3 contract of kernels_bf16.hip / kernels_frame.hip (22-bit operands, f32 accumulation).
//   * the attention is f32 on v_mfma_f32_16x16x4_f32, as attn_temporal_f32_mfma_kernel: the accumulator layout of q, k (lane = token,
//     features 4 g + e) is both operands' layout of S^T = K Q^T contracted in the order (g, e); S^T's (lane = query, keys 4 g + e) is the
//     B operand and v's (lane = feature, tokens 4 g + e) the A operand of O^T = V^T P.  Nothing is re-laid and nothing is rounded.
//   * weight stream (genie_pack_temporal_qkv_f16x3): 48 stages of 16 fragments = (head, q | k | v, K half) x (K-step, 16-feature tile,
//     plane), through the 4-slot LDS-DMA ring of temporal_fused_bf16_kernel.
//   * the output leaves as the proj GEMM's operand planes [hi | lo'] (row-major), a head at a time through the wave's LDS tile.
//   MODE 0 plain causal forward; MODE 1 clean pass of the evaluator (evaluate.py:107-116): the head's k and v accumulators are dumped
//   lane-linear into the layer's cache slice (4 KB per clip, position and head); MODE 2 masked passes: loaded back one head ahead,
//   scores against the cached keys (j < i + shift) and the own key (diagonal), one softmax -- as kernels_fused_prefix.hip.
// vmcnt bookkeeping as there: an acquire allows for every vector-memory operation issued since the needed stage's loads (the ring's two
// younger stages + the cached-fragment loads and the output / dump stores of the last three intervals): they retire in order -- loads AND
// stores on the one vmcnt counter (tools/micro/vmcnt_order_probe.hip: 4.2e8 lane-trials of a cold load followed by 1-6 hot stores and
// s_waitcnt vmcnt(#stores), none consumed an unlanded load; profiles/r05v_vmcnt_order_probe.txt).
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int QA_STAGE = 16384;  // bytes of one stage = 16 fragments of 1 KB
constexpr int QA_NS = 4;         // ring slots
constexpr int QA_RING = QA_NS * QA_STAGE;
constexpr int QA_STAGES = 48;    // per layer

__device__ __forceinline__ void qa_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void qa_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// n is a compile-time constant at every call site once the head loops are unrolled; rounding down only makes the wait stricter
__device__ __forceinline__ void qa_wait_vm_n(int n) {
    if (n >= 14) qa_wait_vm<14>();
    else if (n >= 12) qa_wait_vm<12>();
    else if (n >= 10) qa_wait_vm<10>();
    else qa_wait_vm<8>();
}
__device__ __forceinline__ void qa_wave_lds_fence() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ f32x4 qa_mma(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 qa_mma4(float a, float b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// Fragment reads the compiler does not schedule (it pairs them: read, read, wait, three MFMAs -- one exposed LDS round trip per 48 matrix
// cycles): qa_lds_rd issues a read it knows nothing about, qa_lds_wait2<N> is "all but my N youngest LDS reads have landed", tied to the
// two fragments it guards.  LDS reads return in order; anything else outstanding on lgkmcnt only makes the wait stricter.
template <int OFF>
__device__ __forceinline__ void qa_lds_rd(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void qa_lds_wait2(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int OFF>
__device__ __forceinline__ void qa_ld16(f32x4& dst, unsigned voff, const float* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void qa_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
    split_f16_x4(a.x, a.y, a.z, a.w, h01, h23, l01, l23);
    split_f16_x4(b.x, b.y, b.z, b.w, h45, h67, l45, l67);
    hi = u32x4{h01, h23, h45, h67};
    lo = u32x4{l01, l23, l45, l67};
}

}  // namespace

// Weight stream of one layer (split f16, 48 stages x 16 fragments x 64 lanes x 8 values = 384 K values = 768 KB):
//   stage n: head n / 6, part (n % 6) / 2 (q, k, v), K half n & 1;  fragment f: plane f & 1 (hi, lo'), 16-feature tile (f >> 1) & 1,
//   K-step ks = 4 (n & 1) + (f >> 2):   [lane l][e] = split(Wqkv[part * 256 + head * 32 + tile * 16 + (l & 15)][32 ks + 8 (l >> 4) + e]).plane
__global__ void pack_temporal_qkv_f16x3_kernel(const float* __restrict__ qkv_w, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output value
    if (i >= QA_STAGES * 16 * 64 * 8) return;
    const int e = i & 7, l = (i >> 3) & 63, f = (i >> 9) & 15, n = i >> 13;
    const int head = n / 6, part = (n % 6) >> 1, ks = 4 * (n & 1) + (f >> 2), ft = (f >> 1) & 1;
    uint16_t hi, lo;
    split_f16(qkv_w[(size_t)(part * 256 + head * 32 + ft * 16 + (l & 15)) * 256 + 32 * ks + 8 * (l >> 4) + e], hi, lo);
    out[i] = (f & 1) ? lo : hi;
}

int launch_pack_temporal_qkv_f16x3(const float* qkv_w, uint16_t* out, hipStream_t st) {
    pack_temporal_qkv_f16x3_kernel<<<(QA_STAGES * 16 * 64 * 8) / 256, 256, 0, st>>>(qkv_w, out);
    GENIE_LAUNCH_CHECK("pack_temporal_qkv_f16x3");
    return GENIE_OK;
}

// x: (B, nf, S, 256) f32 (read only).  a16: the attention output's operand planes, (B nf S) x 256 row-major, [hi | lo'] `plane` elements apart.
// kv: this layer's cache slice, [(b S + s) 8 + head][k tile 0 | k tile 1 | v tile 0 | v tile 1][64 lanes][4] floats (MODE 1 written, MODE 2 read).
// A block = 4 G consecutive spatial positions of one clip x 16 frame slots; wave w owns positions G w .. G w + G - 1.  G = 2 (every weight
// fragment read from LDS feeds two groups: the kernel is bound by LDS bytes) where the registers allow: no bias, no cached accumulators.
template <bool QKV_BIAS, int MODE, int G>
__global__ __launch_bounds__(256, 2) void temporal_qkv_attn_f16x3_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wstream,
                                                                         const float* __restrict__ qkv_b, uint16_t* __restrict__ a16,
                                                                         long plane, float* __restrict__ kv, int n_blocks, int S, int nf,
                                                                         int sh, float scale_log2e) {
    static_assert(G == 1 || (MODE != 2 && !QKV_BIAS), "two groups per wave: registers");
    constexpr int D = 256, NH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    float* sbias = reinterpret_cast<float*>(smem + QA_RING);   // qkv bias (768 floats): ds_reads next to the ring, never global loads
    if constexpr (QKV_BIAS) {
        for (int i = tid; i < 768; i += 256) sbias[i] = qkv_b[i];
        __syncthreads();
    }

    auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, QA_STAGES * QA_STAGE, 0x00020000);
    const unsigned voff = (unsigned)lane * 16;
    int n_issue = 0;   // stages issued so far (slot = n & 3)
    int s_pos = 0;     // stream position of the next stage to issue (0 .. 47)
    auto issue_stage = [&]() {
        const int soff = s_pos * QA_STAGE + wid * 4096;
        unsigned char* dst = smem + (n_issue & (QA_NS - 1)) * QA_STAGE + wid * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff,
                                                     soff + j * 1024, 0, 0);
        ++n_issue;
        s_pos = s_pos + 1 == QA_STAGES ? 0 : s_pos + 1;
    };
    int n_use = 0;
    int ex0 = 0, ex1 = 0, ex2 = 0;   // other vector-memory operations issued in the last three intervals between acquires
    auto acquire = [&]() -> const unsigned char* {
        qa_wait_vm_n(8 + ex0 + ex1 + ex2);
        qa_barrier();
        issue_stage();
        __builtin_amdgcn_sched_barrier(0);
        ex0 = ex1; ex1 = ex2; ex2 = 0;
        const unsigned char* p = smem + (n_use & (QA_NS - 1)) * QA_STAGE + lane * 16;
        ++n_use;
        return p;
    };
    auto frag = [&](const unsigned char* stage, int f) { return *reinterpret_cast<const u32x4*>(stage + f * 1024); };

    issue_stage();
    issue_stage();
    issue_stage();

    const int bps = S / (4 * G);  // blocks per clip
    // row-major side of the operand load: lane -> frame slots tt and 8 + tt, columns 4 (lane & 7) .. + 3 of a 32-column slab
    int tt = lane >> 3, cc = (lane & 7) * 4;
    asm volatile("" : "+v"(tt), "+v"(cc));
    const int fa = tt < nf ? tt : nf - 1, fb = 8 + tt < nf ? 8 + tt : nf - 1;
    const int dab = (fb - fa) * S * D;   // the second frame slot's row, relative to the first's (one register instead of a second pointer)
    float* tile = reinterpret_cast<float*>(smem + QA_RING + 4096 + wid * 2304);   // 16 x 36 floats

    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int b = blk / bps, s0 = (blk - b * bps) * (4 * G) + wid * G;   // group grp = position s0 + grp
        int lane_off = fa * S * D + cc;      // (lane-dependent address parts are formed per block behind an empty asm: hoisted out of the
        asm volatile("" : "+v"(lane_off));   // persistent loop they are 64-bit values that live -- or spill -- across the whole main loop)
        const float* xa = x + ((size_t)b * nf * S + s0) * D + lane_off;
        const float* xb = xa + dab;
        float* kvw = kv + ((size_t)b * S + s0) * (NH * 1024);   // the wave's positions: 32 KB each, wave-uniform

        f32x4 kc[2][2], vc[2][2];   // [head parity][feature tile]   cached k, v accumulators (MODE 2; one group)
        if constexpr (MODE == 2) {
            qa_ld16<0>(kc[0][0], voff, kvw);
            qa_ld16<1024>(kc[0][1], voff, kvw);
            qa_ld16<2048>(vc[0][0], voff, kvw);
            qa_ld16<3072>(vc[0][1], voff, kvw);
        }

        u32x4 xhi[G][8], xlo[G][8];
        {
            f32x4 raw[8 * G][2];
            auto load_slab = [&](int i) {   // slab i = (group i >> 3, K-step i & 7: columns 32 (i & 7) ..)
                raw[i][0] = *reinterpret_cast<const f32x4*>(xa + (i >> 3) * D + 32 * (i & 7));
                raw[i][1] = *reinterpret_cast<const f32x4*>(xb + (i >> 3) * D + 32 * (i & 7));
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) load_slab(i);
#pragma unroll
            for (int i = 0; i < 8 * G; ++i) {
                if (i + 4 < 8 * G) load_slab(i + 4);
                *reinterpret_cast<f32x4*>(tile + tt * 36 + cc) = raw[i][0];
                *reinterpret_cast<f32x4*>(tile + (8 + tt) * 36 + cc) = raw[i][1];
                qa_wave_lds_fence();
                qa_split8(*reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g), *reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g + 4),
                          xhi[i >> 3][i & 7], xlo[i >> 3][i & 7]);
                qa_wave_lds_fence();
            }
            qa_wait_vm<0>();   // (also the ring's run-ahead stages, the previous block's stores and MODE 2's first cached accumulators)
            ex0 = ex1 = ex2 = 0;
        }

#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 qv[G][2], kk[G][2], vv[G][2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                f32x4 accm[G][2], accc[G][2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (QKV_BIAS) {
                        const float* bp = sbias + part * D + h * 32 + ft * 16;
                        if (part < 2) b0 = *reinterpret_cast<const f32x4*>(bp + 4 * g);  // lane holds features 4 g .. 4 g + 3 of the tile
                        else b0 = f32x4{bp[r], bp[r], bp[r], bp[r]};                     // lane holds feature r
                    }
#pragma unroll
                    for (int grp = 0; grp < G; ++grp) {
                        accm[grp][ft] = b0;
                        accc[grp][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const unsigned char* stg = acquire();
                    if constexpr (MODE == 2) {
                        if (part == 0 && half == 0 && h + 1 < NH) {   // the next head's cached accumulators: first used six acquires from here
                            const float* p = kvw + (h + 1) * 1024;
                            qa_ld16<0>(kc[(h + 1) & 1][0], voff, p);
                            qa_ld16<1024>(kc[(h + 1) & 1][1], voff, p);
                            qa_ld16<2048>(vc[(h + 1) & 1][0], voff, p);
                            qa_ld16<3072>(vc[(h + 1) & 1][1], voff, p);
                            ex2 += 4;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // (hand-issued fragment reads, 4 pairs in flight, were measured on the one-group form: 3,937-3,967 vs 3,951-3,960 frames/s on
                    // --model c35 f16x3, profiles/r05tq_c35_f16x3_ab.txt -- the kernel is bound by LDS bytes, not by read latency)
#pragma unroll
                    for (int ks4 = 0; ks4 < 4; ++ks4)
#pragma unroll
                        for (int ft = 0; ft < 2; ++ft) {
                            const u32x4 whi = frag(stg, (ks4 * 2 + ft) * 2), wlo = frag(stg, (ks4 * 2 + ft) * 2 + 1);
                            const int ks = 4 * half + ks4;
#pragma unroll
                            for (int grp = 0; grp < G; ++grp) {
                                if (part < 2) {
                                    accm[grp][ft] = qa_mma(whi, xhi[grp][ks], accm[grp][ft]);
                                    accc[grp][ft] = qa_mma(whi, xlo[grp][ks], accc[grp][ft]);
                                    accc[grp][ft] = qa_mma(wlo, xhi[grp][ks], accc[grp][ft]);
                                } else {
                                    accm[grp][ft] = qa_mma(xhi[grp][ks], whi, accm[grp][ft]);
                                    accc[grp][ft] = qa_mma(xlo[grp][ks], whi, accc[grp][ft]);
                                    accc[grp][ft] = qa_mma(xhi[grp][ks], wlo, accc[grp][ft]);
                                }
                            }
                        }
                }
#pragma unroll
                for (int grp = 0; grp < G; ++grp)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) {
                        const f32x4 val = accm[grp][ft] + accc[grp][ft] * (1.0f / 2048.0f);
                        if (part == 0) qv[grp][ft] = val;
                        else if (part == 1) kk[grp][ft] = val;
                        else vv[grp][ft] = val;
                    }
                if constexpr (MODE == 1) {   // dump the accumulators (lane-linear: whole lines)
#pragma unroll
                    for (int grp = 0; grp < G; ++grp) {
                        int l4 = lane * 4;
                        asm volatile("" : "+v"(l4));   // (formed here: see lane_off)
                        float* kd = kvw + grp * (NH * 1024) + h * 1024 + l4;
                        if (part == 1) {
                            *reinterpret_cast<f32x4*>(kd) = kk[grp][0];
                            *reinterpret_cast<f32x4*>(kd + 256) = kk[grp][1];
                            ex2 += 2;
                        } else if (part == 2) {
                            *reinterpret_cast<f32x4*>(kd + 512) = vv[grp][0];
                            *reinterpret_cast<f32x4*>(kd + 768) = vv[grp][1];
                            ex2 += 2;
                        }
                    }
                }
            }
            if constexpr (MODE == 2) {
                // this head's cached accumulators landed at least three acquires ago (header); from here on they are ordinary values
                asm volatile("" : "+v"(kc[h & 1][0]), "+v"(kc[h & 1][1]), "+v"(vc[h & 1][0]), "+v"(vc[h & 1][1]));
            }
#pragma unroll
            for (int grp = 0; grp < G; ++grp) {
                // attention over the frame slots (attention.py:48-58), f32: lane = query frame r, keys 4 g + e
                f32x4 st = {0.f, 0.f, 0.f, 0.f}, so = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st = qa_mma4(MODE == 2 ? kc[h & 1][ft][e] : kk[grp][ft][e], qv[grp][ft][e], st);
                        if constexpr (MODE == 2) so = qa_mma4(kk[grp][ft][e], qv[grp][ft][e], so);   // own keys: the diagonal is used
                    }
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
                f32x4 p, po;
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
                const float inv = 1.0f / sum;
                f32x4 o[2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    o[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[ft] = qa_mma4(MODE == 2 ? vc[h & 1][ft][e] : vv[grp][ft][e], p[e], o[ft]);
                        if constexpr (MODE == 2) o[ft] = qa_mma4(vv[grp][ft][e], po[e], o[ft]);
                    }
                    o[ft] *= inv;
                }
                // the head's 32 output features of the group's 16 tokens -> the proj GEMM's operand planes, row-major through the tile
                *reinterpret_cast<f32x4*>(tile + r * 36 + 4 * g) = o[0];
                *reinterpret_cast<f32x4*>(tile + r * 36 + 16 + 4 * g) = o[1];
                qa_wave_lds_fence();
                {
                    int t2 = lane >> 2, c8 = (lane & 3) * 8;
                    asm volatile("" : "+v"(t2), "+v"(c8));
                    const f32x4 va = *reinterpret_cast<const f32x4*>(tile + t2 * 36 + c8), vb2 = *reinterpret_cast<const f32x4*>(tile + t2 * 36 + c8 + 4);
                    u32x4 hi, lo;
                    qa_split8(va, vb2, hi, lo);
                    uint16_t* dst = a16 + ((size_t)b * nf * S + s0 + grp) * D + h * 32 + (t2 * S * D + c8);
                    if (t2 < nf) {
                        *reinterpret_cast<u32x4*>(dst) = hi;
                        *reinterpret_cast<u32x4*>(dst + plane) = lo;
                    }
                    ex2 += 2;
                }
                qa_wave_lds_fence();
            }
        }
    }
    qa_wait_vm<0>();  // the ring's run-ahead stages must not outlive the workgroup's LDS allocation
}

#ifndef GENIE_VAR_TQA_MIN_CLIPS
#define GENIE_VAR_TQA_MIN_CLIPS 2
#endif
// The geometry the kernel covers.  cache_mode: the pass keeps the k, v accumulators in the layer's cache slice (clean pass / masked pass of
// the evaluator) -- ONE predicate for producer and consumer, as temporal_prefix_fused_takes: fewer frames than the model's T (a cache
// genie_frame_pass could continue never takes this form) and at least 11 (32 KB per position must fit the slice).
bool temporal_qkv_attn_f16x3_takes(const genie_cfg& c, const genie_attn_weights& aw, int B, int model_T, bool cache_mode) {
#ifdef GENIE_VAR_TQA_OFF   // (A/B variant: the unfused launches)
    return false;
#endif
    if (!(aw.fused_w16 && c.precision == GENIE_PREC_F16X3 && c.d_model == 256 && c.num_heads == 8 && c.head_dim == 32 && c.T >= 1 && c.T <= 16 &&
          c.S % 4 == 0 && !c.qk_norm && (long)B * c.S >= GENIE_VAR_TQA_MIN_CLIPS * 256))
        return false;
    return !cache_mode || (c.T >= 11 && c.T < model_T && model_T <= 16);
}

// mode 0: plain causal pass; 1: clean pass (kv written); 2: masked pass (kv read; query slot i sees cached slots j < i + shift and itself)
int launch_temporal_qkv_attn_f16x3(const genie_cfg& c, const genie_attn_weights& aw, const float* x, uint16_t* a16, long plane, float* kv, int B,
                                   int mode, int shift, int model_T, hipStream_t st) {
    if (!temporal_qkv_attn_f16x3_takes(c, aw, B, model_T, mode != 0)) return GENIE_E_UNSUPPORTED;
    GENIE_CHECK_ARG(x && a16 && (mode == 0 || kv) && mode >= 0 && mode <= 2 && (shift == 0 || shift == 1), "temporal_qkv_attn_f16x3: bad argument");
    const bool qb = c.qkv_bias && aw.qkv_b;
#ifndef GENIE_VAR_TQA_G2
#define GENIE_VAR_TQA_G2 1     // two 16-token groups per wave where the registers allow (plain / clean pass, no qkv bias)
#endif
    const int G = (GENIE_VAR_TQA_G2 && mode != 2 && !qb && c.S % 8 == 0) ? 2 : 1;
    const int n_blocks = B * c.S / (4 * G);
    static const int cus = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int grid = n_blocks < 2 * cus ? n_blocks : 2 * cus;
    const double M = (double)B * c.T * c.S;
    ProfScope prof(GENIE_KC_FUSED, M * (2.0 * 256 * 768 + 4.0 * 16 * 256 * (mode == 2 ? 2 : 1)), M * (1024.0 + 1024.0 + (mode ? 2048.0 : 0.0)), st,
                   mode == 0 ? "temporal_qkv_attn_f16x3_kernel<0> (qkv + causal attention over T, split-f16 operands)"
                   : mode == 1 ? "temporal_qkv_attn_f16x3_kernel<1> (clean pass: qkv + causal attention, k / v accumulators out)"
                               : "temporal_qkv_attn_f16x3_kernel<2> (masked pass: qkv + attention over cached accumulators)");
    const size_t lds = QA_RING + 4096 + 4 * 2304;   // ring, bias, one 16 x 36-float tile per wave
    const float sl2e = c.attn_scale * 1.4426950408889634f;
#define QA_LAUNCH(QB_, MODE_, G_)                                                                                                            \
    do {                                                                                                                                 \
        static PerDevice<bool> once;                                                                                                     \
        if (once.needs()) {                                                                                                              \
            (void)hipFuncSetAttribute((const void*)temporal_qkv_attn_f16x3_kernel<QB_, MODE_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            once.set(true);                                                                                                              \
        }                                                                                                                                \
        temporal_qkv_attn_f16x3_kernel<QB_, MODE_, G_><<<grid, 256, lds, st>>>(x, aw.fused_w16, QB_ ? aw.qkv_b : nullptr, a16, plane, kv,   \
                                                                            n_blocks, c.S, c.T, shift, sl2e);                            \
    } while (0)
    if (mode == 0) { if (qb) QA_LAUNCH(true, 0, 1); else if (G == 2) QA_LAUNCH(false, 0, 2); else QA_LAUNCH(false, 0, 1); }
    else if (mode == 1) { if (qb) QA_LAUNCH(true, 1, 1); else if (G == 2) QA_LAUNCH(false, 1, 2); else QA_LAUNCH(false, 1, 1); }
    else { if (qb) QA_LAUNCH(true, 2, 1); else QA_LAUNCH(false, 2, 1); }
#undef QA_LAUNCH
    GENIE_LAUNCH_CHECK("temporal_qkv_attn_f16x3");
    return GENIE_OK;
}

}  // namespace genie
