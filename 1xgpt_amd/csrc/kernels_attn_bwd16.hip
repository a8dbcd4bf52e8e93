// Spatial attention BACKWARD on the bf16 matrix cores, for the bf16 training precision (what `accelerate --mixed_precision bf16`
// computes; the exact and f16x3 precisions keep the f32-MFMA kernel of kernels_train.hip).  S = 256 tokens of one frame,
// head_dim 32 / 64, non-causal (genie/attention.py:36-61 differentiated):
//     P = softmax(scale Q K^T),  dV = P^T dO,  dP = dO V^T,  D_i = sum_j P_ij dP_ij,  dS = P (dP - D),
//     dQ = scale dS K,  dK = scale dS^T Q.
// Operands of every product are rounded to bf16 (Q, K, V, dO, P, dS), accumulation and the softmax are f32.  Nothing of size
// S x S touches HBM.  Two kernels, one workgroup of 8 waves per (frame, head) each, so that every accumulator layout is used
// for the products it feeds WITHOUT transposing score tiles:
//   * attn_bwd16_q_kernel  -- "one query per lane": S^T = K Q^T tiles (lane = query, registers = keys) for the wave's 32
//     queries stay in registers (8 tiles); row max / sum in-lane; D from a second sweep of dP^T = V dO^T; dS^T in registers is
//     directly the A operand of dQ = dS K (contraction over keys), whose B operand is K^T from LDS.  Writes dQ and the row
//     statistics (max in log2 units, 1/sum, D) for the second kernel.
//   * attn_bwd16_kv_kernel -- "one key per lane": S = Q K^T and dP = dO V^T tiles (lane = key, registers = queries) for the
//     wave's 32 keys; P^T and dS^T in registers are the A operands of dV = P^T dO and dK = dS^T Q (contraction over queries),
//     B operands dO^T, Q^T from LDS.
// LDS images: row-major operands as 16-byte slots XOR-swizzled per row (conflict-free ds_read_b128, as in the forward
// kernels); transposed operands [feature][256] with every 16-index group stored {0-3, 8-11 | 4-7, 12-15} -- the order in which
// an MFMA accumulator tile holds its rows -- so a lane's 8 contraction indices are ONE 16-byte unit, units XOR-swizzled with
// the feature row.
#include "common.hpp"
#include "kernels.hpp"

namespace genie {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mma_bf16(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int rowmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// 8 consecutive f32 -> bf16x8
__device__ __forceinline__ s16x8 pack8(const float4& a, const float4& b) {
    s16x8 v;
    v[0] = (short)f32_to_bf16(a.x); v[1] = (short)f32_to_bf16(a.y); v[2] = (short)f32_to_bf16(a.z); v[3] = (short)f32_to_bf16(a.w);
    v[4] = (short)f32_to_bf16(b.x); v[5] = (short)f32_to_bf16(b.y); v[6] = (short)f32_to_bf16(b.z); v[7] = (short)f32_to_bf16(b.w);
    return v;
}
__device__ __forceinline__ s16x8 pack8(const float* v8) {
    s16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (short)f32_to_bf16(v8[i]);
    return v;
}

// Stage a (256, DH) f32 matrix (row stride ld) into LDS as bf16: `rows` = row-major image (may be NULL), `tr` = transposed
// permuted image (may be NULL).  512 threads: thread -> (row, half of the row).
template <int DH>
__device__ __forceinline__ void stage_256xDH(const float* __restrict__ src, long ld, unsigned char* rows, unsigned char* tr, int tid) {
    constexpr int ROWB = DH * 2, SPR = ROWB / 16, RPB = 256 / ROWB, HF = DH / 2;
    const int row = tid >> 1, hh = tid & 1;
    const float* p = src + (size_t)row * ld + hh * HF;
    float v[HF];
#pragma unroll
    for (int i = 0; i < HF / 4; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    if (rows) {
#pragma unroll
        for (int i = 0; i < HF / 8; ++i) {
            const int slot = hh * (HF / 8) + i;
            *reinterpret_cast<s16x8*>(rows + row * ROWB + ((slot ^ ((row / RPB) % SPR)) << 4)) = pack8(v + 8 * i);
        }
    }
    if (tr) {
        // index `row` inside feature row f: group g = row / 16, unit j = (row / 4) & 1, position (row & 3) + 4 * ((row / 8) & 1)
        const int g = row >> 4, kl = row & 15;
        const int u = 2 * g + ((kl >> 2) & 1), pos = (kl & 3) + 4 * (kl >> 3);
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int f = hh * HF + i;
            *reinterpret_cast<uint16_t*>(tr + f * 512 + ((u ^ (f & 31)) << 4) + pos * 2) = f32_to_bf16(v[i]);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- dQ + statistics
template <int DH>
__global__ __launch_bounds__(512, 2) void attn_bwd16_q_kernel(const float* __restrict__ qkv, const float* __restrict__ qk, long qk_ld,
                                                              const float* __restrict__ dO, float* __restrict__ dqkv,
                                                              float* __restrict__ stats, int d, int H, float scale) {
    constexpr int S = 256, ROWB = DH * 2, SPR = ROWB / 16, RPB = 256 / ROWB, KS = DH / 16, NF = DH / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;                 // [256][DH] rows
    unsigned char* sV = sK + S * ROWB;        // [256][DH] rows
    unsigned char* sKT = sV + S * ROWB;       // [DH][256] transposed, permuted
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long bt = blockIdx.x / H;
    const int head = (int)(blockIdx.x - bt * H);
    const size_t row0 = (size_t)bt * S;
    const float* qb = qk + row0 * qk_ld + head * DH;
    const float* kb = qb + d;
    const float* vb = qkv + row0 * 3 * d + 2 * d + head * DH;
    const float* ob = dO + row0 * d + head * DH;
    stage_256xDH<DH>(kb, qk_ld, sK, sKT, tid);
    stage_256xDH<DH>(vb, 3L * d, sV, nullptr, tid);

    // this wave's queries as B operands: lane (r, h) = query 32w + r, features 16 ks + 8 h ..
    s16x8 qf[KS], of[KS];
    {
        const float* qp = qb + (size_t)(w * 32 + r) * qk_ld + 8 * h;
        const float* op = ob + (size_t)(w * 32 + r) * d + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = pack8(*reinterpret_cast<const float4*>(qp + 16 * ks), *reinterpret_cast<const float4*>(qp + 16 * ks + 4));
            of[ks] = pack8(*reinterpret_cast<const float4*>(op + 16 * ks), *reinterpret_cast<const float4*>(op + 16 * ks + 4));
        }
    }
    __syncthreads();
    const float scale_l2 = scale * 1.4426950408889634f;
    auto frag = [&](const unsigned char* base, int row, int ks) {   // row-major image: 16-byte slot 2 ks + h of `row`
        return *reinterpret_cast<const s16x8*>(base + row * ROWB + (((2 * ks + h) ^ ((row / RPB) % SPR)) << 4));
    };
    // ---- sweep 1: S^T tiles (keys 32 jt + rowmap(e, h), query r), kept in registers; row max and sum
    f32x16 sT[8];
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
        f32x16 a;
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = mma_bf16(frag(sK, jt * 32 + r, ks), qf[ks], a);
#pragma unroll
        for (int e = 0; e < 16; ++e) { a[e] *= scale_l2; m = fmaxf(m, a[e]); }
        sT[jt] = a;
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { sT[jt][e] = __builtin_amdgcn_exp2f(sT[jt][e] - m); l += sT[jt][e]; }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    // ---- sweep 2: D = sum_j P dP (dP^T = V dO^T tiles, not kept)
    float dsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; ++e) p[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) p = mma_bf16(frag(sV, jt * 32 + r, ks), of[ks], p);
#pragma unroll
        for (int e = 0; e < 16; ++e) dsum = fmaf(sT[jt][e], p[e], dsum);
    }
    dsum += __shfl_xor(dsum, 32);
    const float D = dsum * inv;
    if (h == 0) {
        float* sp = stats + ((size_t)blockIdx.x * S + w * 32 + r) * 4;
        *reinterpret_cast<float4*>(sp) = make_float4(m, inv, D, 0.f);
    }
    // ---- sweep 3: dS^T = P (dP - D) -> A operand of dQ += dS K (B = K^T from LDS)
    f32x16 dq[NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[ft][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; ++e) p[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) p = mma_bf16(frag(sV, jt * 32 + r, ks), of[ks], p);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {   // 16 keys per MFMA step: registers 8 k2 .. 8 k2 + 7 = unit h of key group 2 jt + k2
            s16x8 a;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * k2 + i;
                a[i] = (short)f32_to_bf16(sT[jt][e] * inv * (p[e] - D));
            }
            const int u = 2 * (2 * jt + k2) + h;
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
                const int f = 32 * ft + r;
                const s16x8 b = *reinterpret_cast<const s16x8*>(sKT + f * 512 + ((u ^ (f & 31)) << 4));
                dq[ft] = mma_bf16(a, b, dq[ft]);
            }
        }
    }
    float* outb = dqkv + row0 * 3 * d + head * DH;
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            outb[(size_t)(w * 32 + rowmap(e, h)) * 3 * d + 32 * ft + r] = dq[ft][e] * scale;
}

// ---------------------------------------------------------------------------------------------------- dK, dV
template <int DH>
__global__ __launch_bounds__(512, 2) void attn_bwd16_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ qk, long qk_ld,
                                                               const float* __restrict__ dO, float* __restrict__ dqkv,
                                                               const float* __restrict__ stats, int d, int H, float scale) {
    constexpr int S = 256, ROWB = DH * 2, SPR = ROWB / 16, RPB = 256 / ROWB, KS = DH / 16, NF = DH / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQ = smem;                  // [256][DH] rows
    unsigned char* sO = sQ + S * ROWB;         // dO rows
    unsigned char* sQT = sO + S * ROWB;        // [DH][256]
    unsigned char* sOT = sQT + DH * 512;       // dO^T
    float* sSt = reinterpret_cast<float*>(sOT + DH * 512);  // [256][4] = (max, 1/sum, D, -)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long bt = blockIdx.x / H;
    const int head = (int)(blockIdx.x - bt * H);
    const size_t row0 = (size_t)bt * S;
    const float* qb = qk + row0 * qk_ld + head * DH;
    const float* kb = qb + d;
    const float* vb = qkv + row0 * 3 * d + 2 * d + head * DH;
    const float* ob = dO + row0 * d + head * DH;
    stage_256xDH<DH>(qb, qk_ld, sQ, sQT, tid);
    stage_256xDH<DH>(ob, (long)d, sO, sOT, tid);
    if (tid < S) *reinterpret_cast<float4*>(sSt + tid * 4) = *reinterpret_cast<const float4*>(stats + ((size_t)blockIdx.x * S + tid) * 4);
    // this wave's keys as B operands: lane (r, h) = key 32w + r, features 16 ks + 8 h ..
    s16x8 kf[KS], vf[KS];
    {
        const float* kp = kb + (size_t)(w * 32 + r) * qk_ld + 8 * h;
        const float* vp = vb + (size_t)(w * 32 + r) * 3 * d + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = pack8(*reinterpret_cast<const float4*>(kp + 16 * ks), *reinterpret_cast<const float4*>(kp + 16 * ks + 4));
            vf[ks] = pack8(*reinterpret_cast<const float4*>(vp + 16 * ks), *reinterpret_cast<const float4*>(vp + 16 * ks + 4));
        }
    }
    __syncthreads();
    const float scale_l2 = scale * 1.4426950408889634f;
    auto frag = [&](const unsigned char* base, int row, int ks) {
        return *reinterpret_cast<const s16x8*>(base + row * ROWB + (((2 * ks + h) ^ ((row / RPB) % SPR)) << 4));
    };
    f32x16 dk[NF], dv[NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[ft][e] = 0.f; dv[ft][e] = 0.f; }
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        // S, dP tiles: rows = queries 32 it + rowmap(e, h), column = key r of this wave
        f32x16 s, p;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; p[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            s = mma_bf16(frag(sQ, it * 32 + r, ks), kf[ks], s);
            p = mma_bf16(frag(sO, it * 32 + r, ks), vf[ks], p);
        }
        float pv[16], ds[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float4 st = *reinterpret_cast<const float4*>(sSt + (it * 32 + rowmap(e, h)) * 4);   // (max, 1/sum, D) of that query
            pv[e] = __builtin_amdgcn_exp2f(fmaf(s[e], scale_l2, -st.x)) * st.y;
            ds[e] = pv[e] * (p[e] - st.z);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const s16x8 ap = pack8(pv + 8 * k2), as = pack8(ds + 8 * k2);
            const int u = 2 * (2 * it + k2) + h;
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
                const int f = 32 * ft + r;
                const int off = f * 512 + ((u ^ (f & 31)) << 4);
                dv[ft] = mma_bf16(ap, *reinterpret_cast<const s16x8*>(sOT + off), dv[ft]);
                dk[ft] = mma_bf16(as, *reinterpret_cast<const s16x8*>(sQT + off), dk[ft]);
            }
        }
    }
    float* outb = dqkv + row0 * 3 * d + head * DH;
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* o = outb + (size_t)(w * 32 + rowmap(e, h)) * 3 * d + 32 * ft + r;
            o[d] = dk[ft][e] * scale;
            o[2 * d] = dv[ft][e];
        }
}

// qkv (M, 3d) f32 saved by the forward, qk = where q and k are read (qk_ld: their row stride; qk-norm variants pass the
// normalised copies), dO (M, d) f32 -> dqkv (M, 3d) f32.  stats: n_bt * H * 256 * 4 floats of scratch.
// GENIE_E_UNSUPPORTED for other geometries (the caller then takes the f32 kernel).
int launch_attn_spatial_bwd_bf16(const float* qkv, const float* qk, long qk_ld, const float* dO, float* dqkv, float* stats, long n_bt,
                                 int S, int d, int H, int Dh, float scale, hipStream_t st) {
    static const int on = study_env("GENIE_ATTN_BWD16", 1);
    if (!on || S != 256 || (Dh != 64 && Dh != 32) || qk_ld % 4 || d % 4) return GENIE_E_UNSUPPORTED;
    if (n_bt <= 0) return GENIE_OK;
    ProfScope prof(GENIE_KC_ATTN_SPATIAL, 14.0 * S * S * Dh * (double)n_bt * H, 4.0 * 10 * S * Dh * (double)n_bt * H, st);
    const unsigned grid = (unsigned)(n_bt * H);
#define BWD16(DH_)                                                                                                        \
    do {                                                                                                                  \
        const int lq = 2 * 256 * DH_ * 2 + DH_ * 512, lkv = 2 * 256 * DH_ * 2 + 2 * DH_ * 512 + 256 * 16;                 \
        (void)hipFuncSetAttribute((const void*)attn_bwd16_q_kernel<DH_>, hipFuncAttributeMaxDynamicSharedMemorySize, lq);  \
        (void)hipFuncSetAttribute((const void*)attn_bwd16_kv_kernel<DH_>, hipFuncAttributeMaxDynamicSharedMemorySize, lkv); \
        attn_bwd16_q_kernel<DH_><<<grid, 512, lq, st>>>(qkv, qk, qk_ld, dO, dqkv, stats, d, H, scale);                    \
        attn_bwd16_kv_kernel<DH_><<<grid, 512, lkv, st>>>(qkv, qk, qk_ld, dO, dqkv, stats, d, H, scale);                  \
    } while (0)
    if (Dh == 64) BWD16(64); else BWD16(32);
#undef BWD16
    GENIE_LAUNCH_CHECK("attn_spatial_bwd_bf16");
    return GENIE_OK;
}

}  // namespace genie
