// Spatial attention on the f16 matrix cores with split operands (the attention counterpart of gemm16<2>):
// every matmul operand x is carried as hi + lo/2048 (two f16), each product as hi.hi + (hi.lo + lo.hi)/2048
// accumulated in f32 -> f32-class scores and outputs at 3 MFMAs of v_mfma_f32_32x32x16_f16 per algorithmic
// MFMA, ~5x less matrix-pipe time than the f32-MFMA kernel (32x32x2 f32 at 64 cycles per K=2).
//
// One workgroup per (sequence of S = 256 tokens, head), 8 waves x 32 queries.
//   LDS:  K  as [key][Dh] f16 hi/lo, rows XOR-swizzled like the GEMM tiles (A operand of S^T = K Q^T)
//         V^T as [d][key] f16 hi/lo, pitch 520 B (B operand of O = P V needs key-contiguous columns)
//   S^T = K Q^T "swapped" so a lane holds 128 scores of ITS query (softmax = in-lane + one cross-half shuffle);
//   the normalised probabilities are split in registers and fed as the A operand of P V: slot s of MFMA m
//   of key tile kt is key kt*32 + 4h + (s&3) + 8*(s>>2) + 16*m, and V^T is read in exactly that order.
#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_h(float a, _Float16& hi, _Float16& lo) {
    _Float16 h = (_Float16)a;
    float hf = (float)h;
    if (fabsf(hf) < 6.103515625e-05f) { h = (_Float16)0.0f; hf = 0.0f; }
    hi = h;
    lo = (_Float16)((a - hf) * 2048.0f);
}

#ifdef GENIE_STUDY
__device__ unsigned long long* g_split_stamps = nullptr;   // GENIE_SPLIT_STAMPS=1: workgroup (0,0,0), wave 0 stamps s_memtime per phase
#define SPLIT_STAMP(i) do { if (g_split_stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_split_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPLIT_STAMP(i) do { } while (0)
#endif

template <int DH>
__global__ __launch_bounds__(512) void attn_spatial_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 int d, float scale, const float* __restrict__ nw,
                                                                 const float* __restrict__ nb,
                                                                 uint16_t* __restrict__ out16, size_t plane, int qw) {
    // qw = query waves per workgroup: 8 = the whole sequence; 1, 2, 4 = the workgroup stages all keys with its 8 waves but
    // only waves < qw go on, with queries (blockIdx.z * qw + wave) * 32 ..: few-sequence launches (batch-1 generate: 8
    // (sequence, head) pairs) spread over 8 / qw times as many CUs
    constexpr int S = 256, NKT = 8;
    constexpr int ROWB = DH * 2, SPR = ROWB / 16, RPB = 256 / ROWB;  // K rows: bytes, 16-B slots, rows per 256 B
    constexpr int K_PLANE = S * ROWB;                                 // bytes
    constexpr int VT_PITCH = 520, VT_PLANE = DH * VT_PITCH;           // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sKh = smem;
    unsigned char* sKl = smem + K_PLANE;
    unsigned char* sVh = smem + 2 * K_PLANE;
    unsigned char* sVl = sVh + VT_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * S;
    const int head = blockIdx.y;
    const float* base = qkv + (size_t)row0 * 3 * d + head * DH;
    SPLIT_STAMP(0);

    // ---- stage K: two adjacent lanes per key row, DH/2 features each (qk-norm statistics span both)
    {
        const int rr = tid >> 1, half = tid & 1;
        const float* kp = base + (size_t)rr * 3 * d + d + half * (DH / 2);
        float kx[DH / 2];
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float4 t = *reinterpret_cast<const float4*>(kp + 4 * c);
            kx[4 * c] = t.x; kx[4 * c + 1] = t.y; kx[4 * c + 2] = t.z; kx[4 * c + 3] = t.w;
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
        for (int sl = 0; sl < DH / 16; ++sl) {  // 8 features = one 16-byte slot
            f16x8 vh, vl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 a, b; split_h(kx[sl * 8 + j], a, b); vh[j] = a; vl[j] = b; }
            const int slot = half * (DH / 16) + sl;
            const int off = rr * ROWB + ((slot ^ ((rr / RPB) % SPR)) << 4);
            *reinterpret_cast<f16x8*>(sKh + off) = vh;
            *reinterpret_cast<f16x8*>(sKl + off) = vl;
        }
    }
    // ---- stage V^T.  Global side like K: two adjacent lanes per key row read its 4*DH bytes contiguously (the former
    // (key pair, 8-feature) tasks touched 32 bytes per 6 KB-strided row and lane: 4x the L2 traffic).  A 32-bit LDS word
    // packs keys (2p, 2p+1) of one feature, so each lane swaps its values with the lane holding the neighbouring key
    // (lane ^ 2); the even-key lane then writes the first half of its features, the odd-key lane the second half.
    {
        const int rr = tid >> 1, half = tid & 1;
        const float* vp = base + (size_t)rr * 3 * d + 2 * d + half * (DH / 2);
        float vx[DH / 2];
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float4 t = *reinterpret_cast<const float4*>(vp + 4 * c);
            vx[4 * c] = t.x; vx[4 * c + 1] = t.y; vx[4 * c + 2] = t.z; vx[4 * c + 3] = t.w;
        }
        const bool odd = rr & 1;
        const int kp2 = rr >> 1;
#pragma unroll
        for (int c = 0; c < DH / 2; ++c) {
            const float other = __shfl_xor(vx[c], 2);
            const bool mine = (c < DH / 4) != odd;  // even key: features [0, DH/4) of its half, odd key: [DH/4, DH/2)
            if (mine) {
                const float a = odd ? other : vx[c], b = odd ? vx[c] : other;  // (key 2p, key 2p+1)
                _Float16 ah, al, bh, bl;
                split_h(a, ah, al);
                split_h(b, bh, bl);
                const int off = (half * (DH / 2) + c) * VT_PITCH + kp2 * 4;
                *reinterpret_cast<uint32_t*>(sVh + off) =
                    (uint32_t)__builtin_bit_cast(uint16_t, ah) | ((uint32_t)__builtin_bit_cast(uint16_t, bh) << 16);
                *reinterpret_cast<uint32_t*>(sVl + off) =
                    (uint32_t)__builtin_bit_cast(uint16_t, al) | ((uint32_t)__builtin_bit_cast(uint16_t, bl) << 16);
            }
        }
    }
    SPLIT_STAMP(1);
    __syncthreads();
    SPLIT_STAMP(2);
    if (qw > 0 && wid >= qw) return;

    const int qb = qw > 0 ? blockIdx.z * qw + wid : blockIdx.z;  // 32 queries per wave (qw = 0: per workgroup)
    // ---- Q fragments: lane (r,h) holds Q[r][16kk + 8h + j], split, scale folded in before the split
    f16x8 qh[DH / 16], ql[DH / 16];
    const float scale_l2 = scale * 1.4426950408889634f;  // scores carried as s*log2(e): softmax = one v_exp_f32 each
    {
        float qf[DH / 2];
        const float* qp = base + (size_t)(qb * 32 + r) * 3 * d + 8 * h;
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk) {
            float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * kk);
            float4 t1 = *reinterpret_cast<const float4*>(qp + 16 * kk + 4);
            qf[8 * kk] = t0.x; qf[8 * kk + 1] = t0.y; qf[8 * kk + 2] = t0.z; qf[8 * kk + 3] = t0.w;
            qf[8 * kk + 4] = t1.x; qf[8 * kk + 5] = t1.y; qf[8 * kk + 6] = t1.z; qf[8 * kk + 7] = t1.w;
        }
        if (nw) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) s += qf[c];
            s += __shfl_xor(s, 32);
            const float mu = s / DH;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < DH / 2; ++c) { float t = qf[c] - mu; v += t * t; }
            v += __shfl_xor(v, 32);
            const float rs = 1.0f / sqrtf(v / DH + 1e-5f);
#pragma unroll
            for (int kk = 0; kk < DH / 16; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cc = 16 * kk + 8 * h + j;
                    qf[8 * kk + j] = (qf[8 * kk + j] - mu) * rs * nw[cc] + nb[cc];
                }
        }
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 a, b;
                split_h(qf[8 * kk + j] * scale_l2, a, b);  // q *= scale (attention.py:48), in log2 units (see softmax)
                qh[kk][j] = a; ql[kk][j] = b;
            }
    }
    SPLIT_STAMP(3);
    if (qw == 0) {
        // ---- KEY-SPLIT mode (a handful of sequences, e.g. one frame of batch-1 generate): the workgroup owns 32 queries and
        // its 8 waves take one 32-key tile each -- 24 MFMAs and 16 exponentials per wave instead of 192 and 128 in one wave --
        // then the partial (max, sum, O) triples are merged like an online softmax, in wave order.
        const int kt = wid;
        f32x16 a0, c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0[e] = 0.f; c0[e] = 0.f; }
        const int rowk = kt * 32 + r;
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk) {
            const int off = rowk * ROWB + (((2 * kk + h) ^ ((rowk / RPB) % SPR)) << 4);
            const f16x8 kh = *reinterpret_cast<const f16x8*>(sKh + off);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(sKl + off);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[kk], a0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kk], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kk], c0, 0, 0, 0);
        }
        float p[16];
        float mw = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) { p[e] = a0[e] + c0[e] * (1.0f / 2048.0f); mw = fmaxf(mw, p[e]); }
        mw = fmaxf(mw, __shfl_xor(mw, 32));
        float lw = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(p[e] - mw); lw += p[e]; }
        lw += __shfl_xor(lw, 32);
        SPLIT_STAMP(4);
        float* stat = reinterpret_cast<float*>(smem + 2 * K_PLANE + 2 * VT_PLANE);  // [8 waves][32 queries][2]
        if (h == 0) { stat[(wid * 32 + r) * 2] = mw; stat[(wid * 32 + r) * 2 + 1] = lw; }
        f32x16 oa[DH / 32], oc[DH / 32];
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { oa[dt][e] = 0.f; oc[dt][e] = 0.f; }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f16x8 ph, pl;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                _Float16 a, b;
                split_h(p[8 * m + s8], a, b);
                ph[s8] = a; pl[s8] = b;
            }
            const int key0 = kt * 32 + 4 * h + 16 * m;
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const int off = (dt * 32 + r) * VT_PITCH + key0 * 2;
                f16x8 vh, vl;
                const f16x4 h0 = *reinterpret_cast<const f16x4*>(sVh + off);
                const f16x4 h1 = *reinterpret_cast<const f16x4*>(sVh + off + 16);
                const f16x4 l0 = *reinterpret_cast<const f16x4*>(sVl + off);
                const f16x4 l1 = *reinterpret_cast<const f16x4*>(sVl + off + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vh[j] = h0[j]; vh[4 + j] = h1[j]; vl[j] = l0[j]; vl[4 + j] = l1[j]; }
                oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, oa[dt], 0, 0, 0);
                oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, oc[dt], 0, 0, 0);
                oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, oc[dt], 0, 0, 0);
            }
        }
        SPLIT_STAMP(5);
        __syncthreads();  // every wave has read its K tile: the K planes become the partial-output scratch [wave][32 q][DH]
        float* op = reinterpret_cast<float*>(smem) + (size_t)wid * 32 * DH;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                op[((e & 3) + 8 * (e >> 2) + 4 * h) * DH + dt * 32 + r] = oa[dt][e] + oc[dt][e] * (1.0f / 2048.0f);
        __syncthreads();
        SPLIT_STAMP(6);
        // merge: thread -> (query, 4 features); waves in order
        constexpr int F4 = DH / 4;
        for (int idx = tid; idx < 32 * F4; idx += 512) {
            const int q = idx / F4, f4 = (idx % F4) * 4;
            float mg = -INFINITY;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) mg = fmaxf(mg, stat[(w8 * 32 + q) * 2]);
            float l = 0.f;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
                const float sc8 = __builtin_amdgcn_exp2f(stat[(w8 * 32 + q) * 2] - mg);
                l += stat[(w8 * 32 + q) * 2 + 1] * sc8;
                const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem) + ((size_t)w8 * 32 + q) * DH + f4);
                o.x += t.x * sc8; o.y += t.y * sc8; o.z += t.z * sc8; o.w += t.w * sc8;
            }
            const float invl = 1.0f / l;
            o.x *= invl; o.y *= invl; o.z *= invl; o.w *= invl;
            const size_t oi = (size_t)(row0 + qb * 32 + q) * d + head * DH + f4;
            if (!out16) *reinterpret_cast<float4*>(out + oi) = o;
            else if (plane) {
                uint32_t h01, h23, l01, l23;
                split_f16_x4(o.x, o.y, o.z, o.w, h01, h23, l01, l23);
                *reinterpret_cast<uint2*>(out16 + oi) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(out16 + plane + oi) = make_uint2(l01, l23);
            } else {
                *reinterpret_cast<uint2*>(out16 + oi) = make_uint2((uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16),
                                                                   (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16));
            }
        }
        SPLIT_STAMP(7);
        return;
    }
    // ---- S^T tiles: sc[kt][e] = score(key kt*32 + (e&3) + 8(e>>2) + 4h, query r)
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x16 a0, c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0[e] = 0.f; c0[e] = 0.f; }
        const int rowk = kt * 32 + r;
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk) {
            const int off = rowk * ROWB + (((2 * kk + h) ^ ((rowk / RPB) % SPR)) << 4);
            const f16x8 kh = *reinterpret_cast<const f16x8*>(sKh + off);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(sKl + off);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[kk], a0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kk], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kk], c0, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kt][e] = a0[e] + c0[e] * (1.0f / 2048.0f);
    }
    // ---- softmax over the 256 keys of query r (128 here, 128 in lane r^32).  The kernel is VALU-bound (128 exponentials
    // and 128 operand splits per lane against 192 MFMAs per wave), so exp is the bare v_exp_f32 on log2-scaled scores
    // (1 ulp) rather than the ~15-instruction expf.
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
    f32x16 oa[DH / 32], oc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oa[dt][e] = 0.f; oc[dt][e] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f16x8 ph, pl;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                _Float16 a, b;
                split_h(sc[kt][8 * m + s8] * inv, a, b);
                ph[s8] = a; pl[s8] = b;
            }
            const int key0 = kt * 32 + 4 * h + 16 * m;  // slots 0..3 -> key0 + 0..3, slots 4..7 -> key0 + 8..11
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                const int off = (dt * 32 + r) * VT_PITCH + key0 * 2;
                f16x8 vh, vl;
                const f16x4 h0 = *reinterpret_cast<const f16x4*>(sVh + off);
                const f16x4 h1 = *reinterpret_cast<const f16x4*>(sVh + off + 16);
                const f16x4 l0 = *reinterpret_cast<const f16x4*>(sVl + off);
                const f16x4 l1 = *reinterpret_cast<const f16x4*>(sVl + off + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vh[j] = h0[j]; vh[4 + j] = h1[j]; vl[j] = l0[j]; vl[4 + j] = l1[j]; }
                oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, oa[dt], 0, 0, 0);
                oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, oc[dt], 0, 0, 0);
                oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, oc[dt], 0, 0, 0);
            }
        }
    // ---- store: row = query (e&3) + 8(e>>2) + 4h, col = feature r
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const size_t oi = (size_t)(row0 + q) * d + head * DH + dt * 32 + r;
            const float v = oa[dt][e] + oc[dt][e] * (1.0f / 2048.0f);
            if (!out16) out[oi] = v;
            else if (plane) { uint16_t hi, lo; split_f16(v, hi, lo); out16[oi] = hi; out16[plane + oi] = lo; }
            else out16[oi] = f32_to_bf16(v);
        }
}

// ------------------------------------------------------------------------------------------------
// The key-split form for a handful of (sequence, head) pairs (one frame of batch-1..4 generate) WITHOUT the LDS staging:
// workgroup = (sequence, head, block of 32 queries), wave w = key tile w.  Every wave reads exactly the fragments its 24 matrix
// instructions need straight from the f32 qkv rows (L2-resident: the qkv GEMM has just written them) in ONE round of loads --
//   K tile   lane (r, h): key 32 w + r, features 16 kk + 8 h .. + 7      (A operand of S^T = K Q^T)
//   Q block  lane (r, h): query 32 qb + r, the same features              (B operand)
//   V tile   lane (r, h): feature 32 dt + r of keys 32 w + 4 h + 16 m + {0..3, 8..11}   (B operand of O = P V, already "transposed")
// -- splits them in registers and merges the eight partial (max, sum, O) triples through LDS in wave order.  The staged kernel
// above spent half of its 31 k cycles converting and transposing ALL 256 keys in every one of the eight query-block workgroups of a
// head and then waited for its Q rows behind the staging barrier (per-phase stamps, GENIE_SPLIT_STAMPS).  No qk-norm here.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512) void attn_spatial_keysplit_kernel(const float* __restrict__ qkv, float* __restrict__ out, int d,
                                                                    float scale, uint16_t* __restrict__ out16, size_t plane) {
    constexpr int S = 256, KK = DH / 16, DT = DH / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem);                          // [8 waves][32 queries][DH]
    float* stat = reinterpret_cast<float*>(smem + 8 * 32 * DH * 4);         // [8 waves][32 queries][2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * S;
    const int head = blockIdx.y, qb = blockIdx.z, kt = wid;
    const size_t ld = (size_t)3 * d;
    const float* base = qkv + (size_t)row0 * ld + head * DH;
    typedef float kf4 __attribute__((ext_vector_type(4)));

    // ---- one round of loads: K tile rows, Q block rows (8 floats per 16-feature step), the V values of both P V steps
    kf4 kf[KK][2], qf[KK][2];
    float vf[2][DT][8];
    {
        const float* kp = base + (size_t)(kt * 32 + r) * ld + d + 8 * h;
        const float* qp = base + (size_t)(qb * 32 + r) * ld + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            kf[kk][0] = *reinterpret_cast<const kf4*>(kp + 16 * kk);
            kf[kk][1] = *reinterpret_cast<const kf4*>(kp + 16 * kk + 4);
            qf[kk][0] = *reinterpret_cast<const kf4*>(qp + 16 * kk);
            qf[kk][1] = *reinterpret_cast<const kf4*>(qp + 16 * kk + 4);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int key = kt * 32 + 4 * h + 16 * m + (s8 & 3) + 8 * (s8 >> 2);
                    vf[m][dt][s8] = base[(size_t)key * ld + 2 * d + dt * 32 + r];
                }
    }
    // ---- S^T tile = K Q^T on split operands (scores carried as s * log2 e: the scale goes into Q before the split)
    const float scale_l2 = scale * 1.4426950408889634f;
    f32x16 a0, c0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; c0[e] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        f16x8 kh, kl, qh, ql;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split_h(kf[kk][j >> 2][j & 3], a, b);
            kh[j] = a; kl[j] = b;
            split_h(qf[kk][j >> 2][j & 3] * scale_l2, a, b);
            qh[j] = a; ql[j] = b;
        }
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh, a0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh, c0, 0, 0, 0);
    }
    float p[16];
    float mw = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) { p[e] = a0[e] + c0[e] * (1.0f / 2048.0f); mw = fmaxf(mw, p[e]); }
    mw = fmaxf(mw, __shfl_xor(mw, 32));
    float lw = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { p[e] = __builtin_amdgcn_exp2f(p[e] - mw); lw += p[e]; }
    lw += __shfl_xor(lw, 32);
    if (h == 0) { stat[(wid * 32 + r) * 2] = mw; stat[(wid * 32 + r) * 2 + 1] = lw; }
    // ---- partial O = P V: register e = 8 m + s8 of the score tile is slot s8 of step m (key 32 w + 4 h + 16 m + (s8&3) + 8 (s8>>2))
    f32x16 oa[DT], oc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oa[dt][e] = 0.f; oc[dt][e] = 0.f; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f16x8 ph, pl;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            _Float16 a, b;
            split_h(p[8 * m + s8], a, b);
            ph[s8] = a; pl[s8] = b;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f16x8 vh, vl;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                _Float16 a, b;
                split_h(vf[m][dt][s8], a, b);
                vh[s8] = a; vl[s8] = b;
            }
            oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, oa[dt], 0, 0, 0);
            oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, oc[dt], 0, 0, 0);
            oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, oc[dt], 0, 0, 0);
        }
    }
    float* op = part + (size_t)wid * 32 * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            op[((e & 3) + 8 * (e >> 2) + 4 * h) * DH + dt * 32 + r] = oa[dt][e] + oc[dt][e] * (1.0f / 2048.0f);
    __syncthreads();
    // ---- merge like an online softmax, waves in order: thread -> (query, 4 features)
    constexpr int F4 = DH / 4;
    for (int idx = tid; idx < 32 * F4; idx += 512) {
        const int q = idx / F4, f4 = (idx % F4) * 4;
        float mg = -INFINITY;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) mg = fmaxf(mg, stat[(w8 * 32 + q) * 2]);
        float l = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
            const float sc8 = __builtin_amdgcn_exp2f(stat[(w8 * 32 + q) * 2] - mg);
            l += stat[(w8 * 32 + q) * 2 + 1] * sc8;
            const kf4 t = *reinterpret_cast<const kf4*>(part + ((size_t)w8 * 32 + q) * DH + f4);
            o.x += t.x * sc8; o.y += t.y * sc8; o.z += t.z * sc8; o.w += t.w * sc8;
        }
        const float invl = 1.0f / l;
        o.x *= invl; o.y *= invl; o.z *= invl; o.w *= invl;
        const size_t oi = (size_t)(row0 + qb * 32 + q) * d + head * DH + f4;
        if (!out16) *reinterpret_cast<float4*>(out + oi) = o;
        else if (plane) {
            uint32_t h01, h23, l01, l23;
            split_f16_x4(o.x, o.y, o.z, o.w, h01, h23, l01, l23);
            *reinterpret_cast<uint2*>(out16 + oi) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(out16 + plane + oi) = make_uint2(l01, l23);
        } else {
            *reinterpret_cast<uint2*>(out16 + oi) = make_uint2((uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16),
                                                               (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16));
        }
    }
}

// Same contract as launch_attn_spatial_f32_mfma (f32 qkv in; f32 / split-f16 / bf16 out); S = 256 only.
int launch_attn_spatial_split(const float* qkv, float* out, int S, long n_seq, int d, int H, int Dh, float scale,
                              const float* nw, const float* nb, hipStream_t st, uint16_t* out16, size_t plane) {
    if (S != 256 || (Dh != 32 && Dh != 64)) return GENIE_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * 256 * Dh * 2 + (size_t)2 * Dh * 520 + 2048;  // K planes, V^T planes, key-split statistics
    // few (sequence, head) pairs: spread one pair over several workgroups -- by keys within the workgroup (qw = 0: 8 query
    // blocks per pair) for a handful, by query blocks of 2 or 4 waves for a few dozen
    static const int ksplit = study_env("GENIE_ATTN_KEYSPLIT", 1);
    // key split up to 256 pairs (32 clips of a one-frame pass): with the direct kernel it is ahead of the staged query-split
    // modes everywhere below the fused chip-filling path (+4 % generate at 6-8 clips, +0.5-1 % at 16-32: profiles/r03_keysplit_pairs.txt)
    static const int ks_pairs = study_env("GENIE_ATTN_KEYSPLIT_MAX_PAIRS", 256);
    // (qk-norm models keep the staged kernel, whose key split pays up to 32 pairs only)
    const int qw = n_seq * H <= (nw ? 32 : ks_pairs) ? (ksplit ? 0 : 1) : (n_seq * H <= 64 ? 2 : (n_seq * H <= 128 ? 4 : 8));
    dim3 grid((unsigned)n_seq, H, qw ? 8 / qw : 8);
    ProfScope prof(GENIE_KC_ATTN_SPATIAL, 4.0 * S * S * Dh * H * (double)n_seq, (double)n_seq * S * H * Dh * 16.0, st);
    static const int direct = study_env("GENIE_ATTN_KEYSPLIT_DIRECT", 1);
    if (qw == 0 && !nw && direct) {   // a handful of pairs, no qk-norm: fragments straight from the qkv rows, no LDS staging
        const size_t lds2 = (size_t)8 * 32 * Dh * 4 + 2048;
        if (Dh == 64) {
            (void)hipFuncSetAttribute((const void*)attn_spatial_keysplit_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            attn_spatial_keysplit_kernel<64><<<grid, 512, lds2, st>>>(qkv, out, d, scale, out16, plane);
        } else {
            (void)hipFuncSetAttribute((const void*)attn_spatial_keysplit_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            attn_spatial_keysplit_kernel<32><<<grid, 512, lds2, st>>>(qkv, out, d, scale, out16, plane);
        }
        GENIE_LAUNCH_CHECK("attn_spatial_keysplit");
        return GENIE_OK;
    }
#ifdef GENIE_STUDY   // per-phase s_memtime stamps of workgroup 0 (synchronises inside the launch: study builds only)
    static const int stamps = study_env("GENIE_SPLIT_STAMPS", 0);
    static unsigned long long* dstamps = nullptr;
    if (stamps && !dstamps) {
        (void)hipMalloc(&dstamps, 8 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_stamps), &dstamps, sizeof(dstamps));
    }
#endif
    if (Dh == 64) {
        (void)hipFuncSetAttribute((const void*)attn_spatial_split_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attn_spatial_split_kernel<64><<<grid, 512, lds, st>>>(qkv, out, d, scale, nw, nb, out16, plane, qw);
#ifdef GENIE_STUDY
        if (stamps && qw == 0) {
            static int shown = 0;
            if (shown++ % 512 == 100) {
                unsigned long long hb[8];
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(hb, dstamps, sizeof(hb), hipMemcpyDeviceToHost);
                fprintf(stderr, "split_stamps (ticks since start): staged %llu barrier %llu qsplit %llu qk+softmax %llu pv %llu partials %llu merge-start %llu end %llu\n",
                        hb[1] - hb[0], hb[2] - hb[0], hb[3] - hb[0], hb[4] - hb[0], hb[5] - hb[0], hb[6] - hb[0], hb[6] - hb[0], hb[7] - hb[0]);
            }
        }
#endif
    } else {
        (void)hipFuncSetAttribute((const void*)attn_spatial_split_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attn_spatial_split_kernel<32><<<grid, 512, lds, st>>>(qkv, out, d, scale, nw, nb, out16, plane, qw);
    }
    GENIE_LAUNCH_CHECK("attn_spatial_split");
    return GENIE_OK;
}

}  // namespace genie
