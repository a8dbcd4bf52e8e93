// Fused sub-blocks of the STBlock for the shipped geometry (magvit_n32_h8_d256: d = 256, 8 heads of 32, T = 16,
// hidden = 1024) in GENIE_PREC_BF16.  At d = 256 every Linear of the block is HBM-bound when run as its own launch
// (43 FLOP per byte for the out-projection), so these kernels keep a token's activations in REGISTERS from the
// first Linear of a sub-block to its residual update and stream only the weights:
//
//   temporal_fused_bf16_kernel   x += proj_t( causal-attention_T( qkv_t( bf16(x) ) ) )   (st_transformer.py:77-78,
//                                attention.py:36-61): replaces qkv GEMM + temporal attention + proj GEMM
//                                (403 + 403 MB of qkv written and re-read, 134 + 134 MB of attention output per layer at 64 clips).
//   mlp_fused_bf16_kernel<MODE>  x += fc2( gelu( fc1( LayerNorm(x) ) ) )   (st_transformer.py:81, 16-25); MODE 1 also writes the NEXT
//                                block's norm1(x) as bf16, MODE 2 instead runs that block's norm1 AND spatial qkv Linear
//                                (st_transformer.py:74, attention.py:37) and writes its attention operand planes.
//   spatial_attn_proj_bf16_kernel  x += proj_s( softmax(q k^T) v )  over the 256 positions of a frame, from those planes
//                                (st_transformer.py:73-74, attention.py:48-60); also the bf16 shadow of x for the temporal kernel.
//   => three launches per block.  Two rules shaped all three: (1) an accumulator layout is made the next
//   product's operand layout (weights packed in the matching K order), so nothing is re-laid between products; (2) every exposed row
//   access to HBM goes row-major through a wave-private LDS tile -- an instruction that touches 32 row PIECES costs like one that
//   moves 8 whole 128-byte lines -- and addresses used only in an epilogue are formed there (behind an empty asm), or the compiler
//   hoists them above the main loop and spills them.
//
// The temporal kernel ("lane = token"):
//   * every matrix instruction is v_mfma_f32_16x16x32_bf16 (16x16x16 for the 16-key P.V product).  A 16-token group is one
//     spatial position's 16 frames; a wave owns G = 2 groups.  A group's operand fragment (lane: token l & 15, k-group l >> 4,
//     8 consecutive k) is loaded ONCE from the bf16 shadow of x and serves as the B operand of the "swapped" products
//     D[feature][token] = W . X^T (q, k, out-projection) and as the A operand of the plain one D[token][feature] (v).
//   * results chain without leaving registers because the 16x16 accumulator layout (lane: column l & 15, rows 4 (l >> 4) + r)
//     of one product IS an operand layout of the next, up to a fixed permutation of the contraction index that the weight
//     packing absorbs:   K', Q' (lane = frame, 8 features) -> S^T = K' Q'^T (one MFMA per head) -> softmax over a lane's 4 keys
//     and the 4 lane groups -> P (lane = query, 4 keys) is the B operand and V (lane = feature, 4 frames) the A operand of
//     O^T = V^T P -> O^T (lane = frame, 8 features of the head) is the B operand of the out-projection's K-step for that head.
//   * the weights arrive as a STREAM of 1 KB fragments in consumption order (genie_pack_temporal_fused_bf16), 16 per 16 KB
//     stage, through a 4-slot LDS ring filled by buffer_load ... lds three stages ahead; a fragment read is one lane-linear
//     ds_read_b128, one workgroup barrier per stage, and each fragment feeds G matrix instructions.
//   * 4 waves per workgroup (128 tokens per block), 2 workgroups per CU (<= 256 registers per lane): one workgroup's
//     HBM phases (operand load, residual read-modify-write) run under the other's matrix work.
// Numerics = the bf16 contract of kernels_bf16.hip / oracle BF16_MFMA: Linear operands bf16 (q, k, v rounded to bf16 as the
// stored temporal qkv is), f32 accumulation, f32 softmax; P is split into two bf16 terms (hi + lo, 16 mantissa bits) so the
// P.V product is f32-class as in the unfused f32 attention kernel.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Ablation knobs exist in the -DGENIE_STUDY build only (GENIE_FUSED_ABL; results are wrong when set).  In the shipping build
// FS_ABL(bit) is a constant false: the kernels sit at the 256-register edge and an extra run-time branch is enough to make the
// register allocator spill.
#ifdef GENIE_STUDY
#define FS_ABL(bit) ((abl & (bit)) != 0)
#else
#define FS_ABL(bit) false
#endif

// Study build: GENIE_FUSED_STAMPS=1 makes wave 0 of every workgroup record s_memrealtime (100 MHz) at four points of each of its
// first 8 blocks (block start, operand rows in registers, main loop done, residual stores issued) plus its hardware id, and
// accumulate wave 0's main-loop cycles (s_memtime) by category: 0 wait (vmcnt + barrier), 1 LDS-DMA issue, 2 matrix section
// (fragment reads + MFMAs), 3 VALU section (GELU / softmax + packing).  Read back with genie_study_fused_stamps
// (tools/fused_timeline.py).  Layout: [512 workgroups][1 + 8 * 4] stamp words, then [512][4] cycle words.
#ifdef GENIE_STUDY
__device__ unsigned long long* g_fs_stamps = nullptr;
#define FS_STAMP(blk_i, k)                                                                                           \
    do {                                                                                                             \
        if (g_fs_stamps && threadIdx.x == 0 && (blk_i) < 8)                                                          \
            g_fs_stamps[(size_t)blockIdx.x * 33 + 1 + (blk_i) * 4 + (k)] = __builtin_amdgcn_s_memrealtime();          \
    } while (0)
#define FS_STAMP_ID()                                                                                                \
    do {                                                                                                             \
        if (g_fs_stamps && threadIdx.x == 0) {                                                                       \
            unsigned hw, xcc;                                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                      \
            g_fs_stamps[(size_t)blockIdx.x * 33] = ((unsigned long long)xcc << 32) | hw;                              \
        }                                                                                                            \
    } while (0)
#define FS_CYC_DECL unsigned long long fs_cyc[4] = {0, 0, 0, 0}, fs_t = __builtin_amdgcn_s_memtime()
#define FS_CYC(cat)                                                                      \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();                      \
        fs_cyc[cat] += n_ - fs_t;                                                        \
        fs_t = n_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#define FS_CYC_RESET() (fs_t = __builtin_amdgcn_s_memtime())
#define FS_CYC_DUMP()                                                                    \
    do {                                                                                 \
        if (g_fs_stamps && threadIdx.x == 0)                                             \
            for (int q_ = 0; q_ < 4; ++q_) g_fs_stamps[33 * 512 + (size_t)blockIdx.x * 4 + q_] = fs_cyc[q_]; \
    } while (0)
#else
#define FS_STAMP(blk_i, k) ((void)0)
#define FS_STAMP_ID() ((void)0)
#define FS_CYC_DECL ((void)0)
#define FS_CYC(cat) ((void)0)
#define FS_CYC_RESET() ((void)0)
#define FS_CYC_DUMP() ((void)0)
#endif

namespace {

constexpr int FS_STAGE = 16384;  // bytes of one stage = 16 fragments of 1 KB
constexpr int FS_NS = 4;         // ring slots
constexpr int FS_RING = FS_NS * FS_STAGE;

__device__ __forceinline__ void fs_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#if defined(GENIE_VAR_T_ABL) && (GENIE_VAR_T_ABL & 16)
#else
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#ifndef GENIE_VAR_M_ABL
#define GENIE_VAR_M_ABL 0   // timing variants of the region loop (results WRONG): 1 no barrier, 2 no LDS-DMA, 4 no vmcnt wait, 8 no GELU, 16 no fragment reads, 32 no MFMA
#endif
#ifndef GENIE_VAR_M_PF
#define GENIE_VAR_M_PF 2   // fragment prefetch depth of the mlp kernel's region loop (0: compiler-scheduled reads; 3 = 256 registers)
#endif
template <int N>
__device__ __forceinline__ void fs_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Fragment reads the compiler does not get to schedule: at ~250 registers it keeps ONE fragment buffer per loop and emits
// ds_read -> s_waitcnt lgkmcnt(0) -> MFMA, so every matrix instruction waits out a full LDS round trip.  fs_lds_rd issues a read
// the compiler knows nothing about, fs_lds_wait<N> is "all but my N youngest LDS reads have landed" tied to the fragment it guards
// (the "+v" makes the MFMA that consumes the fragment depend on the wait).  LDS reads return in order and the compiler's own waits
// can only over-wait; no scalar memory loads may be outstanding in such a section (they share lgkmcnt and return out of order).
template <int OFF>
__device__ __forceinline__ void fs_lds_rd(s16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void fs_lds_wait(s16x8& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}
__device__ __forceinline__ unsigned fs_lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// Lanes of one wave exchanging data through LDS: the compiler reasons per thread and may move a lane's read of another lane's
// slot above its own write -- nothing may cross this point, and every LDS operation issued so far has completed.
__device__ __forceinline__ void fs_wave_lds_fence() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#ifndef GENIE_VAR_T_ABL
#define GENIE_VAR_T_ABL 0   // variant builds only (results wrong): 4 no matrix instructions, 8 no fragment reads, 16 no barriers
#endif
__device__ __forceinline__ f32x4 mma32(const s16x8& a, const s16x8& b, const f32x4& c) {
    if constexpr (GENIE_VAR_T_ABL & 4) {
        f32x4 r = c;
        asm volatile("" : "+v"(r) : "v"(a), "v"(b));
        return r;
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16k(const s16x4& a, const s16x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// two accumulator tiles (features 0-15 | 16-31 of a head, lane = token) -> one 8-value bf16 operand fragment
__device__ __forceinline__ s16x8 pack8(const f32x4& lo, const f32x4& hi) {
    u32x4 p;
    p.x = f32x2_to_bf16x2(lo.x, lo.y);
    p.y = f32x2_to_bf16x2(lo.z, lo.w);
    p.z = f32x2_to_bf16x2(hi.x, hi.y);
    p.w = f32x2_to_bf16x2(hi.z, hi.w);
    return __builtin_bit_cast(s16x8, p);
}
__device__ __forceinline__ s16x4 pack4(const f32x4& v) {
    u32x2 p;
    p.x = f32x2_to_bf16x2(v.x, v.y);
    p.y = f32x2_to_bf16x2(v.z, v.w);
    return __builtin_bit_cast(s16x4, p);
}

}  // namespace

// Weight stream of one layer (bf16, 32 stages x 16 fragments x 64 lanes x 8 values = 256 K values = 512 KB):
//   stage n < 24: head h = n / 3, part = n % 3 (q, k, v); fragment f = 2 ks + ft (ks = 32-wide K-step, ft = 16-feature tile):
//       [lane l][e] = Wqkv[part * 256 + h * 32 + ft * 16 + (l & 15)][32 ks + 8 (l >> 4) + e]
//   stage 24 + h: out-projection K-step of head h; fragment f = 16-column tile ct:
//       [lane l][e] = Wproj[ct * 16 + (l & 15)][32 h + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)],  g = l >> 4
//   (the feature order of the second form is the order in which a lane holds its head's 8 attention outputs).
__global__ void pack_temporal_fused_kernel(const float* __restrict__ qkv_w, const float* __restrict__ proj_w,
                                           uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output value
    if (i >= 32 * 16 * 64 * 8) return;
    const int e = i & 7, l = (i >> 3) & 63, f = (i >> 9) & 15, n = i >> 13;
    const int g = l >> 4;
    float v;
    if (n < 24) {
        const int h = n / 3, part = n % 3, ks = f >> 1, ft = f & 1;
        v = qkv_w[(size_t)(part * 256 + h * 32 + ft * 16 + (l & 15)) * 256 + 32 * ks + 8 * g + e];
    } else {
        const int h = n - 24;
        const int feat = 32 * h + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
        v = proj_w[(size_t)(f * 16 + (l & 15)) * 256 + feat];
    }
    out[i] = f32_to_bf16(v);
}

// x16: (B, 16, S, 256) bf16 shadow of x (read);  x: (B, 16, S, 256) f32, updated in place.
// A block = 8 consecutive spatial positions of one clip x 16 frames; wave w owns positions 2 w, 2 w + 1.
// XF32: the operand fragments come from the f32 rows of x themselves (row-major loads through the wave tile, rounded here) -- no bf16
// shadow of x is read, so the spatial kernel in front need not write one.
template <bool QKV_BIAS, bool XF32>
__global__ __launch_bounds__(256, 2) void temporal_fused_bf16_kernel(const uint16_t* __restrict__ x16, float* __restrict__ x,
                                                                     const uint16_t* __restrict__ wstream,
                                                                     const float* __restrict__ qkv_b,
                                                                     const float* __restrict__ proj_b, int n_blocks, int S,
                                                                     float scale_log2e, int abl) {
    // abl (study build only, GENIE_FUSED_ABL; results are WRONG when set): 1 no in-loop LDS-DMA, 2 no residual read / store
    constexpr int D = 256, T = 16, NH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // biases live in LDS behind the ring (proj 256 floats, then qkv 768): plain global loads next to LDS-DMA in flight make
    // the compiler drain the whole ring at their first use, ds_reads do not
    float* sbias = reinterpret_cast<float*>(smem + FS_RING);
    for (int i = tid; i < 256 + (QKV_BIAS ? 768 : 0); i += 256) sbias[i] = i < 256 ? (proj_b ? proj_b[i] : 0.f) : qkv_b[i - 256];
    __syncthreads();

    FS_CYC_DECL;
    auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, 32 * FS_STAGE, 0x00020000);
    const unsigned voff = (unsigned)lane * 16;
    int n_issue = 0;  // stages issued so far (monotonic; stream position = n & 31, slot = n & 3)
    auto issue_stage = [&]() {
        const int soff = (n_issue & 31) * FS_STAGE + wid * 4096;
        unsigned char* dst = smem + (n_issue & (FS_NS - 1)) * FS_STAGE + wid * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff,
                                                     soff + j * 1024, 0, 0);
        ++n_issue;
    };
    int n_use = 0;  // stages consumed so far
    // stage n_use has landed for every wave, and the slot of stage n_use - 1 is free: refill it three stages ahead
    auto acquire = [&]() -> const unsigned char* {
        FS_CYC(3);
        fs_wait_vm<8>();
        fs_barrier();
        FS_CYC(0);
        if (!FS_ABL(1)) issue_stage();
        FS_CYC(1);
        const unsigned char* p = smem + (n_use & (FS_NS - 1)) * FS_STAGE + lane * 16;
        ++n_use;
        return p;
    };
    auto frag = [&](const unsigned char* stage, int f) {
        if constexpr (GENIE_VAR_T_ABL & 8) {   // (variant: no LDS fragment reads -- a lane-dependent constant instead)
            const short v = (short)(lane + f);
            return s16x8{v, v, v, v, v, v, v, v};
        } else {
            return *reinterpret_cast<const s16x8*>(stage + f * 1024);
        }
    };
#ifdef GENIE_VAR_T_PAIR
    // variant: 32 KB stages (two 16-fragment parts per barrier), two slots: half the barriers, prefetch distance one stage
    int s_issue = 0;   // 32 KB stages requested so far (stream position = s & 15, slot = s & 1)
    auto issue_pair = [&]() {
        const int soff = (s_issue & 15) * 2 * FS_STAGE + wid * 8192;
        unsigned char* dst = smem + (s_issue & 1) * 2 * FS_STAGE + wid * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff,
                                                     soff + j * 1024, 0, 0);
        ++s_issue;
    };
    int s_use = 0;
    const unsigned char* pair_base = nullptr;
    // part p of the block (p = 0..31, a compile-time constant at every call site): even parts open a new 32 KB stage
    auto acquire_part = [&](int p) -> const unsigned char* {
        if (!(p & 1)) {
            fs_wait_vm<0>();
            fs_barrier();
            if (!FS_ABL(1)) issue_pair();
            pair_base = smem + (s_use & 1) * 2 * FS_STAGE + lane * 16;
            ++s_use;
            return pair_base;
        }
        return pair_base + FS_STAGE;
    };
    issue_pair();
#else
    auto acquire_part = [&](int) -> const unsigned char* { return acquire(); };

    issue_stage();
    issue_stage();
    issue_stage();
#endif

    const int bps = S / 8;  // blocks per clip
    FS_STAMP_ID();
    [[maybe_unused]] int blk_i = 0;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x, ++blk_i) {
        FS_STAMP(blk_i, 0);
        const int b = blk / bps, s0 = (blk - b * bps) * 8 + 2 * wid;
        // this lane's token of group grp: frame r, position s0 + grp
        const size_t row0 = ((size_t)b * T + r) * S + s0;
        s16x8 xf[2][8];
        if constexpr (XF32) {
            float* tile = reinterpret_cast<float*>(smem + FS_RING + 4096 + wid * 2304);
            int tt = lane >> 3, cc = (lane & 7) * 4;
            asm volatile("" : "+v"(tt), "+v"(cc));
            const float* xb = x + (((size_t)b * T + tt) * S + s0) * D + cc;
            const size_t half_stride = (size_t)8 * S * D;
            f32x4 raw[16][2];
            auto load_slab = [&](int i) {   // slab i = (grp = i >> 3, K-step i & 7: columns 32 (i & 7) ..)
                const float* p = xb + (i >> 3) * D + 32 * (i & 7);
                raw[i][0] = *reinterpret_cast<const f32x4*>(p);
                raw[i][1] = *reinterpret_cast<const f32x4*>(p + half_stride);
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) load_slab(i);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 4 < 16) load_slab(i + 4);
                *reinterpret_cast<f32x4*>(tile + tt * 36 + cc) = raw[i][0];
                *reinterpret_cast<f32x4*>(tile + (8 + tt) * 36 + cc) = raw[i][1];
                fs_wave_lds_fence();
                xf[i >> 3][i & 7] = pack8(*reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g), *reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * g + 4));
                fs_wave_lds_fence();
            }
            fs_wait_vm<0>();
        } else {
#ifdef GENIE_VAR_T_XTILE
        {   // operand rows as whole 128-byte lines (8 tokens x 64 columns per request), re-laid through the wave's tile
            unsigned char* tile = smem + FS_RING + 4096 + wid * 2304;
            const int tt = lane >> 3, c8 = (lane & 7) * 8;
            const uint16_t* xb = x16 + (((size_t)b * T + tt) * S + s0) * D + c8;
            const size_t half_stride = (size_t)8 * S * D;
            s16x8 raw[2][4][2];
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        raw[grp][c][hf] = *reinterpret_cast<const s16x8*>(xb + grp * D + 64 * c + hf * half_stride);
            fs_wait_vm<0>();
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) *reinterpret_cast<s16x8*>(tile + (8 * hf + tt) * 144 + c8 * 2) = raw[grp][c][hf];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) xf[grp][2 * c + k2] = *reinterpret_cast<const s16x8*>(tile + r * 144 + (32 * k2 + 8 * g) * 2);
                    __builtin_amdgcn_wave_barrier();
                }
        }
#else
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                xf[grp][ks] = *reinterpret_cast<const s16x8*>(x16 + (row0 + grp) * D + 32 * ks + 8 * g);
        fs_wait_vm<0>();
#endif
        }
        FS_STAMP(blk_i, 1);
        FS_CYC_RESET();

        s16x8 oall[2][NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            s16x8 qb[2], kb[2];
            s16x4 vb[2][2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const unsigned char* stg = acquire_part(3 * h + part);
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
                            acc[grp][ft] = part < 2 ? mma32(wf, xf[grp][ks], acc[grp][ft]) : mma32(xf[grp][ks], wf, acc[grp][ft]);
                    }
                FS_CYC(2);
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) {
                    if (part == 0) qb[grp] = pack8(acc[grp][0], acc[grp][1]);
                    else if (part == 1) kb[grp] = pack8(acc[grp][0], acc[grp][1]);
                    else { vb[grp][0] = pack4(acc[grp][0]); vb[grp][1] = pack4(acc[grp][1]); }
                }
            }
            // causal attention over the 16 frames of each group (attention.py:48-58): lane = query frame r, keys 4 g + e
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                f32x4 st = mma32(kb[grp], qb[grp], f32x4{0.f, 0.f, 0.f, 0.f});
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * g + e > r) st[e] = -INFINITY;
                    mx = fmaxf(mx, st[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mxs = mx * scale_log2e;
                float sum = 0.f;
                f32x4 p, plo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = __builtin_amdgcn_exp2f(fmaf(st[e], scale_log2e, -mxs));
                    sum += p[e];
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = __builtin_amdgcn_rcpf(sum);
                const s16x4 phi = pack4(p);
#pragma unroll
                for (int e = 0; e < 4; ++e) plo[e] = p[e] - bf16_to_f32((uint16_t)phi[e]);
                const s16x4 plo16 = pack4(plo);
                f32x4 o[2];
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    o[ft] = mma16k(vb[grp][ft], phi, f32x4{0.f, 0.f, 0.f, 0.f});
                    o[ft] = mma16k(vb[grp][ft], plo16, o[ft]);
                    o[ft] *= inv;
                }
                oall[grp][h] = pack8(o[0], o[1]);
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
            const unsigned char* stg = acquire_part(24 + h);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const s16x8 wf = frag(stg, ct);
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) out[grp][ct] = mma32(wf, oall[grp][h], out[grp][ct]);
            }
            FS_CYC(2);
        }
        // residual update in place: x[token][16 ct + 4 g ..] += out (bias is already in the accumulators).  Four rounds of 8
        // 16-byte pieces; round k + 1's reads are requested before round k's stores (vmcnt retires in order: a wait for reads
        // issued behind a store would also wait for the store).
        FS_STAMP(blk_i, 2);
        if (FS_ABL(2)) continue;
#ifndef GENIE_VAR_T_NO_TILE
        // Through a wave-private LDS tile (16 tokens x 32 columns, rows padded to 144 bytes): the accumulators hold 64-byte pieces of
        // 16 rows per instruction; row-major, an instruction moves 8 whole 128-byte lines (requests, not bytes, are what this
        // kernel's memory side costs).  lane -> token 8 half + (lane >> 3), columns 4 (lane & 7) .. + 3 of the 32-column slab.
        {
            float* tile = reinterpret_cast<float*>(smem + FS_RING + 4096 + wid * 2304);
            const int tt = lane >> 3, cc = (lane & 7) * 4;
            // row of (group grp, token 8 half + tt): ((b T + 8 half + tt) S + s0 + grp) D
            float* xb = x + (((size_t)b * T + tt) * S + s0) * D + cc;
            const size_t half_stride = (size_t)8 * S * D;
#ifndef GENIE_VAR_T_EPF
#define GENIE_VAR_T_EPF 1      // residual slabs requested ahead of their use
#endif
            constexpr int EPF = GENIE_VAR_T_EPF;
            f32x4 rs[EPF + 1][2];
            auto load_slab = [&](int i, f32x4* dst) {   // slab i = (grp = i >> 3, columns 32 (i & 7) ..)
                const float* p = xb + (i >> 3) * D + 32 * (i & 7);
                dst[0] = *reinterpret_cast<const f32x4*>(p);
                dst[1] = *reinterpret_cast<const f32x4*>(p + half_stride);
            };
#pragma unroll
            for (int i = 0; i < EPF; ++i) load_slab(i, rs[i % (EPF + 1)]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int grp = i >> 3, sl = i & 7;
                if (i + EPF < 16) load_slab(i + EPF, rs[(i + EPF) % (EPF + 1)]);
                *reinterpret_cast<f32x4*>(tile + r * 36 + 4 * g) = out[grp][2 * sl];
                *reinterpret_cast<f32x4*>(tile + r * 36 + 16 + 4 * g) = out[grp][2 * sl + 1];
                fs_wave_lds_fence();
                float* p = xb + grp * D + 32 * sl;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (8 * hf + tt) * 36 + cc);
                    *reinterpret_cast<f32x4*>(p + hf * half_stride) = rs[i % (EPF + 1)][hf] + v;
                }
                fs_wave_lds_fence();
            }
        }
#else
        float* xr0 = x + row0 * D + 4 * g;
        f32x4 res[2][8];
        auto load_round = [&](int k, f32x4* dst) {
            float* xr = xr0 + (k >> 1) * D + (k & 1) * 128;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[c] = *reinterpret_cast<const f32x4*>(xr + c * 16);
        };
        load_round(0, res[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k + 1 < 4) load_round(k + 1, res[(k + 1) & 1]);
            float* xr = xr0 + (k >> 1) * D + (k & 1) * 128;
#pragma unroll
            for (int c = 0; c < 8; ++c) res[k & 1][c] += out[k >> 1][(k & 1) * 8 + c];
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(xr + c * 16) = res[k & 1][c];
        }
#endif
        FS_STAMP(blk_i, 3);
    }
    FS_CYC_DUMP();
    fs_wait_vm<0>();  // the ring's run-ahead stages must not outlive the workgroup's LDS allocation
}

#ifdef GENIE_STUDY
static unsigned long long* g_fs_stamps_host_ptr = nullptr;
static void fs_stamps_prepare() {
    static const int on = study_env("GENIE_FUSED_STAMPS", 0);
    if (!on) return;
    if (!g_fs_stamps_host_ptr) {
        (void)hipMalloc(&g_fs_stamps_host_ptr, sizeof(unsigned long long) * 37 * 512);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fs_stamps), &g_fs_stamps_host_ptr, sizeof(g_fs_stamps_host_ptr));
    }
    (void)hipMemset(g_fs_stamps_host_ptr, 0, sizeof(unsigned long long) * 37 * 512);
}
extern "C" int genie_study_fused_stamps(unsigned long long* out_host, int n_words) {   // stamps of the LAST fused launch
    if (!g_fs_stamps_host_ptr || !out_host) return -1;
    (void)hipDeviceSynchronize();
    const int n = n_words < 37 * 512 ? n_words : 37 * 512;
    return hipMemcpy(out_host, g_fs_stamps_host_ptr, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) == hipSuccess ? n : -1;
}
#else
static void fs_stamps_prepare() {}
#endif

int launch_pack_temporal_fused(const float* qkv_w, const float* proj_w, uint16_t* out, hipStream_t st) {
    pack_temporal_fused_kernel<<<(32 * 16 * 64 * 8) / 256, 256, 0, st>>>(qkv_w, proj_w, out);
    GENIE_LAUNCH_CHECK("pack_temporal_fused");
    return GENIE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// mlp_fused_bf16_kernel:  x += fc2( gelu( fc1( LayerNorm(x) ) ) )   (st_transformer.py:16-25, 81) for d = 256, hidden = 1024.
// Replaces LayerNorm + fc1 GEMM + fc2 GEMM: the normalised operand and the 1024-wide hidden never leave the registers
// (134 MB of LayerNorm output and 2 x 537 MB of hidden per layer at 64 clips).
//   * v_mfma_f32_32x32x16_bf16, lane = token (l & 31), k-group l >> 5.  A wave owns 32 consecutive rows of x: it reads them as
//     f32 fragments (8 consecutive channels per K-step), normalises in registers (row statistics = in-lane sums + one
//     cross-half exchange) and keeps the 16 bf16 K-step fragments for the whole block.
//   * the hidden is produced in chunks of 32 units by the swapped product  D[unit][token] = W1 . X^T  (bias in the accumulator's
//     initial value); after GELU the accumulator's registers 0-7 / 8-15 ARE the two K-step fragments (lane = token, 8 units) of
//     the swapped second product  D[column][token] += W2 . H^T, whose 8 accumulator tiles hold the token's 256 outputs.
//   * weight stream (genie_pack_mlp_fused_bf16): 64 stages of 16 fragments, [A_0 | B_0 | A_1 | B_1 | ...] (A_c = fc1 rows of
//     chunk c over the 16 K-steps, B_c = fc2 columns over chunk c's two K-steps).  The main loop runs in 33 REGIONS: region j
//     holds the first product of chunk j next to the second product of chunk j - 1 -- independent matrix work to interleave
//     (the fc1 chain runs on ONE accumulator) and to cover the GELU of the chunk before.  The ring is two 32 KB region slots
//     [A half | B half]: region j + 1 is requested right after region j's barrier (the slot of region j - 1 is free then) and
//     waited for with vmcnt(0) one region later -- one barrier and one uniform protocol step per region.
//   * 4 waves / 128 rows per block, 2 workgroups per CU.
// Numerics = the bf16 contract (oracle BF16_MFMA): LayerNorm f32 -> bf16 operand, f32 accumulate + bias, erf-GELU in f32
// (gelu_erf_fast, |error| 1.5e-7) -> bf16 operand, f32 accumulate + bias + f32 residual.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void pack_mlp_fused_kernel(const float* __restrict__ fc1_w, const float* __restrict__ fc2_w, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per output value
    if (i >= 64 * 16 * 64 * 8) return;
    const int e = i & 7, l = (i >> 3) & 63, f = (i >> 9) & 15, n = i >> 13;
    const int h = l >> 5, rr = l & 31;
    const int c = n >> 1;   // stage 2c = A_c, stage 2c + 1 = B_c
    float v;
    if (!(n & 1)) {
        // fragment f = 2 ct + kk contracts the features a lane holds in registers 8 kk .. 8 kk + 7 of accumulator tile ct
        // (the LayerNorm output is packed straight from accumulator-layout registers, as the GELU output is for fc2)
        v = fc1_w[(size_t)(32 * c + rr) * 256 + 32 * (f >> 1) + 16 * (f & 1) + (e & 3) + 8 * (e >> 2) + 4 * h];
    } else {
        const int kk = f >> 3, ct = f & 7;                                                 // fragment f = 8 kk + ct
        v = fc2_w[(size_t)(32 * ct + rr) * 1024 + 32 * c + 16 * kk + (e & 3) + 8 * (e >> 2) + 4 * h];
    }
    out[i] = f32_to_bf16(v);
}

int launch_pack_mlp_fused(const float* fc1_w, const float* fc2_w, uint16_t* out, hipStream_t st) {
    pack_mlp_fused_kernel<<<(64 * 16 * 64 * 8) / 256, 256, 0, st>>>(fc1_w, fc2_w, out);
    GENIE_LAUNCH_CHECK("pack_mlp_fused");
    return GENIE_OK;
}

namespace {
__device__ __forceinline__ f32x16 mma32x32(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
constexpr int ML_OFF_LNG = FS_RING, ML_OFF_LNB = FS_RING + 1024, ML_OFF_B1 = FS_RING + 2048, ML_OFF_B2 = FS_RING + 6144;
constexpr int ML_OFF_NXG = FS_RING + 7168, ML_OFF_NXB = FS_RING + 8192;
constexpr int ML_LDS = FS_RING + 9216;
}  // namespace

// x16_out (optional) receives a bf16 copy of the updated rows: the plain shadow of x (LNOUT = false; the readout's operand after
// the last block), or -- LNOUT = true -- LayerNorm(x; nx_g, nx_b), i.e. the NEXT block's norm1 output, the operand of its
// spatial qkv Linear (st_transformer.py:73): the row is complete in this lane pair's registers, so that LayerNorm launch goes.
// MODE 2: instead of that bf16 row, the next block's spatial q | k | v^T operand planes (its LayerNorm AND its qkv Linear happen here:
// 24 more 16 KB weight stages per block through the region ring's idle halves; the planes leave through a wave tile as whole lines).
template <int MODE>
__global__ __launch_bounds__(256, 2) void mlp_fused_bf16_kernel(float* __restrict__ x, const uint16_t* __restrict__ wstream,
                                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                                const float* __restrict__ b1, const float* __restrict__ b2,
                                                                uint16_t* __restrict__ x16_out, const float* __restrict__ nx_g,
                                                                const float* __restrict__ nx_b, int n_blocks, float eps, int abl,
                                                                const uint16_t* __restrict__ qstream, long P, float qscale) {
    constexpr bool LNOUT = MODE >= 1, QKV = MODE == 2;   // (MODE 2: x16_out = the plane buffer)
    // abl (study build only, GENIE_FUSED_ABL; results are WRONG when set): 1 no in-loop LDS-DMA, 2 no residual read / store,
    // 4 no GELU, 8 no LayerNorm prologue loads
    constexpr int D = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    {   // parameter tables in LDS (ds_reads do not disturb the LDS-DMA stream's vmcnt bookkeeping)
        float* t = reinterpret_cast<float*>(smem + FS_RING);
        for (int i = tid; i < 1792; i += 256)
            t[i] = i < 256 ? ln_g[i] : i < 512 ? ln_b[i - 256] : i < 1536 ? (b1 ? b1[i - 512] : 0.f) : (b2 ? b2[i - 1536] : 0.f);
        if constexpr (LNOUT) { t[1792 + tid] = nx_g[tid]; t[2048 + tid] = nx_b[tid]; }
        __syncthreads();
    }
    const float* s_g = reinterpret_cast<const float*>(smem + ML_OFF_LNG);
    const float* s_b = reinterpret_cast<const float*>(smem + ML_OFF_LNB);
    const float* s_b1 = reinterpret_cast<const float*>(smem + ML_OFF_B1);
    const float* s_b2 = reinterpret_cast<const float*>(smem + ML_OFF_B2);

    auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, 64 * FS_STAGE, 0x00020000);
    const unsigned voff = (unsigned)lane * 16;
    // request the halves of region j (A_j if j < 32, B_{j-1} if j >= 1) into region slot j & 1; wave w moves pieces 4w .. 4w + 3
    auto issue_region = [&](int j) {
        if (FS_ABL(1) && j > 1) return;
        unsigned char* dst = smem + (j & 1) * 2 * FS_STAGE + wid * 4096;
        if (j < 32) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, voff,
                                                         (2 * j) * FS_STAGE + wid * 4096 + q * 1024, 0, 0);
        }
        if (j >= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + FS_STAGE + q * 1024), 16,
                                                         voff, (2 * j - 1) * FS_STAGE + wid * 4096 + q * 1024, 0, 0);
        }
    };
    auto frag = [&](const unsigned char* stage, int f) { return *reinterpret_cast<const s16x8*>(stage + f * 1024); };
    const unsigned char* lbase = smem + lane * 16;

    issue_region(0);

    FS_STAMP_ID();
    FS_CYC_DECL;
    [[maybe_unused]] int blk_i = 0;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x, ++blk_i) {
        FS_STAMP(blk_i, 0);
        // ---- the wave's 32 rows: row-major from HBM (8 whole 128-byte segments per request), through the wave's tile (ring slot 1: free
        // from this block's last region barrier to the next block's first one) into ACCUMULATOR layout.  They stay there: `out` starts as
        // the residual row and fc2 accumulates on top, so the epilogue reads nothing (the rows were read twice before: 268 MB per launch).
        // LayerNorm from the same registers; its output is packed per accumulator half-tile -- fc1's fragments are packed in that K order.
        f32x16 out[8];
        s16x8 xf[16];
        {
            float* tile = reinterpret_cast<float*>(smem + 2 * FS_STAGE + wid * 8192);
            int rr = lane >> 3, cc = (lane & 7) * 4;
            asm volatile("" : "+v"(rr), "+v"(cc));   // (addresses derived from them are formed here, per block -- not hoisted and spilled)
            const float* xw = x + ((size_t)blk * 128 + wid * 32 + rr) * D + cc;
            f32x4 raw[8][4];
            auto load_slab = [&](int ct) {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) raw[ct][i2] = *reinterpret_cast<const f32x4*>(xw + (size_t)(8 * i2) * D + 32 * ct);
            };
#ifndef GENIE_VAR_M_TILE_PF
#define GENIE_VAR_M_TILE_PF 2   // slabs requested ahead (4 measured the same)
#endif
#pragma unroll
            for (int ct = 0; ct < GENIE_VAR_M_TILE_PF; ++ct) load_slab(ct);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                if (ct + GENIE_VAR_M_TILE_PF < 8) load_slab(ct + GENIE_VAR_M_TILE_PF);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) *reinterpret_cast<f32x4*>(tile + (rr + 8 * i2) * 36 + cc) = raw[ct][i2];
                fs_wave_lds_fence();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(tile + r * 36 + 8 * j + 4 * h);
                    out[ct][4 * j] = t4.x; out[ct][4 * j + 1] = t4.y; out[ct][4 * j + 2] = t4.z; out[ct][4 * j + 3] = t4.w;
                }
                fs_wave_lds_fence();
            }
            float sum = 0.f;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int i = 0; i < 16; i += 4) sum += (out[ct][i] + out[ct][i + 1]) + (out[ct][i + 2] + out[ct][i + 3]);
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / D);
            float sq = 0.f;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const float d0 = out[ct][i] - mean, d1 = out[ct][i + 1] - mean, d2 = out[ct][i + 2] - mean, d3 = out[ct][i + 3] - mean;
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            sq += __shfl_xor(sq, 32);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + eps);
            const float* s_g4 = s_g + 4 * h;
            const float* s_b4 = s_b + 4 * h;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x4 y[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int j = 2 * kk + q;
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(s_g4 + 32 * ct + 8 * j), bv = *reinterpret_cast<const f32x4*>(s_b4 + 32 * ct + 8 * j);
                        y[q] = (f32x4{out[ct][4 * j], out[ct][4 * j + 1], out[ct][4 * j + 2], out[ct][4 * j + 3]} - mean) * rstd * gv + bv;
                    }
                    xf[2 * ct + kk] = pack8(y[0], y[1]);
                }
        }
        fs_wait_vm<0>();
        FS_STAMP(blk_i, 1);
        auto bias1 = [&](int c) {
            f32x16 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_b1 + 32 * c + 8 * j + 4 * h);
                a[4 * j] = bv.x; a[4 * j + 1] = bv.y; a[4 * j + 2] = bv.z; a[4 * j + 3] = bv.w;
            }
            return a;
        };
        auto gelu_pack = [&](const f32x16& a, s16x8& h0, s16x8& h1) {
            float gz[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
#ifdef GENIE_VAR_M_GELU_AS   // (variant: the 1.5e-7 GELU of the GEMM epilogues)
                gz[i] = FS_ABL(4) ? a[i] : gelu_erf_fast(a[i]); gz[i + 1] = FS_ABL(4) ? a[i + 1] : gelu_erf_fast(a[i + 1]);
#else
                const genie_f2 g2 = (FS_ABL(4) || (GENIE_VAR_M_ABL & 8)) ? genie_f2{a[i], a[i + 1]} : gelu_erf_poly2(genie_f2{a[i], a[i + 1]});
                gz[i] = g2[0]; gz[i + 1] = g2[1];
#endif
            }
            h0 = pack8(f32x4{gz[0], gz[1], gz[2], gz[3]}, f32x4{gz[4], gz[5], gz[6], gz[7]});
            h1 = pack8(f32x4{gz[8], gz[9], gz[10], gz[11]}, f32x4{gz[12], gz[13], gz[14], gz[15]});
        };

        // region 0: fc1 of chunk 0
        FS_CYC_RESET();
        fs_barrier();
        FS_CYC(0);
        issue_region(1);
        FS_CYC(1);
        f32x16 acc1 = bias1(0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc1 = mma32x32(frag(lbase, ks), xf[ks], acc1);
        FS_CYC(2);
        s16x8 hk0, hk1;
        gelu_pack(acc1, hk0, hk1);
        FS_CYC(3);
        // regions 1..31: fc1 of chunk j next to fc2 of chunk j - 1
        for (int j = 1; j < 32; ++j) {
            if constexpr (!(GENIE_VAR_M_ABL & 4)) fs_wait_vm<0>();
            if constexpr (!(GENIE_VAR_M_ABL & 1)) fs_barrier();
            FS_CYC(0);
            if constexpr (!(GENIE_VAR_M_ABL & 2)) issue_region(j + 1);
            FS_CYC(1);
            const unsigned char* sa = lbase + (j & 1) * 2 * FS_STAGE;
            acc1 = bias1(j);
#if GENIE_VAR_M_PF > 0
            {   // the region's 32 fragments [A_0 B_0 A_1 B_1 ...] through GENIE_VAR_M_PF rotating buffers
                constexpr int PF = GENIE_VAR_M_PF;
                const unsigned la = fs_lds_addr(sa);
                s16x8 fb[PF];
#define ML_OFF(n) ((((n) & 1) ? FS_STAGE : 0) + ((n) >> 1) * 1024)
#define ML_STEP(n)                                                                                                       \
    {                                                                                                                    \
        if constexpr (!(GENIE_VAR_M_ABL & 16)) fs_lds_wait<((31 - (n)) < (PF - 1) ? (31 - (n)) : (PF - 1))>(fb[(n) % PF]); \
        if constexpr ((GENIE_VAR_M_ABL & 32) != 0) { asm volatile("" : "+v"(acc1) : "v"(fb[(n) % PF])); }              \
        else if constexpr (((n) & 1) != 0) out[((n) >> 1) & 7] = mma32x32(fb[(n) % PF], ((n) >> 4) ? hk1 : hk0, out[((n) >> 1) & 7]); \
        else acc1 = mma32x32(fb[(n) % PF], xf[(n) >> 1], acc1);                                                         \
        if constexpr ((n) + PF < 32 && !(GENIE_VAR_M_ABL & 16)) fs_lds_rd<ML_OFF(((n) + PF) & 31)>(fb[(n) % PF], la);   \
    }
                if constexpr ((GENIE_VAR_M_ABL & 16) != 0) { fb[0] = xf[0]; fb[1 % PF] = xf[1]; }   /* (timing variant: no fragment reads) */
                else fs_lds_rd<ML_OFF(0)>(fb[0], la);
                if constexpr (PF > 1 && !(GENIE_VAR_M_ABL & 16)) fs_lds_rd<ML_OFF(1)>(fb[1 % PF], la);
                if constexpr (PF > 2) fs_lds_rd<ML_OFF(2)>(fb[2 % PF], la);
                if constexpr (PF > 3) fs_lds_rd<ML_OFF(3)>(fb[3 % PF], la);
                ML_STEP(0) ML_STEP(1) ML_STEP(2) ML_STEP(3) ML_STEP(4) ML_STEP(5) ML_STEP(6) ML_STEP(7)
                ML_STEP(8) ML_STEP(9) ML_STEP(10) ML_STEP(11) ML_STEP(12) ML_STEP(13) ML_STEP(14) ML_STEP(15)
                ML_STEP(16) ML_STEP(17) ML_STEP(18) ML_STEP(19) ML_STEP(20) ML_STEP(21) ML_STEP(22) ML_STEP(23)
                ML_STEP(24) ML_STEP(25) ML_STEP(26) ML_STEP(27) ML_STEP(28) ML_STEP(29) ML_STEP(30) ML_STEP(31)
#undef ML_STEP
#undef ML_OFF
            }
#else
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc1 = mma32x32(frag(sa, i), xf[i], acc1);
                out[i & 7] = mma32x32(frag(sa + FS_STAGE, i), (i >> 3) ? hk1 : hk0, out[i & 7]);
            }
#endif
            FS_CYC(2);
            gelu_pack(acc1, hk0, hk1);
            FS_CYC(3);
        }
        // region 32: fc2 of chunk 31; the next block's region 0 is requested here (this block's slot 1 is free after the barrier)
        fs_wait_vm<0>();
        fs_barrier();
        issue_region(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i & 7] = mma32x32(frag(lbase + FS_STAGE, i), (i >> 3) ? hk1 : hk0, out[i & 7]);
        [[maybe_unused]] auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qstream, 0, 24 * FS_STAGE, 0x00020000);
        // qkv stage n (part n / 8 of q | k | v, head n % 8; 16 fragments) into slot 0's second half (even n) or slot 1's first half (odd n)
        [[maybe_unused]] auto issue_q = [&](int n) {
            unsigned char* dst = smem + ((n & 1) ? 2 * FS_STAGE : FS_STAGE) + wid * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, voff,
                                                         n * FS_STAGE + wid * 4096 + q * 1024, 0, 0);
        };
        if constexpr (QKV) {
            fs_barrier();      // every wave is done with region 32's fragments (slot 0, second half)
            issue_q(0);
        }
        FS_STAMP(blk_i, 2);
        if (FS_ABL(2)) { asm volatile("" :: "v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7])); continue; }
        {
            // `out` holds x + fc2(...); + bias = the updated row.  Out row-major through the wave's tile (ring slot 1): 8 whole row
            // segments per request instead of 32 pieces of 32 (16) bytes; LNOUT: also the row normalised with the next block's norm1.
            float* tile = reinterpret_cast<float*>(smem + 2 * FS_STAGE + wid * 8192);
            float* stat = tile + 32 * 36;      // (mean, rstd) of the wave's 32 tokens
            int rr = lane >> 3, cc = (lane & 7) * 4;
            size_t roff = ((size_t)blk * 128 + wid * 32 + rr) * D + cc;
            asm volatile("" : "+v"(roff));     // the row addresses are formed HERE: hoisted above the region loop they are 16 spilled registers
            float* xw = x + roff;
            uint16_t* xw16 = x16_out + roff;
            [[maybe_unused]] int w16 = x16_out != nullptr;
            asm volatile("" : "+v"(w16));   // a per-lane predicate: the store loop is masked, not duplicated (the duplicate spilled)
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs_[4] = {1.f, 1.f, 1.f, 1.f};
            [[maybe_unused]] s16x8 yk[16];
            if constexpr (LNOUT) {   // LayerNorm statistics of the updated row (two-pass, as layer_norm_fast_kernel)
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(s_b2 + 32 * ct + 8 * j + 4 * h);
                        out[ct][4 * j] += bv.x; out[ct][4 * j + 1] += bv.y; out[ct][4 * j + 2] += bv.z; out[ct][4 * j + 3] += bv.w;
                    }
                float sum = 0.f;
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; i += 4) sum += (out[ct][i] + out[ct][i + 1]) + (out[ct][i + 2] + out[ct][i + 3]);
                sum += __shfl_xor(sum, 32);
                const float mean = sum * (1.0f / D);
                float sq = 0.f;
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        const float d0 = out[ct][i] - mean, d1 = out[ct][i + 1] - mean, d2 = out[ct][i + 2] - mean, d3 = out[ct][i + 3] - mean;
                        sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                sq += __shfl_xor(sq, 32);
                const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + eps);
                if constexpr (QKV) {   // the normalised row as K-step fragments, straight from accumulator layout (as the prologue's)
                    const float* s_ng4 = reinterpret_cast<const float*>(smem + ML_OFF_NXG) + 4 * h;
                    const float* s_nb4 = reinterpret_cast<const float*>(smem + ML_OFF_NXB) + 4 * h;
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            f32x4 y[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int j = 2 * kk + q;
                                const f32x4 gv = *reinterpret_cast<const f32x4*>(s_ng4 + 32 * ct + 8 * j), bv = *reinterpret_cast<const f32x4*>(s_nb4 + 32 * ct + 8 * j);
                                y[q] = (f32x4{out[ct][4 * j], out[ct][4 * j + 1], out[ct][4 * j + 2], out[ct][4 * j + 3]} - mean) * rstd * gv + bv;
                            }
                            yk[2 * ct + kk] = pack8(y[0], y[1]);
                        }
                } else {
                    {
                        int rl = r;
                        asm volatile("" : "+v"(rl));   // (its address is formed here, not kept in a spilled register since kernel entry)
                        if (h == 0) { stat[2 * rl] = mean; stat[2 * rl + 1] = rstd; }
                    }
                    fs_wave_lds_fence();
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) { mu[i2] = stat[2 * (rr + 8 * i2)]; rs_[i2] = stat[2 * (rr + 8 * i2) + 1]; }
                }
            }
            const float* s_ng = reinterpret_cast<const float*>(smem + ML_OFF_NXG) + cc;
            const float* s_nb = reinterpret_cast<const float*>(smem + ML_OFF_NXB) + cc;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 o4 = f32x4{out[ct][4 * j], out[ct][4 * j + 1], out[ct][4 * j + 2], out[ct][4 * j + 3]};
                    if constexpr (!LNOUT) o4 += *reinterpret_cast<const f32x4*>(s_b2 + 32 * ct + 8 * j + 4 * h);   // (LNOUT: added above, for the statistics)
                    *reinterpret_cast<f32x4*>(tile + r * 36 + 8 * j + 4 * h) = o4;
                }
                fs_wave_lds_fence();
                f32x4 gv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (LNOUT && !QKV) { gv = *reinterpret_cast<const f32x4*>(s_ng + 32 * ct); bv = *reinterpret_cast<const f32x4*>(s_nb + 32 * ct); }
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * i2) * 36 + cc);
                    *reinterpret_cast<f32x4*>(xw + (size_t)(8 * i2) * D + 32 * ct) = vv;
                    if constexpr (QKV) {
                    } else if constexpr (LNOUT) {
                        const f32x4 y = (vv - mu[i2]) * rs_[i2] * gv + bv;
                        *reinterpret_cast<s16x4*>(xw16 + (size_t)(8 * i2) * D + 32 * ct) = pack4(y);
                    } else {
                        if (w16) *reinterpret_cast<s16x4*>(xw16 + (size_t)(8 * i2) * D + 32 * ct) = pack4(vv);
                    }
                }
                fs_wave_lds_fence();
            }
            if constexpr (QKV) {
                // ---- the next block's spatial qkv Linear on the normalised rows (attention.py:37 of st_transformer.py:74's call), written as the
                // operand planes spatial_attn_proj_bf16_kernel streams: Q (x scale log2 e) and K head-major [(seq, head)][pos][32],
                // V^T [(seq, head)][feature][256 keys, each 16-group stored {0-3, 8-11, 4-7, 12-15}] -- the formats of gemm16_pp's G16X_QKV.
                short* tl = reinterpret_cast<short*>(smem + 3 * FS_STAGE + wid * 4096);   // 32 rows x 40 shorts (64 bytes + pad)
                const long row0 = (long)blk * 128 + wid * 32;
                const long seq = row0 >> 8;
                const int pos0 = (int)(row0 & 255);
                uint16_t* planes = x16_out;
                int ln = lane;
                asm volatile("" : "+v"(ln));       // (everything lane-derived below is formed here, per block: hoisted to kernel entry it is spilled)
                const int t4 = ln >> 2, pc = (ln & 3) * 8, rq = ln & 31, hq = ln >> 5;
#ifndef GENIE_VAR_M_QWAIT
#define GENIE_VAR_M_QWAIT 2     // (0: every stage also waits out the previous stage's two plane stores)
#endif
                for (int n = 0; n < 24; ++n) {
                    // stage n's four LDS-DMA loads were requested at the top of stage n - 1; the only younger vector-memory operations are
                    // that stage's two plane stores (vmcnt retires in order) -- they stay in flight.  n = 0: behind the row stores above.
                    if (n == 0) fs_wait_vm<0>();
                    else fs_wait_vm<GENIE_VAR_M_QWAIT>();
                    fs_barrier();      // stage n landed for every wave; the other half is free (n = 0: everyone's phase above is over too)
                    if (n + 1 < 24) issue_q(n + 1);
                    const unsigned char* sb = smem + ((n & 1) ? 2 * FS_STAGE : FS_STAGE) + ln * 16;
                    const int part = n >> 3, ht = n & 7;
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                    // the stage's 16 fragments through 8 rotating buffers, reads the compiler does not schedule (it pairs them: read, read,
                    // wait, MFMA, wait, MFMA -- one LDS round trip per two matrix instructions, 44 % of the matrix rate; GENIE_VAR_M_QPF 0)
#ifndef GENIE_VAR_M_QPF
#define GENIE_VAR_M_QPF 8
#endif
#if GENIE_VAR_M_QPF > 0
                    const unsigned lq = fs_lds_addr(sb);
                    s16x8 qf[GENIE_VAR_M_QPF];
#define QS_SWAPPED(f) acc = mma32x32(qf[(f) % GENIE_VAR_M_QPF], yk[f], acc);
#define QS_PLAIN(f) acc = mma32x32(yk[f], qf[(f) % GENIE_VAR_M_QPF], acc);
#define QS_STEP(f, OP)                                                                                                            \
    {                                                                                                                             \
        fs_lds_wait<((15 - (f)) < (GENIE_VAR_M_QPF - 1) ? (15 - (f)) : (GENIE_VAR_M_QPF - 1))>(qf[(f) % GENIE_VAR_M_QPF]);          \
        OP(f)                                                                                                                     \
        if constexpr ((f) + GENIE_VAR_M_QPF < 16) fs_lds_rd<((f) + GENIE_VAR_M_QPF) * 1024>(qf[(f) % GENIE_VAR_M_QPF], lq);        \
    }
#define QS_ALL(OP)                                                                                                                 \
    {                                                                                                                             \
        fs_lds_rd<0>(qf[0], lq);                                                                                                  \
        if constexpr (GENIE_VAR_M_QPF > 1) fs_lds_rd<1024>(qf[1 % GENIE_VAR_M_QPF], lq);                                          \
        if constexpr (GENIE_VAR_M_QPF > 2) { fs_lds_rd<2048>(qf[2 % GENIE_VAR_M_QPF], lq); fs_lds_rd<3072>(qf[3 % GENIE_VAR_M_QPF], lq); } \
        if constexpr (GENIE_VAR_M_QPF > 4) {                                                                                      \
            fs_lds_rd<4096>(qf[4 % GENIE_VAR_M_QPF], lq); fs_lds_rd<5120>(qf[5 % GENIE_VAR_M_QPF], lq);                             \
            fs_lds_rd<6144>(qf[6 % GENIE_VAR_M_QPF], lq); fs_lds_rd<7168>(qf[7 % GENIE_VAR_M_QPF], lq);                             \
        }                                                                                                                         \
        QS_STEP(0, OP) QS_STEP(1, OP) QS_STEP(2, OP) QS_STEP(3, OP) QS_STEP(4, OP) QS_STEP(5, OP) QS_STEP(6, OP) QS_STEP(7, OP)     \
        QS_STEP(8, OP) QS_STEP(9, OP) QS_STEP(10, OP) QS_STEP(11, OP) QS_STEP(12, OP) QS_STEP(13, OP) QS_STEP(14, OP) QS_STEP(15, OP) \
    }
#else
#define QS_SWAPPED(f) acc = mma32x32(frag(sb, f), yk[f], acc);
#define QS_PLAIN(f) acc = mma32x32(yk[f], frag(sb, f), acc);
#define QS_ALL(OP)                                                                                                                 \
    { OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) }
#endif
                    if (part < 2) {    // swapped: D[feature][token]
                        QS_ALL(QS_SWAPPED)
                        const float sc = part == 0 ? qscale : 1.0f;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<s16x4*>(tl + rq * 40 + 8 * j + 4 * hq) = pack4(f32x4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]} * sc);
                        fs_wave_lds_fence();
                        uint16_t* dst = planes + (size_t)part * P + ((size_t)(seq * 8 + ht) * 256 + pos0) * 32;
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps)
                            *reinterpret_cast<s16x8*>(dst + (t4 + 16 * ps) * 32 + pc) = *reinterpret_cast<const s16x8*>(tl + (t4 + 16 * ps) * 40 + pc);
                        fs_wave_lds_fence();
                    } else {           // plain: D[token][feature] -- a lane holds 16 keys of ONE feature, in the planes' key order
                        QS_ALL(QS_PLAIN)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2)
                            *reinterpret_cast<s16x8*>(tl + rq * 40 + 16 * g2 + 8 * hq) =
                                pack8(f32x4{acc[8 * g2], acc[8 * g2 + 1], acc[8 * g2 + 2], acc[8 * g2 + 3]},
                                      f32x4{acc[8 * g2 + 4], acc[8 * g2 + 5], acc[8 * g2 + 6], acc[8 * g2 + 7]});
                        fs_wave_lds_fence();
                        uint16_t* dst = planes + (size_t)2 * P + ((size_t)(seq * 8 + ht) * 32) * 256 + pos0;
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps)
                            *reinterpret_cast<s16x8*>(dst + (size_t)(t4 + 16 * ps) * 256 + pc) = *reinterpret_cast<const s16x8*>(tl + (t4 + 16 * ps) * 40 + pc);
                        fs_wave_lds_fence();
                    }
                }
                fs_barrier();          // slot 1 and the tiles are free before anyone's next prologue writes its tile there
            }
        }
        FS_STAMP(blk_i, 3);
    }
    FS_CYC_DUMP();
    fs_wait_vm<0>();
}

// the next block's spatial qkv weights (768, 256) as 24 stages of 16 fragments (stage = part x head; fragment f = 2 ct + kk contracts the
// features of accumulator half-tile (ct, kk), as pack_mlp_fused_kernel's fc1 fragments)
__global__ void pack_spatial_qkv_kernel(const float* __restrict__ qkv_w, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 24 * 16 * 64 * 8) return;
    const int e = i & 7, l = (i >> 3) & 63, f = (i >> 9) & 15, n = i >> 13;
    const int h = l >> 5, rr = l & 31;
    out[i] = f32_to_bf16(qkv_w[(size_t)((n >> 3) * 256 + (n & 7) * 32 + rr) * 256 + 32 * (f >> 1) + 16 * (f & 1) + (e & 3) + 8 * (e >> 2) + 4 * h]);
}

int launch_pack_spatial_qkv(const float* qkv_w, uint16_t* out, hipStream_t st) {
    pack_spatial_qkv_kernel<<<(24 * 16 * 64 * 8) / 256, 256, 0, st>>>(qkv_w, out);
    GENIE_LAUNCH_CHECK("pack_spatial_qkv");
    return GENIE_OK;
}

// x += Mlp(LayerNorm(x)) on (rows, 256); x16_out (optional): bf16 shadow of the result, or -- when nx_g / nx_b are given --
// LayerNorm(result; nx_g, nx_b) as bf16.  GENIE_E_UNSUPPORTED outside the geometry.
#ifndef GENIE_VAR_S_MIN_SEQ
#define GENIE_VAR_S_MIN_SEQ 128   // fewest sequences the fused spatial kernel takes (one workgroup each; measured: 64 sequences lose 15 %, 128 gain 5 %, 192 gain 9 % over attention + proj GEMM)
#endif
// (mode 2 of the fused MLP kernel writes planes for any consumer of the format: the stand-alone attention kernel takes them below
// the fused spatial kernel's threshold, so no lower bound is needed on the producer side)
#define GENIE_VAR_S_MIN_SEQ_DECL 0
#ifndef GENIE_VAR_M_MIN_CLIPS
#define GENIE_VAR_M_MIN_CLIPS 2   // fewest clips' worth of rows (4,096 each; same measurement)
#endif
// The spatial operand planes of n_seq sequences are addressed with 32-bit scalar offsets: one plane (n_seq * 256 * 256 bf16 values)
// plus a tile of slack must stay below 2^31 bytes.  ONE predicate for the producer (fused MLP kernel, mode 2) and the consumer
// (spatial_attn_proj): planes that are written can always be read (n_seq < ~15,200).
static bool spatial_planes_addressable(long n_seq) { return (double)n_seq * 256 * 256 * 2 + 4096.0 * 256 < 2.0e9; }
static int device_cus() { return device_cu_count(); }   // (common.hpp: cached per device)

int launch_mlp_fused_bf16(const genie_cfg& c, const genie_layer_weights& lw, float* x, uint16_t* x16_out, long rows, hipStream_t st,
                          const float* nx_g, const float* nx_b, const uint16_t* nx_qkv_stream, uint16_t* planes) {
    if (c.precision != GENIE_PREC_BF16 || !lw.mlp_fused_w16 || c.d_model != 256 || c.hidden != 1024 || c.qk_norm || rows % 128 || rows < GENIE_VAR_M_MIN_CLIPS * 4096 || !lw.norm2_w ||
        !lw.norm2_b)
        return GENIE_E_UNSUPPORTED;
    GENIE_CHECK_ARG((nx_g == nullptr) == (nx_b == nullptr) && (!nx_g || x16_out || planes), "mlp_fused: next-norm parameters need both pointers and an output");
    // planes: the next block's spatial operand planes instead of its norm1 output (needs its qkv fragment stream, no qkv bias,
    // sequences of 256 tokens -- a wave's 32 rows never straddle one -- and the scalar offsets of the planes inside 2^31)
    const bool qkv = planes && nx_qkv_stream && nx_g && !c.qkv_bias && c.S == 256 && c.num_heads == 8 && c.head_dim == 32 && rows % 256 == 0 &&
                     spatial_planes_addressable(rows / 256) && rows / 256 >= GENIE_VAR_S_MIN_SEQ_DECL;
    if (planes && !qkv) return GENIE_E_UNSUPPORTED;
    const int n_blocks = (int)(rows / 128);
    const int cus = device_cus();
    const int grid = n_blocks < 2 * cus ? n_blocks : 2 * cus;
    ProfScope prof(GENIE_KC_FUSED, (double)rows * (4.0 * 256 * 1024 + (qkv ? 2.0 * 256 * 768 : 0.0)),
                   (double)rows * (2048.0 + (qkv ? 1536.0 : x16_out ? 512.0 : 0.0)), st,
                   qkv ? "mlp_fused_bf16_kernel<2> (LayerNorm + fc1 + GELU + fc2 + residual + next block's LayerNorm + spatial qkv planes)"
                       : "mlp_fused_bf16_kernel (LayerNorm + fc1 + GELU + fc2 + residual)");
    const float* fb1 = c.mlp_bias ? lw.fc1_b : nullptr;
    const float* fb2 = c.mlp_bias ? lw.fc2_b : nullptr;
    const int abl = study_env("GENIE_FUSED_ABL", 0);
    fs_stamps_prepare();
    if (qkv) {
        { static PerDevice<bool> once; if (once.needs()) { (void)hipFuncSetAttribute((const void*)mlp_fused_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS); once.set(true); } }
        mlp_fused_bf16_kernel<2><<<grid, 256, ML_LDS, st>>>(x, lw.mlp_fused_w16, lw.norm2_w, lw.norm2_b, fb1, fb2, planes, nx_g, nx_b, n_blocks,
                                                            1e-5f, abl, nx_qkv_stream, rows * 256, c.attn_scale * 1.4426950408889634f);
    } else if (nx_g) {
        { static PerDevice<bool> once; if (once.needs()) { (void)hipFuncSetAttribute((const void*)mlp_fused_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS); once.set(true); } }
        mlp_fused_bf16_kernel<1><<<grid, 256, ML_LDS, st>>>(x, lw.mlp_fused_w16, lw.norm2_w, lw.norm2_b, fb1, fb2, x16_out, nx_g, nx_b,
                                                            n_blocks, 1e-5f, abl, nullptr, 0, 0.f);
    } else {
        { static PerDevice<bool> once; if (once.needs()) { (void)hipFuncSetAttribute((const void*)mlp_fused_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS); once.set(true); } }
        mlp_fused_bf16_kernel<0><<<grid, 256, ML_LDS, st>>>(x, lw.mlp_fused_w16, lw.norm2_w, lw.norm2_b, fb1, fb2, x16_out, nullptr,
                                                            nullptr, n_blocks, 1e-5f, abl, nullptr, 0, 0.f);
    }
    GENIE_LAUNCH_CHECK("mlp_fused_bf16");
    return GENIE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// spatial_attn_proj_bf16_kernel:  x += proj_s( softmax(q k^T) v )  over the 256 tokens of a frame, all 8 heads (attention.py:36-61
// with causal = False, st_transformer.py:73-74), from the operand planes [Q * scale * log2e | K | V^T] the spatial qkv GEMM writes
// (launch_gemm16_pp with G16X_QKV; layouts in kernels_attn_dma.hip).  Replaces the attention launch + the out-projection GEMM:
// the attention output (134 MB written + read per layer at 64 clips) never exists and the residual row is updated once.
//   * one workgroup of 8 waves per (b, t) sequence, persistent over sequences; wave w owns queries 32 w .. 32 w + 31 of EVERY head,
//     so a token's output row accumulates in its lane pair's registers across the heads (8 tiles of 32 columns, 128 registers).
//   * per head: K (16 KB), V^T (16 KB) and the out-projection's 16 fragments for that head (16 KB) arrive by LDS-DMA in a
//     double buffer together with the head's Q rows (16 KB: a wave reads its own 32 rows back as fragments; held in registers one
//     head ahead they cost 16 registers the kernel does not have); the next head's 64 KB are requested at this head's barrier,
//     across sequence boundaries.
//   * with 128 registers taken by the output row there is no room for a query's 256 scores (128 registers per lane), so the
//     softmax is ONLINE over eight tiles of 32 keys: S^T tile (2 MFMAs, lane = query, 16 scores) -> running max / sum update,
//     O rescale -> P (bf16) is the B operand and the V^T fragment the A operand of O^T += V^T P^T (2 MFMAs, lane = query, 16
//     features) -> after the last tile O / sum, rounded to bf16, is the B operand of the head's two out-projection K-steps.
// Numerics = the bf16 attention contract of attn_spatial_dma_kernel (q, k, v, p, o rounded to bf16, f32 accumulation and softmax);
// the probabilities are taken against the running maximum instead of the row maximum (same value up to the rounding of p).
#ifndef GENIE_VAR_S_ABL
#define GENIE_VAR_S_ABL 0   // variant builds only (results wrong): 1 no LDS-DMA after the first item, 2 no residual epilogue,
#endif                      // 4 no matrix instructions, 8 no exponentials, 16 no LDS fragment reads
namespace {
__device__ __forceinline__ f32x16 sa_mma(const s16x8& a, const s16x8& b, const f32x16& c) {
    if constexpr (GENIE_VAR_S_ABL & 4) {
        f32x16 r_ = c;
        asm volatile("" : "+v"(r_) : "v"(a), "v"(b));
        return r_;
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float sa_exp2(float v) {
    if constexpr (GENIE_VAR_S_ABL & 8) return v * 0.001f;
    return __builtin_amdgcn_exp2f(v);
}
__device__ __forceinline__ s16x8 sa_frag(const unsigned char* p, int lane) {
    if constexpr (GENIE_VAR_S_ABL & 16) { const short v = (short)lane; return s16x8{v, v, v, v, v, v, v, v}; }
    return *reinterpret_cast<const s16x8*>(p);
}
constexpr int SA_BUF = 4 * 16384;                 // K | V^T | Wp fragments | Q of one head
constexpr int SA_LDS = 2 * SA_BUF + 1024;         // double buffer + out-projection bias
}  // namespace

// Wp stream of one layer (bf16, 8 heads x 16 fragments x 64 lanes x 8 values = 64 K values = 128 KB): head hd, fragment f = 8 kk + ct:
//   [lane l][e] = Wproj[32 ct + (l & 31)][32 hd + 16 kk + (e & 3) + 8 (e >> 2) + 4 (l >> 5)]
__global__ void pack_spatial_proj_kernel(const float* __restrict__ proj_w, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * 16 * 64 * 8) return;
    const int e = i & 7, l = (i >> 3) & 63, f = (i >> 9) & 15, hd = i >> 13;
    const int kk = f >> 3, ct = f & 7, h = l >> 5;
    out[i] = f32_to_bf16(proj_w[(size_t)(32 * ct + (l & 31)) * 256 + 32 * hd + 16 * kk + (e & 3) + 8 * (e >> 2) + 4 * h]);
}

int launch_pack_spatial_proj(const float* proj_w, uint16_t* out, hipStream_t st) {
    pack_spatial_proj_kernel<<<(8 * 16 * 64 * 8) / 256, 256, 0, st>>>(proj_w, out);
    GENIE_LAUNCH_CHECK("pack_spatial_proj");
    return GENIE_OK;
}

#ifndef GENIE_VAR_S_RES_AT_START
#define GENIE_VAR_S_RESEND 1   // the residual row joins in the epilogue, row-major (variant: as the accumulators' initial value)
#endif
__global__ __launch_bounds__(512, 2) void spatial_attn_proj_bf16_kernel(const uint16_t* __restrict__ qkv16, long P,
                                                                        const uint16_t* __restrict__ wstream,
                                                                        const float* __restrict__ proj_b, float* __restrict__ x,
                                                                        uint16_t* __restrict__ x16, long n_seq, int stagger) {
    constexpr int D = 256, NH = 8, DH = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    float* sbias = reinterpret_cast<float*>(smem + 2 * SA_BUF);
    if (tid < 256) sbias[tid] = proj_b ? proj_b[tid] : 0.f;
    __syncthreads();

    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qkv16, 0, -1, 0x00020000);
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv16 + (size_t)P), 0, -1, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv16 + (size_t)2 * P), 0, -1, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, NH * 16384, 0x00020000);
    // per-lane source offsets of this wave's LDS-DMA pieces (two K, two V^T, two Wp per head), conventions of kernels_attn_dma.hip:
    //   K chunk (64 keys x 64 bytes): piece = 16 rows, the 16-byte slot of a row XOR (row / 4) % 4
    //   V^T chunk (32 features x 128 bytes = 64 keys): piece = 8 feature rows, slot XOR (feature / 2) % 8
    unsigned voK[2], voV[2];
    int pcK[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = 2 * wid + j;                  // piece 0..15 of the head's K (and of its V^T): chunk pc >> 2, part pc & 3
        pcK[j] = pc;
        const int pp = pc & 3;
        {
            const int row = pp * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3);
            voK[j] = (unsigned)((row * DH + slot * 8) * 2);
        }
        {
            const int f = pp * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((f >> 1) & 7);
            voV[j] = (unsigned)((f * 256 + slot * 8) * 2);
        }
    }
    // item = (sequence, head) in the order this workgroup walks them
    auto issue_item = [&](long seq, int hd, int buf) {
        unsigned char* base = smem + buf * SA_BUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = pcK[j] >> 2, pp = pcK[j] & 3;
            const int soK = (int)((((seq * NH + hd) * 256 + c * 64) * (long)DH) * 2);
            const int soV = (int)((((seq * NH + hd) * DH) * 256 + c * 64) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(base + c * 4096 + pp * 1024), 16,
                                                     voK[j], soK, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(base + 16384 + c * 4096 + pp * 1024),
                                                     16, voV[j], soV, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + 32768 + pcK[j] * 1024), 16,
                                                     (unsigned)lane * 16, hd * 16384 + pcK[j] * 1024, 0, 0);
            // Q rows 16 (2 wid + j) .. + 15 of the head (this wave's own queries), same row image as a K chunk
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(base + 49152 + pcK[j] * 1024), 16,
                                                     voK[j], soK, 0, 0);
        }
    };
    // fragment read offsets (bytes inside a chunk): K rows, V^T rows
    unsigned offK[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) offK[kk] = r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
    const unsigned offV = r * 128;
    const int vsw = (r >> 1) & 7;

    const long seq0 = blockIdx.x, step = gridDim.x;
    if (seq0 >= n_seq) return;
    // Staggered start: every workgroup does identical work, so the chip would run in lockstep -- all CUs in their attention phases
    // with HBM idle, then all 256 bursts of 640 KB (residual rows in, updated rows and their shadow out) at once, which must drain
    // before anyone passes the next head's wait (measured: the kernel's time was compute + bytes / HBM bandwidth).  Workgroup b
    // waits (b mod 8) x stagger ticks of 10 ns before its first sequence, which spreads the bursts over the sequence period for good.
    if (stagger > 0) {
        const unsigned long long wait = (unsigned long long)(blockIdx.x & 7) * (unsigned long long)stagger;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    issue_item(seq0, 0, 0);
    int buf = 0;
    for (long seq = seq0; seq < n_seq; seq += step) {
        // The residual row: either the accumulators' initial value (GENIE_VAR_S_RES_AT_START: 32 row PIECES of 32 bytes per request, issued
        // here and first needed at the end of head 0 -- fully hidden, 320-325 us), or -- shipped -- added in the epilogue from row-major
        // loads (8 whole 128-byte segments per request, one column tile ahead of its use: 317-318 us, config 2 -0.5 % same-box).
        [[maybe_unused]] float* xrow = x + ((size_t)seq * 256 + wid * 32 + r) * D + 4 * h;
        f32x16 out[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#ifdef GENIE_VAR_S_RESEND
                const f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};   // (variant: the residual joins in the epilogue, row-major)
#else
                const f32x4 xv = (GENIE_VAR_S_ABL & 64) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(xrow + 32 * ct + 8 * j);
#endif
                out[ct][4 * j] = xv.x; out[ct][4 * j + 1] = xv.y; out[ct][4 * j + 2] = xv.z; out[ct][4 * j + 3] = xv.w;
            }
#pragma unroll 1
        for (int hd = 0; hd < NH; ++hd) {
            // this head's 64 KB have landed for every wave; the other buffer is free.  Head 0 of every sequence but the first: the
            // 64 stores of the previous sequence's epilogue are YOUNGER than this head's 8 LDS-DMA pieces (requested at that
            // sequence's last barrier) and may stay in flight -- vmcnt retires in order, so 63 outstanding means the pieces are in.
            // The count must not exceed the number of epilogue operations that are ALWAYS issued: 32 f32 stores, plus -- in the shipped
            // form, GENIE_VAR_S_RESEND -- the 32 residual loads of the row-major epilogue (the 32 bf16 stores only exist with x16).
#ifdef GENIE_VAR_S_RESEND
            constexpr int YOUNGER_MIN = 64;
#else
            constexpr int YOUNGER_MIN = 32;   // (GENIE_VAR_S_RES_AT_START: the residual loads sit in front of the head loop)
#endif
            if (hd == 0 && seq != seq0) fs_wait_vm<YOUNGER_MIN - 1>(); else fs_wait_vm<0>();
            fs_barrier();
            {
                const bool last_h = hd == NH - 1;
                const long nseq = last_h ? seq + step : seq;
                const int nhd = last_h ? 0 : hd + 1;
                if (nseq < n_seq && !(GENIE_VAR_S_ABL & 1)) issue_item(nseq, nhd, buf ^ 1);
            }
            const unsigned char* kb = smem + buf * SA_BUF;
            s16x8 qf[2];   // B operand of S^T = K Q^T: lane = query r, 8 features 16 kk + 8 h ..
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const s16x8*>(kb + 49152 + wid * 2048 + offK[kk]);
            const unsigned char* vb = kb + 16384;
            const unsigned char* wb = kb + 32768 + lane * 16;
            f32x16 o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
            float m = -1.0e30f, l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {        // 32 keys per step (chunk c = kt / 2 of the K / V^T images, tile t = kt % 2)
                const int c = kt >> 1, t = kt & 1;
                f32x16 sc;
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    sc = sa_mma(sa_frag(kb + c * 4096 + t * 2048 + offK[kk], lane), qf[kk], sc);
                float mc = sc[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mc = fmaxf(mc, sc[e]);
                mc = fmaxf(mc, __shfl_xor(mc, 32));
                const float mn = fmaxf(m, mc);
                const float alpha = sa_exp2(m - mn);
                m = mn;
                float ls = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sc[e] = sa_exp2(sc[e] - mn);
                    ls += sc[e];
                }
                l = l * alpha + ls;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] *= alpha;
                // O^T += V^T P^T: MFMA mm takes the lane's scores 8 mm .. 8 mm + 7 of the tile (keys 16 mm + (s & 3) + 8 (s >> 2) + 4 h:
                // the order the V^T planes store every 16-key group in)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const s16x8 pa = pack8(f32x4{sc[8 * mm], sc[8 * mm + 1], sc[8 * mm + 2], sc[8 * mm + 3]},
                                           f32x4{sc[8 * mm + 4], sc[8 * mm + 5], sc[8 * mm + 6], sc[8 * mm + 7]});
                    const s16x8 vf = sa_frag(vb + c * 4096 + offV + (((t * 4 + 2 * mm + h) ^ vsw) << 4), lane);
                    o = sa_mma(vf, pa, o);
                }
            }
            l += __shfl_xor(l, 32);
            const float inv = __builtin_amdgcn_rcpf(l);
            const s16x8 ob0 = pack8(f32x4{o[0], o[1], o[2], o[3]} * inv, f32x4{o[4], o[5], o[6], o[7]} * inv);
            const s16x8 ob1 = pack8(f32x4{o[8], o[9], o[10], o[11]} * inv, f32x4{o[12], o[13], o[14], o[15]} * inv);
#pragma unroll
            for (int f = 0; f < 16; ++f)
                out[f & 7] = sa_mma(sa_frag(wb + f * 1024, lane), (f >> 3) ? ob1 : ob0, out[f & 7]);
            buf ^= 1;
        }
        // ---- epilogue: x = out + bias (the row already holds x + all heads' projections) and its bf16 shadow (the temporal
        // sub-block's operand).  A lane owns a TOKEN, so a store straight from the accumulators writes 32-byte pieces of 32 different
        // rows -- measured: those requests, not their bytes, were most of the epilogue's cost (the same bytes as lane-linear stores:
        // -40 us of 350).  So each 32-column tile goes through a wave-private 4.5 KB LDS tile (rows padded to 144 bytes) and leaves as
        // whole 128-byte lines: 8 rows per store instruction.  The tile lives in the buffer of the head just finished (one extra
        // barrier per sequence makes sure every wave is done reading it).
        if constexpr (GENIE_VAR_S_ABL & 2) { asm volatile("" :: "v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7])); continue; }
        fs_barrier();
        {
            float* tile = reinterpret_cast<float*>(smem + (buf ^ 1) * SA_BUF + wid * 8192);   // (buf was toggled after the last head)
            size_t soff = ((size_t)seq * 256 + wid * 32) * D;
            asm volatile("" : "+s"(soff));   // (address arithmetic stays here: hoisted above the head loop it is spilled)
            float* xw = x + soff;
            uint16_t* xw16 = x16 + soff;
            int w16 = x16 != nullptr;
            asm volatile("" : "+v"(w16));   // per-lane predicate: the store is masked, the loop not duplicated
            // (an opaque copy of the lane id: what the epilogue derives from it -- rr, cc, the tile row offset r * 36 -- is formed HERE;
            // derived from `lane` itself the compiler computed (lane & 7) * 4 and (lane & 31) * 144 in front of the head loop, spilled
            // them at 256 registers and reloaded them here behind an s_waitcnt vmcnt(0) that also drained the next head's LDS-DMA)
            int le = lane;
            asm volatile("" : "+v"(le));
            const int rr = le >> 3, cc = (le & 7) * 4;        // row-major side: row rr + 8 i, columns cc .. cc + 3 of the tile
            const int re = le & 31, he = le >> 5;             // accumulator side: (r, h) of this lane
#ifdef GENIE_VAR_S_RESEND
#ifndef GENIE_VAR_S_EPF
#define GENIE_VAR_S_EPF 1      // residual column tiles requested ahead of their use
#endif
            constexpr int EPF = GENIE_VAR_S_EPF;
            f32x4 rs[EPF + 1][4];
            auto load_res = [&](int ct_, f32x4* dst) {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) dst[i2] = *reinterpret_cast<const f32x4*>(xw + (size_t)(rr + 8 * i2) * D + 32 * ct_ + cc);
            };
#pragma unroll
            for (int i = 0; i < EPF; ++i) load_res(i, rs[i % (EPF + 1)]);
#endif
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
#ifdef GENIE_VAR_S_RESEND
                if (ct + EPF < 8) load_res(ct + EPF, rs[(ct + EPF) % (EPF + 1)]);
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(tile + re * 36 + 8 * j + 4 * he) =
                        f32x4{out[ct][4 * j], out[ct][4 * j + 1], out[ct][4 * j + 2], out[ct][4 * j + 3]} +
                        *reinterpret_cast<const f32x4*>(sbias + 32 * ct + 8 * j + 4 * he);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
#ifdef GENIE_VAR_S_RESEND
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * i2) * 36 + cc) + rs[ct % (EPF + 1)][i2];
#else
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * i2) * 36 + cc);
#endif
                    *reinterpret_cast<f32x4*>(xw + (size_t)(rr + 8 * i2) * D + 32 * ct + cc) = v;
                    if (w16) *reinterpret_cast<s16x4*>(xw16 + (size_t)(rr + 8 * i2) * D + 32 * ct + cc) = pack4(v);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    fs_wait_vm<0>();
}

// x += proj_s(attention_S(planes)) and x16 = bf16(x) for n_seq sequences of 256 tokens; GENIE_E_UNSUPPORTED outside d 256 / 8 x 32
int launch_spatial_attn_proj_bf16(const genie_cfg& c, const genie_attn_weights& aw, const uint16_t* qkv16, float* x, uint16_t* x16,
                                  long n_seq, hipStream_t st) {
    if (c.precision != GENIE_PREC_BF16 || !aw.fused_w16 || c.d_model != 256 || c.num_heads != 8 || c.head_dim != 32 || c.S != 256 || c.qk_norm ||
        n_seq < GENIE_VAR_S_MIN_SEQ)
        return GENIE_E_UNSUPPORTED;
    const long P = n_seq * 256 * 256;
    if (!spatial_planes_addressable(n_seq)) return GENIE_E_UNSUPPORTED;   // 32-bit scalar offsets inside the plane descriptors
    const int cus = device_cus();
    const unsigned grid = (unsigned)(n_seq < cus ? n_seq : cus);
    const double M = (double)n_seq * 256;
    ProfScope prof(GENIE_KC_FUSED, M * (4.0 * 256 * 256 + 2.0 * 256 * 256), M * (3 * 512.0 + 2048.0 + 512.0), st,
                   "spatial_attn_proj_bf16_kernel (attention over S, all heads + proj + residual)");
    { static PerDevice<bool> once; if (once.needs()) { (void)hipFuncSetAttribute((const void*)spatial_attn_proj_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SA_LDS); once.set(true); } }
#ifndef GENIE_VAR_S_STAGGER
#define GENIE_VAR_S_STAGGER 0     // (measured: 3-15 us per class only adds the delay -- profiles/r04_fused_experiments.txt)
#endif
    const int stagger = n_seq >= 2 * (long)grid ? study_env("GENIE_FUSED_STAGGER_S", GENIE_VAR_S_STAGGER) : 0;
    spatial_attn_proj_bf16_kernel<<<grid, 512, SA_LDS, st>>>(qkv16, P, aw.fused_w16, c.proj_bias ? aw.proj_b : nullptr, x, x16, n_seq,
                                                             stagger);
    GENIE_LAUNCH_CHECK("spatial_attn_proj_bf16");
    return GENIE_OK;
}

#ifndef GENIE_VAR_T_MIN_CLIPS
#define GENIE_VAR_T_MIN_CLIPS 2   // (measured at 1 / 2 / 3 / 4 / 6 clips: 1 clip is 14 % slower fused, from 2 clips on 6-30 % faster)
#endif
// x += proj_t(attention_T(qkv_t(x16))) on dense (B, 16, S, 256) buffers; GENIE_E_UNSUPPORTED for any other geometry
bool temporal_fused_takes(const genie_cfg& c, const genie_attn_weights& aw, int B) {
    // (precision: in GENIE_PREC_F16X3 `fused_w16` is the split-f16 qkv stream of kernels_fused_f16x3.hip, not this kernel's bf16
    // [qkv | proj] stream; qk_norm: the shipped config -- genie/configs/magvit_n32_h8_d256.json -- has LayerNorm blocks, and the d = 256
    // fused kernels are specialisations for it: a qk-norm model of this width runs the unfused launches)
    return c.precision == GENIE_PREC_BF16 && aw.fused_w16 && c.d_model == 256 && c.num_heads == 8 && c.head_dim == 32 && c.T == 16 &&
           c.S % 8 == 0 && !c.qk_norm && (long)B * c.S >= GENIE_VAR_T_MIN_CLIPS * 256;
}

int launch_temporal_fused_bf16(const genie_cfg& c, const genie_attn_weights& aw, const uint16_t* x16, float* x, int B,
                               hipStream_t st) {
    if (!temporal_fused_takes(c, aw, B)) return GENIE_E_UNSUPPORTED;
    const int n_blocks = B * c.S / 8;
    const int cus = device_cus();
    const int grid = n_blocks < 2 * cus ? n_blocks : 2 * cus;
    const double M = (double)B * c.T * c.S;
    ProfScope prof(GENIE_KC_FUSED, M * (2.0 * 256 * 1024 + 4.0 * 16 * 256), M * (512.0 + 2048.0), st,
                   "temporal_fused_bf16_kernel (qkv + causal attention over T + proj + residual)");
#ifndef GENIE_VAR_T_NO_TILE
    const size_t lds = FS_RING + 4096 + 4 * 2304;   // ring, biases, one 16 x 36-float tile per wave
#else
    const size_t lds = FS_RING + 4096;
#endif
    fs_stamps_prepare();
    const float sl2e = c.attn_scale * 1.4426950408889634f;
    const int abl = study_env("GENIE_FUSED_ABL", 0);
    const float* pb = c.proj_bias ? aw.proj_b : nullptr;
#define T_LAUNCH(QB_, XF_)                                                                                                        \
    do {                                                                                                                           \
        (void)hipFuncSetAttribute((const void*)temporal_fused_bf16_kernel<QB_, XF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        temporal_fused_bf16_kernel<QB_, XF_><<<grid, 256, lds, st>>>(x16, x, aw.fused_w16, QB_ ? aw.qkv_b : nullptr, pb, n_blocks, c.S, sl2e, abl); \
    } while (0)
    const bool qb = c.qkv_bias && aw.qkv_b;
    if (!x16) { if (qb) T_LAUNCH(true, true); else T_LAUNCH(false, true); }     // x16 == NULL: operands from the f32 rows
    else { if (qb) T_LAUNCH(true, false); else T_LAUNCH(false, false); }
#undef T_LAUNCH
    GENIE_LAUNCH_CHECK("temporal_fused_bf16");
    return GENIE_OK;
}

}  // namespace genie
