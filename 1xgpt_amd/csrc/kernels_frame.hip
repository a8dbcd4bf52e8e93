// kernels_frame.hip -- the ONE-FRAME passes of generate (genie_frame_pass: 256 rows per clip and frame), GENIE_PREC_F16X3.
//
// Reference: generate.py:81-95 calls STMaskGIT.maskgit_generate once per new frame; every MaskGIT step of it is one forward
// (st_mask_git.py:163-169) through STBlock (st_transformer.py:70-83).  With the temporal KV cache only the rows of the frame being
// decoded go through the stack: M = 256 rows per clip, so every Linear is a 256 x N x K problem whose weights (4 bytes per
// parameter as split f16) outweigh its activations, and a launch lasts a few microseconds.
//
// What bounded the round-3 kernels of these passes (gemm16_sm, attn_spatial_keysplit, attn_temporal_single) was not bytes but
// REQUESTS: their operand loads were MFMA-fragment shaped -- lane (r, h) reads 16 bytes of row r -- so ONE load instruction
// touched 32 (f16 rows) to 64 (f32 rows) different 128-byte lines for 1 KB of data, and the vector memory pipeline charges per
// line touched (36-40 GB/s per CU measured, bf16 vs f16x3 and K = 512 vs 2048 on one line: profiles/r03_sm_prefetch_ab.txt).
// Everything here moves WHOLE LINES:
//
//   * 16-bit operands live in FRAGMENT ORDER ("fr"): a matrix X[rows][K] is cut into blocks of 32 rows x 64 k; a block is
//     NPL planes x 4 MFMA steps of one 1 KB FRAGMENT = the 64 lanes' 16-byte operand pieces of v_mfma_f32_32x32x16_f16
//     (lane 32 h + r: row r, k = 16 step + 8 h .. + 7).  A fragment load is one instruction over 1 KB of contiguous memory
//     (8 lines), a wave's operands of one block are 8 KB contiguous.  Weights are packed once (genie_pack_frame_w16);
//     activations are written in this order by the producing kernel's epilogue.
//   * f32 rows (the residual stream in front of a LayerNorm) are read as whole rows: 16 lanes x 16 bytes per row and 64-k
//     block, the row statistics stay inside the wave, the normalised row goes through LDS into fragment order.
//   * the spatial attention reads [Q scale log2e | K | V^T] fragments that the qkv Linear's epilogue wrote for it; the temporal
//     decode attention reads the KV cache as whole 256-byte head slices (4 frames per instruction).
//
// Kernels: gemm16_fr_kernel (nn.Linear with in-workgroup split-K over 8 waves, the whole contraction in flight; epilogues:
// f32 rows / residual update (+ operand copy) / GELU operand / attention operand planes), attn_spatial_fr_kernel,
// attn_temporal_fr_kernel, pack_frame_w16_kernel.  Driver: st_block_frame_f16x3 (one STBlock of a frame pass).
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int FR = 512;        // 16-bit elements of one fragment (64 lanes x 8)
constexpr int NPL = 2;         // planes of the split-f16 operands: hi, lo'

__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// hi / lo' fragments of eight f32 values (element e in half e&1 of dword e>>1)
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
    uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
    split_f16_x4(v[0], v[1], v[2], v[3], h01, h23, l01, l23);
    split_f16_x4(v[4], v[5], v[6], v[7], h45, h67, l45, l67);
    hi = u32x4{h01, h23, h45, h67};
    lo = u32x4{l01, l23, l45, l67};
}

// sum over the 16 lanes of a DPP row (lanes 16 i .. 16 i + 15), the same bits in every lane of the row: four row rotations
// (v_add with a DPP operand each) instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));  // row_ror:1
    return v;
}

// sum over aligned groups of 8 lanes (a head of 32 features at 4 per lane), the same bits in every lane of the group
__device__ __forceinline__ float lanes8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    return v;
}
// The reference's default attention variant (qk_norm, attention.py:31-34, 42-47): LayerNorm over the head_dim features of ONE head
// (eps 1e-5, biased variance, one affine shared by q, k and all heads), for a row-major phase in which a lane holds 4 consecutive
// features and a head's features sit in Dh / 4 consecutive aligned lanes (16 = one DPP row, or 8).
__device__ __forceinline__ f32x4 head_layer_norm(f32x4 v, int Dh, const f32x4& gm, const f32x4& bt) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    s = Dh == 64 ? row16_sum(s) : lanes8_sum(s);
    const float mean = s * (1.0f / (float)Dh);
    v -= mean;
    float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    q = Dh == 64 ? row16_sum(q) : lanes8_sum(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / (float)Dh) + 1e-5f);
    return v * rstd * gm + bt;
}

// Hand-placed LDS fragment reads for the LDS-tiled kernel: left to the compiler every group of matrix instructions sits behind an
// s_waitcnt lgkmcnt(0) (a full LDS round trip per group); fr_lds_rd issues a read the compiler knows nothing about, fr_lds_wait<N>
// is "all but my N youngest LDS reads have landed", tied to the fragment it guards (LDS reads return in order; no scalar load may
// be outstanding in such a section: it shares the counter).
template <int OFF>
__device__ __forceinline__ void fr_lds_rd(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void fr_lds_wait(u32x4& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}

// every operand request of these kernels is issued before the first wait; the scheduler must not sink loads below the
// matrix instructions to save registers (a launch is one memory round trip long)
#define FR_PIN_LOADS() __builtin_amdgcn_sched_barrier(0)

// element offset of fragment (plane p, step s) of block (rb, kb) in a fragment-ordered operand with KB k-blocks per row block
__host__ __device__ inline size_t fr_frag(long rb, int kb, int KB, int p, int s) {
    return ((((size_t)rb * KB + kb) * NPL + p) * 4 + s) * FR;
}
}  // namespace

enum { FR_EPI_F32 = 0, FR_EPI_RES = 1, FR_EPI_GELU = 2, FR_EPI_QKVS = 3 };

struct FrGemmArgs {
    // A operand: fragment-ordered rows (LNF == false) ...
    const uint16_t* A;
    int a_group, a_mul, a_off;   // row-block remap: block rb of the problem is block ((rb / a_group) * a_mul + a_off) * a_group + rb % a_group of A
    // ... or LayerNorm(x) of f32 rows (LNF == true)
    const float* X;
    long ldx;
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
    const uint16_t* W;           // fragment-ordered (N, K)
    const float* bias;           // (N) or NULL
    float alpha;
    int M, N, K;
    float* Cf;                   // EPI_F32: out rows; EPI_RES: the residual stream, updated in place
    long ldc, rows_per_batch, strideC;   // row -> Cf + (row / rows_per_batch) * strideC + (row % rows_per_batch) * ldc
    uint16_t* C16;               // EPI_RES (optional) / EPI_GELU: fragment-ordered operand copy of the output, N / 64 k-blocks
    uint16_t* qkvs;              // EPI_QKVS: attention operand planes
    float qscale;                // EPI_QKVS: softmax scale * log2(e), folded into Q
    int H, S, Dh;                // EPI_QKVS: heads, rows per sequence (256), head_dim (64 or 32 columns per head)
    const float* qn_g;           // EPI_QKVS, qk_norm: gamma | beta (Dh floats each) of the per-head LayerNorm of q and k, or NULL
    const float* qn_b;
};

// The spatial attention's operand planes (FR_EPI_QKVS), written from a finished 32 x 32 sub-tile of the qkv output that sits in LDS
// (`tl`: its first element, row pitch `pitch` floats): rows = the 32 tokens of row block rb, columns col0 .. col0 + 31 of the 3 d wide
// output = 32 features of ONE (q | k | v, head) slice (head_dim 64: half a head, head_dim 32: a whole one).  Per (sequence, head):
//   Q, K: [block of 32 tokens][plane][step 0 .. Dh / 16) fragments, lane (r, h): token r, features 16 step + 8 h .. + 7  (Q scaled)
//   V^T:  [key tile kt][feature tile dt < Dh / 32][step m][plane] fragments, lane (r, h): feature 32 dt + r of keys
//         32 kt + 16 m + 4 h + {0..3, 8..11} -- the key order in which the S^T accumulators hold the probabilities
// `sub` = 0, 1: the two fragments the sub-tile fills (Q / K: steps (col0 % Dh) / 16 + sub; V^T: m = sub); fl = lane of the fragment.
__device__ __forceinline__ void qkvs_store(const FrGemmArgs& a, const float* tl, int pitch, int rb, int col0, int sub, int fl) {
    const int r = fl & 31, h = fl >> 5;
    const int d = a.H * a.Dh, KS = a.Dh >> 4;
    const int which = col0 / d, head = (col0 % d) / a.Dh, o = col0 % a.Dh;
    const int blocks = a.S / 32;
    const long seq = rb / blocks;
    const int blk = rb % blocks;
    const size_t unit = (size_t)blocks * NPL * KS * FR;     // one of Q | K | V^T of a (sequence, head)
    uint16_t* base = a.qkvs + ((size_t)(seq * a.H + head) * 3 + which) * unit;
    float v[8];
    uint16_t* dst;
    int lo_off;
    if (which < 2) {
        const float sc = which == 0 ? a.qscale : 1.0f;
        const float* src = tl + r * pitch + 16 * sub + 8 * h;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * sc, v1 = *reinterpret_cast<const f32x4*>(src + 4) * sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
        dst = base + ((size_t)blk * NPL * KS + (o >> 4) + sub) * FR + fl * 8;
        lo_off = KS * FR;
    } else {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) v[s8] = tl[(16 * sub + 4 * h + (s8 & 3) + 8 * (s8 >> 2)) * pitch + r];
        dst = base + ((size_t)(blk * (a.Dh >> 5) + (o >> 5)) * 2 + sub) * NPL * FR + fl * 8;
        lo_off = FR;
    }
    u32x4 hi, lo;
    split8(v, hi, lo);
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + lo_off) = lo;
}

// ------------------------------------------------------------------------------------------------------------------------------
// C = epilogue(alpha * A . W^T + bias): workgroup = (32 MI) rows x (32 NJ) columns, NW waves split K (wave w: 64-k blocks w, w + NW, ..),
// NKB blocks per wave, RING of them in flight.  Two accumulators per tile as in gemm16_sm (hi.hi | hi.lo' + lo'.hi, the second
// scaled by 2^-11 at the end: valid for any operand magnitude).  MI = 2 (two-frame passes, M >= 512): the same number of workgroups as
// a one-frame pass, every weight fragment feeds both row blocks -- with MI = 1 the 384-512 workgroups of such a pass ran as two rounds
// (one workgroup per CU: LDS and registers) and the pass cost 1.7x a one-frame pass (profiles/r05b_gen1_kernel_stats.txt).
// A row's result does not depend on MI (same k order per wave, same wave order in the reduce).
// ------------------------------------------------------------------------------------------------------------------------------
template <int NW, int MI, int NJ, int NKB, int EPI, bool LNF>
__global__ __launch_bounds__(NW * 64, 1) __attribute__((amdgpu_waves_per_eu(NW >= 4 ? NW / 4 : 1, NW >= 4 ? NW / 4 : 1)))
void gemm16_fr_kernel(const FrGemmArgs a) {
    constexpr int NT = NW * 64, TM = 32 * MI, TN = 32 * NJ, PITCH = TN + 4;
    // k-blocks of a wave in flight (fc2: 4 per wave).  Three in flight (231 registers) measured +0.3 %: fc2's 10 us are its 512 KB per
    // workgroup on 128 CUs, not the second round trip (profiles/r05q_fc2_ring3_ab.txt)
    constexpr int RING = NKB > 1 ? 2 : 1;
    static_assert(!LNF || NKB == 1, "LayerNorm-fused A: the waves tile one row of K = 64 NW");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                                   // [NW][TM][PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = a.N / TN;
    const int rb0 = (blockIdx.x / nt) * MI, jt = blockIdx.x % nt;   // first 32-row block, column tile
    const int m0 = rb0 * 32, n0 = jt * TN;
    const int KB = a.K / 64;

    // (every kernel argument the request phase needs, in SGPRs before the first request: fetched lazily they were three to four
    // dependent scalar-cache round trips spread between the requests)
    asm volatile("" :: "s"(a.A), "s"(a.X), "s"(a.ln_g), "s"(a.ln_b), "s"(a.W), "s"(a.bias), "s"(a.Cf), "s"(a.ldx), "s"(a.ldc),
                 "s"(a.K), "s"(a.N), "s"(a.a_group), "s"(a.a_mul), "s"(a.a_off));
    // ---- requests first, oldest = needed first (loads return in order): bias and LayerNorm parameters, (LNF) the f32 rows, the
    // weight fragments of this wave's first RING blocks, the residual rows
    constexpr int C4 = TN / 4, ITEMS = TM * C4, PASSES = (ITEMS + NT - 1) / NT;
    static_assert(NT % C4 == 0, "a thread keeps its columns over the passes of the row-major phase");
    f32x4 pre_b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) pre_b = *reinterpret_cast<const f32x4*>(a.bias + n0 + (tid % C4) * 4);
    f32x4 pre_qg = f32x4{0.f, 0.f, 0.f, 0.f}, pre_qb = pre_qg;   // qk-norm: the affine of this thread's 4 features of a head
    const bool qkn = EPI == FR_EPI_QKVS && a.qn_g != nullptr && n0 < 2 * a.H * a.Dh;   // (a q or k column tile: workgroup-uniform)
    if (qkn) {
        pre_qg = *reinterpret_cast<const f32x4*>(a.qn_g + ((tid % C4) * 4) % a.Dh);
        pre_qb = *reinterpret_cast<const f32x4*>(a.qn_b + ((tid % C4) * 4) % a.Dh);
    }
    f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LNF) {   // gamma | beta: one float4 per thread (K / 2 <= NT float4s), staged through LDS
        if (tid < a.K / 4) gbv = reinterpret_cast<const f32x4*>(a.ln_g)[tid];
        else if (tid < a.K / 2) gbv = reinterpret_cast<const f32x4*>(a.ln_b)[tid - a.K / 4];
    }
    constexpr int RW = TM / NW;                 // rows per wave in the LayerNorm prologue
    constexpr int G = RW / 4 > 0 ? RW / 4 : 1;  // groups of 4 rows (a wave instruction = 4 rows x 64 floats)
    f32x4 xv[LNF ? G : 1][LNF ? NW : 1];
    if constexpr (LNF) {
        static_assert(!LNF || RW % 4 == 0, "LayerNorm prologue: 4 rows per instruction");
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float* xr = a.X + (size_t)(m0 + RW * wid + 4 * g + (lane >> 4)) * a.ldx + 4 * (lane & 15);
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) xv[g][kb] = *reinterpret_cast<const f32x4*>(xr + 64 * kb);
        }
    }
    u32x4 wf[RING][NJ][NPL][4], af[RING][MI][NPL][4];
    auto load_w = [&](int slot, int kb) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint16_t* src = a.W + fr_frag((long)jt * NJ + j, kb, KB, 0, 0) + lane * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                // (plain loads: a weight tile is read by the pass's 8+ row-block workgroups of the same XCD, so it should stay in L2 --
                // non-temporal loads measured -2.7 % at 1 clip, -4 % at 2: profiles/r05p_nt_weight_loads_ab.txt)
                for (int s = 0; s < 4; ++s) wf[slot][j][p][s] = *reinterpret_cast<const u32x4*>(src + (p * 4 + s) * FR);
        }
    };
    auto load_a = [&](int slot, int kb) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rb = rb0 + i;
            const long arb = ((long)(rb / a.a_group) * a.a_mul + a.a_off) * a.a_group + rb % a.a_group;
            const uint16_t* src = a.A + fr_frag(arb, kb, KB, 0, 0) + lane * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int s = 0; s < 4; ++s) af[slot][i][p][s] = *reinterpret_cast<const u32x4*>(src + (p * 4 + s) * FR);
        }
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) {
        if constexpr (!LNF) load_a(i, wid + NW * i);
        load_w(i, wid + NW * i);
    }

    // the residual rows of the epilogue (EPI_RES) go out now as well: one float4 per thread and pass of the row-major phase
    f32x4 pre_r[EPI == FR_EPI_RES ? PASSES : 1];
    if constexpr (EPI == FR_EPI_RES) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int idx = tid + q * NT;
            const int rl = idx / C4, c4 = (idx % C4) * 4;
            pre_r[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < ITEMS) pre_r[q] = *reinterpret_cast<const f32x4*>(a.Cf + (size_t)(m0 + rl) * a.ldc + n0 + c4);
        }
    }
    FR_PIN_LOADS();

    f32x16 accm[MI][NJ], accc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[i][j][e] = 0.f; accc[i][j][e] = 0.f; }
    // column tile j outermost: its weight fragments are dead after its 12 MI instructions, before tile j + 1's accumulators start
    auto compute = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    accm[i][j] = mma16(af[slot][i][0][s], wf[slot][j][0][s], accm[i][j]);
                    accc[i][j] = mma16(af[slot][i][0][s], wf[slot][j][1][s], accc[i][j]);
                    accc[i][j] = mma16(af[slot][i][1][s], wf[slot][j][0][s], accc[i][j]);
                }
    };

    if constexpr (LNF) {
        // ---- LayerNorm in the wave: lane (row q = lane / 16 of each 4-row group, columns 4 (lane % 16) .. + 3 of every 64-k block);
        // the normalised rows go to LDS in fragment order (+16 B per half fragment: conflict-free 8-byte stores).  The staging area
        // shares its LDS with the partial tiles of the reduce (barrier between the last fragment read and the first partial store).
        constexpr int HS = 528, SS = 2 * HS, KBS = NPL * 4 * SS;            // bytes
        unsigned char* alds = smem;                                          // [MI row blocks][NW k-blocks][NPL][4 steps][2 halves][32 rows x 16 B (+16)]
        float* gb = reinterpret_cast<float*>(smem + (size_t)MI * NW * KBS); // gamma | beta, K floats each
        if (tid < a.K / 2) reinterpret_cast<f32x4*>(gb)[tid] = gbv;
        __syncthreads();
        const float invK = 1.0f / (float)(64 * NW);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sx = 0.f;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) sx += (xv[g][kb][0] + xv[g][kb][1]) + (xv[g][kb][2] + xv[g][kb][3]);
            sx = row16_sum(sx);
            const float mean = sx * invK;
            float q = 0.f;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[g][kb][e] -= mean; q = fmaf(xv[g][kb][e], xv[g][kb][e], q); }
            q = row16_sum(q);
            const float rstd = 1.0f / sqrtf(q * invK + a.ln_eps);
            const int rt = RW * wid + 4 * g + (lane >> 4);                   // row of the workgroup's tile
            const int r = rt & 31;
            const int c = lane & 15;                                         // columns 4 c .. 4 c + 3 of the block
            unsigned char* dst = alds + (size_t)(rt >> 5) * NW * KBS + (c >> 2) * SS + ((c >> 1) & 1) * HS + r * 16 + (c & 1) * 8;
#pragma unroll
            for (int kb = 0; kb < NW; ++kb) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gb + 64 * kb + 4 * c);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(gb + a.K + 64 * kb + 4 * c);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = xv[g][kb][e] * rstd * gm[e] + bt[e];
                uint32_t h01, h23, l01, l23;
                split_f16_x4(y[0], y[1], y[2], y[3], h01, h23, l01, l23);
                *reinterpret_cast<u32x2*>(dst + (size_t)kb * KBS) = u32x2{h01, h23};
                *reinterpret_cast<u32x2*>(dst + (size_t)kb * KBS + 4 * SS) = u32x2{l01, l23};
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned char* src = alds + ((size_t)i * NW + wid) * KBS + (lane >> 5) * HS + (lane & 31) * 16;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int s = 0; s < 4; ++s) af[0][i][p][s] = *reinterpret_cast<const u32x4*>(src + (p * 4 + s) * SS);
        }
        compute(0);
        __syncthreads();   // every wave has its A fragments in registers: the staging area becomes the partial tiles
    } else {
#pragma unroll
        for (int i = 0; i < NKB; ++i) {
            compute(i % RING);
            if (i + RING < NKB) {
                load_a(i % RING, wid + NW * (i + RING));
                load_w(i % RING, wid + NW * (i + RING));
            }
        }
    }

    // ---- the NW partial tiles meet in LDS (accumulator element e of lane (r, h): row 8 (e >> 2) + 4 h + (e & 3), column r)
    {
        const int r = lane & 31, h = lane >> 5;
        float* mine = red + (size_t)wid * TM * PITCH;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    mine[(32 * i + 8 * (e >> 2) + 4 * h + (e & 3)) * PITCH + 32 * j + r] = accm[i][j][e] + accc[i][j][e] * (1.0f / 2048.0f);
    }
    __syncthreads();
    // ---- row-major phase: thread -> (row, 4 columns); sums the waves in order (bit-reproducible), bias, GELU / residual; f32
    // rows go out here (whole 128-byte lines), the finished values return to slab 0 for the fragment-order phase
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        const int idx = tid + q * NT;
        if (idx < ITEMS) {
            const int rl = idx / C4, c4 = (idx % C4) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(red + rl * PITCH + c4);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(red + ((size_t)w * TM + rl) * PITCH + c4);
            v = v * a.alpha + pre_b;
            if constexpr (EPI == FR_EPI_GELU) {
                const genie_f2 g0 = gelu_erf_fast2(genie_f2{v[0], v[1]}), g1 = gelu_erf_fast2(genie_f2{v[2], v[3]});
                v = f32x4{g0[0], g0[1], g1[0], g1[1]};
            }
            if constexpr (EPI == FR_EPI_QKVS) {   // (every pass is full: ITEMS % NT == 0 for 64-column tiles, so all lanes of a head take part)
                if (qkn) v = head_layer_norm(v, a.Dh, pre_qg, pre_qb);
            }
            if constexpr (EPI == FR_EPI_RES) v += pre_r[q];
            if constexpr (EPI == FR_EPI_F32 || EPI == FR_EPI_RES) {
                const long row = m0 + rl;
                float* dst = a.Cf + (size_t)(row / a.rows_per_batch) * a.strideC + (size_t)(row % a.rows_per_batch) * a.ldc + n0 + c4;
                *reinterpret_cast<f32x4*>(dst) = v;
            }
            if constexpr (EPI != FR_EPI_F32) *reinterpret_cast<f32x4*>(red + rl * PITCH + c4) = v;
        }
    }
    if constexpr (EPI == FR_EPI_F32) return;
    if (EPI == FR_EPI_RES && !a.C16) return;
    __syncthreads();
    if constexpr (EPI == FR_EPI_RES || EPI == FR_EPI_GELU) {
        // ---- fragment-order phase: thread -> (row block, step sl of the tile's columns, lane of the fragment): 8 values of one row
        const int KBo = a.N / 64;
        for (int idx = tid; idx < MI * (TN / 16) * 64; idx += NT) {
            const int fl = idx & 63, r = fl & 31, h = fl >> 5;
            const int sl = (idx >> 6) % (TN / 16), i = (idx >> 6) / (TN / 16);
            const float* src = red + (32 * i + r) * PITCH + 16 * sl + 8 * h;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            u32x4 hi, lo;
            split8(v, hi, lo);
            const int col = n0 + 16 * sl;
            uint16_t* dst = a.C16 + fr_frag(rb0 + i, col >> 6, KBo, 0, (col & 63) >> 4) + fl * 8;
            *reinterpret_cast<u32x4*>(dst) = hi;
            *reinterpret_cast<u32x4*>(dst + 4 * FR) = lo;
        }
    } else if constexpr (EPI == FR_EPI_QKVS) {
        // ---- the spatial attention's operand planes (qkvs_store): item = (row block, 32-column half of the tile, fragment, lane)
        static_assert(EPI != FR_EPI_QKVS || TN == 64, "attention operand planes: 64-column tiles");
        for (int idx = tid; idx < MI * 256; idx += NT) {
            const int i = idx >> 8, j = (idx >> 7) & 1, sub = (idx >> 6) & 1, fl = idx & 63;
            qkvs_store(a, red + (size_t)32 * i * PITCH + 32 * j, PITCH, rb0 + i, n0 + 32 * j, sub, fl);
        }
    }
}

// Wave-local epilogue of the LDS-tiled kernels: a wave's (32 MI) x (32 NJ) accumulator tile -> its LDS tile -> whole rows out (bias,
// GELU / residual, f32 rows, fragment-ordered operand copy, or the spatial attention's operand planes).  rbw0: the wave's first
// 32-row block, n0: its first column.  `tile`: [32 MI][32 NJ + 4] floats of LDS owned by this wave.
template <int MI, int NJ, int EPI>
__device__ __forceinline__ void frm_epilogue(const FrGemmArgs& a, const f32x16 (&accm)[MI][NJ], const f32x16 (&accc)[MI][NJ], float* tile,
                                             int rb0, int n0, int lane) {
    constexpr int TNW = 32 * NJ, PITCH = TNW + 4;
    // ---- epilogue, wave-local: accumulators -> [32 MI][PITCH] tile (element e of lane (r, h): row 8 (e >> 2) + 4 h + (e & 3), column r)
    {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    tile[(32 * i + 8 * (e >> 2) + 4 * h + (e & 3)) * PITCH + 32 * j + r] = accm[i][j][e] + accc[i][j][e] * (1.0f / 2048.0f);
    }
    __builtin_amdgcn_wave_barrier();
    const int m0 = rb0 * 32;
    // row-major phase: lane -> (row of a group of 64 / C4 rows, 4 columns): bias, GELU / residual, f32 rows out, values back to the tile
    constexpr int C4 = TNW / 4, RPI = 64 / C4;
    {
        const int c4 = (lane % C4) * 4, rl0 = lane / C4;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + c4);
        f32x4 qg = f32x4{0.f, 0.f, 0.f, 0.f}, qb = qg;
        // qk-norm: the wave's tile holds whole heads (the launcher picks 64-column wave tiles for head_dim 64)
        const bool qkn = EPI == FR_EPI_QKVS && a.qn_g != nullptr && n0 < 2 * a.H * a.Dh;
        if (qkn && TNW % a.Dh != 0) __builtin_trap();   // (launch_frm_any never pairs 32-column wave tiles with heads of 64 under qk-norm)
        if (qkn) {
            qg = *reinterpret_cast<const f32x4*>(a.qn_g + c4 % a.Dh);
            qb = *reinterpret_cast<const f32x4*>(a.qn_b + c4 % a.Dh);
        }
#pragma unroll 4
        for (int it = 0; it < 32 * MI / RPI; ++it) {
            const int rl = it * RPI + rl0;
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + rl * PITCH + c4) * a.alpha + bv;
            if constexpr (EPI == FR_EPI_GELU) {
                const genie_f2 g0 = gelu_erf_fast2(genie_f2{v[0], v[1]}), g1 = gelu_erf_fast2(genie_f2{v[2], v[3]});
                v = f32x4{g0[0], g0[1], g1[0], g1[1]};
            }
            if constexpr (EPI == FR_EPI_QKVS) {
                if (qkn) v = head_layer_norm(v, a.Dh, qg, qb);
            }
            if constexpr (EPI == FR_EPI_F32 || EPI == FR_EPI_RES) {
                const long row = m0 + rl;
                float* dst = a.Cf + (size_t)(row / a.rows_per_batch) * a.strideC + (size_t)(row % a.rows_per_batch) * a.ldc + n0 + c4;
                if constexpr (EPI == FR_EPI_RES) v += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = v;
            }
            if constexpr (EPI != FR_EPI_F32) *reinterpret_cast<f32x4*>(tile + rl * PITCH + c4) = v;
        }
    }
    if constexpr (EPI == FR_EPI_F32) return;
    if (EPI == FR_EPI_RES && !a.C16) return;
    __builtin_amdgcn_wave_barrier();
    const int fl = lane, r = fl & 31, h = fl >> 5;
    if constexpr (EPI == FR_EPI_RES || EPI == FR_EPI_GELU) {
        const int KBo = a.N / 64;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int sl = 0; sl < TNW / 16; ++sl) {
                const float* src = tile + (32 * i + r) * PITCH + 16 * sl + 8 * h;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                u32x4 hi, lo;
                split8(v, hi, lo);
                const int col = n0 + 16 * sl;
                uint16_t* dst = a.C16 + fr_frag(rb0 + i, col >> 6, KBo, 0, (col & 63) >> 4) + fl * 8;
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + 4 * FR) = lo;
            }
    } else if constexpr (EPI == FR_EPI_QKVS) {
        // (qkvs_store: every 32 x 32 sub-tile of the wave's tile fills two fragments of its (q | k | v, head) slice)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
                    qkvs_store(a, tile + (size_t)32 * i * PITCH + 32 * j, PITCH, rb0 + i, n0 + 32 * j, sub, fl);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same Linear for passes of >= 1024 rows (4+ clips per pass, the 8-frame prompt of generate): there the register-direct
// kernel above re-reads every weight fragment once per 32-64 rows and runs in several rounds, and the row-major mid-size kernels
// (gemm16_nt / gemm16_v2: two LDS stages of 64 k) wait out one memory round trip per stage (8 stages at K = 512: 24-37 us for
// 3-6 GFLOP, profiles/r05a_gen{8,16}_kernel_stats.txt).  Here: workgroup tile (32 WM MI) x (32 WN NJ), NST stages of 32 k in an
// LDS ring filled by LDS-DMA -- a stage's operand pieces ARE fragments (1 KB contiguous in global memory, lane-linear in LDS: no
// swizzle, conflict-free ds_read_b128), NST - 1 stages in flight, one barrier per stage, counted vmcnt.  Wave (wm, wn) owns a
// (32 MI) x (32 NJ) tile; epilogues as above but wave-local (no split-K reduce): accumulators -> the wave's LDS tile -> whole rows.
// ------------------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MI, int NJ, int NST, int EPI>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm16_frm_kernel(const FrGemmArgs a) {
    constexpr int NWV = WM * WN, RBA = WM * MI, RBW = WN * NJ, SLOTS = RBA + RBW;
    constexpr int STAGE_B = SLOTS * 4096;                 // [slot][plane][step of the half block] x 1 KB
    constexpr int PIECES = SLOTS * 4, PPW = PIECES / NWV;
    static_assert(PIECES % NWV == 0, "a stage's pieces divide evenly over the waves");
    constexpr int TNW = 32 * NJ, PITCH = TNW + 4;         // a wave's epilogue tile: [32 MI][PITCH] floats
    static_assert(NWV * 32 * MI * PITCH * 4 <= NST * STAGE_B, "the epilogue tiles live in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int nt = a.N / (32 * RBW);
    const int rbA0 = (blockIdx.x / nt) * RBA, rbW0 = (blockIdx.x % nt) * RBW;
    const int KB = a.K / 64, nk = a.K / 32;

    // ---- this wave's share of a stage: PPW pieces (slot, plane, step); global fragment base of half block 0
    const uint16_t* gp[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid * PPW + i, slot = q >> 2, pl = (q >> 1) & 1, s2 = q & 1;
        if (slot < RBA) {
            const int rb = rbA0 + slot;
            const long arb = ((long)(rb / a.a_group) * a.a_mul + a.a_off) * a.a_group + rb % a.a_group;
            gp[i] = a.A + fr_frag(arb, 0, KB, pl, s2) + lane * 8;
        } else {
            gp[i] = a.W + fr_frag(rbW0 + slot - RBA, 0, KB, pl, s2) + lane * 8;
        }
    }
    auto issue = [&](int hk) {   // half block hk (32 k) -> ring slot hk % NST
        const size_t koff = ((size_t)(hk >> 1) * NPL * 4 + 2 * (hk & 1)) * FR;
        unsigned char* dst = smem + (size_t)(hk % NST) * STAGE_B + (size_t)wid * PPW * 1024;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp[i] + koff),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);

    f32x16 accm[MI][NJ], accc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[i][j][e] = 0.f; accc[i][j][e] = 0.f; }

    for (int k = 0; k < nk; ++k) {
        // stage k has landed when at most the pieces of the min(NST - 2, nk - 1 - k) younger stages are outstanding
        const int younger = nk - 1 - k < NST - 2 ? nk - 1 - k : NST - 2;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave's pieces of stage k are visible; the slot of stage k - 1 is free
        // the refill of the freed slot, then all 4 (MI + NJ) fragment reads of the stage before the first matrix instruction (the
        // waits then count down in step with the MFMAs instead of draining before every group)
        if (k + NST - 1 < nk) issue(k + NST - 1);
        const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem + (unsigned)(k % NST) * STAGE_B + lane * 16;
        const unsigned aA = sbase + (unsigned)(wm * MI) * 4096, aW = sbase + (unsigned)(RBA + wn * NJ) * 4096;
        u32x4 af[2][MI][NPL], wf[2][NJ][NPL];
        // read order per step: A hi, W hi (-> hi.hi group), W lo' (-> hi.lo'), A lo' (-> lo'.hi)
#define FRM_READ_STEP(S2)                                                                                                  \
        do {                                                                                                               \
            if constexpr (MI >= 1) fr_lds_rd<0 * 4096 + 0 * 2048 + S2 * 1024>(af[S2][0][0], aA);                           \
            if constexpr (MI >= 2) fr_lds_rd<1 * 4096 + 0 * 2048 + S2 * 1024>(af[S2][MI >= 2 ? 1 : 0][0], aA);             \
            if constexpr (NJ >= 1) fr_lds_rd<0 * 4096 + 0 * 2048 + S2 * 1024>(wf[S2][0][0], aW);                           \
            if constexpr (NJ >= 2) fr_lds_rd<1 * 4096 + 0 * 2048 + S2 * 1024>(wf[S2][NJ >= 2 ? 1 : 0][0], aW);             \
            if constexpr (NJ >= 1) fr_lds_rd<0 * 4096 + 1 * 2048 + S2 * 1024>(wf[S2][0][1], aW);                           \
            if constexpr (NJ >= 2) fr_lds_rd<1 * 4096 + 1 * 2048 + S2 * 1024>(wf[S2][NJ >= 2 ? 1 : 0][1], aW);             \
            if constexpr (MI >= 1) fr_lds_rd<0 * 4096 + 1 * 2048 + S2 * 1024>(af[S2][0][1], aA);                           \
            if constexpr (MI >= 2) fr_lds_rd<1 * 4096 + 1 * 2048 + S2 * 1024>(af[S2][MI >= 2 ? 1 : 0][1], aA);             \
        } while (0)
        static_assert(MI <= 2 && NJ <= 2, "fragment read macro covers up to 2 x 2 tiles per wave");
        FRM_READ_STEP(0);
        FRM_READ_STEP(1);
#undef FRM_READ_STEP
        __builtin_amdgcn_s_setprio(1);
        constexpr int R = 2 * (MI + NJ);   // reads per step
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // hi.hi needs the step's first MI + NJ reads
            if (s2 == 0) fr_lds_wait<2 * R - (MI + NJ)>(wf[0][NJ - 1][0]); else fr_lds_wait<R - (MI + NJ)>(wf[1][NJ - 1][0]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accm[i][j] = mma16(af[s2][i][0], wf[s2][j][0], accm[i][j]);
            __builtin_amdgcn_sched_barrier(0);   // (the matrix instructions stay between their waits)
            if (s2 == 0) fr_lds_wait<2 * R - (MI + 2 * NJ)>(wf[0][NJ - 1][1]); else fr_lds_wait<R - (MI + 2 * NJ)>(wf[1][NJ - 1][1]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accc[i][j] = mma16(af[s2][i][0], wf[s2][j][1], accc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (s2 == 0) fr_lds_wait<R>(af[0][MI - 1][1]); else fr_lds_wait<0>(af[1][MI - 1][1]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accc[i][j] = mma16(af[s2][i][1], wf[s2][j][0], accc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();   // the ring is free: it becomes the waves' epilogue tiles

    // ---- epilogue, wave-local
    frm_epilogue<MI, NJ, EPI>(a, accm, accc, reinterpret_cast<float*>(smem) + (size_t)wid * 32 * MI * PITCH, rbA0 + wm * MI,
                              (rbW0 + wn * NJ) * 32, lane);
}

// LayerNorm of f32 rows -> fragment-ordered split operand (the mid-size passes; below 1024 rows the Linear does it in its prologue):
// workgroup = one 32-row block, NW = K / 64 waves, wave w normalises rows 4 w' .. and, after the barrier, writes k-block w.
template <int NW>
__global__ __launch_bounds__(NW * 64) void ln_fr_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, float eps, uint16_t* __restrict__ out16) {
    constexpr int K = 64 * NW, RW = 32 / NW, G = RW / 4 > 0 ? RW / 4 : 1;
    static_assert(RW % 4 == 0, "4 rows per instruction");
    constexpr int HS = 528, SS = 2 * HS, KBS = NPL * 4 * SS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* gb = reinterpret_cast<float*>(smem + (size_t)NW * KBS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long rb = blockIdx.x;
    f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < K / 4) gbv = reinterpret_cast<const f32x4*>(ln_g)[tid];
    else if (tid < K / 2) gbv = reinterpret_cast<const f32x4*>(ln_b)[tid - K / 4];
    f32x4 xv[G][NW];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float* xr = X + (size_t)(rb * 32 + RW * wid + 4 * g + (lane >> 4)) * ldx + 4 * (lane & 15);
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) xv[g][kb] = *reinterpret_cast<const f32x4*>(xr + 64 * kb);
    }
    FR_PIN_LOADS();
    if (tid < K / 2) reinterpret_cast<f32x4*>(gb)[tid] = gbv;
    __syncthreads();
    const float invK = 1.0f / (float)K;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float sx = 0.f;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) sx += (xv[g][kb][0] + xv[g][kb][1]) + (xv[g][kb][2] + xv[g][kb][3]);
        sx = row16_sum(sx);
        const float mean = sx * invK;
        float q = 0.f;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[g][kb][e] -= mean; q = fmaf(xv[g][kb][e], xv[g][kb][e], q); }
        q = row16_sum(q);
        const float rstd = 1.0f / sqrtf(q * invK + eps);
        const int r = RW * wid + 4 * g + (lane >> 4);
        const int c = lane & 15;
        unsigned char* dst = smem + (c >> 2) * SS + ((c >> 1) & 1) * HS + r * 16 + (c & 1) * 8;
#pragma unroll
        for (int kb = 0; kb < NW; ++kb) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gb + 64 * kb + 4 * c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(gb + K + 64 * kb + 4 * c);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = xv[g][kb][e] * rstd * gm[e] + bt[e];
            uint32_t h01, h23, l01, l23;
            split_f16_x4(y[0], y[1], y[2], y[3], h01, h23, l01, l23);
            *reinterpret_cast<u32x2*>(dst + (size_t)kb * KBS) = u32x2{h01, h23};
            *reinterpret_cast<u32x2*>(dst + (size_t)kb * KBS + 4 * SS) = u32x2{l01, l23};
        }
    }
    __syncthreads();
    const unsigned char* src = smem + (size_t)wid * KBS + (lane >> 5) * HS + (lane & 31) * 16;
    uint16_t* o = out16 + fr_frag(rb, wid, NW, 0, 0) + lane * 8;
#pragma unroll
    for (int ps = 0; ps < NPL * 4; ++ps) *reinterpret_cast<u32x4*>(o + ps * FR) = *reinterpret_cast<const u32x4*>(src + ps * SS);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Spatial attention of a frame pass (attention.py:36-61 over the S = 256 positions of one frame, head_dim DH = 64 or 32): workgroup =
// (sequence, head, block of 32 queries), wave kt = the 32-key tile kt.  All operand fragments of a wave (Q block, K tile, V^T tile;
// hi and lo' planes: 24 at DH 64) are requested at once, nothing is converted or transposed here (the qkv Linear's epilogue did
// that, qkvs_store).  S^T = K Q^T so a lane holds 16 scores of ITS query; exp2 softmax per tile; the eight partial
// (max, sum, O) triples merge through LDS in wave order; O leaves in the fragment order of the out-projection's A operand.
// ------------------------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_spatial_fr_kernel(const uint16_t* __restrict__ qkvs, uint16_t* __restrict__ out16,
                                                                  int H) {
    constexpr int KS = DH / 16, DT = DH / 32, PITCH = DH + 4, BLK = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem);                               // [8 waves][32 queries][PITCH]
    float* stat = part + 8 * 32 * PITCH;                                        // [8 waves][32 queries][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int kt = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long seq = blockIdx.x;
    const int head = blockIdx.y, qb = blockIdx.z;
    const size_t unit = (size_t)BLK * NPL * KS * FR;
    const uint16_t* base = qkvs + (size_t)(seq * H + head) * 3 * unit + lane * 8;
    u32x4 qf[NPL][KS], kf[NPL][KS], vf[DT][2][NPL];
    {
        const uint16_t* qp = base + (size_t)qb * NPL * KS * FR;
        const uint16_t* kp = base + unit + (size_t)kt * NPL * KS * FR;
        const uint16_t* vp = base + 2 * unit + (size_t)kt * DT * 2 * NPL * FR;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kf[p][s] = *reinterpret_cast<const u32x4*>(kp + (p * KS + s) * FR);
                qf[p][s] = *reinterpret_cast<const u32x4*>(qp + (p * KS + s) * FR);
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < NPL; ++p) vf[dt][m][p] = *reinterpret_cast<const u32x4*>(vp + ((dt * 2 + m) * NPL + p) * FR);
    }
    FR_PIN_LOADS();
    f32x16 a0, c0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; c0[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        a0 = mma16(kf[0][s], qf[0][s], a0);
        c0 = mma16(kf[0][s], qf[1][s], c0);
        c0 = mma16(kf[1][s], qf[0][s], c0);
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
    if (h == 0) { stat[(kt * 32 + r) * 2] = mw; stat[(kt * 32 + r) * 2 + 1] = lw; }
    f32x16 oa[DT], oc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oa[dt][e] = 0.f; oc[dt][e] = 0.f; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        u32x4 ph, pl;
        split8(p + 8 * m, ph, pl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oa[dt] = mma16(ph, vf[dt][m][0], oa[dt]);
            oc[dt] = mma16(ph, vf[dt][m][1], oc[dt]);
            oc[dt] = mma16(pl, vf[dt][m][0], oc[dt]);
        }
    }
    float* op = part + (size_t)kt * 32 * PITCH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            op[((e & 3) + 8 * (e >> 2) + 4 * h) * PITCH + dt * 32 + r] = oa[dt][e] + oc[dt][e] * (1.0f / 2048.0f);
    __syncthreads();
    // ---- merge like an online softmax, waves in order; thread -> (step, lane of the output fragment): 8 features of one query
    if (tid < KS * 64) {
        const int sl = tid >> 6, fl = tid & 63, q = fl & 31, hh = fl >> 5;
        float mg = -INFINITY;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) mg = fmaxf(mg, stat[(w8 * 32 + q) * 2]);
        float l = 0.f;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {
            const float sc8 = __builtin_amdgcn_exp2f(stat[(w8 * 32 + q) * 2] - mg);
            l += stat[(w8 * 32 + q) * 2 + 1] * sc8;
            const float* src = part + ((size_t)w8 * 32 + q) * PITCH + 16 * sl + 8 * hh;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += t0[e] * sc8; o[4 + e] += t1[e] * sc8; }
        }
        const float invl = 1.0f / l;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= invl;
        u32x4 hi, lo;
        split8(o, hi, lo);
        // the head's features are columns head * DH .. of the out-projection's A operand (K = H * DH, 64-k blocks)
        const int col = head * DH + 16 * sl;
        uint16_t* dst = out16 + fr_frag(seq * BLK + qb, col >> 6, (H * DH) >> 6, 0, (col & 63) >> 4) + fl * 8;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 4 * FR) = lo;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Temporal decode attention of a frame pass (attention.py:36-61 with the causal mask, one query frame per row): one wave per
// (clip, frame of the pass, position, head).  The query frame t = t0 + f attends cache slots 0..t of its position; the cache
// is the (B, T, S, 3 d) f32 qkv of the earlier passes, slot t written by this pass's qkv Linear.  Whole head slices per request
// (DH floats = LPF = DH / 4 lanes x 16 bytes): an instruction fetches FPI = 64 / LPF frames (lane = (frame, 4 features)), so K and
// V are 16 / FPI loads each; scores are 4-feature partial dot products reduced over the LPF lanes of a frame, the softmax runs over
// the register copies and the lane groups, P.V reduces over the groups.  head_dim 64 / 32, T <= 16; qk-norm (qn_g != NULL): the raw cached q / k
// slices go through the per-head LayerNorm on read (head_layer_norm: a frame's head slice = the LPF lanes of its group).
// ------------------------------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_temporal_fr_kernel(const float* __restrict__ cache, uint16_t* __restrict__ out16,
                                                               long n_items, int T, int S, int t0, int nf, int d, int H, float scale,
                                                               const float* __restrict__ qn_g, const float* __restrict__ qn_b) {
    constexpr int LPF = DH / 4, FPI = 64 / LPF, NI = 16 / FPI;     // lanes per frame, frames per instruction, instructions
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int head = (int)(item % H);
    const long row = item / H;                       // (b, f, s) of the dense pass layout
    const long bf = row / S;
    const int s = (int)(row - bf * S);
    const long b = bf / nf;
    const int f = (int)(bf - b * nf), t = t0 + f;
    const int g = lane / LPF, c = lane % LPF;
    const size_t tok = (size_t)S * 3 * d;
    const float* hb = cache + ((size_t)(b * T) * S + s) * 3 * d + head * DH + 4 * c;
    f32x4 qv = *reinterpret_cast<const f32x4*>(hb + (size_t)t * tok);
    f32x4 kv[NI], vv[NI];
    f32x4 qg = f32x4{0.f, 0.f, 0.f, 0.f}, qb = qg;
    if (qn_g) { qg = *reinterpret_cast<const f32x4*>(qn_g + 4 * c); qb = *reinterpret_cast<const f32x4*>(qn_b + 4 * c); }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = FPI * i + g;
        const float* src = hb + (size_t)(j <= t ? j : t) * tok;    // (frames past t re-read frame t: their probability is 0)
        kv[i] = *reinterpret_cast<const f32x4*>(src + d);
        vv[i] = *reinterpret_cast<const f32x4*>(src + 2 * d);
    }
    FR_PIN_LOADS();
    if (qn_g) {   // qk-norm (attention.py:42-47): the cache holds the raw q and k; a frame's head slice = the LPF lanes of its group
        qv = head_layer_norm(qv, DH, qg, qb);
#pragma unroll
        for (int i = 0; i < NI; ++i) kv[i] = head_layer_norm(kv[i], DH, qg, qb);
    }
    float sc[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float part = fmaf(qv[3], kv[i][3], fmaf(qv[2], kv[i][2], fmaf(qv[1], kv[i][1], qv[0] * kv[i][0])));
        if constexpr (LPF == 16) part = row16_sum(part);
        else {
#pragma unroll
            for (int o = 1; o < LPF; o <<= 1) part += __shfl_xor(part, o);
        }
        sc[i] = (FPI * i + g <= t) ? part * scale : -INFINITY;
        mx = fmaxf(mx, sc[i]);
    }
#pragma unroll
    for (int o = LPF; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { sc[i] = (FPI * i + g <= t) ? expf(sc[i] - mx) : 0.f; sum += sc[i]; }
#pragma unroll
    for (int o = LPF; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) o += vv[i] * (sc[i] * inv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int x = LPF; x < 64; x <<= 1) o[e] += __shfl_xor(o[e], x);
    }
    if (g != 0) return;
    // lane c: features 4 c .. 4 c + 3 of the head = columns col .. col + 3 of the out-projection's A operand: 64-k block col / 64,
    // step (col % 64) / 16, half (col % 16) / 8, elements col % 8 .. + 3 of the lane's 16-byte piece
    uint32_t h01, h23, l01, l23;
    split_f16_x4(o[0], o[1], o[2], o[3], h01, h23, l01, l23);
    const int col = head * DH + 4 * c;
    uint16_t* dst = out16 + fr_frag(row >> 5, col >> 6, (H * DH) >> 6, 0, (col & 63) >> 4) + (32 * ((col >> 3) & 1) + (int)(row & 31)) * 8 + (col & 7);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
    *reinterpret_cast<u32x2*>(dst + 4 * FR) = u32x2{l01, l23};
}

// ------------------------------------------------------------------------------------------------------------------------------
// ONE-frame passes below 2,048 rows: the temporal qkv Linear AND the temporal decode attention in one launch (round 6).
// Workgroup = (32-row block of the pass, head h): it computes q_h | k_h | v_h of its rows -- three phases of the register-direct
// split-K Linear above (same k-block per wave, same wave order in the reduce, same bias add: the cache rows it writes carry the bits
// gemm16_fr_kernel<NW, ., DH / 32, 1, FR_EPI_F32> writes) -- and then attends: rows 4 w .. 4 w + 3 go to wave w with the lane mapping and
// the arithmetic of attn_temporal_fr_kernel (lane (g, c): frames FPI i + g, features 4 c .. + 3), q and the frame-t slices of k, v
// from the workgroup's LDS tiles, slots j < t from the cache (requested right behind the last phase's matrix instructions, so their
// round trip hides under that phase's reduce).  A workgroup streams 3 x DH x K x 4 B of weights (384 KB at d = 512) instead of 128 KB
// per workgroup of the plain Linear; what it saves is one launch + its boundary and the attention kernel's own memory round trip.
// ------------------------------------------------------------------------------------------------------------------------------
struct FrQkvtArgs {
    const uint16_t* A;       // fragment-ordered operand copy of x: (M / 32) row blocks x (K / 64) k-blocks
    const uint16_t* W;       // fragment-ordered temporal qkv weight (3 d, K)
    const float* bias;       // (3 d) or NULL
    float* cache;            // this layer's (B, T, S, 3 d) f32 cache slice
    uint16_t* out16;         // attention output, fragment order (the out-projection's A operand)
    const float* qn_g;       // qk-norm affine (DH floats each) or NULL
    const float* qn_b;
    int K, H, S, T, t;       // contraction, heads, rows per frame, cache slots per clip, the frame of this pass
    float scale;
};

template <int NW, int DH>
__global__ __launch_bounds__(NW * 64, 1) __attribute__((amdgpu_waves_per_eu(NW >= 4 ? NW / 4 : 1, NW >= 4 ? NW / 4 : 1)))
void qkvt_attn_fr_kernel(const FrQkvtArgs a) {
    constexpr int NT = NW * 64, NJ = DH / 32, TN = DH, PITCH = TN + 4, C4 = TN / 4, ITEMS = 32 * C4, PASSES = (ITEMS + NT - 1) / NT;
    constexpr int LPF = DH / 4, FPI = 64 / LPF, NI = 16 / FPI, RPW = 32 / NW;   // attention: lanes per frame, frames per instruction, instructions; rows per wave
    static_assert(NT % C4 == 0, "a thread keeps its columns over the passes of the row-major phase");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                       // [NW][32][PITCH] partial tiles
    float* til = red + (size_t)NW * 32 * PITCH;                        // [3][32][PITCH]: q | k | v of the rows, this head
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = blockIdx.x, head = blockIdx.y;
    const int KB = a.K / 64, d = a.H * DH;
    asm volatile("" :: "s"(a.A), "s"(a.W), "s"(a.bias), "s"(a.cache), "s"(a.K), "s"(a.H), "s"(a.S), "s"(a.T), "s"(a.t));
    // ---- requests: biases of the three phases, A fragments of this wave's k-block, the weight fragments of q and k
    f32x4 pre_b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pre_b[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) pre_b[p] = *reinterpret_cast<const f32x4*>(a.bias + p * d + head * DH + (tid % C4) * 4);
    }
    u32x4 af[NPL][4], wf[2][NJ][NPL][4];
    {
        const uint16_t* src = a.A + fr_frag(rb, wid, KB, 0, 0) + lane * 8;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[p][s] = *reinterpret_cast<const u32x4*>(src + (p * 4 + s) * FR);
    }
    auto load_w = [&](int slot, int ph) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint16_t* src = a.W + fr_frag((long)(ph * d + head * DH) / 32 + j, wid, KB, 0, 0) + lane * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int s = 0; s < 4; ++s) wf[slot][j][p][s] = *reinterpret_cast<const u32x4*>(src + (p * 4 + s) * FR);
        }
    };
    load_w(0, 0);
    load_w(1, 1);
    FR_PIN_LOADS();
    // the pass's rows are (clip b, frame t, position s): row block rb = b * (S / 32) + s / 32
    const int blocks = a.S / 32;
    const long b = rb / blocks;
    const int s0 = (rb % blocks) * 32;
    const size_t tok = (size_t)a.S * 3 * d;                                     // one cache slot of a clip
    float* const slot_t = a.cache + ((size_t)(b * a.T + a.t) * a.S + s0) * 3 * d;   // row s0 of slot t
    f32x4 kv[RPW][NI], vv[RPW][NI];
    const int g = lane / LPF, c = lane % LPF;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        f32x16 accm[NJ], accc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accm[j][e] = 0.f; accc[j][e] = 0.f; }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                accm[j] = mma16(af[0][s], wf[ph & 1][j][0][s], accm[j]);
                accc[j] = mma16(af[0][s], wf[ph & 1][j][1][s], accc[j]);
                accc[j] = mma16(af[1][s], wf[ph & 1][j][0][s], accc[j]);
            }
        if (ph == 0) { load_w(0, 2); FR_PIN_LOADS(); }
        if (ph == 2) {
            // the cached k, v slices of this wave's rows (slots j < t of the same clip and position), one head slice per frame
#pragma unroll
            for (int r4 = 0; r4 < RPW; ++r4) {
                const float* hb = a.cache + ((size_t)(b * a.T) * a.S + s0 + RPW * wid + r4) * 3 * d + head * DH + 4 * c;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = FPI * i + g;
                    kv[r4][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    vv[r4][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (j < a.t) {
                        kv[r4][i] = *reinterpret_cast<const f32x4*>(hb + (size_t)j * tok + d);
                        vv[r4][i] = *reinterpret_cast<const f32x4*>(hb + (size_t)j * tok + 2 * d);
                    }
                }
            }
            FR_PIN_LOADS();
        }
        {   // partial tile (accumulator element e of lane (r, h): row 8 (e >> 2) + 4 h + (e & 3), column r)
            const int r = lane & 31, h = lane >> 5;
            float* mine = red + (size_t)wid * 32 * PITCH;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[(8 * (e >> 2) + 4 * h + (e & 3)) * PITCH + 32 * j + r] = accm[j][e] + accc[j][e] * (1.0f / 2048.0f);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int idx = tid + q * NT;
            if (idx < ITEMS) {
                const int rl = idx / C4, c4 = (idx % C4) * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(red + rl * PITCH + c4);
#pragma unroll
                for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 32 + rl) * PITCH + c4);
                v = v * 1.0f + pre_b[ph];
                *reinterpret_cast<f32x4*>(slot_t + (size_t)rl * 3 * d + ph * d + head * DH + c4) = v;
                *reinterpret_cast<f32x4*>(til + ((size_t)ph * 32 + rl) * PITCH + c4) = v;
            }
        }
        __syncthreads();
    }
    // ---- decode attention of rows RPW wid .. (attn_temporal_fr_kernel's arithmetic, lane for lane)
    f32x4 qg = f32x4{0.f, 0.f, 0.f, 0.f}, qb = qg;
    if (a.qn_g) { qg = *reinterpret_cast<const f32x4*>(a.qn_g + 4 * c); qb = *reinterpret_cast<const f32x4*>(a.qn_b + 4 * c); }
    const int t = a.t;
#pragma unroll
    for (int r4 = 0; r4 < RPW; ++r4) {
        const int rl = RPW * wid + r4;
        f32x4 qv = *reinterpret_cast<const f32x4*>(til + (size_t)rl * PITCH + 4 * c);
        const f32x4 kt = *reinterpret_cast<const f32x4*>(til + ((size_t)32 + rl) * PITCH + 4 * c);
        const f32x4 vt = *reinterpret_cast<const f32x4*>(til + ((size_t)64 + rl) * PITCH + 4 * c);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (FPI * i + g >= t) { kv[r4][i] = kt; vv[r4][i] = vt; }        // frame t itself (frames past t repeat it: their probability is 0)
        if (a.qn_g) {
            qv = head_layer_norm(qv, DH, qg, qb);
#pragma unroll
            for (int i = 0; i < NI; ++i) kv[r4][i] = head_layer_norm(kv[r4][i], DH, qg, qb);
        }
        float sc[NI];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float part = fmaf(qv[3], kv[r4][i][3], fmaf(qv[2], kv[r4][i][2], fmaf(qv[1], kv[r4][i][1], qv[0] * kv[r4][i][0])));
            if constexpr (LPF == 16) part = row16_sum(part);
            else {
#pragma unroll
                for (int o = 1; o < LPF; o <<= 1) part += __shfl_xor(part, o);
            }
            sc[i] = (FPI * i + g <= t) ? part * a.scale : -INFINITY;
            mx = fmaxf(mx, sc[i]);
        }
#pragma unroll
        for (int o = LPF; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) { sc[i] = (FPI * i + g <= t) ? expf(sc[i] - mx) : 0.f; sum += sc[i]; }
#pragma unroll
        for (int o = LPF; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) o += vv[r4][i] * (sc[i] * inv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int x = LPF; x < 64; x <<= 1) o[e] += __shfl_xor(o[e], x);
        }
        if (g == 0) {
            uint32_t h01, h23, l01, l23;
            split_f16_x4(o[0], o[1], o[2], o[3], h01, h23, l01, l23);
            const int col = head * DH + 4 * c;
            uint16_t* dst = a.out16 + fr_frag(rb, col >> 6, d >> 6, 0, (col & 63) >> 4) + (32 * ((col >> 3) & 1) + rl) * 8 + (col & 7);
            *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
            *reinterpret_cast<u32x2*>(dst + 4 * FR) = u32x2{l01, l23};
        }
    }
}

// f32 rows (M, K) -> fragment-ordered split operand (the activation split: split8).  qk-norm blocks have no LayerNorm in front of their
// spatial qkv Linear (norm1 = Identity, st_transformer.py:44): the first block of a pass reads this copy of the embedded rows, every later
// Linear reads the copy its producer's epilogue wrote.  One thread per 16-byte piece of the hi plane.
__global__ void cast_fr_kernel(const float* __restrict__ X, uint16_t* __restrict__ out16, long M, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KB = K / 64;
    if (i >= (size_t)(M / 32) * KB * 256) return;
    const int fl = (int)(i & 63), s = (int)((i >> 6) & 3);
    const size_t blk = i >> 8;
    const int kb = (int)(blk % KB);
    const long rb = (long)(blk / KB);
    const float* p = X + (size_t)(rb * 32 + (fl & 31)) * K + 64 * kb + 16 * s + 8 * (fl >> 5);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    u32x4 hi, lo;
    split8(v, hi, lo);
    uint16_t* o = out16 + fr_frag(rb, kb, KB, 0, s) + fl * 8;
    *reinterpret_cast<u32x4*>(o) = hi;
    *reinterpret_cast<u32x4*>(o + 4 * FR) = lo;
}

// f32 (N, K) row-major -> fragment order, split f16 planes (the split of genie_pack_split_f16: hi flushed below the f16 normal range)
__global__ void pack_frame_w16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int N, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte piece of the hi plane
    const int KB = K / 64;
    const size_t total = (size_t)(N / 32) * KB * 4 * 64;
    if (i >= total) return;
    const int fl = (int)(i & 63), s = (int)((i >> 6) & 3);
    const size_t blk = i >> 8;
    const int kb = (int)(blk % KB);
    const long rb = (long)(blk / KB);
    const int r = fl & 31, h = fl >> 5;
    const float* p = src + (size_t)(rb * 32 + r) * K + 64 * kb + 16 * s + 8 * h;
    uint16_t hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split_f16(p[e], hi[e], lo[e]);
    uint16_t* o = dst + fr_frag(rb, kb, KB, 0, s) + fl * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = hi[e]; o[4 * FR + e] = lo[e]; }
}

int launch_pack_frame_w16(const float* src, uint16_t* dst, int N, int K, hipStream_t st) {
    GENIE_CHECK_SHAPE(N % 32 == 0 && K % 64 == 0 && N > 0 && K > 0, "pack_frame_w16: (N, K) = (%d, %d) must be multiples of (32, 64)", N, K);
    const size_t total = (size_t)(N / 32) * (K / 64) * 256;
    pack_frame_w16_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src, dst, N, K);
    GENIE_LAUNCH_CHECK("pack_frame_w16");
    return GENIE_OK;
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
namespace {
template <int NW, int MI, int NJ, int NKB, int EPI, bool LNF>
int launch_fr(const FrGemmArgs& a, hipStream_t st) {
    constexpr int TM = 32 * MI, TN = 32 * NJ;
    size_t lds = (size_t)NW * TM * (TN + 4) * 4;
    if (LNF) {   // the LayerNorm staging area shares the LDS of the partial tiles
        const size_t stage = (size_t)MI * NW * NPL * 4 * 1056 + (size_t)2 * a.K * 4;
        lds = lds > stage ? lds : stage;
    }
    static PerDevice<bool> attr_set;
    if (attr_set.needs()) {
        (void)hipFuncSetAttribute((const void*)gemm16_fr_kernel<NW, MI, NJ, NKB, EPI, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.set(true);
    }
    if (a.M % TM || a.N % TN) {   // (no tail handling: a ragged width would leave columns unwritten)
        set_error("gemm16_fr: (M, N) = (%d, %d) must be multiples of the (%d, %d) tile", a.M, a.N, TM, TN);
        return GENIE_E_UNSUPPORTED;
    }
    const unsigned grid = (unsigned)((a.M / TM) * (a.N / TN));
    gemm16_fr_kernel<NW, MI, NJ, NKB, EPI, LNF><<<grid, NW * 64, lds, st>>>(a);
    GENIE_LAUNCH_CHECK("gemm16_fr");
    return GENIE_OK;
}

// dispatch on the width: K = 64 NW (NKB = 1) for the Linears fed by d_model, K = 256 NW (NKB = 4) for fc2; wide2: 64-row
// workgroups for the 64-column tiles of a pass of >= 512 rows (two frames, or two clips)
template <int NJ, int NKB, int EPI, bool LNF>
int launch_fr_w(int nw, const FrGemmArgs& a, hipStream_t st) {
    // (the LayerNorm-fused Linears keep 32-row workgroups: two of them fit a CU since the staging area shares the reduce's LDS, and a
    // 64-row workgroup does twice the LayerNorm work behind one barrier -- same-box A/B: equal at 512 rows, +2-3 % at 1,024,
    // profiles/r05o_ln_mi1_ab.txt)
    const bool wide2 = NJ == 2 && NKB == 1 && a.M >= 512 && a.M % 64 == 0 && !LNF;
    if constexpr (NJ == 2 && NKB == 1) {
        if (wide2) {
            switch (nw) {
                case 8: return launch_fr<8, 2, NJ, NKB, EPI, LNF>(a, st);
                case 4: return launch_fr<4, 2, NJ, NKB, EPI, LNF>(a, st);
                case 2: return launch_fr<2, 2, NJ, NKB, EPI, LNF>(a, st);
                default: break;
            }
        }
    }
    switch (nw) {
        case 8: return launch_fr<8, 1, NJ, NKB, EPI, LNF>(a, st);
        case 4: return launch_fr<4, 1, NJ, NKB, EPI, LNF>(a, st);
        case 2: return launch_fr<2, 1, NJ, NKB, EPI, LNF>(a, st);
        default: break;
    }
    set_error("gemm16_fr: width %d not covered", 64 * nw);
    return GENIE_E_UNSUPPORTED;
}
}  // namespace

namespace {
template <int WM, int WN, int MI, int NJ, int NST, int EPI>
int launch_frm(const FrGemmArgs& a, hipStream_t st) {
    constexpr int BM = 32 * WM * MI, BN = 32 * WN * NJ;
    constexpr size_t lds = (size_t)NST * (WM * MI + WN * NJ) * 4096;
    if (a.M % BM || a.N % BN || a.K % 64) return GENIE_E_UNSUPPORTED;
    static PerDevice<bool> attr_set;
    if (attr_set.needs()) {
        (void)hipFuncSetAttribute((const void*)gemm16_frm_kernel<WM, WN, MI, NJ, NST, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.set(true);
    }
    gemm16_frm_kernel<WM, WN, MI, NJ, NST, EPI><<<(unsigned)((a.M / BM) * (a.N / BN)), 64 * WM * WN, lds, st>>>(a);
    GENIE_LAUNCH_CHECK("gemm16_frm");
    return GENIE_OK;
}
// 128 x 128 tiles where they fill the chip, 128 x 64 for the narrow outputs (N = d_model)
template <int EPI>
int launch_frm_any(const FrGemmArgs& a, hipStream_t st) {
    const long t128 = (long)(a.M / 128) * (a.N / 128);
    // the wide Linears (N >= 1024) from 4,096 rows: 256 x 128 tiles, 64 x 64 per wave -- a 32 k stage is 24 matrix instructions per
    // wave (0.64 us per CU), so the two stages in flight cover twice the round trip that 128 x 128 tiles (12 per stage) cover:
    // qkv 32.6 -> 29.4 us, fc1 38.6 -> 33.2 at 4,096 rows, fc1 73.4 -> 64.9 at 8,192; at 2,048 rows (96 tiles) they lose
    // (17.7 -> 26.6): profiles/r05j_frame_gemm.txt against r05i
    static const int big = study_env("GENIE_FRM_256", 1);
    if constexpr (EPI == FR_EPI_QKVS) {   // qk-norm: a wave's tile must hold whole heads -> 64-column wave tiles (128 x 128 per workgroup of 4 waves)
        if (a.qn_g) return launch_frm<2, 2, 2, 2, 4, EPI>(a, st);
    }
    if (big && a.N >= 1024 && a.N % 128 == 0 && a.M % 256 == 0 && a.M >= 4096) return launch_frm<4, 2, 2, 2, 3, EPI>(a, st);
    if (a.N % 128 == 0 && (EPI == FR_EPI_QKVS || t128 >= 160 || a.N >= 1024)) return launch_frm<2, 4, 2, 1, 4, EPI>(a, st);
    if constexpr (EPI != FR_EPI_QKVS) {
        // the narrow outputs (N = d): 128 x 64 tiles, or -- while those leave CUs idle (fc2 at 2,048 rows: 128 tiles) -- 64 x 64 tiles on
        // 4-wave workgroups, two per CU (64 KB of ring each)
        static const int small = study_env("GENIE_FRM_64", 1);
        const long t64w = (long)(a.M / 128) * (a.N / 64);
        if (small && t64w < 192 && a.M % 64 == 0 && a.N % 64 == 0) return launch_frm<2, 2, 1, 1, 4, EPI>(a, st);
        return launch_frm<4, 2, 1, 1, 5, EPI>(a, st);
    }
    return GENIE_E_UNSUPPORTED;
}
int launch_ln_fr(const float* x, long ldx, const float* g, const float* b, float eps, uint16_t* out16, int M, int K, hipStream_t st) {
    const size_t lds = (size_t)(K / 64) * NPL * 4 * 1056 + (size_t)2 * K * 4;
#define LN_FR(NW_)                                                                                                           \
    case NW_: {                                                                                                              \
        static PerDevice<bool> attr_set;                                                                                     \
        if (attr_set.needs()) { (void)hipFuncSetAttribute((const void*)ln_fr_kernel<NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set.set(true); } \
        ln_fr_kernel<NW_><<<(unsigned)(M / 32), NW_ * 64, lds, st>>>(x, ldx, g, b, eps, out16);                              \
        break;                                                                                                               \
    }
    switch (K / 64) {
        LN_FR(8) LN_FR(4) LN_FR(2)
        default: set_error("ln_fr: width %d not covered", K); return GENIE_E_UNSUPPORTED;
    }
#undef LN_FR
    GENIE_LAUNCH_CHECK("ln_fr");
    return GENIE_OK;
}
}  // namespace

namespace {
template <int NW, int DH>
int launch_qkvt_attn_t(const FrQkvtArgs& q, unsigned row_blocks, unsigned heads, hipStream_t st) {
    constexpr size_t lds = ((size_t)NW * 32 + 3 * 32) * (DH + 4) * 4;
    static PerDevice<bool> attr_set;
    if (attr_set.needs()) {
        (void)hipFuncSetAttribute((const void*)qkvt_attn_fr_kernel<NW, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.set(true);
    }
    qkvt_attn_fr_kernel<NW, DH><<<dim3(row_blocks, heads), NW * 64, lds, st>>>(q);
    GENIE_LAUNCH_CHECK("qkvt_attn_fr");
    return GENIE_OK;
}
int launch_qkvt_attn(int nw, int dh, const FrQkvtArgs& q, unsigned row_blocks, unsigned heads, hipStream_t st) {
    if (dh == 64) {
        if (nw == 8) return launch_qkvt_attn_t<8, 64>(q, row_blocks, heads, st);
        if (nw == 4) return launch_qkvt_attn_t<4, 64>(q, row_blocks, heads, st);
        // (2 waves x 16 rows each: the cached k / v slices of 16 rows do not fit the registers -- 568 bytes of scratch; that width keeps the two launches)
    } else if (dh == 32) {
        if (nw == 8) return launch_qkvt_attn_t<8, 32>(q, row_blocks, heads, st);
        if (nw == 4) return launch_qkvt_attn_t<4, 32>(q, row_blocks, heads, st);
        if (nw == 2) return launch_qkvt_attn_t<2, 32>(q, row_blocks, heads, st);
    }
    set_error("qkvt_attn_fr: width %d / head_dim %d not covered", 64 * nw, dh);
    return GENIE_E_UNSUPPORTED;
}
}  // namespace

bool frame_path_takes(const genie_cfg& c, const genie_layer_weights& lw, long rows) {
    static const int on = study_env("GENIE_FRAME_KERNELS", 1);
    // LayerNorm blocks need norm1 / norm2; qk-norm blocks (norm1 = norm2 = Identity, st_transformer.py:44,67) the two per-head affines
    const bool norms = c.qk_norm ? (lw.spatial.norm_w && lw.spatial.norm_b && lw.temporal.norm_w && lw.temporal.norm_b)
                                 : (lw.norm1_w && lw.norm1_b && lw.norm2_w && lw.norm2_b);
    return on && c.precision == GENIE_PREC_F16X3 && c.S == 256 && (c.head_dim == 64 || c.head_dim == 32) &&
           c.d_model == c.num_heads * c.head_dim &&
           (c.d_model == 512 || c.d_model == 256 || c.d_model == 128) && c.hidden == 4 * c.d_model && rows % 256 == 0 && rows <= 16384 &&
           lw.spatial.frame_w16 && lw.temporal.frame_w16 && lw.mlp_frame_w16 && norms;
}

// One STBlock (st_transformer.py:70-83) of a frame pass: B clips x nf frames (slots t0 .. t0 + nf - 1 of the cache) x S rows.
//   x   f32 rows (the residual stream)            xs  = w.xn : operand copy of x (fragment order)
//   as  = w.aux: attention outputs                 big = w.big: spatial attention operand planes, then the MLP hidden
// rows from which a pass runs the LDS-tiled kernels (gemm16_frm, separate LayerNorm) instead of the register-direct ones
static long frm_min_rows() {
    static const long v = study_env("GENIE_FRM_MIN_ROWS", 2048);
    return v;
}

int st_block_frame_f16x3(const genie_cfg& c, const genie_layer_weights& lw, float* x, Workspace& w, int B, int nf, bool want_xs,
                         hipStream_t st) {
    const int d = c.d_model, S = c.S, H = c.num_heads, hid = c.hidden;
    const int M = B * nf * S, nw = d / 64;
    const bool mid = M >= frm_min_rows() && M % 128 == 0 && d % 128 == 0;
    // (the out-projections, N = d: the register-direct kernel is ahead up to twice that many rows -- 7.5 against 10.4 us at 2,048
    // rows, 17.7 against 14.4 at 4,096: profiles/r05e_frame_gemm.txt)
    const bool mid_proj = mid && M >= 2 * frm_min_rows();
    uint16_t* xs = (uint16_t*)w.xn;
    uint16_t* as = (uint16_t*)w.aux;
    uint16_t* big = (uint16_t*)w.big;
    const uint16_t* wq_s = lw.spatial.frame_w16;
    const uint16_t* wp_s = wq_s + (size_t)3 * d * d * NPL;
    const uint16_t* wq_t = lw.temporal.frame_w16;
    const uint16_t* wp_t = wq_t + (size_t)3 * d * d * NPL;
    const uint16_t* w1 = lw.mlp_frame_w16;
    const uint16_t* w2 = w1 + (size_t)hid * d * NPL;
    FrGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.a_group = 1; a.a_mul = 1; a.a_off = 0;
    a.alpha = 1.0f;
    a.M = M;
    a.rows_per_batch = M;
    // (Tried and dropped in round 6: extra workgroups that touch the NEXT Linear's weight tiles, XCD-targeted, so that its loads would hit
    // L2 / the Infinity Cache: neutral at 4 prefetch workgroups per XCD, slower below -- a prefetch workgroup streams at the same per-CU
    // rate as everybody else and outlives the Linear it rides in; profiles/r06e_weight_prefetch_ab.txt.)
    // ---- spatial: LayerNorm + qkv -> attention operand planes; attention; out-projection + residual (+ operand copy of x)
    // (qk-norm blocks: no LayerNorm in front -- the Linear reads the operand copy of x -- and q, k leave through the per-head LayerNorm)
    const bool qkn = c.qk_norm != 0;
    {
        FrGemmArgs g = a;
        g.W = wq_s; g.bias = c.qkv_bias ? lw.spatial.qkv_b : nullptr; g.N = 3 * d; g.K = d;
        g.qkvs = big; g.qscale = c.attn_scale * 1.4426950408889634f; g.H = H; g.S = S; g.Dh = c.head_dim;
        if (qkn) { g.qn_g = lw.spatial.norm_w; g.qn_b = lw.spatial.norm_b; g.A = xs; }
        if (mid) {
            if (!qkn) {
                ProfScope prof(GENIE_KC_LAYERNORM, 8.0 * M * d, 8.0 * M * d, st, "ln_fr_kernel");
                GENIE_TRY(launch_ln_fr(x, d, lw.norm1_w, lw.norm1_b, 1e-5f, as, M, d, st));
                g.A = as;
            }
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * 3.0 * d * d, 4.0 * M * d + 4.0 * 3 * d * d + 4.0 * M * 3 * d, st,
                           "gemm16_frm_kernel (qkv -> attention operand planes)");
            GENIE_TRY(launch_frm_any<FR_EPI_QKVS>(g, st));
        } else if (qkn) {
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * 3.0 * d * d, 4.0 * M * d + 4.0 * 3 * d * d + 4.0 * M * 3 * d, st,
                           "gemm16_fr_kernel (qkv + qk-norm -> attention operand planes)");
            GENIE_TRY((launch_fr_w<2, 1, FR_EPI_QKVS, false>(nw, g, st)));
        } else {
            g.X = x; g.ldx = d; g.ln_g = lw.norm1_w; g.ln_b = lw.norm1_b; g.ln_eps = 1e-5f;
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * 3.0 * d * d, 4.0 * M * d + 4.0 * 3 * d * d + 4.0 * M * 3 * d, st,
                           "gemm16_fr_kernel (LayerNorm + qkv -> attention operand planes)");
            GENIE_TRY((launch_fr_w<2, 1, FR_EPI_QKVS, true>(nw, g, st)));
        }
    }
    {
        ProfScope prof(GENIE_KC_ATTN_SPATIAL, 4.0 * S * S * c.head_dim * H * (double)(B * nf), (double)B * nf * S * d * 16.0, st,
                       "attn_spatial_fr_kernel");
        const dim3 grid((unsigned)(B * nf), H, 8);
        if (c.head_dim == 64) {
            const size_t lds = (size_t)8 * 32 * 68 * 4 + 8 * 32 * 2 * 4;
            static PerDevice<bool> attr_set;
            if (attr_set.needs()) { (void)hipFuncSetAttribute((const void*)attn_spatial_fr_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set.set(true); }
            attn_spatial_fr_kernel<64><<<grid, 512, lds, st>>>(big, as, H);
        } else {
            const size_t lds = (size_t)8 * 32 * 36 * 4 + 8 * 32 * 2 * 4;
            attn_spatial_fr_kernel<32><<<grid, 512, lds, st>>>(big, as, H);
        }
        GENIE_LAUNCH_CHECK("attn_spatial_fr");
    }
    {
        FrGemmArgs g = a;
        g.A = as; g.W = wp_s; g.bias = c.proj_bias ? lw.spatial.proj_b : nullptr; g.N = d; g.K = d;
        g.Cf = x; g.ldc = d; g.C16 = xs;
        ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)d * d, 4.0 * M * d * 4 + 4.0 * d * d, st,
                       mid_proj ? "gemm16_frm_kernel (proj + residual)" : "gemm16_fr_kernel (proj + residual)");
        if (mid_proj) GENIE_TRY(launch_frm_any<FR_EPI_RES>(g, st));
        else GENIE_TRY((launch_fr_w<1, 1, FR_EPI_RES, false>(nw, g, st)));
    }
    // ---- temporal: qkv -> cache slots t0 .. ; decode attention over the cache; out-projection + residual
    // one-frame passes on the register-direct kernels: both as ONE launch (qkvt_attn_fr_kernel)
    // MEASURED SLOWER and therefore off in the shipping library (profiles/r06d_qkvt_attn_merge_ab.txt, same box: batch 1 4.67 against 4.40 ms
    // per frame at 2 steps, 16.35 against 14.63 at 8): 64 workgroups streaming 384 KB each run at the per-CU rate of a fragment stream
    // (bytes in flight / latency: ~64 GB/s), three reduce rounds deep -- more than the launch + boundary + round trip the merge removes.
    // -DGENIE_VAR_QKVT_ATTN builds the variant (python 1xgpt_amd/build.py --variant qkvt -DGENIE_VAR_QKVT_ATTN); bit-identical results.
#ifdef GENIE_VAR_QKVT_ATTN
    static const int merged_on = 1;
#else
    static const int merged_on = study_env("GENIE_FRAME_QKVT_ATTN", 0);
#endif
    const bool merged_t = merged_on && nf == 1 && !mid && (nw == 8 || nw == 4 || (nw == 2 && c.head_dim == 32)) && w.frame_T <= 16;
    if (merged_t) {
        FrQkvtArgs q;
        q.A = xs; q.W = wq_t; q.bias = c.qkv_bias ? lw.temporal.qkv_b : nullptr; q.cache = w.fcache; q.out16 = as;
        q.qn_g = qkn ? lw.temporal.norm_w : nullptr; q.qn_b = qkn ? lw.temporal.norm_b : nullptr;
        q.K = d; q.H = H; q.S = S; q.T = w.frame_T; q.t = w.frame_t; q.scale = c.attn_scale;
        ProfScope prof(GENIE_KC_GEMM, 2.0 * M * 3.0 * d * d + 4.0 * (w.frame_t + 1) * c.head_dim * (double)M * H,
                       4.0 * M * d + 4.0 * 3 * d * d + 4.0 * M * 3 * d, st, "qkvt_attn_fr_kernel (temporal qkv -> cache + decode attention)");
        GENIE_TRY(launch_qkvt_attn(nw, c.head_dim, q, (unsigned)(M / 32), (unsigned)H, st));
    } else {
    {
        FrGemmArgs g = a;
        g.A = xs; g.W = wq_t; g.bias = c.qkv_bias ? lw.temporal.qkv_b : nullptr; g.N = 3 * d; g.K = d;
        g.Cf = w.fcache + (size_t)w.frame_t * S * 3 * d; g.ldc = 3 * d; g.rows_per_batch = (long)nf * S;
        g.strideC = (long)w.frame_T * S * 3 * d;
        ProfScope prof(GENIE_KC_GEMM, 2.0 * M * 3.0 * d * d, 4.0 * M * d + 4.0 * 3 * d * d + 4.0 * M * 3 * d, st,
                       mid ? "gemm16_frm_kernel (temporal qkv -> cache)" : "gemm16_fr_kernel (temporal qkv -> cache)");
        if (mid) GENIE_TRY(launch_frm_any<FR_EPI_F32>(g, st));
        else GENIE_TRY((launch_fr_w<2, 1, FR_EPI_F32, false>(nw, g, st)));
    }
    {
        const long n = (long)M * H;
        ProfScope prof(GENIE_KC_ATTN_TEMPORAL, 4.0 * (w.frame_t + nf) * c.head_dim * (double)n, (double)n * c.head_dim * 4.0 * (2 * (w.frame_t + nf) + 2), st,
                       "attn_temporal_fr_kernel");
        if (c.head_dim == 64)
            attn_temporal_fr_kernel<64><<<(unsigned)((n + 3) / 4), 256, 0, st>>>(w.fcache, as, n, w.frame_T, S, w.frame_t, nf, d, H, c.attn_scale,
                                                                                 qkn ? lw.temporal.norm_w : nullptr, qkn ? lw.temporal.norm_b : nullptr);
        else
            attn_temporal_fr_kernel<32><<<(unsigned)((n + 3) / 4), 256, 0, st>>>(w.fcache, as, n, w.frame_T, S, w.frame_t, nf, d, H, c.attn_scale,
                                                                                 qkn ? lw.temporal.norm_w : nullptr, qkn ? lw.temporal.norm_b : nullptr);
        GENIE_LAUNCH_CHECK("attn_temporal_fr");
    }
    }
    {
        FrGemmArgs g = a;
        g.A = as; g.W = wp_t; g.bias = c.proj_bias ? lw.temporal.proj_b : nullptr; g.N = d; g.K = d;
        g.Cf = x; g.ldc = d; g.C16 = qkn ? xs : nullptr;   // (qk-norm: fc1 has no LayerNorm in front and reads the operand copy of x)
        ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)d * d, 4.0 * M * d * 3 + 4.0 * d * d, st,
                       mid_proj ? "gemm16_frm_kernel (proj + residual)" : "gemm16_fr_kernel (proj + residual)");
        if (mid_proj) GENIE_TRY(launch_frm_any<FR_EPI_RES>(g, st));
        else GENIE_TRY((launch_fr_w<1, 1, FR_EPI_RES, false>(nw, g, st)));
    }
    // ---- MLP: LayerNorm + fc1 + GELU -> hidden operand; fc2 + residual (+ operand copy for the readout)
    {
        FrGemmArgs g = a;
        g.W = w1; g.bias = c.mlp_bias ? lw.fc1_b : nullptr; g.N = hid; g.K = d; g.C16 = big;
        if (qkn) g.A = xs;
        if (mid) {
            if (!qkn) {
                ProfScope prof(GENIE_KC_LAYERNORM, 8.0 * M * d, 8.0 * M * d, st, "ln_fr_kernel");
                GENIE_TRY(launch_ln_fr(x, d, lw.norm2_w, lw.norm2_b, 1e-5f, as, M, d, st));
                g.A = as;
            }
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)hid * d, 4.0 * M * d + 4.0 * hid * d + 4.0 * M * hid, st,
                           "gemm16_frm_kernel (fc1 + GELU)");
            GENIE_TRY(launch_frm_any<FR_EPI_GELU>(g, st));
        } else if (qkn) {
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)hid * d, 4.0 * M * d + 4.0 * hid * d + 4.0 * M * hid, st,
                           "gemm16_fr_kernel (fc1 + GELU)");
            GENIE_TRY((launch_fr_w<2, 1, FR_EPI_GELU, false>(nw, g, st)));
        } else {
            g.X = x; g.ldx = d; g.ln_g = lw.norm2_w; g.ln_b = lw.norm2_b; g.ln_eps = 1e-5f;
            ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)hid * d, 4.0 * M * d + 4.0 * hid * d + 4.0 * M * hid, st,
                           "gemm16_fr_kernel (LayerNorm + fc1 + GELU)");
            GENIE_TRY((launch_fr_w<2, 1, FR_EPI_GELU, true>(nw, g, st)));
        }
    }
    {
        FrGemmArgs g = a;
        g.A = big; g.W = w2; g.bias = c.mlp_bias ? lw.fc2_b : nullptr; g.N = d; g.K = hid;
        g.Cf = x; g.ldc = d; g.C16 = (want_xs || qkn) ? xs : nullptr;   // (qk-norm: the next block's spatial qkv reads it)
        ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)hid * d, 4.0 * M * hid + 4.0 * hid * d + 4.0 * M * d * 2, st,
                       mid ? "gemm16_frm_kernel (fc2 + residual)" : "gemm16_fr_kernel (fc2 + residual)");
        if (mid) GENIE_TRY(launch_frm_any<FR_EPI_RES>(g, st));
        else GENIE_TRY((launch_fr_w<1, 4, FR_EPI_RES, false>(nw, g, st)));
    }
    return GENIE_OK;
}

// In front of the first block of a frame pass: qk-norm blocks read the operand copy of the embedded rows (cast_fr_kernel)
int frame_prepare_f16x3(const genie_cfg& c, const float* x, Workspace& w, int B, int nf, hipStream_t st) {
    if (!c.qk_norm) return GENIE_OK;
    const long M = (long)B * nf * c.S;
    const size_t total = (size_t)(M / 32) * (c.d_model / 64) * 256;
    cast_fr_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, (uint16_t*)w.xn, M, c.d_model);
    GENIE_LAUNCH_CHECK("cast_fr");
    return GENIE_OK;
}

// y (M, N) f32 = A . W^T + bias from fragment-ordered split operands (C ABI genie_frame_linear: the kernel-level test and
// micro-benchmark entry).  mode 0: the block driver's own choice, 1: register-direct kernel, 2: LDS-tiled kernel.
int launch_frame_linear(const uint16_t* A, const uint16_t* W, const float* bias, float* y, int M, int N, int K, int mode, hipStream_t st) {
    GENIE_CHECK_SHAPE(M % 32 == 0 && N % 64 == 0 && K % 64 == 0, "frame_linear: (M, N, K) = (%d, %d, %d) must be multiples of (32, 64, 64)", M, N, K);
    FrGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.a_group = 1; g.a_mul = 1; g.a_off = 0;
    g.W = W; g.bias = bias; g.alpha = 1.0f; g.M = M; g.N = N; g.K = K;
    g.Cf = y; g.ldc = N; g.rows_per_batch = M; g.strideC = 0;
    const bool mid = mode == 2 || (mode == 0 && M >= frm_min_rows() && M % 128 == 0);
    ProfScope prof(GENIE_KC_GEMM, 2.0 * M * (double)N * K, 4.0 * M * K + 4.0 * N * K + 4.0 * M * N, st, "gemm16_fr(m)_kernel (frame_linear)");
    if (mid) return launch_frm_any<FR_EPI_F32>(g, st);
    if (K <= 512) return launch_fr_w<2, 1, FR_EPI_F32, false>(K / 64, g, st);
    set_error("frame_linear: K = %d has no register-direct f32-output kernel", K);
    return GENIE_E_UNSUPPORTED;
}

// out_x_proj on frame f_out of a frame pass from the fragment-ordered operand copy of x: logits (B, S, V) f32 token-major
int readout_frame_f16x3(const genie_cfg& c, const genie_weights& wt, Workspace& w, int B, int nf, int f_out, float* logits, hipStream_t st) {
    const int d = c.d_model, V = c.factored_vocab * c.num_factored;
    FrGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const uint16_t*)w.xn; g.a_group = c.S / 32; g.a_mul = nf; g.a_off = f_out;
    g.W = wt.out_frame_w16; g.bias = wt.out_b; g.alpha = c.readout_mult;
    g.M = B * c.S; g.N = V; g.K = d;
    g.Cf = logits; g.ldc = V; g.rows_per_batch = g.M; g.strideC = 0;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * g.M * (double)V * d, 4.0 * g.M * d + 4.0 * V * d + 4.0 * g.M * V, st, "gemm16_fr(m)_kernel (readout)");
    if (g.M >= frm_min_rows() && g.M % 128 == 0 && V % 128 == 0) return launch_frm_any<FR_EPI_F32>(g, st);
    return launch_fr_w<2, 1, FR_EPI_F32, false>(d / 64, g, st);
}

}  // namespace genie
