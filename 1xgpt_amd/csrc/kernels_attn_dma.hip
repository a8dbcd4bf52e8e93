// Spatial attention (softmax(q k^T) v over the S = 256 tokens of one frame, per head; genie/attention.py:36-61 with
// causal = False) as a persistent, LDS-DMA-fed kernel over the operand planes the QKV GEMM writes
// (launch_gemm16_pp with G16X_OUT16 | G16X_QKV):
//     qkv16 = [Q planes | K planes | V^T planes], 16-bit, NPL planes each (NPL = 2: f16 split pairs hi + lo'/2048, NPL = 1: bf16)
//     Q, K   head-major [(sequence, head)][256 rows][head_dim], Q already multiplied by scale * log2(e)
//     V^T    [(sequence, head)][feature][256 keys], the keys of every 16-key group stored {0-3, 8-11, 4-7, 12-15}
// so nothing is converted, split or transposed here: every global byte reaches LDS by buffer_load ... lds.
//
// One workgroup (8 waves, 32 queries each) per CU walks the (sequence, head) items  bid, bid + grid, ...  An item is 8 chunks
// of 64 keys -- K0..K3 then V0..V3 (16 KB each at head_dim 64 with split operands) -- that stream through a 4-slot ring
// (chunk c of an item always lands in slot c & 3); the chunk three phases ahead is issued at every phase barrier, across item
// boundaries, so HBM never waits for compute.  The 32 query rows of a wave live in its private 8 KB of LDS (next item's Q is
// fetched as soon as this item's fragments are in registers); a 4 KB per-wave scratch transposes the output tile.
//   LDS = 4 x 16 KB ring + 8 x 8 KB Q + 8 x 4 KB scratch = 160 KB.
// Arithmetic (same as the kernel it replaces, kernels_attn16.hip): S^T = K Q^T "swapped" so that a lane holds 128 scores of ITS
// query; softmax in registers on log2-scaled scores (v_exp_f32); the un-normalised probabilities are the A operand of P V and
// the row sum divides the output.  Split operands run 3 matrix instructions per product into ONE accumulator scaled by 2^11:
//   acc' += a_hi (2048 b_hi) + a_hi b_lo' + a_lo' b_hi    (2048 q_hi / 2048 p_hi are exact: |q * scale| < 32, p <= 1).
// vmcnt bookkeeping is per wave and dynamic (issue counters in SGPRs): a wait names how many younger operations may stay in
// flight, which differs for the first / last items of a workgroup.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ void wait_vm_dyn(int n) {  // n: wave-uniform number of youngest VMEM operations allowed in flight
#define GENIE_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        GENIE_VM_CASE(0) GENIE_VM_CASE(1) GENIE_VM_CASE(2) GENIE_VM_CASE(3) GENIE_VM_CASE(4) GENIE_VM_CASE(5)
        GENIE_VM_CASE(6) GENIE_VM_CASE(7) GENIE_VM_CASE(8) GENIE_VM_CASE(9) GENIE_VM_CASE(10) GENIE_VM_CASE(11)
        GENIE_VM_CASE(12) GENIE_VM_CASE(13) GENIE_VM_CASE(14) GENIE_VM_CASE(15) GENIE_VM_CASE(16) GENIE_VM_CASE(17)
        GENIE_VM_CASE(18) GENIE_VM_CASE(19) GENIE_VM_CASE(20) GENIE_VM_CASE(21) GENIE_VM_CASE(22) GENIE_VM_CASE(23)
        GENIE_VM_CASE(24) GENIE_VM_CASE(25) GENIE_VM_CASE(26) GENIE_VM_CASE(27) GENIE_VM_CASE(28) GENIE_VM_CASE(29)
        GENIE_VM_CASE(30) GENIE_VM_CASE(31) GENIE_VM_CASE(32)
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;  // stricter than asked: always safe
    }
#undef GENIE_VM_CASE
}
__device__ __forceinline__ void attn_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int NPL>
__device__ __forceinline__ f32x16 mma_k16(const s16x8& a, const s16x8& b, const f32x16& c) {
    if constexpr (NPL == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

}  // namespace

// STAMP (study knob GENIE_ATTN_STAMPS=1): wave 0 of every workgroup accumulates s_memtime deltas per phase -- counted wait,
// barrier, compute -- and adds them to stamps[0..24] (p*3 + {wait, barrier, compute}, [24] = output section) at the end.
template <int DH, int NPL, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void attn_spatial_dma_kernel(const uint16_t* __restrict__ qkv16, long P, int d, int H,
                                                                  long n_items, uint16_t* __restrict__ out16, long out_plane,
                                                                  unsigned long long* stamps = nullptr) {
    constexpr int ROWB = DH * 2;                // bytes of one K / Q row of this head
    constexpr int SPR = ROWB / 16;              // 16-byte slots per K / Q row
    constexpr int RPB = 256 / ROWB;             // K / Q rows per 256-byte bank row
    constexpr int KK = DH / 16;                 // k16 steps of q.k
    constexpr int NDT = DH / 32;                // 32-wide feature tiles of the output
    constexpr int KPL = 64 * ROWB;              // bytes of one K chunk plane (64 keys)
    constexpr int VPL = DH * 128;               // bytes of one V^T chunk plane (DH features x 64 keys)
    constexpr int CHUNK = NPL * KPL;            // = NPL * VPL
    constexpr int QW = NPL * 32 * ROWB;         // bytes of one wave's Q tile
    constexpr int OFF_Q = 4 * CHUNK, OFF_S = OFF_Q + 8 * QW;
    constexpr int PCC = CHUNK / 1024;           // LDS-DMA pieces per chunk
    constexpr int PCQ = QW / 1024;              // ... per wave Q tile
    constexpr int NST = NDT * 2 * NPL;          // output stores per lane per item
    static_assert(KPL == VPL && PCC >= 1 && PCQ >= 1, "chunk geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint16_t* Qb = qkv16;
    const uint16_t* Kb = qkv16 + (size_t)NPL * P;
    const uint16_t* Vb = qkv16 + (size_t)2 * NPL * P;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, -1, 0x00020000);
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, -1, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, -1, 0x00020000);

    // ---- per-lane source offsets of the LDS-DMA pieces (bytes, relative to the item / chunk base in the scalar offset)
    constexpr int NPW = (PCC + 7) / 8;          // chunk pieces per wave (waves beyond PCC issue none)
    static_assert(NPW <= 2 && PCQ <= 8, "piece tables");
    unsigned voK[2], voV[2];  // literal bounds: a captured array whose bound is a dependent constexpr local makes the
                                      // HOST-side instantiation of the kernel silently invalid (no device stub is emitted)
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int pc = wid + 8 * j;             // piece of the chunk image [plane][rows]
        const int pl = pc / (KPL / 1024), pp = pc % (KPL / 1024);
        {   // K: 1024 / ROWB rows per piece
            const int row = pp * (1024 / ROWB) + lane / SPR;
            const int slot = (lane % SPR) ^ ((row / RPB) % SPR);
            voK[j] = (unsigned)((size_t)pl * P * 2 + ((size_t)row * DH + slot * 8) * 2);
        }
        {   // V^T: 8 feature rows of 128 bytes (64 keys) per piece, 16-byte slot XOR (feature / 2) % 8
            const int f = pp * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((f >> 1) & 7);
            voV[j] = (unsigned)((size_t)pl * P * 2 + ((size_t)f * 256 + slot * 8) * 2);
        }
    }
    // Q pieces: piece j = (plane pl, rows pp * (1024 / ROWB) ..) of this wave's tile.  The lane part of the source offset depends on pp only
    // through the slot swizzle ((row / RPB) % SPR = (4 pp + lane-part) % SPR: two distinct values at most), the rest is wave-uniform and
    // rides in the scalar offset -- 2 address registers instead of PCQ (8 at head_dim 64 with split operands)
    constexpr int RPP = 1024 / ROWB;            // rows per piece
    unsigned voQ2[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
        const int row = q2 * RPP + lane / SPR;
        const int slot = (lane % SPR) ^ ((row / RPB) % SPR);
        voQ2[q2] = (unsigned)(((size_t)(lane / SPR) * DH + slot * 8) * 2);
    }
    static_assert((2 * RPP / RPB) % SPR == 0 || SPR == 8, "swizzle period of the Q pieces");
    int issued = 0;                              // VMEM operations this wave has issued so far
    int idx_slot[4] = {0, 0, 0, 0};              // value of `issued` right after the chunk now owning slot s was issued
    int idx_q = 0;
    auto issue_chunk = [&](long item, int c) {   // c = 0..3: K chunk c; 4..7: V chunk c - 4
        const long seq = item / H;
        const int head = (int)(item - seq * H);
        const bool isv = c >= 4;
        const int soff = isv ? (int)((((seq * H + head) * DH) * 256 + (c - 4) * 64) * 2)
                             : (int)((((seq * H + head) * 256 + c * 64) * (long)DH) * 2);
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            if (wid + 8 * j < PCC) {
                unsigned char* dst = smem + (c & 3) * CHUNK + (wid + 8 * j) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isv ? rsV : rsK, (__attribute__((address_space(3))) void*)dst, 16,
                                                         isv ? voV[j] : voK[j], soff, 0, 0);
                ++issued;
            }
        }
        idx_slot[c & 3] = issued;
    };
    auto issue_q = [&](long item, int j0, int j1) {   // pieces [j0, j1) of this wave's Q tile
        const long seq = item / H;
        const int head = (int)(item - seq * H);
        const int soff = (int)(((seq * H + head) * 256 * (long)DH) * 2);
#pragma unroll
        for (int j = 0; j < PCQ; ++j) {
            if (j >= j0 && j < j1) {
                const int pl = j / (32 * ROWB / 1024), pp = j % (32 * ROWB / 1024);
                const unsigned srow = (unsigned)((size_t)pl * P * 2) + (unsigned)((wid * 32 + pp * RPP) * ROWB);   // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(smem + OFF_Q + wid * QW + j * 1024),
                                                         16, voQ2[pp & 1], (int)((unsigned)soff + srow), 0, 0);
                ++issued;
            }
        }
        idx_q = issued;   // (after the last piece: every Q piece is at least this old)
    };

    // fragment read offsets
    unsigned offK[4];                            // K / Q rows: lane (r, h) reads slot 2 kk + h of row r (+ 32 per key tile)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) offK[kk] = r * ROWB + (((2 * kk + h) ^ ((r / RPB) % SPR)) << 4);
    unsigned offV[2];                            // V^T rows: feature dt * 32 + r; unit h of a 16-key group = this lane's 8 keys
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) offV[dt] = (dt * 32 + r) * 128;
    const int vsw = (r >> 1) & 7;                // slot swizzle of this lane's feature rows ((dt*32 + r) / 2) % 8 = (r / 2) % 8
    float* ct = reinterpret_cast<float*>(smem + OFF_S + wid * 4096);

    const long item0 = blockIdx.x, step = gridDim.x;
    if (item0 >= n_items) return;
    unsigned tacc[25];
    if constexpr (STAMP) {
#pragma unroll
        for (int i = 0; i < 25; ++i) tacc[i] = 0;
    }
    issue_chunk(item0, 0);
    issue_q(item0, 0, PCQ);
    issue_chunk(item0, 1);
    issue_chunk(item0, 2);

    for (long item = item0; item < n_items; item += step) {
        const long nxt = item + step;
        const bool has_next = nxt < n_items;
        f32x16 sc[8];
        f32x16 oacc[NDT];
        s16x8 qf[NPL][KK], qup[KK];
        // ONLINE softmax per 64-key chunk (round 6): chunk c's 32 scores of this lane are exponentiated against the chunk's OWN maximum
        // mloc[c] one phase after its matrix instructions -- VALU work that now sits between the next chunk's matrix instructions instead of
        // in one block behind the last S tile with the matrix pipe idle -- and rescaled by fsc[c] = 2^(mloc[c] - max) where the probabilities
        // are packed for P V (folded into the 2^11 scaling of the split, so the rescale costs nothing there).
        float mloc[4], psum[4], fsc[4];
        float inv = 0.f;
        constexpr float UNS = NPL == 2 ? 1.0f / 2048.0f : 1.0f;
        auto softmax_chunk = [&](int c) {    // tiles 2c, 2c + 1 -> un-normalised probabilities relative to the chunk maximum
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) m = fmaxf(m, sc[2 * c + t][e]);
            m = fmaxf(m, __shfl_xor(m, 32));
            const float mc = m * UNS;
            mloc[c] = mc;
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sc[2 * c + t][e] = __builtin_amdgcn_exp2f(fmaf(sc[2 * c + t][e], UNS, -mc));
                    sum += sc[2 * c + t][e];
                }
            psum[c] = sum;
            // (pinned here: left alone, the compiler sinks all 96 exponentials of chunks 0-2 down to their first use in phase 4)
            asm volatile("" : "+v"(sc[2 * c]), "+v"(sc[2 * c + 1]), "+v"(psum[c]), "+v"(mloc[c]));
        };
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // ---- chunk p of this item (and, for p = 0, this wave's Q tile) must have landed; then every wave's share has
            int need = issued - idx_slot[p & 3];
            if (p == 0) need = min(need, issued - idx_q);
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
            if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
            wait_vm_dyn(need);
            if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
            attn_barrier();
            if constexpr (STAMP) ts2 = __builtin_amdgcn_s_memtime();
            // slot (p - 1) & 3 is free now: refill it with the chunk three phases ahead
            if (p + 3 < 8) issue_chunk(item, p + 3);
            else if (has_next) issue_chunk(nxt, p + 3 - 8);
            // next item's Q tile: this item's Q fragments are in registers since phase 0; the 8 pieces are spread over phases
            // 1..4 (LDS-DMA issue is the expensive part of a phase: all 8 in phase 1 made it 1.5x as long as its neighbours)
            constexpr int QPP = (PCQ + 3) / 4;
            if (p >= 1 && p <= 4 && has_next && (p - 1) * QPP < PCQ) issue_q(nxt, (p - 1) * QPP, min(p * QPP, PCQ));
            const unsigned char* sl = smem + (p & 3) * CHUNK;
            if (p == 0) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
                        qf[pl][kk] = *reinterpret_cast<const s16x8*>(smem + OFF_Q + wid * QW + pl * 32 * ROWB + offK[kk]);
            }
            if (p < 4) {
                // (2048 q_hi is re-made per phase -- 16 packed multiplies -- instead of living in 16 registers through the S phases:
                // the online softmax's in-flight values need them)
                if constexpr (NPL == 2) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
                        qup[kk] = __builtin_bit_cast(s16x8, __builtin_bit_cast(f16x8, qf[0][kk]) * (_Float16)2048.0f);
                    asm volatile("" : "+v"(qup[0]), "+v"(qup[KK - 1]));
                }
                // ---- S^T tiles 2p, 2p+1: sc[kt][e] = score(key kt*32 + (e&3) + 8(e>>2) + 4h, query r) (x 2048 when split)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x16 a;
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const s16x8 kh = *reinterpret_cast<const s16x8*>(sl + t * 32 * ROWB + offK[kk]);
                        if constexpr (NPL == 2) {
                            const s16x8 kl = *reinterpret_cast<const s16x8*>(sl + KPL + t * 32 * ROWB + offK[kk]);
                            a = mma_k16<2>(kh, qup[kk], a);
                            a = mma_k16<2>(kh, qf[1][kk], a);
                            a = mma_k16<2>(kl, qf[0][kk], a);
                        } else {
                            a = mma_k16<1>(kh, qf[0][kk], a);
                        }
                    }
                    sc[(p & 3) * 2 + t] = a;
                }
#ifdef GENIE_VAR_ATTN_BLOCK_SOFTMAX   // (A/B variant: all exponentials in one block behind the last S tile, as before round 6)
                if (p == 3) { softmax_chunk(0); softmax_chunk(1); softmax_chunk(2); }
#else
                if (p >= 1) softmax_chunk(p - 1);     // (independent of this phase's matrix instructions: the scheduler interleaves them)
#endif
            } else {
                if (p == 4) {
                    // chunk 3's maximum closes the row maximum (both halves of the row: mloc is already lane-pair uniform); its
                    // exponentials are only needed in phase 7 but cost nothing here, between this phase's matrix instructions
                    softmax_chunk(3);
                    const float mx = fmaxf(fmaxf(mloc[0], mloc[1]), fmaxf(mloc[2], mloc[3]));
                    float sum = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        fsc[c] = __builtin_amdgcn_exp2f(mloc[c] - mx);
                        sum = fmaf(psum[c], fsc[c], sum);
                    }
                    sum += __shfl_xor(sum, 32);
                    inv = 1.0f / sum;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
                }
                // ---- O += P V over keys (p-4)*64 .. +63: slot s of MFMA m of key tile kt is key kt*32 + 4h + (s&3) + 8(s>>2) + 16m
                const float fchunk = fsc[(p - 4) & 3];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int kt = (p - 4) * 2 + t;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        s16x8 pa, pb, pc;  // split: 2048 p_hi, p_hi, p_lo';  bf16: pa only
                        if constexpr (NPL == 2) {
                            f16x8 up, hi, lo;
#pragma unroll
                            for (int s2 = 0; s2 < 8; s2 += 2) {
                                const f32x2v pv = {sc[kt][8 * m + s2], sc[kt][8 * m + s2 + 1]};
                                const f32x2v ps = pv * (2048.0f * fchunk);   // p = 2^(s - chunk max) * 2^(chunk max - row max), scaled 2^11
                                const f16x2v u2 = __builtin_convertvector(ps, f16x2v);
                                const f32x2v rem = ps - __builtin_convertvector(u2, f32x2v);
                                const f16x2v l2 = __builtin_convertvector(rem, f16x2v);
                                up[s2] = u2[0]; up[s2 + 1] = u2[1];
                                lo[s2] = l2[0]; lo[s2 + 1] = l2[1];
                            }
                            hi = up * (_Float16)(1.0f / 2048.0f);
                            pa = __builtin_bit_cast(s16x8, up); pb = __builtin_bit_cast(s16x8, hi); pc = __builtin_bit_cast(s16x8, lo);
                        } else {
#pragma unroll
                            for (int s2 = 0; s2 < 8; ++s2) pa[s2] = (short)f32_to_bf16(sc[kt][8 * m + s2] * fchunk);
                        }
                        // the planes store the keys of a 16-key group as {0-3, 8-11 | 4-7, 12-15}: unit h is exactly the 8 keys
                        // lane half h feeds to MFMA m (slots 0..3 -> key0 + 0..3, slots 4..7 -> key0 + 8..11): one ds_read_b128
                        const int s0 = ((t * 4 + 2 * m + h) ^ vsw) << 4;
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt) {
                            const s16x8 vh = *reinterpret_cast<const s16x8*>(sl + offV[dt] + s0);
                            if constexpr (NPL == 2) {
                                const s16x8 vl = *reinterpret_cast<const s16x8*>(sl + VPL + offV[dt] + s0);
                                oacc[dt] = mma_k16<2>(pa, vh, oacc[dt]);
                                oacc[dt] = mma_k16<2>(pb, vl, oacc[dt]);
                                oacc[dt] = mma_k16<2>(pc, vh, oacc[dt]);
                            } else {
                                oacc[dt] = mma_k16<1>(pa, vh, oacc[dt]);
                            }
                        }
                    }
                }
            }
            if constexpr (STAMP) {
                asm volatile("s_nop 0" ::: "memory");
                const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
                tacc[p * 3] += (unsigned)(ts1 - ts0);
                tacc[p * 3 + 1] += (unsigned)(ts2 - ts1);
                tacc[p * 3 + 2] += (unsigned)(ts3 - ts2);
            }
        }
        unsigned long long tso = 0;
        if constexpr (STAMP) tso = __builtin_amdgcn_s_memtime();
        // ---- output: oacc[dt][e] = O(query (e&3) + 8(e>>2) + 4h, feature dt*32 + r); through the wave's scratch to whole rows
        const long seq = item / H;
        const int head = (int)(item - seq * H);
        const float oscale = (NPL == 2 ? 1.0f / 2048.0f : 1.0f) * inv;  // inv belongs to query r: applied on the write side
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // the row sum is per QUERY: this lane holds queries (e&3) + 8(e>>2) + 4h, whose 1/sum lives in lane (that query)
                const int qrow = (e & 3) + 8 * (e >> 2) + 4 * h;
                ct[qrow * 32 + r] = oacc[dt][e];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2), fq = (lane & 3) * 8;
                const float4 a = *reinterpret_cast<const float4*>(ct + row * 32 + fq);
                const float4 b = *reinterpret_cast<const float4*>(ct + row * 32 + fq + 4);
                const float sr = __shfl(oscale, row);  // 1/sum of query `row` (lane `row` owns query row)
                const size_t oi = (size_t)(seq * 256 + wid * 32 + row) * d + head * DH + dt * 32 + fq;
                typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                if constexpr (NPL == 2) {
                    uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
                    split_f16_x4(a.x * sr, a.y * sr, a.z * sr, a.w * sr, h01, h23, l01, l23);
                    split_f16_x4(b.x * sr, b.y * sr, b.z * sr, b.w * sr, h45, h67, l45, l67);
                    const u4v th = {h01, h23, h45, h67}, tl = {l01, l23, l45, l67};
                    *reinterpret_cast<u4v*>(out16 + oi) = th;
                    *reinterpret_cast<u4v*>(out16 + out_plane + oi) = tl;
                } else {
                    const u4v t = {(uint32_t)f32_to_bf16(a.x * sr) | ((uint32_t)f32_to_bf16(a.y * sr) << 16),
                                   (uint32_t)f32_to_bf16(a.z * sr) | ((uint32_t)f32_to_bf16(a.w * sr) << 16),
                                   (uint32_t)f32_to_bf16(b.x * sr) | ((uint32_t)f32_to_bf16(b.y * sr) << 16),
                                   (uint32_t)f32_to_bf16(b.z * sr) | ((uint32_t)f32_to_bf16(b.w * sr) << 16)};
                    *reinterpret_cast<u4v*>(out16 + oi) = t;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        issued += NST;
        if constexpr (STAMP) tacc[24] += (unsigned)(__builtin_amdgcn_s_memtime() - tso);
    }
    if constexpr (STAMP) {
        if (tid == 0 && stamps) {
#pragma unroll
            for (int i = 0; i < 25; ++i) atomicAdd(&stamps[i], (unsigned long long)tacc[i]);
        }
    }
}

// qkv16: the planes written by launch_gemm16_pp(G16X_OUT16 | G16X_QKV) for n_seq sequences of 256 tokens (M = 256 n_seq rows);
// out16: (M, d) row-major, split planes [hi | lo] at out_plane (npl = 2) or bf16 (npl = 1).
int launch_attn_spatial_dma(int npl, const uint16_t* qkv16, long n_seq, int d, int H, int Dh, uint16_t* out16, size_t out_plane,
                            hipStream_t st) {
    if ((Dh != 64 && Dh != 32) || d != H * Dh || n_seq <= 0) return GENIE_E_UNSUPPORTED;
    const long P = n_seq * 256 * (long)d;
    if ((double)P * 2 * npl + 4096.0 * d >= 4.0e9) return GENIE_E_UNSUPPORTED;  // 32-bit offsets inside a buffer descriptor
    const long items = n_seq * H;
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
    ProfScope prof(GENIE_KC_ATTN_SPATIAL, 4.0 * 256 * 256 * Dh * (double)items, (double)items * 256 * Dh * 2.0 * npl * 4.0, st);
#define ATTN_LAUNCH(DH_, NPL_)                                                                                            \
    do {                                                                                                                  \
        constexpr int lds = 4 * NPL_ * 64 * DH_ * 2 + 8 * NPL_ * 32 * DH_ * 2 + 8 * 4096;                                 \
        (void)hipFuncSetAttribute((const void*)attn_spatial_dma_kernel<DH_, NPL_>,                                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                       \
        attn_spatial_dma_kernel<DH_, NPL_><<<grid, 512, lds, st>>>(qkv16, P, d, H, items, out16, (long)out_plane);        \
    } while (0)
#ifdef GENIE_STUDY   // per-phase s_memtime stamps: allocates and synchronises inside the launch -- study builds only
    static const int stamp = study_env("GENIE_ATTN_STAMPS", 0);
    if (stamp && Dh == 64 && npl == 2) {
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 25 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbuf, 0, 25 * sizeof(unsigned long long), st);
        constexpr int lds = 4 * 2 * 64 * 64 * 2 + 8 * 2 * 32 * 64 * 2 + 8 * 4096;
        (void)hipFuncSetAttribute((const void*)attn_spatial_dma_kernel<64, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attn_spatial_dma_kernel<64, 2, true><<<grid, 512, lds, st>>>(qkv16, P, d, H, items, out16, (long)out_plane, dbuf);
        unsigned long long hbuf[25];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hbuf, dbuf, sizeof(hbuf), hipMemcpyDeviceToHost);
        const double per = 1.0 / (double)items;  // ticks per item (wave 0 of the workgroup that ran it)
        fprintf(stderr, "attn_stamps items=%ld per-item ticks:", items);
        for (int p8 = 0; p8 < 8; ++p8)
            fprintf(stderr, " p%d[w %.0f b %.0f c %.0f]", p8, hbuf[p8 * 3] * per, hbuf[p8 * 3 + 1] * per, hbuf[p8 * 3 + 2] * per);
        fprintf(stderr, " out %.0f\n", hbuf[24] * per);
        GENIE_LAUNCH_CHECK("attn_spatial_dma_stamps");
        return GENIE_OK;
    }
#endif
    if (Dh == 64 && npl == 2) ATTN_LAUNCH(64, 2);
    else if (Dh == 64) ATTN_LAUNCH(64, 1);
    else if (npl == 2) ATTN_LAUNCH(32, 2);
    else ATTN_LAUNCH(32, 1);
#undef ATTN_LAUNCH
    GENIE_LAUNCH_CHECK("attn_spatial_dma");
    return GENIE_OK;
}

}  // namespace genie
