// gemm16_pp_kernel: the 16-bit NT GEMM of the GENIE forward, rebuilt around a two-group ("ping-pong") phase schedule.
//
//   C[M,N] (+)= epilogue( alpha * A[M,K] . W[N,K]^T + bias )      (nn.Linear; st_transformer.py:16-25, attention.py:27-29)
//
//   * 256x256 block tile, 8 waves = 2 (M) x 4 (N), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16 tiles (128 accumulator
//     registers), two waves per SIMD: waves 0-3 (group 0, rows 0-127) and waves 4-7 (group 1, rows 128-255).
//   * A K-tile is 128 bytes of every row: 64 bf16/f16 values (NPL = 1) or 32 values x [hi | lo] f16 planes (NPL = 2).
//     It lives in LDS as FOUR 16 KB half-tiles ordered by WHEN the waves read them, not by row:
//        A-early = the first 64 rows of each group's 128,  A-late = the other 64,
//        B-early = the first 32 columns of each wave's 64, B-late = the other 32.
//     Two K-tiles (128 KB) are resident; the LDS image is lane-linear (LDS-DMA), the 16-byte slot of a row is XOR-swizzled
//     with (row/2)%8 on the SOURCE address and on the fragment read: conflict-free ds_read_b128.
//   * Every K-tile is 4 phases, one 64x32 quadrant of the wave tile each: (a0,c0) (a0,c1) (a1,c1) (a1,c0).  A phase is
//        LOAD: fragment reads for the quadrant (12 / 4 / 8 / 0 ds_read_b128) + 2 LDS-DMA pieces (one half-tile per
//              phase per workgroup) + a COUNTED s_waitcnt vmcnt   -> s_barrier ->
//        MFMA: 8 (NPL = 1) or 12 (NPL = 2) matrix instructions   -> s_barrier
//     Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA part while the other is in its
//     LOAD part: the matrix pipe sees back-to-back clusters, the LDS/DMA traffic of one wave hides under the other's.
//   * Hazard rules used (interval = time between two workgroup barriers; group 0 runs LOAD(p) in interval 2p, group 1
//     in 2p+1): a half-tile last read in LOAD(p) is re-staged in LOAD(p+2) or later; data waited for in LOAD(p) (vmcnt
//     before the barrier, by every wave) is read in LOAD(p+1) or later.  Schedule per K-tile t:
//        q0 stages B-late(t+1)   waits vmcnt(8) -> B-late(t) landed  (read in q1)
//        q1 stages A-late(t+1)   waits vmcnt(8) -> A-late(t) landed  (read in q2)
//        q2 stages A-early(t+2)  -
//        q3 stages B-early(t+2)  waits vmcnt(8) -> A-early(t+1), B-early(t+1) landed (read in q0 of t+1)
//     i.e. four half-tiles (8 loads per lane) stay in flight across every barrier.
//   * NPL = 2 ("f16x3"): operands are split pairs a = hi + lo'/2048.  The three products run into ONE accumulator
//     scaled by 2^11:  acc' += ah.(2048 bh) + ah.bl' + al'.bh  (2048 bh is a packed-f16 multiply on the B fragment in
//     registers: exact for |w| < 32); the epilogue multiplies by alpha / 2048.  TERMS = 2 drops ah.bl' (weights rounded
//     to f16), NPL = 1 with F16 reads only the hi plane (plain f16) -- both for the accuracy/speed study of DESIGN.md.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ABL & 32: wave 0 of every workgroup stamps s_memtime at start / first data / end of main loop / end (debug study)
__device__ unsigned long long* g_pp_timing = nullptr;

namespace {

constexpr int PP_HT = 16384;    // bytes of one half-tile
constexpr int PP_BUF = 65536;   // bytes of one K-tile (4 half-tiles: A-early, A-late, B-early, B-late)

template <int N, bool SKIP = false>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (!SKIP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// LDS accesses of the epilogue go through ext_vector types, never through HIP's float4 STRUCT: with the struct's TBAA the compiler's
// waitcnt pass puts an s_waitcnt vmcnt(0) in front of the first such ds_read / ds_write while an LDS-DMA is in flight, and the next
// tile's K-tile 0 -- requested just before the epilogue -- was drained there (a memory round trip per tile with the matrix pipe idle).
typedef float pp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_ld4(const float* p) {
    const pp_f4 v = *reinterpret_cast<const pp_f4*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<pp_f4*>(p) = pp_f4{a, b, c, d}; }

template <bool F16>
__device__ __forceinline__ f32x16 mma16(const s16x8& a, const s16x8& b, const f32x16& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

}  // namespace

// ABL: timing-only ablations for the study in DESIGN.md (results are wrong when ABL != 0): 1 no in-loop LDS-DMA,
// 2 no in-loop fragment reads, 4 no vmcnt waits, 8 no output stores, 16 no matrix instructions
// EPI >= 0: the epilogue's flag word (G16X_*) is a compile-time constant (the model's four Linear flavours get their own
// instantiation: straight-line epilogue, loads hoisted, no per-row branches); EPI = -1: flags are read at run time.
// GELU of an epilogue.  LOWP (bf16 operands, the value leaves only as bf16): the polynomial form -- its 1.3e-5 is 0.7 % of a bf16
// half-ulp (common.hpp); otherwise the 1.5e-7 form.
template <bool LOWP>
__device__ __forceinline__ genie_f2 pp_gelu2(genie_f2 z) {
#ifdef GENIE_VAR_PP_GELU_AS   // (variant: the 1.5e-7 form everywhere)
    return gelu_erf_fast2(z);
#else
    if constexpr (LOWP) return gelu_erf_poly2(z);
    else return gelu_erf_fast2(z);
#endif
}

template <int NPL, int TERMS, bool F16, int ABL = 0, int SCHED = 0, int EPI = -1>
__global__ __launch_bounds__(512, 2) void gemm16_pp_kernel(const uint16_t* __restrict__ A, long lda, long planeA,
                                                            const uint16_t* __restrict__ W, long ldw, long planeW,
                                                            const float* __restrict__ bias, float* __restrict__ Cf,
                                                            uint16_t* __restrict__ C16, long plane16, long ldc, int M, int N,
                                                            int K, int flags, float alpha, long strideA, long strideC,
                                                            const float* Rf, long strideW, float qscale, int head_dim,
                                                            const float* __restrict__ qn_g, const float* __restrict__ qn_b) {
    constexpr int KK = NPL == 1 ? 4 : 2;  // k16 steps per K-tile
    constexpr int BK = 16 * KK;
    constexpr int BM = 256, BN = 256;
    constexpr int OFF_AE = 0, OFF_AL = PP_HT, OFF_BE = 2 * PP_HT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int r = lane & 31, h = lane >> 5;

    // PERSISTENT: one workgroup per CU walks tiles bid, bid + gridDim.x, ... (the hardware hands block b to XCD b % 8, so the
    // walk stays on one XCD).  XCD-aware tile order: the 8 m-tiles of a group go to the 8 XCDs and each XCD walks the n-tiles
    // of ITS m-tile, so an A row panel is fetched into one L2 only.
    const int mt = M / BM, nt_n = N / BN;
    const int ntiles = mt * nt_n;
    const int full = (mt / 8) * 8 * nt_n;
    auto tile_origin = [&](int bid, int& m0_, int& n0_) {
        int m_tile, n_tile;
        if (bid < full) {
            const int grp = bid / (8 * nt_n), rem = bid - grp * 8 * nt_n;
            m_tile = grp * 8 + (rem & 7);
            n_tile = rem >> 3;
        } else {
            const int rem = bid - full;
            m_tile = (mt / 8) * 8 + rem / nt_n;
            n_tile = rem % nt_n;
        }
        m0_ = m_tile * BM;
        n0_ = n_tile * BN;
    };
    A += (size_t)blockIdx.y * strideA;
    W += (size_t)blockIdx.y * strideW;

    // ---- staging: LDS-DMA through buffer descriptors (voffset per lane, K / late-half offset in the scalar offset)
    int bid = blockIdx.x, m0, n0;
    tile_origin(bid, m0, n0);
    auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, -1, 0x00020000);
    auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, -1, 0x00020000);
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rho = (wid * 2 + j) * 8 + (lane >> 3);          // row of the half-tile image this lane fills
        const int s = (lane & 7) ^ ((rho >> 1) & 7);               // logical 16-byte slot that lands in physical slot lane&7
        const unsigned ka = NPL == 1 ? s * 16 : (unsigned)((s >> 2) * planeA * 2 + (s & 3) * 16);
        const unsigned kb = NPL == 1 ? s * 16 : (unsigned)((s >> 2) * planeW * 2 + (s & 3) * 16);
        voffA[j] = (unsigned)(((rho >> 6) * 128 + (rho & 63)) * lda * 2) + ka;
        voffB[j] = (unsigned)(((rho >> 5) * 64 + (rho & 31)) * ldw * 2) + kb;
    }
    const int lateA = (int)(64 * lda * 2), lateB = (int)(32 * ldw * 2);
    bool in_loop = false;
    // one 8 KB piece (j = 0, 1) of a half-tile: ht 0 A-early, 1 A-late, 2 B-early, 3 B-late; all arguments wave-uniform
    auto piece = [&](int ht, int j, int buf, int kt) {
        if constexpr (ABL & 1) { if (in_loop) return; }
        // (ABL & 256, study: every K-tile re-reads K-tile 0's addresses -- the same DMA instructions and LDS writes, all L2 hits)
        const int soff = ((ABL & 256) ? 0 : kt * BK * 2) + (ht == 1 ? lateA : ht == 3 ? lateB : 0);
        unsigned char* dst = smem + buf * PP_BUF + ht * PP_HT + wid * 2048 + j * 1024;
        // ABL & 64 / & 128 (study): non-temporal policy (aux = 2) on the A / W stream
        constexpr int AUXA = (ABL & 64) ? 2 : 0, AUXW = (ABL & 128) ? 2 : 0;
        if (ht < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voffA[j], soff, 0, AUXA);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voffB[j], soff, 0, AUXW);
    };
    auto stage = [&](int ht, int buf, int kt) {
        piece(ht, 0, buf, kt);
        piece(ht, 1, buf, kt);
    };

    // ---- fragment read offsets: lane (r, h) reads 16-byte slot 2c + h of its row, c = plane * KK + k16 step
    unsigned offA[4], offB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned o = r * 128 + (((2 * c + h) ^ ((r >> 1) & 7)) << 4);
        offA[c] = o + wm * 8192;            // rows wm*64 .. of A-early / A-late
        offB[c] = o + wn * 4096 + OFF_BE;   // rows wn*32 .. of B-early / B-late
    }

    f32x16 acc[4][2];

    s16x8 fa[2][4];      // the current 64-row A sub-tile: [row tile][c]
    s16x8 fb[2][4];      // both 32-column B sub-tiles:    [sub][c]
    s16x8 fup[2][2];     // NPL = 2: 2048 * hi plane of the B sub-tiles [sub][kk]

    auto read_a = [&](int buf, int late) {
        if constexpr (ABL & 2) { if (in_loop) return; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                fa[i][c] = *reinterpret_cast<const s16x8*>(smem + buf * PP_BUF + (late ? OFF_AL : OFF_AE) + i * 4096 + offA[c]);
    };
    auto read_b = [&](int buf, int late) {
        if constexpr (ABL & 2) { if (in_loop) return; }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            fb[late][c] = *reinterpret_cast<const s16x8*>(smem + buf * PP_BUF + (late ? PP_HT : 0) + offB[c]);
    };
    auto scale_b = [&](int sub) {
        if constexpr (NPL == 2) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                fup[sub][kk] = __builtin_bit_cast(s16x8, __builtin_bit_cast(f16x8, fb[sub][kk]) * (_Float16)2048.0f);
        }
    };
    auto mma = [&](int asub, int csub) {
        f32x16& c0 = acc[asub * 2][csub];
        f32x16& c1 = acc[asub * 2 + 1][csub];
        if constexpr (ABL & 16) {
            asm volatile("" : "+v"(c0), "+v"(c1));
            return;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if constexpr (NPL == 1) {
                c0 = mma16<F16>(fa[0][kk], fb[csub][kk], c0);
                c1 = mma16<F16>(fa[1][kk], fb[csub][kk], c1);
            } else {
                c0 = mma16<true>(fa[0][kk], fup[csub][kk], c0);
                c1 = mma16<true>(fa[1][kk], fup[csub][kk], c1);
                if constexpr (TERMS == 3) {
                    c0 = mma16<true>(fa[0][kk], fb[csub][KK + kk], c0);
                    c1 = mma16<true>(fa[1][kk], fb[csub][KK + kk], c1);
                }
                c0 = mma16<true>(fa[0][KK + kk], fb[csub][kk], c0);
                c1 = mma16<true>(fa[1][KK + kk], fb[csub][kk], c1);
            }
        }
    };
    auto mfma_part = [&](int asub, int csub, int scale_sub) {
        wg_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (scale_sub >= 0) scale_b(scale_sub);
        __builtin_amdgcn_s_setprio(1);
        mma(asub, csub);
        __builtin_amdgcn_s_setprio(0);
    };
    // SCHED 2 (study knob GENIE_PP_SCHED=2, run-time-flag epilogue only): the phase's two LDS-DMA pieces are issued from INSIDE
    // the matrix cluster (after the first pair of MFMAs of k-step 0 and of k-step KK/2) instead of in the LOAD part the other
    // wave group waits on.  Measured (r2g_sched2.log): identical throughput to SCHED 0 on every shape, f16x3 and bf16 -- the
    // launch rate is pinned by the board's power-managed clock, not by where the VMEM issue sits.  ht < 0: nothing to stage.
    auto mfma_part_dma = [&](int asub, int csub, int scale_sub, int ht, int buf, int kt) {
        wg_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (scale_sub >= 0) scale_b(scale_sub);
        f32x16& c0 = acc[asub * 2][csub];
        f32x16& c1 = acc[asub * 2 + 1][csub];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if constexpr (NPL == 1) {
                c0 = mma16<F16>(fa[0][kk], fb[csub][kk], c0);
                c1 = mma16<F16>(fa[1][kk], fb[csub][kk], c1);
                if (ht >= 0 && (kk == 0 || kk == KK / 2)) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(ht, kk == 0 ? 0 : 1, buf, kt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                c0 = mma16<true>(fa[0][kk], fup[csub][kk], c0);
                c1 = mma16<true>(fa[1][kk], fup[csub][kk], c1);
                if (ht >= 0 && (kk == 0 || kk == KK / 2)) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(ht, kk == 0 ? 0 : 1, buf, kt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (TERMS == 3) {
                    c0 = mma16<true>(fa[0][kk], fb[csub][KK + kk], c0);
                    c1 = mma16<true>(fa[1][kk], fb[csub][KK + kk], c1);
                }
                c0 = mma16<true>(fa[0][KK + kk], fb[csub][kk], c0);
                c1 = mma16<true>(fa[1][KK + kk], fb[csub][kk], c1);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // MODE 0: steady state, 1: second-to-last K-tile (stages for tile t+1 only), 2: last K-tile (stages nothing).
    // SCHED 0: one half-tile per phase (q0 B-late(t+1), q1 A-late(t+1), q2 A-early(t+2), q3 B-early(t+2)).
    // SCHED 1: the LDS-DMA issue (the expensive part of a LOAD) is moved away from the phases that carry the fragment
    //          reads: q0 none (12 reads), q1 B-late(t+1) + half of A-late(t+1) (4 reads), q2 the other half (8 reads),
    //          q3 A-early(t+2) + B-early(t+2) (no reads).  Same hazard rules; waits: q0 vmcnt(6), q1 (7), q3 (8).
    auto ktile = [&](auto bufc, auto modec, int t) {
        constexpr int BUF = decltype(bufc)::value, MODE = decltype(modec)::value;
        constexpr bool NOWAIT = (ABL & 4) != 0;
        if constexpr (SCHED == 2) {
            // same half-tile per phase as SCHED 0, issued half a phase later (inside the MFMA part), so every counted wait
            // names one stage (2 loads) fewer: q0 B-late(t) landed <- younger: A-late(t), A-early(t+1), B-early(t+1) = 6; ...
            read_a(BUF, 0);
            read_b(BUF, 0);
            if constexpr (MODE < 2) wait_vmcnt<6, NOWAIT>(); else wait_vmcnt<2>();
            mfma_part_dma(0, 0, 0, MODE < 2 ? 3 : -1, BUF ^ 1, t + 1);
            wg_barrier();
            read_b(BUF, 1);
            if constexpr (MODE < 2) wait_vmcnt<6, NOWAIT>(); else wait_vmcnt<0>();
            mfma_part_dma(0, 1, 1, MODE < 2 ? 1 : -1, BUF ^ 1, t + 1);
            wg_barrier();
            read_a(BUF, 1);
            mfma_part_dma(1, 1, -1, MODE == 0 ? 0 : -1, BUF, t + 2);
            wg_barrier();
            if constexpr (MODE == 0) wait_vmcnt<6, NOWAIT>();
            if constexpr (MODE == 1) wait_vmcnt<4>();
            mfma_part_dma(1, 0, -1, MODE == 0 ? 2 : -1, BUF, t + 2);
            if (MODE < 2 || wm == 0) wg_barrier();
            return;
        }
        // q0: quadrant (a0, c0)
        read_a(BUF, 0);
        read_b(BUF, 0);
        if constexpr (SCHED == 0) {
            if constexpr (MODE < 2) stage(3, BUF ^ 1, t + 1);
            if constexpr (MODE < 2) wait_vmcnt<8, NOWAIT>(); else wait_vmcnt<2>();
        } else {
            if constexpr (MODE < 2) wait_vmcnt<6, NOWAIT>(); else wait_vmcnt<2>();
        }
        mfma_part(0, 0, 0);
        wg_barrier();
        // q1: quadrant (a0, c1)
        read_b(BUF, 1);
        if constexpr (SCHED == 0) {
            if constexpr (MODE < 2) stage(1, BUF ^ 1, t + 1);
            if constexpr (MODE < 2) wait_vmcnt<8, NOWAIT>(); else wait_vmcnt<0>();
        } else {
            if constexpr (MODE < 2) { stage(3, BUF ^ 1, t + 1); piece(1, 0, BUF ^ 1, t + 1); }
            if constexpr (MODE < 2) wait_vmcnt<7, NOWAIT>(); else wait_vmcnt<0>();
        }
        mfma_part(0, 1, 1);
        wg_barrier();
        // q2: quadrant (a1, c1)
        read_a(BUF, 1);
        if constexpr (SCHED == 0) {
            if constexpr (MODE == 0) stage(0, BUF, t + 2);
        } else {
            if constexpr (MODE < 2) piece(1, 1, BUF ^ 1, t + 1);
        }
        mfma_part(1, 1, -1);
        wg_barrier();
        // q3: quadrant (a1, c0)
        if constexpr (SCHED == 0) {
            if constexpr (MODE == 0) stage(2, BUF, t + 2);
        } else {
            if constexpr (MODE == 0) { stage(0, BUF, t + 2); stage(2, BUF, t + 2); }
        }
        if constexpr (MODE == 0) wait_vmcnt<8, NOWAIT>();
        if constexpr (MODE == 1) wait_vmcnt<4>();
        mfma_part(1, 0, -1);
        if (MODE < 2 || wm == 0) wg_barrier();  // group 1 skips the very last barrier (it entered one barrier late)
    };

    const int nk = K / BK;  // even, >= 2 (launcher)
    {   // De-phasing (flags bits 8..11 = number of phase groups G, 0/1 = off): every CU runs its tiles back to back and all
        // tiles take the same time, so without this all 256 CUs reach their epilogue together and the 256 KB x 256 store
        // burst runs at the HBM write rate while the matrix pipes idle.  The first workgroup of each CU (block ids < 256, one
        // workgroup per CU) waits phase/G of a main loop before it starts, phase = CU index within its XCD mod G; the offset
        // then persists for the whole launch.
        const int G = (flags >> 8) & 15;
        if (G > 1 && bid < 256) {
            const long wait = (long)((bid >> 3) % G) * ((long)nk * 8 * (NPL == 1 ? 288 : 420) / G);
            const long t0 = (long)__builtin_amdgcn_s_memtime();
            while ((long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }
    unsigned long long tstamp[4];
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (Cf) Cf += (size_t)blockIdx.y * strideC;
    if (C16) C16 += (size_t)blockIdx.y * strideC;
    const float* Rsrc = Rf ? Rf + (size_t)blockIdx.y * strideC : Cf;
    // the epilogue's transpose scratch: 8 KB per wave in the SECOND K-tile buffer (the first one receives the next tile's
    // K-tile 0 while the epilogue runs)
    float* ct = reinterpret_cast<float*>(smem + PP_BUF + wid * 8192);
    const int fl = EPI >= 0 ? EPI : flags;
    const bool do_gelu = fl & G16X_GELU, do_acc = fl & G16X_ACCUM;
    constexpr bool kBf16 = NPL == 1 && !F16;   // bf16 operands (GENIE_PREC_BF16)
    const bool out16 = fl & G16X_OUT16, outf = fl & G16X_OUTF32, nts = fl & G16X_NT;
    const float ascale = NPL == 2 ? alpha * (1.0f / 2048.0f) : alpha;

    // The tile's 256 bias values travel with its K-tile 0, by LDS-DMA into one of two 1 KB slots behind the ring (wave 0, one piece,
    // issued first = oldest: every counted wait below covers it): an ordinary global load in the epilogue would make the compiler
    // wait vmcnt(0) for it, i.e. for the next tile's K-tile 0 requested just before.
    const auto rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)(bias ? (const void*)bias : (const void*)W), 0, -1, 0x00020000);
    float* const bias_lds = reinterpret_cast<float*>(smem + 2 * PP_BUF);
    int bias_slot = 0;
    // G16X_QKNORM (the reference's default attention variant, attention.py:31-34, 42-47): gamma | beta of the per-head LayerNorm of q
    // and k (head_dim floats each, shared by all heads) sit behind the bias slots for the whole launch -- plain loads, before the
    // first LDS-DMA is requested (a global load later on would make the compiler drain the ring); visible after the first barrier
    constexpr bool kQkNorm = EPI >= 0 && (EPI & G16X_QKNORM) != 0;
    float* const norm_lds = bias_lds + 512;
    if constexpr (kQkNorm) {
        if (tid < head_dim) { norm_lds[tid] = qn_g[tid]; norm_lds[64 + tid] = qn_b[tid]; }
    }
    auto stage_bias = [&](int slot, int n0_) {
        if (bias && wid == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, (__attribute__((address_space(3))) void*)(bias_lds + slot * 256), 16,
                                                     (unsigned)(lane * 16), n0_ * 4, 0, 0);
    };
    // K-tile 0 of the first tile; every later tile's K-tile 0 is staged BEFORE the previous tile's epilogue, so that its
    // loads run ahead of the 256 KB of output stores instead of queueing behind them
    stage_bias(0, n0);
    stage(0, 0, 0);
    stage(2, 0, 0);
    stage(3, 0, 0);
    stage(1, 0, 0);
    for (;;) {
        if constexpr (ABL & 32) tstamp[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        in_loop = false;
        stage(0, 1, 1);
        stage(2, 1, 1);
        wait_vmcnt<8>();  // A-early(0), B-early(0) of this wave have landed (outstanding output stores of the previous tile only
                          // make a counted wait stricter: loads retire in order among loads)
        wg_barrier();
        if constexpr (ABL & 32) tstamp[1] = __builtin_amdgcn_s_memtime();
        if (wm == 1) wg_barrier();  // group 1 runs one barrier behind group 0
        if constexpr (ABL & 2) {  // fragments are read once, here
            wait_vmcnt<0>();
            wg_barrier();
            read_a(0, 0); read_b(0, 0); read_b(0, 1);
            scale_b(0); scale_b(1);
            wg_barrier();
        }
        in_loop = true;
        for (int t = 0; t + 2 < nk; t += 2) {
            ktile(I0{}, I0{}, t);
            ktile(I1{}, I0{}, t + 1);
        }
        ktile(I0{}, I1{}, nk - 2);
        ktile(I1{}, I2{}, nk - 1);
        in_loop = false;

        // ---- all LDS-DMA has landed and every fragment read has completed before any wave gets here (tail waits above)
        if constexpr (ABL & 32) tstamp[2] = __builtin_amdgcn_s_memtime();
        const int m0e = m0, n0e = n0;            // this tile's origin, for the epilogue
        const float* const bl = bias_lds + bias_slot * 256 + wn * 64;   // this tile's bias, the wave's 64 columns
        const int tile_id = bid;
        // residual-accumulate flavours: the FIRST round's residual rows are requested before the next tile's K-tile 0 (loads retire in
        // order: requested after it, their wait would also be a wait for those 64 KB)
        constexpr bool kResidualFirst = EPI >= 0 && (EPI & G16X_ACCUM) != 0 && (EPI & G16X_QKV) == 0 &&
                                        !((EPI & G16X_OUT16) != 0 && (EPI & (G16X_OUTF32 | G16X_ACCUM)) == 0);
        float4 res0[8];
        if constexpr (kResidualFirst) {
            const int colr = n0e + wn * 64 + ((lane & 15) << 2);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = m0e + wm * 128 + it * 4 + (lane >> 4);
                res0[it] = *reinterpret_cast<const float4*>(Rsrc + (size_t)row * ldc + colr);
            }
        }
        bid += gridDim.x;
        const bool more = bid < ntiles;
        if (more) {
            tile_origin(bid, m0, n0);
            rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, -1, 0x00020000);
            rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, -1, 0x00020000);
            bias_slot ^= 1;
            stage_bias(bias_slot, n0);
            stage(0, 0, 0);
            stage(2, 0, 0);
            stage(3, 0, 0);
            stage(1, 0, 0);
        }

        // ---- epilogue.  The accumulators hold one COLUMN per lane; each wave transposes its tile through its own 8 KB of
        // the ring's second buffer, 32 rows at a time, and then works on whole rows: 16 lanes x float4 = one 256-byte row
        // segment per quarter-wave for the residual read, the f32 store and the 16-bit operand store.
#ifndef GENIE_VAR_QKV_ABL
#define GENIE_VAR_QKV_ABL 0    // timing variants (results WRONG): 1 no plane stores, 2 no epilogue at all, 4 plain (not nt) stores
#endif
#define QKV_STORE(val, rs, vo, so)                                                                              \
    do {                                                                                                        \
        if constexpr ((GENIE_VAR_QKV_ABL & 1) != 0) asm volatile("" ::"v"(val));                                \
        else __builtin_amdgcn_raw_buffer_store_b128(val, rs, vo, so, (GENIE_VAR_QKV_ABL & 4) ? 0 : 2 /* nt */);  \
    } while (0)
        if constexpr (EPI >= 0 && (EPI & G16X_QKV) != 0 && (GENIE_VAR_QKV_ABL & 2) != 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[q][j]));
        } else if constexpr (EPI >= 0 && (EPI & G16X_QKV) != 0) {
            // ---- spatial-attention operand layout (N = 3d, d % 256 == 0, so a 256-column tile is all-Q, all-K or all-V; the
            // 256 rows of a tile are exactly one (clip, frame) sequence).  C16 holds 3*NPL planes of M*d 16-bit values:
            //   [Q planes | K planes | V^T planes], Q and K head-major [(sequence, head)][256 rows][head_dim] with Q multiplied by qscale (= scale * log2 e),
            //   V^T as [(sequence, head)][feature][256 keys] (keys of a 16-group stored {0-3, 8-11, 4-7, 12-15}): what
            //   kernels_attn_dma.hip streams straight into LDS.
            const int dm = N / 3;
            const int hd_shift = __builtin_ctz((unsigned)head_dim);   // (the launcher admits head_dim | 64 only)
            const int which = n0e / dm;                      // 0 Q, 1 K, 2 V  (block-uniform)
            const size_t P = (size_t)plane16;               // = M * d
            uint16_t* base = C16 + (size_t)which * NPL * P;
            if (which < 2) {
                // rows of 8 lanes x 8 columns: one 16-byte store per 16-bit plane and lane (the 8-byte form issued twice as
                // many store instructions for the same bytes, and the epilogue is store-ISSUE bound).  The transpose scratch
                // XORs its 16-byte slot with (row & 1) so that the two rows a 16-lane pass reads use disjoint banks.
                // Addresses: one buffer descriptor per tile at the sequence's block of the plane, ONE 32-bit lane offset, the
                // (round, row group) part as a scalar offset.  (Sixteen 64-bit lane addresses per branch, hoisted out of the
                // persistent loop, spilled -- and a scratch reload is a vmcnt wait behind the next tile's prefetch and every
                // earlier store.)  head_dim is a power of two (32 / 64 on this path): shifts, no integer division.
                const float qs = which == 0 ? qscale : 1.0f;
                const int c8 = (lane & 7) << 3;
                const int lrow = lane >> 3;
                const int colq = n0e - which * dm + wn * 64 + c8;
                const int hq = colq >> hd_shift, fq = colq & (head_dim - 1);
                const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)(m0e >> 8) * 256 * dm), 0, -1, 0x00020000);
                const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(base + P + (size_t)(m0e >> 8) * 256 * dm), 0, -1, 0x00020000);
                const int voff = ((((hq << 8) + wm * 128 + lrow) << hd_shift) + fq) * 2;
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                if (bias) { b0 = lds_ld4(bl + c8); b1 = lds_ld4(bl + c8 + 4); }
                float4 g0 = b0, g1 = b0, e0 = b0, e1 = b0;   // qk-norm: gamma / beta of this lane's 8 features
                if constexpr (kQkNorm) {
                    g0 = lds_ld4(norm_lds + fq); g1 = lds_ld4(norm_lds + fq + 4);
                    e0 = lds_ld4(norm_lds + 64 + fq); e1 = lds_ld4(norm_lds + 64 + fq + 4);
                }
                const float inv_hd = 1.0f / (float)head_dim;
                // sum over the lanes that hold one head's features of a row: 8 lanes x 8 columns (head_dim 64) or 4 (head_dim 32);
                // every lane of the group ends with the same bits (commutative pairings)
                auto head_sum = [&](float t) {
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
                    if (head_dim == 64)
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, false));  // row_half_mirror
                    return t;
                };
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int rw = (e & 3) + 8 * (e >> 2) + 4 * h;
                            ct[rw * 64 + ((j * 32 + r) ^ ((rw & 1) << 2))] = acc[q][j][e];
                        }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rl = it * 8 + lrow;
                        // head-major: [(sequence, head)][row of the sequence][feature] -- a head's 256 rows are one contiguous
                        // 256 * head_dim block, which is what the attention kernel's LDS-DMA pieces walk
                        // (element index ((sequence * heads + hq) * 256 + row) * head_dim + fq, row = wm * 128 + q * 32 + rl)
                        const int soff = ((q * 32 + it * 8) << hd_shift) * 2;
                        const int sw = (rl & 1) << 2;
                        float4 v = lds_ld4(ct + rl * 64 + (c8 ^ sw));
                        float4 u = lds_ld4(ct + rl * 64 + ((c8 + 4) ^ sw));
                        if constexpr (kQkNorm) {
                            // q = LN_head(q), k = LN_head(k): two-pass statistics over the head's features of this row (eps 1e-5,
                            // biased variance: nn.LayerNorm), then the shared affine, then (q only) scale * log2 e
                            v.x = v.x * ascale + b0.x; v.y = v.y * ascale + b0.y; v.z = v.z * ascale + b0.z; v.w = v.w * ascale + b0.w;
                            u.x = u.x * ascale + b1.x; u.y = u.y * ascale + b1.y; u.z = u.z * ascale + b1.z; u.w = u.w * ascale + b1.w;
                            const float mean = head_sum(((v.x + v.y) + (v.z + v.w)) + ((u.x + u.y) + (u.z + u.w))) * inv_hd;
                            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean; u.x -= mean; u.y -= mean; u.z -= mean; u.w -= mean;
                            const float var = head_sum(((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) +
                                                       ((u.x * u.x + u.y * u.y) + (u.z * u.z + u.w * u.w))) * inv_hd;
                            const float rstd = 1.0f / sqrtf(var + 1e-5f);
                            v.x = (v.x * rstd * g0.x + e0.x) * qs; v.y = (v.y * rstd * g0.y + e0.y) * qs;
                            v.z = (v.z * rstd * g0.z + e0.z) * qs; v.w = (v.w * rstd * g0.w + e0.w) * qs;
                            u.x = (u.x * rstd * g1.x + e1.x) * qs; u.y = (u.y * rstd * g1.y + e1.y) * qs;
                            u.z = (u.z * rstd * g1.z + e1.z) * qs; u.w = (u.w * rstd * g1.w + e1.w) * qs;
                        } else {
                        v.x = (v.x * ascale + b0.x) * qs; v.y = (v.y * ascale + b0.y) * qs;
                        v.z = (v.z * ascale + b0.z) * qs; v.w = (v.w * ascale + b0.w) * qs;
                        u.x = (u.x * ascale + b1.x) * qs; u.y = (u.y * ascale + b1.y) * qs;
                        u.z = (u.z * ascale + b1.z) * qs; u.w = (u.w * ascale + b1.w) * qs;
                        }
                        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                        if constexpr (NPL == 1) {
                            const u4v t = {(uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16),
                                           (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16),
                                           (uint32_t)f32_to_bf16(u.x) | ((uint32_t)f32_to_bf16(u.y) << 16),
                                           (uint32_t)f32_to_bf16(u.z) | ((uint32_t)f32_to_bf16(u.w) << 16)};
                            QKV_STORE(t, rsH, voff, soff);
                        } else {
                            uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
                            split_f16_x4(v.x, v.y, v.z, v.w, h01, h23, l01, l23);
                            split_f16_x4(u.x, u.y, u.z, u.w, h45, h67, l45, l67);
                            const u4v th = {h01, h23, h45, h67}, tl = {l01, l23, l45, l67};
                            QKV_STORE(th, rsH, voff, soff);
                            QKV_STORE(tl, rsL, voff, soff);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                // V tiles: the accumulators already hold one COLUMN (feature) per lane with 4 consecutive rows (keys) per
                // register group, so they go to LDS as [feature][32 keys] with float4 writes (16-byte slot XOR (feature & 7):
                // conflict-free on both sides) and come back as 8 consecutive keys of one feature per lane = one 16-byte
                // store per 16-bit plane.
                // element index ((sequence * heads + head) * head_dim + f) * 256 + key = (sequence * d + feature column) * 256 + key
                const int fl4 = lane >> 2, kq = lane & 3;
                const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)(m0e >> 8) * 256 * dm), 0, -1, 0x00020000);
                const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(base + P + (size_t)(m0e >> 8) * 256 * dm), 0, -1, 0x00020000);
                const int voff = (((n0e - 2 * dm + wn * 64 + fl4) << 8) + wm * 128 + kq * 8) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = j * 32 + r, ks = 2 * g + h;
                            lds_st4(ct + c * 32 + ((ks ^ (c & 7)) << 2), acc[q][j][4 * g], acc[q][j][4 * g + 1], acc[q][j][4 * g + 2],
                                    acc[q][j][4 * g + 3]);
                        }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int c = it * 16 + fl4;                                 // feature column inside the wave's 64
                        // key order inside a 16-key group is the PV operand's own: unit j (16 bytes) = keys {4j..4j+3, 8+4j..8+4j+3},
                        // so the attention kernel fetches a lane's 8 keys with ONE conflict-free ds_read_b128
                        const int ga = 4 * (kq >> 1) + (kq & 1);
                        const float4 a = lds_ld4(ct + c * 32 + ((ga ^ (c & 7)) << 2));
                        const float4 b = lds_ld4(ct + c * 32 + (((ga + 2) ^ (c & 7)) << 2));
                        const float bb = bias ? bl[c] : 0.f;
                        const int soff = (((it * 16) << 8) + q * 32) * 2;
                        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                        const float v0 = a.x * ascale + bb, v1 = a.y * ascale + bb, v2 = a.z * ascale + bb, v3 = a.w * ascale + bb;
                        const float v4 = b.x * ascale + bb, v5 = b.y * ascale + bb, v6 = b.z * ascale + bb, v7 = b.w * ascale + bb;
                        if constexpr (NPL == 1) {
                            const u4v t = {(uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16),
                                           (uint32_t)f32_to_bf16(v2) | ((uint32_t)f32_to_bf16(v3) << 16),
                                           (uint32_t)f32_to_bf16(v4) | ((uint32_t)f32_to_bf16(v5) << 16),
                                           (uint32_t)f32_to_bf16(v6) | ((uint32_t)f32_to_bf16(v7) << 16)};
                            QKV_STORE(t, rsH, voff, soff);
                        } else {
                            uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
                            split_f16_x4(v0, v1, v2, v3, h01, h23, l01, l23);
                            split_f16_x4(v4, v5, v6, v7, h45, h67, l45, l67);
                            const u4v th = {h01, h23, h45, h67}, tl = {l01, l23, l45, l67};
                            QKV_STORE(th, rsH, voff, soff);
                            QKV_STORE(tl, rsL, voff, soff);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else if constexpr (EPI >= 0 && (EPI & G16X_OUT16) != 0 && (EPI & (G16X_OUTF32 | G16X_ACCUM)) == 0) {
            // 16-bit output only (fc1: GELU + operand planes): 8 columns per lane
            // rows of 8 lanes x 8 columns (see the Q / K path above): 16-byte stores for the 16-bit planes, two adjacent float4
            // for the f32 output and the residual read
            const int c8 = (lane & 7) << 3;
            const int col8 = n0e + wn * 64 + c8;
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (bias) { b0 = lds_ld4(bl + c8); b1 = lds_ld4(bl + c8 + 4); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rw = (e & 3) + 8 * (e >> 2) + 4 * h;
                        ct[rw * 64 + ((j * 32 + r) ^ ((rw & 1) << 2))] = acc[q][j][e];
                    }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = it * 8 + (lane >> 3);
                    const int row = m0e + wm * 128 + q * 32 + rl;
                    const int sw = (rl & 1) << 2;
                    float4 v = lds_ld4(ct + rl * 64 + (c8 ^ sw));
                    float4 u = lds_ld4(ct + rl * 64 + ((c8 + 4) ^ sw));
                    v.x = v.x * ascale + b0.x; v.y = v.y * ascale + b0.y; v.z = v.z * ascale + b0.z; v.w = v.w * ascale + b0.w;
                    u.x = u.x * ascale + b1.x; u.y = u.y * ascale + b1.y; u.z = u.z * ascale + b1.z; u.w = u.w * ascale + b1.w;
                    auto gelu4 = [](float4& t, auto lowp) {
                        const genie_f2 g0 = pp_gelu2<decltype(lowp)::value>(genie_f2{t.x, t.y}), g1 = pp_gelu2<decltype(lowp)::value>(genie_f2{t.z, t.w});
                        t.x = g0[0]; t.y = g0[1]; t.z = g1[0]; t.w = g1[1];
                    };
                    if (do_gelu) {
                        if (kBf16 && !outf) { gelu4(v, std::true_type{}); gelu4(u, std::true_type{}); }
                        else { gelu4(v, std::false_type{}); gelu4(u, std::false_type{}); }
                    }
                    const size_t idx = (size_t)row * ldc + col8;
                    if (do_acc) {
                        const float4 o0 = *reinterpret_cast<const float4*>(Rsrc + idx);
                        const float4 o1 = *reinterpret_cast<const float4*>(Rsrc + idx + 4);
                        v.x += o0.x; v.y += o0.y; v.z += o0.z; v.w += o0.w;
                        u.x += o1.x; u.y += o1.y; u.z += o1.z; u.w += o1.w;
                    }
                    if constexpr (ABL & 8) {
                        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
                        continue;
                    }
                    if (outf) {
                        if (nts) {
                            typedef float nt4 __attribute__((ext_vector_type(4)));
                            nt4 t0 = {v.x, v.y, v.z, v.w}, t1 = {u.x, u.y, u.z, u.w};
                            __builtin_nontemporal_store(t0, reinterpret_cast<nt4*>(Cf + idx));
                            __builtin_nontemporal_store(t1, reinterpret_cast<nt4*>(Cf + idx + 4));
                        } else {
                            *reinterpret_cast<float4*>(Cf + idx) = v;
                            *reinterpret_cast<float4*>(Cf + idx + 4) = u;
                        }
                    }
                    if (out16) {
                        if (fl & G16X_GELU16) { gelu4(v, std::integral_constant<bool, kBf16>{}); gelu4(u, std::integral_constant<bool, kBf16>{}); }
                        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
                        auto st4 = [&](uint16_t* p_, uint32_t a, uint32_t b, uint32_t c, uint32_t dd) {
                            u4v t = {a, b, c, dd};
                            if (nts) __builtin_nontemporal_store(t, reinterpret_cast<u4v*>(p_));
                            else *reinterpret_cast<u4v*>(p_) = t;
                        };
                        const bool split_out = EPI >= 0 ? (NPL == 2) : (plane16 != 0);  // compile-time in the EPI instantiations
                        if (!split_out) {  // bf16 output
                            st4(C16 + idx, (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16),
                                (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16),
                                (uint32_t)f32_to_bf16(u.x) | ((uint32_t)f32_to_bf16(u.y) << 16),
                                (uint32_t)f32_to_bf16(u.z) | ((uint32_t)f32_to_bf16(u.w) << 16));
                        } else {             // split f16 planes [hi | lo]
                            uint32_t h01, h23, l01, l23, h45, h67, l45, l67;
                            split_f16_x4(v.x, v.y, v.z, v.w, h01, h23, l01, l23);
                            split_f16_x4(u.x, u.y, u.z, u.w, h45, h67, l45, l67);
                            st4(C16 + idx, h01, h23, h45, h67);
                            st4(C16 + (size_t)plane16 + idx, l01, l23, l45, l67);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next round
            }
        } else {
            // f32 output (alone or with 16-bit planes): 4 columns per lane, 16 lanes per row -- every f32 store instruction
            // covers whole 256-byte row segments (8 columns per lane would interleave two instructions inside each 32 bytes:
            // measured 10-25 % slower on the f32-output flavours)
            const int c4 = (lane & 15) << 2;
            const int col = n0e + wn * 64 + c4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = lds_ld4(bl + c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ct[((e & 3) + 8 * (e >> 2) + 4 * h) * 64 + j * 32 + r] = acc[q][j][e];
                // the round's eight residual rows are requested together, ahead of the stores: the residual and the output may
                // be the same buffer, so the compiler would keep each read behind the previous row's store (one exposed memory
                // round trip per row); the rows of a round are distinct, so reading them all first is safe in place too
                float4 res[8];
                if (do_acc) {
                    if (kResidualFirst && q == 0) {
#pragma unroll
                        for (int it = 0; it < 8; ++it) res[it] = res0[it];
                    } else {
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int row = m0e + wm * 128 + q * 32 + it * 4 + (lane >> 4);
                            res[it] = *reinterpret_cast<const float4*>(Rsrc + (size_t)row * ldc + col);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rl = it * 4 + (lane >> 4);
                    const int row = m0e + wm * 128 + q * 32 + rl;
                    float4 v = lds_ld4(ct + rl * 64 + c4);
                    v.x = v.x * ascale + bv.x; v.y = v.y * ascale + bv.y; v.z = v.z * ascale + bv.z; v.w = v.w * ascale + bv.w;
                    if (do_gelu) {
                        if (kBf16 && !outf) {
                            const genie_f2 g0 = pp_gelu2<true>(genie_f2{v.x, v.y}), g1 = pp_gelu2<true>(genie_f2{v.z, v.w});
                            v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
                        } else {
                            const genie_f2 g0 = pp_gelu2<false>(genie_f2{v.x, v.y}), g1 = pp_gelu2<false>(genie_f2{v.z, v.w});
                            v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
                        }
                    }
                    const size_t idx = (size_t)row * ldc + col;
                    if (do_acc) {
                        const float4 o = res[it];
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if constexpr (ABL & 8) {
                        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                        continue;
                    }
                    if (outf) {
                        if (nts) {
                            typedef float nt4 __attribute__((ext_vector_type(4)));
                            nt4 t = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(t, reinterpret_cast<nt4*>(Cf + idx));
                        } else {
                            *reinterpret_cast<float4*>(Cf + idx) = v;
                        }
                    }
                    if (out16) {
                        if (fl & G16X_GELU16) {
                            const genie_f2 g0 = pp_gelu2<kBf16>(genie_f2{v.x, v.y}), g1 = pp_gelu2<kBf16>(genie_f2{v.z, v.w});
                            v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
                        }
                        typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                        auto st2 = [&](uint16_t* p_, uint32_t a, uint32_t b) {
                            u2v t = {a, b};
                            if (nts) __builtin_nontemporal_store(t, reinterpret_cast<u2v*>(p_));
                            else *reinterpret_cast<u2v*>(p_) = t;
                        };
                        const bool split_out = EPI >= 0 ? (NPL == 2) : (plane16 != 0);  // compile-time in the EPI instantiations
                        if (!split_out) {  // bf16 output
                            st2(C16 + idx, (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16),
                                (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16));
                        } else {             // split f16 planes [hi | lo]
                            uint32_t h01, h23, l01, l23;
                            split_f16_x4(v.x, v.y, v.z, v.w, h01, h23, l01, l23);
                            st2(C16 + idx, h01, h23);
                            st2(C16 + (size_t)plane16 + idx, l01, l23);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next round
            }
        }
        if constexpr (ABL & 32) {
            tstamp[3] = __builtin_amdgcn_s_memtime();
            if (tid == 0 && g_pp_timing) {
                unsigned long long* o = g_pp_timing + 4 * (size_t)tile_id;
                o[0] = tstamp[0]; o[1] = tstamp[1]; o[2] = tstamp[2]; o[3] = tstamp[3];
            }
        }
        if (!more) break;
        // every wave is done with its transpose scratch before K-tile 1 of the next tile is staged over it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
    }
}

// Returns GENIE_E_UNSUPPORTED when the problem does not fit this kernel's tiling (the caller falls back to the
// gemm16_v2 / gemm16_nt kernels of kernels_bf16.hip).  npl = 1: bf16 operands (f16 = 0) or the hi plane of split
// operands (f16 = 1); npl = 2: split-f16 operands, terms = 3 (f32-class) or 2.
int launch_gemm16_pp(int npl, int terms, int f16, const uint16_t* A, long lda, long planeA, const uint16_t* W, long ldw,
                     long planeW, const float* bias, const float* Rf, float* Cf, uint16_t* C16, long plane16, long ldc, int M,
                     int N, int K, int flags, float alpha, hipStream_t st, int batch, long strideA, long strideW, long strideC,
                     float qscale, int head_dim, const float* qn_g, const float* qn_b) {
    const int bk = npl == 1 ? 64 : 32;
    if ((flags & G16X_QKNORM) && (!(flags & G16X_QKV) || !qn_g || !qn_b || (head_dim != 32 && head_dim != 64))) return GENIE_E_UNSUPPORTED;
    if (flags & G16X_QKV) {  // C16 = [Q | K | V^T] operand planes of the spatial attention (plane16 = M * d, ldc unused)
        if (N % 3 || (N / 3) % 256 || M % 256 || batch != 1 || head_dim <= 0 || (N / 3) % head_dim || 64 % head_dim ||
            !(flags & G16X_OUT16) || (flags & (G16X_ACCUM | G16X_GELU | G16X_OUTF32 | G16X_GELU16)) || f16 != (npl == 2) ||
            terms != 3)
            return GENIE_E_UNSUPPORTED;
        static const bool study_off = study_env("GENIE_PP_ABL", 0) || !study_env("GENIE_PP_EPI", 1) || study_env("GENIE_PP_SCHED", 0);
        if (study_off) return GENIE_E_UNSUPPORTED;  // study knobs: no QKV variant
        flags |= G16X_NT;
    }
    if (!kStudyBuild && ((npl == 1 && f16) || (npl == 2 && terms != 3))) return GENIE_E_UNSUPPORTED;  // study-only forms
    if (M < 256 || M % 256 || N % 256 || K % (2 * bk) || K < 2 * bk) return GENIE_E_UNSUPPORTED;
    if (lda % 8 || ldw % 8 || ldc % 4 || ((flags & G16X_OUT16) && !(flags & (G16X_QKV | G16X_OUTF32 | G16X_ACCUM)) && ldc % 8))
        return GENIE_E_UNSUPPORTED;  // (16-byte stores of the 16-bit-only epilogue)
    // 32-bit byte offsets inside one tile's buffer descriptor
    if ((double)(npl - 1) * planeA * 2 + 256.0 * lda * 2 + 2.0 * K >= 4.0e9) return GENIE_E_UNSUPPORTED;
    if ((double)(npl - 1) * planeW * 2 + 256.0 * ldw * 2 + 2.0 * K >= 4.0e9) return GENIE_E_UNSUPPORTED;
    const long tiles = (long)(M / 256) * (N / 256) * batch;
    static const long min_tiles = study_env("GENIE_GEMM16_PP_MIN_TILES", (int)(192));
    if (tiles < min_tiles) return GENIE_E_UNSUPPORTED;
    const double mn = (double)M * N * batch;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * mn * K,
                   2.0 * npl * ((double)M * K * batch + (double)N * K) +
                       mn * ((flags & G16X_ACCUM ? 4 : 0) + (flags & G16X_OUTF32 ? 4 : 0) +
                             (flags & G16X_OUT16 ? (plane16 ? 4 : 2) : 0)),
                   st, npl == 2 ? (terms == 3 ? "gemm16_pp_kernel<2,3,true,...> (256x256 tile, two-group phase schedule, 3x v_mfma_f32_32x32x16_f16 per algorithmic MFMA into one accumulator)" : "gemm16_pp_kernel<2,2,...> (STUDY: 2 of 3 split terms)")
                                 : (f16 ? "gemm16_pp_kernel<1,1,true,...> (STUDY: plain f16)" : "gemm16_pp_kernel<1,1,false,...> (256x256 tile, two-group phase schedule, v_mfma_f32_32x32x16_bf16)"));
    // persistent: one workgroup per CU (128 KB of LDS: one fits), each walking tiles bid, bid + grid.x, ...
    static const bool persist = study_env("GENIE_PP_PERSIST", 1) != 0;   // 0: one workgroup per tile (the non-persistent launch, for A/B runs)
    const int n_cu = persist ? device_cu_count() : 1 << 30;            // (per device: common.hpp)
    const long tiles_x = (long)(M / 256) * (N / 256);
    const dim3 grid((unsigned)(tiles_x < n_cu ? tiles_x : n_cu), (unsigned)batch);
    static const int stagger = study_env("GENIE_PP_STAGGER", 0);
    static const long stagger_min = study_env("GENIE_PP_STAGGER_MIN_TILES", (int)(1024));
    if (stagger > 1 && batch == 1 && tiles >= stagger_min) flags |= (stagger & 15) << 8;
    constexpr size_t lds = 2 * PP_BUF + 2048 + 512;   // ring + two 1 KB bias slots + gamma | beta of the qk-norm epilogue
#define PP_LAUNCH(NPL_, TERMS_, F16_, ABL_, SCHED_)                                                                       \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)gemm16_pp_kernel<NPL_, TERMS_, F16_, ABL_, SCHED_>,                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
        gemm16_pp_kernel<NPL_, TERMS_, F16_, ABL_, SCHED_><<<grid, 512, lds, st>>>(                                       \
            A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf,     \
            strideW, qscale, head_dim, qn_g, qn_b);                                                                                                     \
    } while (0)
#define PP_LAUNCH_ABL(ABL_)                                                                                               \
    do {                                                                                                                  \
        if (npl == 1) PP_LAUNCH(1, 1, false, ABL_, 0);                                                                    \
        else PP_LAUNCH(2, 3, true, ABL_, 0);                                                                              \
    } while (0)
    static const int sched = study_env("GENIE_PP_SCHED", 0);
#ifdef GENIE_STUDY
    static const int abl = study_env("GENIE_PP_ABL", 0);
    const size_t n_wg = (size_t)tiles_x;
    static unsigned long long* tbuf = nullptr;
    if ((abl & 32) && !tbuf) {
        (void)hipMalloc(&tbuf, sizeof(unsigned long long) * 4 * 65536);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pp_timing), &tbuf, sizeof(tbuf));
    }
    if ((abl & 32) && n_wg > 65536) return GENIE_E_UNSUPPORTED;   // the stamp buffer holds 65536 tiles
    if (abl == 1) PP_LAUNCH_ABL(1);
    else if (abl == 2) PP_LAUNCH_ABL(2);
    else if (abl == 3) PP_LAUNCH_ABL(3);
    else if (abl == 8) PP_LAUNCH_ABL(8);
    else if (abl == 16) PP_LAUNCH_ABL(16);
    else if (abl == 11) PP_LAUNCH_ABL(11);
    else if (abl == 32) PP_LAUNCH_ABL(32);
    else if (abl == 256) PP_LAUNCH_ABL(256);   // energy study: all operand reads are L2 hits
    else if (abl == 96) PP_LAUNCH_ABL(96);     // stamps + nt on the A stream
    else if (abl == 160) PP_LAUNCH_ABL(160);   // stamps + nt on the W stream
    else if (abl == 224) PP_LAUNCH_ABL(224);   // stamps + nt on both
    else if (abl == 33) {  // stamps with the compile-time OUTF32 | NT epilogue
        if (npl == 1) { (void)hipFuncSetAttribute((const void*)gemm16_pp_kernel<1, 1, false, 32, 0, G16X_OUTF32 | G16X_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            gemm16_pp_kernel<1, 1, false, 32, 0, G16X_OUTF32 | G16X_NT><<<grid, 512, lds, st>>>(A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf, strideW, qscale, head_dim, qn_g, qn_b); }
        else { (void)hipFuncSetAttribute((const void*)gemm16_pp_kernel<2, 3, true, 32, 0, G16X_OUTF32 | G16X_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            gemm16_pp_kernel<2, 3, true, 32, 0, G16X_OUTF32 | G16X_NT><<<grid, 512, lds, st>>>(A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf, strideW, qscale, head_dim, qn_g, qn_b); }
    }
    else
#endif
    {
        // compile-time epilogues for the model's Linear flavours (qkv / readout: OUTF32; proj, fc2: ACCUM | OUTF32 [| OUT16];
        // fc1: GELU | OUT16; bf16 temporal qkv: OUT16; the training forward's fc1: OUTF32 | OUT16 | GELU16), each with and without
        // non-temporal stores; anything else takes the run-time-flag kernel (which spills: 80-236 bytes per lane)
        const int e = flags & 255;
#define PP_EPI(NPL_, F16_, E_)                                                                                            \
        case E_: {                                                                                                        \
            (void)hipFuncSetAttribute((const void*)gemm16_pp_kernel<NPL_, (NPL_ == 2 ? 3 : 1), F16_, 0, PP_SCHED, E_>,    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            gemm16_pp_kernel<NPL_, (NPL_ == 2 ? 3 : 1), F16_, 0, PP_SCHED, E_><<<grid, 512, lds, st>>>(                   \
                A, lda, planeA, W, ldw, planeW, bias, Cf, C16, plane16, ldc, M, N, K, flags, alpha, strideA, strideC, Rf,  \
                strideW, qscale, head_dim, qn_g, qn_b);                                                                                                 \
            done = true;                                                                                                  \
        } break;
#define PP_EPI_ALL(NPL_, F16_)                                                                                            \
        switch (e) {                                                                                                      \
            PP_EPI(NPL_, F16_, G16X_OUTF32)                                                                               \
            PP_EPI(NPL_, F16_, G16X_OUTF32 | G16X_NT)                                                                     \
            PP_EPI(NPL_, F16_, G16X_ACCUM | G16X_OUTF32)                                                                  \
            PP_EPI(NPL_, F16_, G16X_ACCUM | G16X_OUTF32 | G16X_NT)                                                        \
            PP_EPI(NPL_, F16_, G16X_ACCUM | G16X_OUTF32 | G16X_OUT16)                                                     \
            PP_EPI(NPL_, F16_, G16X_ACCUM | G16X_OUTF32 | G16X_OUT16 | G16X_NT)                                           \
            PP_EPI(NPL_, F16_, G16X_OUT16)                                                                                \
            PP_EPI(NPL_, F16_, G16X_OUT16 | G16X_NT)                                                                      \
            PP_EPI(NPL_, F16_, G16X_GELU | G16X_OUT16)                                                                    \
            PP_EPI(NPL_, F16_, G16X_GELU | G16X_OUT16 | G16X_NT)                                                          \
            PP_EPI(NPL_, F16_, G16X_OUT16 | G16X_QKV | G16X_NT)                                                           \
            PP_EPI(NPL_, F16_, G16X_OUT16 | G16X_QKV | G16X_NT | G16X_QKNORM)   /* + per-head LayerNorm of q and k */      \
            PP_EPI(NPL_, F16_, G16X_OUTF32 | G16X_OUT16 | G16X_GELU16)         /* training forward fc1 */                 \
            PP_EPI(NPL_, F16_, G16X_OUTF32 | G16X_OUT16 | G16X_GELU16 | G16X_NT)                                          \
            default: break;                                                                                               \
        }
        bool done = false;
        static const int epi = study_env("GENIE_PP_EPI", 1);
        const bool out_kind_ok = !(flags & G16X_OUT16) || (flags & G16X_QKV) || (npl == 2 ? plane16 != 0 : plane16 == 0);
        if (epi && sched == 0 && terms == 3 && out_kind_ok) {
#define PP_SCHED 0
            if (npl == 1 && !f16) { PP_EPI_ALL(1, false) }
            else if (npl == 2) { PP_EPI_ALL(2, true) }
#undef PP_SCHED
        }
#undef PP_EPI_ALL
#undef PP_EPI
        if (done) {
        }
#ifdef GENIE_STUDY
        else if (npl == 1 && !f16 && sched) { if (sched == 2) PP_LAUNCH(1, 1, false, 0, 2); else PP_LAUNCH(1, 1, false, 0, 1); }
        else if (npl == 1 && f16) PP_LAUNCH(1, 1, true, 0, 0);
        else if (npl == 2 && terms == 2) PP_LAUNCH(2, 2, true, 0, 0);
        else if (npl == 2 && sched) { if (sched == 2) PP_LAUNCH(2, 3, true, 0, 2); else PP_LAUNCH(2, 3, true, 0, 1); }
#endif
        else if (npl == 1 && !f16) PP_LAUNCH(1, 1, false, 0, 0);
        else if (npl == 2 && terms == 3) PP_LAUNCH(2, 3, true, 0, 0);
        else return GENIE_E_UNSUPPORTED;   // plain-f16 / 2-term forms exist in the study build only
    }
#undef PP_LAUNCH
#undef PP_LAUNCH_ABL
#ifdef GENIE_STUDY
    if ((abl & 32) && n_wg <= 16384 && batch == 1) {  // debug study: average the stamps of this launch (synchronises!)
        static unsigned long long host[4 * 16384];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, tbuf, sizeof(unsigned long long) * 4 * n_wg, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < n_wg; ++i) {
            pro += (double)(host[4 * i + 1] - host[4 * i]);
            loop += (double)(host[4 * i + 2] - host[4 * i + 1]);
            epi += (double)(host[4 * i + 3] - host[4 * i + 2]);
            if (host[4 * i] < t0) t0 = host[4 * i];
            if (host[4 * i + 3] > t1) t1 = host[4 * i + 3];
        }
        fprintf(stderr, "pp_timing npl=%d M=%d N=%d K=%d wgs=%zu: prologue %.0f  main loop %.0f  epilogue+drain %.0f  (s_memtime "
                        "ticks per workgroup, 100 MHz clock?)  span %.0f\n",
                npl, M, N, K, n_wg, pro / n_wg, loop / n_wg, epi / n_wg, (double)(t1 - t0));
    }
#endif
    GENIE_LAUNCH_CHECK("gemm16_pp");
    return GENIE_OK;
}

}  // namespace genie
