// Weight-gradient GEMM of the bf16 training step straight from the ROW-MAJOR operands (no transposed copies):
//   dW[N][K] (beta * dW +)= alpha * sum_m dY[m][n] * X[m][k]        (reference: autograd of nn.Linear, st_transformer.py:16-83)
// Both operands are contracted over their ROW index, so the matrix-core fragments (8 consecutive contraction elements per lane)
// are columns of the staged tiles: they are fetched with ds_read_b64_tr_b16, the CDNA4 transposing LDS read -- within a
// 16-lane group lane c supplies the address of an 8-byte chunk, and lane i receives element (i & 3) of the chunks 4j + (i >> 2),
// j = 0..3.  With chunk c = (row c >> 2, columns 4 (c & 3) .. + 3) lane i ends up with rows 0..3 of column i: four consecutive
// contraction elements of its own column.  Two such reads make one bf16x8 operand of v_mfma_f32_32x32x16_bf16.
//   tile 128 (n) x 128 (k), 4 waves of 64 x 64, 64 token rows per stage (16 KB per operand), 2 stages, 2 workgroups per CU;
//   LDS rows are 256 B; their 32-byte units are XOR-swizzled with (row & 7) on the DMA source address so the four rows a
//   16-lane group reads fall into different banks.
//   The token axis is cut into `ns` slabs (one per workgroup row), the slabs of a weight are added in order by
//   slab_reduce_kernel: fixed summation order, bit-reproducible.  Slabs that share an XCD read the same operand rows.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace genie {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void wgrad16_tn_kernel(const uint16_t* __restrict__ dY, long ldy,
                                                            const uint16_t* __restrict__ X, long ldx, float* __restrict__ out,
                                                            int N, int K, int rows_per_slab, int ns, float alpha) {
    constexpr int TS = 64;               // token rows per stage
    constexpr int OP_B = TS * 256;       // one operand's stage: 64 rows x 128 columns bf16
    constexpr int STAGE_B = 2 * OP_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid >> 1, wk = wid & 1;
    const int tiles_k = K >> 7, tiles = (N >> 7) * tiles_k;
    int slab, tile;
    {
        const int id = blockIdx.x;
        if ((ns & 7) == 0) {  // workgroup id & 7 = XCD: the tiles of a slab share an L2
            const int xcd = id & 7, j = id >> 3;
            slab = xcd + 8 * (j / tiles);
            tile = j % tiles;
        } else {
            slab = id / tiles;
            tile = id - slab * tiles;
        }
    }
    const int n0 = (tile / tiles_k) << 7, k0 = (tile % tiles_k) << 7;
    const long m_begin = (long)slab * rows_per_slab;

    // ---- DMA: chunk c = wid * 4 + i (1 KB = 4 rows), lane -> row c*4 + (lane >> 4), physical 16-byte slot lane & 15
    long offY[4], offX[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m_local = (wid * 4 + i) * 4 + (lane >> 4);
        const int lu = ((lane & 15) >> 1) ^ (m_local & 7);  // logical 32-byte unit held by this physical slot
        const int col = lu * 16 + (lane & 1) * 8;
        offY[i] = (long)m_local * ldy + n0 + col;
        offX[i] = (long)m_local * ldx + k0 + col;
    }
    auto issue = [&](int st) {
        unsigned char* base = smem + (st & 1) * STAGE_B;
        const uint16_t* y = dY + (size_t)(m_begin + (long)st * TS) * ldy;
        const uint16_t* x = X + (size_t)(m_begin + (long)st * TS) * ldx;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(y + offY[i]),
                                             (__attribute__((address_space(3))) void*)(base + (wid * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + offX[i]),
                                             (__attribute__((address_space(3))) void*)(base + OP_B + (wid * 4 + i) * 1024), 16, 0,
                                             0);
    };

    // ---- fragment addresses (bytes inside an operand's stage, K-step 0): two transposing reads per 32-column block
    const int g = lane >> 4, c = lane & 15;
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = 8 * (g >> 1) + (c >> 2) + 4 * q;
            const int colA = wn * 64 + b * 32 + 16 * (g & 1) + 4 * (c & 3);
            const int colB = wk * 64 + b * 32 + 16 * (g & 1) + 4 * (c & 3);
            fa[b][q] = row * 256 + (((colA >> 4) ^ (row & 7)) << 5) + (colA & 15) * 2;
            fb[b][q] = row * 256 + (((colB >> 4) ^ (row & 7)) << 5) + (colB & 15) * 2;
        }
    auto frag = [&](const unsigned char* base, const int (&f)[2]) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + f[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + f[1]));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nst = rows_per_slab / TS;
    issue(0);
    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage st has landed everywhere; everyone is done with stage st - 1
        if (st + 1 < nst) issue(st + 1);
        const unsigned char* sy = smem + (st & 1) * STAGE_B;
        const unsigned char* sx = sy + OP_B;
#pragma unroll
        for (int kk = 0; kk < TS / 16; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = frag(sy + kk * 16 * 256, fa[i]);
                b[i] = frag(sx + kk * 16 * 256, fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- the slab's tile: lanes hold consecutive k, so the stores of a register are 128-byte runs
    float* o = out + (size_t)slab * N * K;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                o[(size_t)n * K + k0 + wk * 64 + j * 32 + r] = alpha * acc[i][j][e];
            }
}

// GENIE_E_UNSUPPORTED (nothing launched) unless N % 128 == 0, K % 128 == 0, Mtok % 64 == 0 and 16-byte aligned rows.
int launch_wgrad16_tn(const uint16_t* dY, long ldy, const uint16_t* X, long ldx, float* dW, int Mtok, int N, int K, float alpha,
                      float beta, float* slabs, size_t slab_floats, hipStream_t st) {
    if (N <= 0 || K <= 0 || Mtok <= 0 || N % 128 || K % 128 || Mtok % 64 || ldy % 8 || ldx % 8) return GENIE_E_UNSUPPORTED;
    if ((size_t)N * K > slab_floats) return GENIE_E_UNSUPPORTED;
    GENIE_CHECK_ARG(beta == 0.f || beta == 1.f, "wgrad: beta must be 0 or 1");
    const int tiles = (N / 128) * (K / 128);
    static const int want = study_env("GENIE_TN_WGS", 512);  // 2 per CU: 256 / 1024 / 2048 measured 5 / 2 / 8 % slower
    int ns = 1;
    while (ns < 64 && tiles * ns < want && Mtok % (64 * ns * 2) == 0 && (size_t)(ns * 2) * N * K <= slab_floats) ns *= 2;
    const size_t lds = 2 * 2 * 64 * 256;
    ProfScope prof(GENIE_KC_GEMM, 2.0 * Mtok * (double)N * K, 2.0 * Mtok * ((double)N + K) + 8.0 * ns * (double)N * K, st);
    (void)hipFuncSetAttribute((const void*)wgrad16_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    wgrad16_tn_kernel<<<tiles * ns, 256, lds, st>>>(dY, ldy, X, ldx, slabs, N, K, Mtok / ns, ns, alpha);
    GENIE_LAUNCH_CHECK("wgrad16_tn");
    return launch_slab_reduce(slabs, ns, (size_t)N * K, dW, beta, st);
}

}  // namespace genie
