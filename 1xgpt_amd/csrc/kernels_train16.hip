// 16-bit operand plumbing of the training step's matrix-core contractions (GENIE_PREC_BF16 / GENIE_PREC_F16X3).
// The NT GEMM of kernels_bf16.hip wants both operands with the contraction axis contiguous.  Forward and dgrad get
// that from row-major activations and (transposed) weight copies; wgrad contracts over TOKENS, so it needs dY^T and
// X^T: the kernels here make the 16-bit copies -- bf16, or f16 split planes [hi | lo] with a ~ hi + lo/2048 -- in
// both orientations in one pass over the f32 source (64x64 tiles through LDS, 128-byte rows on both sides).
#include "kernels.hpp"

namespace genie {

template <int NPL>
__device__ __forceinline__ void to16(float v, uint16_t& hi, uint16_t& lo) {
    if constexpr (NPL == 1) { hi = f32_to_bf16(v); lo = 0; }
    else split_f16(v, hi, lo);
}

// MODE 0: v = in;  MODE 1: v = in * gelu'(z), also written back to `in` (f32 dz for the bias gradient)
__device__ __forceinline__ float gelu_grad16(float z) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * z * z) * 0.39894228040143267794f;
    return cdf + z * pdf;
}

// in (rows, cols) f32 row-major (leading dim ld) -> out16 (rows, cols) [optional] and out16T (cols, rows)
template <int NPL, int MODE>
__global__ __launch_bounds__(256) void cast_transpose_kernel(float* __restrict__ in, long ld, const float* __restrict__ z,
                                                             uint16_t* __restrict__ out16, uint16_t* __restrict__ out16T,
                                                             int rows, int cols, float* __restrict__ colpart) {
    __shared__ uint16_t tile[NPL][64][66];
    __shared__ float csum[4][64];
    float cs = 0.f;  // column c0+tx over this thread's 16 rows (bias gradient partial)
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t plane = (size_t)rows * cols;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int rl = ty * 16 + i;
        const size_t src = (size_t)(r0 + rl) * ld + c0 + tx;
        float v = in[src];
        if constexpr (MODE == 1) {
            v *= gelu_grad16(z[src]);
            in[src] = v;
        }
        cs += v;
        uint16_t hi, lo;
        to16<NPL>(v, hi, lo);
        tile[0][rl][tx] = hi;
        if constexpr (NPL == 2) tile[1][rl][tx] = lo;
        if (out16) {
            const size_t dst = (size_t)(r0 + rl) * cols + c0 + tx;
            out16[dst] = hi;
            if constexpr (NPL == 2) out16[plane + dst] = lo;
        }
    }
    if (colpart) csum[ty][tx] = cs;
    __syncthreads();
    if (colpart && ty == 0)
        colpart[(size_t)blockIdx.y * cols + c0 + tx] = ((csum[0][tx] + csum[1][tx]) + csum[2][tx]) + csum[3][tx];
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int cl = ty * 16 + i;
        const size_t dst = (size_t)(c0 + cl) * rows + r0 + tx;
        out16T[dst] = tile[0][tx][cl];
        if constexpr (NPL == 2) out16T[plane + dst] = tile[1][tx][cl];
    }
}

// in (rows, cols) f32 -> bf16 copy in the SAME orientation (times gelu'(z) in MODE 1) and, per 64-row block, the column sums
// (bias gradient partials, same [rows/64][cols] layout as cast_transpose_kernel): the dY operand of launch_wgrad16_tn, which
// needs no transposed copy.  Thread = 8 consecutive columns (two 16-byte loads, one 16-byte store) of 8 rows.
template <int MODE>
__global__ __launch_bounds__(256) void cast_rows_colsum_kernel(const float* __restrict__ in, long ld, const float* __restrict__ z,
                                                               uint16_t* __restrict__ out16, int cols,
                                                               float* __restrict__ colpart) {
    __shared__ float csum[8][256];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 256;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
        const size_t src = (size_t)(r0 + ty + 8 * i) * ld + c0 + tx * 8;
        const float4 a = *reinterpret_cast<const float4*>(in + src), b = *reinterpret_cast<const float4*>(in + src + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if constexpr (MODE == 1) {
            const float4 za = *reinterpret_cast<const float4*>(z + src), zb = *reinterpret_cast<const float4*>(z + src + 4);
            const float zz[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_grad16(zz[e]);
        }
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[2 * e] += v[2 * e];
            cs[2 * e + 1] += v[2 * e + 1];
            pk[e] = f32x2_to_bf16x2(v[2 * e], v[2 * e + 1]);
        }
        *reinterpret_cast<uint4*>(out16 + (size_t)(r0 + ty + 8 * i) * cols + c0 + tx * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    if (!colpart) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[ty][tx * 8 + e] = cs[e];
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) a += csum[t][threadIdx.x];
    colpart[(size_t)blockIdx.y * cols + c0 + threadIdx.x] = a;
}
// GENIE_E_UNSUPPORTED unless rows % 64 == 0, cols % 256 == 0, ld % 4 == 0
int launch_cast_rows16(const float* in, long ld, const float* z, uint16_t* out16, int rows, int cols, hipStream_t st,
                       float* colpart) {
    if (rows % 64 || cols % 256 || ld % 4) return GENIE_E_UNSUPPORTED;
    if (!rows || !cols) return GENIE_OK;
    dim3 grid(cols / 256, rows / 64);
    ProfScope prof(GENIE_KC_OTHER, 0.0, (double)rows * cols * (4.0 + (z ? 4.0 : 0.0) + 2.0), st);
    if (z) cast_rows_colsum_kernel<1><<<grid, 256, 0, st>>>(in, ld, z, out16, cols, colpart);
    else cast_rows_colsum_kernel<0><<<grid, 256, 0, st>>>(in, ld, nullptr, out16, cols, colpart);
    GENIE_LAUNCH_CHECK("cast_rows16");
    return GENIE_OK;
}

// 16-bit (rows, cols) planes -> (cols, rows) planes
template <int NPL>
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ outT,
                                                          int rows, int cols) {
    __shared__ uint16_t tile[NPL][64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t plane = (size_t)rows * cols;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int rl = ty * 16 + i;
        const size_t src = (size_t)(r0 + rl) * cols + c0 + tx;
        tile[0][rl][tx] = in[src];
        if constexpr (NPL == 2) tile[1][rl][tx] = in[plane + src];
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int cl = ty * 16 + i;
        const size_t dst = (size_t)(c0 + cl) * rows + r0 + tx;
        outT[dst] = tile[0][tx][cl];
        if constexpr (NPL == 2) outT[plane + dst] = tile[1][tx][cl];
    }
}

static int check64(int rows, int cols, const char* what) {
    GENIE_CHECK_SHAPE(rows % 64 == 0 && cols % 64 == 0, "%s: rows=%d and cols=%d must be multiples of 64", what, rows, cols);
    return GENIE_OK;
}

// mode 0: plain cast; mode 1: in *= gelu'(z) first (in place).  out16 may be NULL.
int launch_cast_transpose16(int npl, float* in, long ld, const float* z, uint16_t* out16, uint16_t* out16T, int rows,
                            int cols, hipStream_t st, float* colpart) {
    GENIE_TRY(check64(rows, cols, "cast_transpose16"));
    if (!rows || !cols) return GENIE_OK;
    dim3 grid(cols / 64, rows / 64);
    ProfScope prof(GENIE_KC_OTHER, 0.0, (double)rows * cols * (4.0 + (z ? 8.0 : 0.0) + 2.0 * npl * (out16 ? 2 : 1)), st);
    if (npl == 1) {
        if (z) cast_transpose_kernel<1, 1><<<grid, 256, 0, st>>>(in, ld, z, out16, out16T, rows, cols, colpart);
        else cast_transpose_kernel<1, 0><<<grid, 256, 0, st>>>(in, ld, nullptr, out16, out16T, rows, cols, colpart);
    } else {
        if (z) cast_transpose_kernel<2, 1><<<grid, 256, 0, st>>>(in, ld, z, out16, out16T, rows, cols, colpart);
        else cast_transpose_kernel<2, 0><<<grid, 256, 0, st>>>(in, ld, nullptr, out16, out16T, rows, cols, colpart);
    }
    GENIE_LAUNCH_CHECK("cast_transpose16");
    return GENIE_OK;
}
int launch_transpose16(int npl, const uint16_t* in, uint16_t* outT, int rows, int cols, hipStream_t st) {
    GENIE_TRY(check64(rows, cols, "transpose16"));
    if (!rows || !cols) return GENIE_OK;
    dim3 grid(cols / 64, rows / 64);
    ProfScope prof(GENIE_KC_OTHER, 0.0, (double)rows * cols * 4.0 * npl, st);
    if (npl == 1) transpose16_kernel<1><<<grid, 256, 0, st>>>(in, outT, rows, cols);
    else transpose16_kernel<2><<<grid, 256, 0, st>>>(in, outT, rows, cols);
    GENIE_LAUNCH_CHECK("transpose16");
    return GENIE_OK;
}
// f32 -> 16-bit planes, same orientation
int launch_cast16(int npl, const float* src, uint16_t* dst, size_t n, hipStream_t st) {
    return npl == 1 ? launch_pack_bf16(src, dst, n, st) : launch_split_f16(src, dst, n, n, st);
}

}  // namespace genie
