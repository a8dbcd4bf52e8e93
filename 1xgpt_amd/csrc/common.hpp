// Shared helpers for the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/genie_hip.h"

namespace genie {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

#define GENIE_CHECK_ARG(cond, ...)        \
    do {                                  \
        if (!(cond)) {                    \
            genie::set_error(__VA_ARGS__); \
            return GENIE_E_ARG;           \
        }                                 \
    } while (0)

#define GENIE_CHECK_SHAPE(cond, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            genie::set_error(__VA_ARGS__); \
            return GENIE_E_SHAPE;         \
        }                                 \
    } while (0)

#define GENIE_LAUNCH_CHECK(what)                                                    \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            genie::set_error("%s: HIP error %s", what, hipGetErrorString(e__));      \
            return GENIE_E_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

#define GENIE_TRY(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != GENIE_OK) return rc__; \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// (value, index) arg-max with "first max wins" (torch.argmax tie rule).
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o);
        int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__device__ __forceinline__ float gelu_erf(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }
// The same GELU for the 16-bit GEMM epilogues, where the library erff (~35 VALU instructions with both branches of
// its range split executed) made the fc1 epilogue cost 16 % (f16x3) to 45 % (bf16) of the whole GEMM: erf by
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. 2 ulp of the 1 + erf it feeds) on v_rcp_f32 / v_exp_f32, 14 instructions.
__device__ __forceinline__ float gelu_erf_fast(float z) {
    const float x = z * 0.70710678118654752440f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = p * t * __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);  // erfc(|x|)
    return 0.5f * z * (x >= 0.f ? 2.0f - q : q);                                       // 1 + erf(x)
}

// round-to-nearest-even f32 -> bf16 bits
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// f32 -> (hi, lo) f16 pair with a ~ hi + lo/2048 (GENIE_PREC_F16X3).  hi is flushed to zero below the f16 normal
// range so that nothing depends on how the matrix core treats f16 subnormals.
__device__ __forceinline__ void split_f16(float a, uint16_t& hi, uint16_t& lo) {
    _Float16 h = (_Float16)a;
    float hf = (float)h;
    if (fabsf(hf) < 6.103515625e-05f) { h = (_Float16)0.0f; hf = 0.0f; }
    _Float16 l = (_Float16)((a - hf) * 2048.0f);
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, l);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace genie
