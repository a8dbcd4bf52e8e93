// Shared helpers for the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

// Two GELUs at once in the packed-f32 form the GEMM epilogues want (v_pk_fma_f32 / v_pk_mul_f32: 2 lanes of math per issue
// slot).  Same approximation as gelu_erf_fast; the sign handling is folded away:  with q = erfc(|x|), x = z/sqrt(2),
//   z >= 0: 0.5 z (2 - q) = z - 0.5 |z| q       z < 0: 0.5 z q = -0.5 |z| q        =>  gelu(z) = max(z, 0) - (|x| / sqrt(2)) q
typedef float genie_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ genie_f2 gelu_erf_fast2(genie_f2 z) {
    const genie_f2 x = z * 0.70710678118654752440f;
    const genie_f2 ax = {fabsf(x[0]), fabsf(x[1])};
    auto splat = [](float c) { return genie_f2{c, c}; };
    const genie_f2 u = __builtin_elementwise_fma(ax, splat(0.3275911f), splat(1.0f));
    const genie_f2 t = {__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
    genie_f2 p = __builtin_elementwise_fma(t, splat(1.061405429f), splat(-1.453152027f));
    p = __builtin_elementwise_fma(p, t, splat(1.421413741f));
    p = __builtin_elementwise_fma(p, t, splat(-0.284496736f));
    p = __builtin_elementwise_fma(p, t, splat(0.254829592f));
    const genie_f2 a2 = ax * ax * -1.4426950408889634f;
    const genie_f2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
    const genie_f2 w = (p * t) * (e * (ax * 0.70710678118654752440f));
    const genie_f2 zp = {fmaxf(z[0], 0.f), fmaxf(z[1], 0.f)};
    return zp - w;
}

// Two GELUs for a consumer that rounds them to bf16 at once (the fused MLP kernel's hidden): no transcendental, packed f32.
//   gelu(z) = z * Phi(z),  Phi(z) - 1/2 = zc * P(zc^2),  zc = clamp(z, -4.25, 4.25),  P of degree 8 (weighted minimax fit of
//   (Phi(z) - 1/2) / z on [0, 4.25], tools/fit_gelu_poly.py, tests/test_gelu_poly.py):  |Phi error| <= 1.3e-5 for every z (0 < Phi < 1
//   also beyond the clamp), |gelu error| <= 5.3e-5 absolute; relative 1.4e-5 for z > 0.25 -- 0.7 % of a bf16 half-ulp, fewer than 1
//   value in 100 rounds to the neighbouring bf16; on the negative side (|gelu| <= 0.17) the absolute bound holds but the relative
//   one does not (Phi itself is small there): ~7 % of those values land on a neighbouring bf16, 1-2 ulp at z < -3.
//   13 instructions per PAIR (2 v_med3, v_pk_mul, 9 v_pk_fma, v_pk_mul) against 2 x 15 with two quarter-rate ones for
//   gelu_erf_fast.  NOT for the parity-grade modes (their GELU stays at 1.5e-7).
__device__ __forceinline__ genie_f2 gelu_erf_poly2(genie_f2 z) {
    auto splat = [](float c) { return genie_f2{c, c}; };
    const genie_f2 zc = {__builtin_amdgcn_fmed3f(z[0], -4.25f, 4.25f), __builtin_amdgcn_fmed3f(z[1], -4.25f, 4.25f)};
    const genie_f2 s = zc * zc;
    genie_f2 p = __builtin_elementwise_fma(s, splat(5.564818051e-11f), splat(-5.327728037e-09f));
    p = __builtin_elementwise_fma(p, s, splat(2.255418963e-07f));
    p = __builtin_elementwise_fma(p, s, splat(-5.626413895e-06f));
    p = __builtin_elementwise_fma(p, s, splat(9.341857367e-05f));
    p = __builtin_elementwise_fma(p, s, splat(-1.108560245e-03f));
    p = __builtin_elementwise_fma(p, s, splat(9.815969504e-03f));
    p = __builtin_elementwise_fma(p, s, splat(-6.634449214e-02f));
    p = __builtin_elementwise_fma(p, s, splat(3.989023268e-01f));
    return z * __builtin_elementwise_fma(zc, p, splat(0.5f));
}

// GELU of a 16-bit GEMM epilogue.  LOWP = the value leaves ONLY as bf16 (the MLP hidden of GENIE_PREC_BF16): the polynomial form in
// EVERY bf16 kernel (fused MLP, gemm16_pp / v2 / nt / sm), so that a clip's hidden does not depend on which kernel its batch
// size selects; otherwise (split-f16 operands, f32 outputs) the 1.5e-7 form.
template <bool LOWP>
__device__ __forceinline__ genie_f2 gelu16_2(genie_f2 z) {
    if constexpr (LOWP) return gelu_erf_poly2(z);
    else return gelu_erf_fast2(z);
}
template <bool LOWP>
__device__ __forceinline__ float gelu16_1(float z) { return gelu16_2<LOWP>(genie_f2{z, z})[0]; }

// round-to-nearest-even f32 -> bf16 bits: v_cvt_pk_bf16_f32 (gfx950), one instruction instead of the five of the integer
// emulation  u += 0x7FFF + ((u >> 16) & 1); u >>= 16  -- the same rounding for every finite value
typedef __bf16 genie_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float a, float b) {  // element 0 in the low half
    const genie_f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, genie_bf2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(f32x2_to_bf16x2(f, 0.f) & 0xFFFFu); }
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

// The same split for four values at once, on packed conversions (v_cvt_pk_f16_f32: 3 VALU instructions per element instead of
// ~10) and WITHOUT the flush of a subnormal hi: gfx950's f16 matrix instructions take subnormal inputs exactly (probed on
// MI355X: A- and B-side subnormals down to 2^-24 give exact sums, tools/probe_f16_subnormals.py), so hi + lo/2048 is the same
// 22-bit value either way.  Returns hi pairs in h01 / h23 and lo pairs in l01 / l23 (element 0 in the low half).
__device__ __forceinline__ void split_f16_x4(float a0, float a1, float a2, float a3, uint32_t& h01, uint32_t& h23,
                                             uint32_t& l01, uint32_t& l23) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v x = {a0, a1}, y = {a2, a3};
    const h2v hx = __builtin_convertvector(x, h2v), hy = __builtin_convertvector(y, h2v);
    const f2v rx = (x - __builtin_convertvector(hx, f2v)) * 2048.0f, ry = (y - __builtin_convertvector(hy, f2v)) * 2048.0f;
    const h2v lx = __builtin_convertvector(rx, h2v), ly = __builtin_convertvector(ry, h2v);
    h01 = __builtin_bit_cast(uint32_t, hx); h23 = __builtin_bit_cast(uint32_t, hy);
    l01 = __builtin_bit_cast(uint32_t, lx); l23 = __builtin_bit_cast(uint32_t, ly);
}

// Study knobs (ablations that make results WRONG, reduced-precision GEMM terms, in-launch stamps that allocate and synchronise)
// exist only in a -DGENIE_STUDY build (GENIE_STUDY=1 python 1xgpt_amd/build.py -> libgenie_hip_study.so).  In the shipping
// library study_env() is a constant: a leaked environment variable cannot change what a launch computes.
#ifdef GENIE_STUDY
inline int study_env(const char* name, int dflt) {
    const char* e = getenv(name);
    if (e && atoi(e) != dflt) {
        static thread_local char seen[512] = "";
        if (!strstr(seen, name)) {
            fprintf(stderr, "libgenie_hip (STUDY BUILD): %s=%s overrides the default %d -- results of this process are not the product's\n", name, e, dflt);
            if (strlen(seen) + strlen(name) + 2 < sizeof(seen)) { strcat(seen, name); strcat(seen, ";"); }
        }
    }
    return e ? atoi(e) : dflt;
}
constexpr bool kStudyBuild = true;
#else
constexpr int study_env(const char*, int dflt) { return dflt; }
constexpr bool kStudyBuild = false;
#endif

// One-time per-DEVICE launch setup (hipFuncSetAttribute(MaxDynamicSharedMemorySize), the CU count that sizes a persistent grid): a
// process may drive several devices and call from several host threads, so "once per process" is wrong on the second device.
//   static PerDevice<bool> ready;  if (ready.needs()) { hipFuncSetAttribute(...); ready.set(true); }
// needs() is true until set() has been called on the current device; two racing threads both run the (idempotent) setup, neither
// launches before it has run.
constexpr int GENIE_MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (d < 0 || d >= GENIE_MAX_DEVICES) ? 0 : d;
}
template <typename T>
struct PerDevice {
    T v[GENIE_MAX_DEVICES] = {};
    volatile bool ok[GENIE_MAX_DEVICES] = {};
    bool needs() const { return !__atomic_load_n(&ok[current_device()], __ATOMIC_ACQUIRE); }
    void set(T x) { const int d = current_device(); v[d] = x; __atomic_store_n(&ok[d], true, __ATOMIC_RELEASE); }
    T get() const { return v[current_device()]; }
};
// multiprocessor count of the CURRENT device (cached per device)
inline int device_cu_count() {
    static PerDevice<int> cus;
    if (cus.needs()) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, current_device());
        cus.set(n > 0 ? n : 256);
    }
    return cus.get();
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace genie
