// Does the 1.5 MiB frame stride of the temporal KV cache (S * 3 d * 4 bytes at S = 256, d = 512) cost the decode attention kernel
// memory-channel conflicts?  Emulates its access pattern: one wave per (row, head) item reads, for each of 16 frames, 256 bytes of K and
// 256 bytes of V at  base + frame * stride + row * 6144 + head * 256  (+ 2048 / 4096 for k / v), 16 lanes x 16 bytes per frame and
// 4 frames per instruction.  usage: stride_probe   (prints microseconds per launch for several strides; rows = 256 * clips)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
int main_strides();
int main_b();
__global__ __launch_bounds__(256) void probe(const float* __restrict__ cache, float* __restrict__ out, long n_items, long stride_f, int t) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int head = (int)(item % 8);
    const long row = item / 8;
    const int g = lane >> 4, c = lane & 15;
    const float* hb = cache + row * 1536 + head * 64 + 4 * c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = 4 * i + g;
        const float* src = hb + (long)(j <= t ? j : t) * stride_f;
        acc += *reinterpret_cast<const f32x4*>(src + 512);
        acc += *reinterpret_cast<const f32x4*>(src + 1024);
    }
    if (acc[0] == 123.456f) out[item] = acc[1];
}
// layout B: the frames of one row adjacent -- [row][frame][k 512 | v 512] floats: 64 KB contiguous per row over 16 frames
__global__ __launch_bounds__(256) void probe_rowmajor_frames(const float* __restrict__ cache, float* __restrict__ out, long n_items, int T, int t) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int head = (int)(item % 8);
    const long row = item / 8;
    const int g = lane >> 4, c = lane & 15;
    const float* hb = cache + row * (long)T * 1024 + head * 64 + 4 * c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = 4 * i + g;
        const float* src = hb + (long)(j <= t ? j : t) * 1024;
        acc += *reinterpret_cast<const f32x4*>(src);
        acc += *reinterpret_cast<const f32x4*>(src + 512);
    }
    if (acc[0] == 123.456f) out[item] = acc[1];
}
// layout C: everything one wave reads contiguous -- [row][head][frame][k 64 | v 64] floats: 8 KB per (row, head)
__global__ __launch_bounds__(256) void probe_wave_contiguous(const float* __restrict__ cache, float* __restrict__ out, long n_items, int T, int t) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int g = lane >> 4, c = lane & 15;
    const float* hb = cache + item * (long)T * 128 + 4 * c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = 4 * i + g;
        const float* src = hb + (long)(j <= t ? j : t) * 128;
        acc += *reinterpret_cast<const f32x4*>(src);
        acc += *reinterpret_cast<const f32x4*>(src + 64);
    }
    if (acc[0] == 123.456f) out[item] = acc[1];
}
int main() {
    for (int clips : {1, 16}) {
        const long rows = 256L * clips, n = rows * 8;
        float *cache, *out, *thrash;
        hipMalloc(&cache, (size_t)rows * 16 * 1024 * 4);
        hipMalloc(&out, n * 4);
        hipMalloc(&thrash, 512 << 20);
        hipMemset(cache, 0, (size_t)rows * 16 * 1024 * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int t : {3, 15}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemsetAsync(thrash, rep, 512 << 20, 0);
                hipEventRecord(e0, 0);
                probe_wave_contiguous<<<(unsigned)((n + 3) / 4), 256, 0, 0>>>(cache, out, n, 16, t);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("clips %2d  t %2d  layout [row][head][frame][k | v] (a wave's 8 KB contiguous)  %7.1f us\n", clips, t, best * 1e3);
        }
        hipFree(cache); hipFree(out); hipFree(thrash);
    }
    return main_b();
}
int main_b() {
    for (int clips : {1, 16}) {
        const long rows = 256L * clips, n = rows * 8;
        float *cache, *out, *thrash;
        hipMalloc(&cache, (size_t)rows * 16 * 1024 * 4);
        hipMalloc(&out, n * 4);
        hipMalloc(&thrash, 512 << 20);
        hipMemset(cache, 0, (size_t)rows * 16 * 1024 * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int t : {3, 15}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemsetAsync(thrash, rep, 512 << 20, 0);
                hipEventRecord(e0, 0);
                probe_rowmajor_frames<<<(unsigned)((n + 3) / 4), 256, 0, 0>>>(cache, out, n, 16, t);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("clips %2d  t %2d  layout [row][frame][k | v] (frames of a row adjacent)  %7.1f us\n", clips, t, best * 1e3);
        }
        hipFree(cache); hipFree(out); hipFree(thrash);
    }
    return main_strides();
}
int main_strides() {
    const int T = 16;
    for (int clips : {1, 16}) {
        const long rows = 256L * clips, n = rows * 8;
        for (long pad_bytes : {0L, 256L, 4096L, 6144L, 65536L + 6144L}) {
            const long stride_f = (rows * 1536 * 4 + pad_bytes) / 4;   // floats between frames of one clip-batch layout (B folded into rows)
            float *cache, *out;
            hipMalloc(&cache, (size_t)T * stride_f * 4 + (1 << 20));
            hipMalloc(&out, n * 4);
            hipMemset(cache, 0, (size_t)T * stride_f * 4);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int t : {3, 15}) {
                // a second buffer walk between launches keeps the cache lines cold (weights of a layer would do that in the model)
                float* thrash; hipMalloc(&thrash, 512 << 20);
                float best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    hipMemsetAsync(thrash, rep, 512 << 20, 0);
                    hipEventRecord(e0, 0);
                    probe<<<(unsigned)((n + 3) / 4), 256, 0, 0>>>(cache, out, n, stride_f, t);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                hipFree(thrash);
                printf("clips %2d  t %2d  frame stride %9ld B (pad %6ld)  %7.1f us\n", clips, t, stride_f * 4, pad_bytes, best * 1e3);
            }
            hipFree(cache); hipFree(out);
        }
    }
    return 0;
}
