// Do vector-memory loads and stores retire IN ORDER with each other on gfx950 (one vmcnt counter)?  The fused prefix-cache kernels
// (csrc/kernels_fused_prefix.hip, kernels_fused_f16x3.hip) count stores issued AFTER a load among the operations a counted wait may leave
// in flight -- correct only if a younger store cannot retire before the older load.  (LLVM's gfx9 model says so: loads and stores are one
// event type on vmcnt.)  Probe: every lane issues ONE slow load (a cold line of a 2 GB buffer, inline asm so the compiler adds no wait),
// then NST fast stores (a hot 4 KB scratch line set per wave), then s_waitcnt vmcnt(NST) and immediately consumes the loaded register.
// If stores could overtake the load, vmcnt <= NST would be reached with the load still in flight and the consumed value would be the
// register's old content (a sentinel).  Counts sentinel reads over many waves and repetitions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NST>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, unsigned* __restrict__ bad, size_t n_cold, unsigned salt) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    // a pseudo-random cold address per lane (whole lanes of a wave hit different DRAM pages: slow)
    size_t idx = ((size_t)gid * 2654435761u + (size_t)salt * 40503u) % n_cold;
    const unsigned* p = cold + idx;
    unsigned v = 0xDEADBEEFu;   // sentinel: the loaded data never has this value
    unsigned* h = hot + (size_t)(gid & ~63u) * 16 + (gid & 63u);
    asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < NST; ++i) asm volatile("global_store_dword %0, %1, off" ::"v"(h + 64 * i), "v"(gid + i) : "memory");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(NST));
    unsigned got;
    asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "v"(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (got == 0xDEADBEEFu) atomicAdd(bad, 1u);
    else if (got != (unsigned)(idx & 0x7FFFFFFFu)) atomicAdd(bad + 1, 1u);   // (wrong data of any other kind)
}


template <int NST>
__global__ __launch_bounds__(256) void control(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, unsigned* __restrict__ bad, size_t n_cold, unsigned salt) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    size_t idx = ((size_t)gid * 2654435761u + (size_t)salt * 40503u) % n_cold;
    const unsigned* p = cold + idx;
    unsigned v = 0xDEADBEEFu;
    unsigned* h = hot + (size_t)(gid & ~63u) * 16 + (gid & 63u);
    asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(p) : "memory");
#pragma unroll
    for (int i = 0; i < NST; ++i) asm volatile("global_store_dword %0, %1, off" ::"v"(h + 64 * i), "v"(gid + i) : "memory");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(NST + 1));   // the load itself may still be in flight
    unsigned got;
    asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "v"(v));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (got == 0xDEADBEEFu) atomicAdd(bad, 1u);
}

__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)(i & 0x7FFFFFFFu);
}

int main() {
    const size_t n_cold = (size_t)512 << 20;   // 2 GB of dwords
    unsigned *cold, *hot, *bad;
    hipMalloc(&cold, n_cold * 4);
    const int blocks = 8192;
    hipMalloc(&hot, (size_t)blocks * 256 * 16 * 4);
    hipMalloc(&bad, 8);
    fill<<<4096, 256>>>(cold, n_cold);
    hipDeviceSynchronize();
    unsigned tot[2] = {0, 0};
    long launched = 0;
    for (int rep = 0; rep < 200; ++rep) {
        hipMemset(bad, 0, 8);
        switch (rep % 4) {
            case 0: probe<1><<<blocks, 256>>>(cold, hot, bad, n_cold, rep); break;
            case 1: probe<2><<<blocks, 256>>>(cold, hot, bad, n_cold, rep); break;
            case 2: probe<4><<<blocks, 256>>>(cold, hot, bad, n_cold, rep); break;
            default: probe<6><<<blocks, 256>>>(cold, hot, bad, n_cold, rep); break;
        }
        unsigned b[2];
        hipMemcpy(b, bad, 8, hipMemcpyDeviceToHost);
        tot[0] += b[0]; tot[1] += b[1];
        launched += (long)blocks * 256;
    }
    printf("loads followed by 1 / 2 / 4 / 6 younger stores, s_waitcnt vmcnt(#stores), value consumed at once: %ld lane-trials\n", launched);
    printf("  load not landed (sentinel read): %u     other wrong data: %u\n", tot[0], tot[1]);
    printf("  => %s\n", tot[0] == 0 && tot[1] == 0 ? "younger stores never retired ahead of the older load: vmcnt is in order across loads and stores"
                                                    : "OUT OF ORDER retirement observed");
    // control: the same probe with vmcnt(NST + 1) (one more operation may stay in flight = the load itself): sentinels MUST appear
    hipMemset(bad, 0, 8);
    control<4><<<blocks, 256>>>(cold, hot, bad, n_cold, 12345);
    unsigned c[2];
    hipMemcpy(c, bad, 8, hipMemcpyDeviceToHost);
    printf("control (vmcnt(#stores + 1): the load may still be in flight): %u of %ld lanes read the sentinel -- the probe does see an unlanded load\n", c[0], (long)blocks * 256);
    return 0;
}
