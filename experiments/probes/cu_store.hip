// How many bytes per clock can ONE CU store?  (round 6: the NT epilogue of a 256 x 256 tile is serialised with its CU's K loop, and its
// store tail -- 128 KiB of bf16, 256 KiB for the two-output forms -- has measured ~13-19 B/clk/CU in the kernels.  Is that the CU's own
// store path or the chip's write bandwidth shared by 256 CUs?)  G blocks (one per CU) of 512 threads; every wave writes its own slice of a
// per-block region with 16-byte stores (plain / non-temporal), `iters` passes over `kib` KiB per block; G = 8, 64, 256.  Prints B/clk/CU
// from s_memtime and GB/s per CU / chip from wall time.   hipcc --offload-arch=gfx950 -O3 cu_store.hip -o cu_store && ./cu_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool NT>
__global__ __launch_bounds__(512) void store_stream(char* __restrict__ buf, size_t bytes_per_block, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = buf + (size_t)blockIdx.x * bytes_per_block;
    const size_t per_wave = bytes_per_block / 8;
    char* w = base + (size_t)wave * per_wave + lane * 16;
    const u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
        for (size_t off = 0; off + 1024 <= per_wave; off += 1024) {
            if (NT) __builtin_nontemporal_store(v, (u32x4*)(w + off));
            else *(u32x4*)(w + off) = v;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <bool NT>
static void run(const char* name, char* buf, int G, int kib, int iters, unsigned long long* clk) {
    const size_t bpb = (size_t)kib * 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(store_stream<NT>, dim3(G), dim3(512), 0, 0, buf, bpb, iters, clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(store_stream<NT>, dim3(G), dim3(512), 0, 0, buf, bpb, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    (void)hipMemcpy(h, clk, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ticks = 0;
    for (int b = 0; b < G; ++b) ticks += (double)h[b];
    ticks /= G;
    const double bytes = (double)bpb * iters;
    printf("%-14s G=%3d  %5d KiB x %4d per block: %6.1f B per s_memtime tick per CU, wall %8.1f us = %6.1f GB/s per CU, %6.2f TB/s chip\n", name, G, kib, iters,
           bytes / ticks, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * G / (ms * 1e-3) / 1e12);
}

int main() {
    char* buf; unsigned long long* clk;
    const size_t total = (size_t)256 * 16 * 1024 * 1024;  // 16 MiB per block: past every cache
    (void)hipMalloc(&buf, total); (void)hipMalloc(&clk, 256 * 8);
    const int Gs[3] = {8, 64, 256};
    for (int gi = 0; gi < 3; ++gi) {
        // the epilogue's burst: 128 KiB / 256 KiB per block, written once per "tile" into fresh addresses (16 MiB per block = 128 / 64 bursts)
        run<true>("nt 128 KiB", buf, Gs[gi], 128, 1, clk);
        run<true>("nt 256 KiB", buf, Gs[gi], 256, 1, clk);
        run<true>("nt  16 MiB", buf, Gs[gi], 16384, 1, clk);
        run<false>("plain 16 MiB", buf, Gs[gi], 16384, 1, clk);
    }
    return 0;
}
