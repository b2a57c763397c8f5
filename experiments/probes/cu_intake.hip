// How many bytes per clock does ONE CU take in from an L2-resident (or HBM-cold) buffer?  (round 4: the question behind the ring
// kernel's 0.55 us per 32 KiB stage.)  G blocks (one per CU), W waves each; every wave streams its own slice with U independent
// 16-byte loads per lane in flight -- into registers (mode 0) or into LDS by LDS-DMA (mode 1).  Prints B/clk/CU from the shader clock
// and GB/s from wall time.   hipcc --offload-arch=gfx950 -O3 cu_intake.hip -o cu_intake && ./cu_intake
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

template <int U, int MODE>
__global__ __launch_bounds__(1024) void intake(const char* __restrict__ buf, size_t bytes_per_block, int iters, float* sink,
                                               unsigned long long* clocks) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * bytes_per_block;
    const size_t per_wave = bytes_per_block / nw;
    const char* wbase = base + (size_t)wave * per_wave;
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (size_t off = 0; off + (size_t)U * 1024 <= per_wave; off += (size_t)U * 1024) {
            if (MODE == 0) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *(const f32x4*)(wbase + off + u * 1024 + lane * 16);
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(wbase + off + u * 1024 + lane * 16),
                                                     (LDS_PTR(void))(smem + (wave * U + u) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) acc[0] += *(const float*)(smem + threadIdx.x * 4);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e33f) sink[0] = acc[0];
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

// the GEMM operand pattern: a wave instruction fetches 8 rows x 128 B (lane l: row l >> 3, 16-B chunk (l & 7) [^ row & 7]) of a row-major
// [rows][row_bytes] matrix and walks k in 128-B steps; the block's waves take 8-row groups of a `rows`-row panel, U steps in flight
template <int U, int MODE, bool SWZ>
__global__ __launch_bounds__(1024) void intake_rows(const char* __restrict__ buf, int rows, int row_bytes, int iters, float* sink,
                                                    unsigned long long* clocks) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * rows * row_bytes;
    f32x4 acc = {0, 0, 0, 0};
    const int r = lane >> 3, c = SWZ ? ((lane & 7) ^ (r & 7)) : (lane & 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
        for (int g = wave; g < rows / 8; g += nw) {
            const char* p = base + (size_t)(g * 8 + r) * row_bytes + c * 16;
            for (int k0 = 0; k0 + U * 128 <= row_bytes; k0 += U * 128) {
                if (MODE == 0) {
                    f32x4 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = *(const f32x4*)(p + k0 + u * 128);
#pragma unroll
                    for (int u = 0; u < U; ++u) acc += v[u];
                } else if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(p + k0 + u * 128), (LDS_PTR(void))(smem + (wave * U + u) * 1024), 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {  // M0 rewritten in front of every piece, 32-bit lane offsets from a scalar base (the production form)
                    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_PTR(char))smem + (unsigned)(wave * U) * 1024u);
                    const unsigned long long sb = (unsigned long long)(base + k0);
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
                    const char* ub = (const char*)(((unsigned long long)hi << 32) | lo);
                    const unsigned voff = (unsigned)((g * 8 + r) * row_bytes + c * 16);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(voff), "s"(ub), "s"(lds0 + u * 1024u), "i"(0) : "memory");
                        ub += 128;
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) acc[0] += *(const float*)(smem + threadIdx.x * 4);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e33f) sink[0] = acc[0];
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}
template <int U, int MODE, bool SWZ>
static void run_rows(int G, int W, int rows, int row_bytes, int iters, const char* what, char* buf, float* sink, unsigned long long* clk) {
    hipFuncSetAttribute((const void*)intake_rows<U, MODE, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    intake_rows<U, MODE, SWZ><<<G, W * 64, 65536>>>(buf, rows, row_bytes, 1, sink, clk);
    hipEventRecord(e0);
    intake_rows<U, MODE, SWZ><<<G, W * 64, 65536>>>(buf, rows, row_bytes, iters, sink, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * (row_bytes / (U * 128) * (U * 128)) * iters;
    printf("%-10s G=%3d W=%2d U=%2d panel %4d rows x %5d B x %4d: wall %8.1f us = %6.1f GB/s per CU, %6.2f TB/s chip\n", what, G, W, U, rows,
           row_bytes, iters, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * G / (ms * 1e-3) / 1e12);
}

// the ring kernel's stage pattern: per 64-deep stage the block's 8 waves fetch a 128-row A panel slice and a 128-row B panel slice
// (wave w: rows (8 t + w) * 8 .. + 7, t = 0, 1, of each) at k0, then k0 += 128 B.  BAR: one barrier per stage, D stages in flight.
template <int D, bool BAR>
__global__ __launch_bounds__(512) void intake_stage(const char* __restrict__ A, const char* __restrict__ B, int row_bytes, int iters, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* a = A + (size_t)blockIdx.x * 128 * row_bytes;
    const char* b = B + (size_t)(blockIdx.x % 6) * 128 * row_bytes;
    const int r = lane >> 3, c = (lane & 7) ^ (r & 7);
    const int nk = row_bytes / 128;
    int issued = 0;
    auto issue = [&](int s) {
        const int k0 = (s % nk) * 128, slot = s % (D + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = (t * 8 + wave) * 8 + r;
            __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(a + (size_t)row * row_bytes + k0 + c * 16), (LDS_PTR(void))(smem + slot * 32768 + (t * 8 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(b + (size_t)row * row_bytes + k0 + c * 16), (LDS_PTR(void))(smem + slot * 32768 + 16384 + (t * 8 + wave) * 1024), 16, 0, 0);
        }
        ++issued;
    };
    const int total = nk * iters;
    for (int s = 0; s < D && s < total; ++s) issue(s);
    for (int s = 0; s < total; ++s) {
        // stage s landed: (issued - s - 1) stages of 4 pieces may stay in flight
        const int younger = issued - s - 1;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
        if (issued < total) issue(issued);
    }
    float v = *(const float*)(smem + threadIdx.x * 4);
    if (v == 1.2345e33f) sink[0] = v;
}
template <int D, bool BAR>
static void run_stage(int G, int row_bytes, int iters, const char* what, char* buf, float* sink) {
    hipFuncSetAttribute((const void*)intake_stage<D, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    char* B = buf + ((size_t)512 << 20);
    intake_stage<D, BAR><<<G, 512, (D + 1) * 32768>>>(buf, B, row_bytes, 1, sink);
    hipEventRecord(e0);
    intake_stage<D, BAR><<<G, 512, (D + 1) * 32768>>>(buf, B, row_bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double stages = (double)(row_bytes / 128) * iters;
    printf("%-28s G=%3d row %5d B: %7.1f us, %6.3f us per 32 KiB stage = %6.1f GB/s per CU\n", what, G, row_bytes, ms * 1e3, ms * 1e3 / stages, 32768.0 * stages / (ms * 1e-3) / 1e9);
}

template <int U, int MODE>
static void run(int G, int W, size_t bytes_per_block, int iters, const char* what, char* buf, float* sink, unsigned long long* clk) {
    hipFuncSetAttribute((const void*)intake<U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    intake<U, MODE><<<G, W * 64, 65536>>>(buf, bytes_per_block, 1, sink, clk);  // warm (L2 / MALL)
    hipEventRecord(e0);
    intake<U, MODE><<<G, W * 64, 65536>>>(buf, bytes_per_block, iters, sink, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), clk, G * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto x : h) c += (double)x; c /= G;
    const double bytes = (double)(bytes_per_block / (W * (size_t)U * 1024) * (W * (size_t)U * 1024)) * iters;
    printf("%-8s G=%3d W=%2d U=%2d slice %6zu KiB x %4d: %6.1f B per s_memtime tick per CU (%9.0f ticks per block), wall %8.1f us = %6.1f GB/s per CU, %6.2f TB/s chip\n",
           what, G, W, U, bytes_per_block >> 10, iters, bytes / c, c, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * G / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf; float* sink; unsigned long long* clk;
    hipMalloc(&buf, total); hipMemset(buf, 1, total); hipMalloc(&sink, 64); hipMalloc(&clk, 8 * 1024);
    for (int G : {8, 48, 256}) {
        for (int W : {4, 8, 16}) {
            // L2-resident: 128 KiB per block re-read 200 times (256 blocks: 32 MiB = 4 MiB per XCD); cold: 4 MiB per block once
            run<8, 0>(G, W, 128 << 10, 200, "regs hot", buf, sink, clk);
            run<8, 1>(G, W, 128 << 10, 200, "dma  hot", buf, sink, clk);
        }
        run<8, 0>(G, 8, (size_t)4 << 20, 1, "regs cold", buf, sink, clk);
        run<8, 1>(G, 8, (size_t)4 << 20, 1, "dma  cold", buf, sink, clk);
        run<4, 0>(G, 8, 128 << 10, 200, "regs hot", buf, sink, clk);
        run<16, 0>(G, 8, 128 << 10, 200, "regs hot", buf, sink, clk);
    }
    printf("# the ring kernel's stage pattern (A panel per block, B panel shared by the blocks of a tile column), panels re-read 20 times\n");
    for (int G : {48, 256})
        for (int rb : {1536, 6144, 16384}) {
            run_stage<1, true>(G, rb, 20, "1 in flight, barrier", buf, sink);
            run_stage<3, true>(G, rb, 20, "3 in flight, barrier", buf, sink);
            run_stage<3, false>(G, rb, 20, "3 in flight, no barrier", buf, sink);
            run_stage<4, true>(G, rb, 20, "4 in flight, barrier", buf, sink);
        }
    printf("# GEMM operand pattern (8 rows x 128 B per wave instruction), L2-resident panels re-read 100 times\n");
    for (int G : {48, 256})
        for (int rb : {1536, 4608, 6144}) {
            run_rows<4, 0, false>(G, 8, 128, rb, 100, "regs", buf, sink, clk);
            run_rows<4, 1, false>(G, 8, 128, rb, 100, "dma", buf, sink, clk);
            run_rows<4, 1, true>(G, 8, 128, rb, 100, "dma swz", buf, sink, clk);
            run_rows<12, 1, true>(G, 8, 128, rb, 100, "dma swz", buf, sink, clk);
            run_rows<4, 2, true>(G, 8, 128, rb, 100, "dma m0 asm", buf, sink, clk);
            run_rows<12, 2, true>(G, 8, 128, rb, 100, "dma m0 asm", buf, sink, clk);
        }
    return 0;
}
