// Does the chip hold its clock under a full-chip bf16 MFMA stream, and does the MFMA SHAPE matter?  (round 6: the in-kernel clock samples
// of the bench step read ~1.99 GHz of the sheet's 2.4 under the GEMM load -- the step is power-limited.)  256 blocks x 512 threads (two
// waves per SIMD, as the GEMM kernels), every wave runs ITERS x 64 back-to-back MFMAs on 32 independent accumulator tiles (128 registers,
// the GEMM's budget) with operands that differ per lane (random bits, so the datapath toggles like real data; zero operands as the
// second arm).  Arms: v_mfma_f32_16x16x32_bf16 (the GEMM kernels' shape: 2 x 256 operand elements per 16 384 FLOP) against
// v_mfma_f32_32x32x16_bf16 (2 x 512 per 32 768: half the operand register reads per FLOP); optionally with the GEMM's LDS fragment
// traffic beside them (24 ds_read_b128 per 64 MFMAs of the 16x16 shape).  Prints TFLOP/s from wall time and the clock from
// s_memtime / s_memrealtime inside the kernel.   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(512, 2) void mfma_stream(const u32x4* __restrict__ seed, int iters, float* sink, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63;
    // operands: 4 A + 4 B fragment registers, refreshed from LDS when LDS is on (the image is the seed data)
    bf16x8 a[4], b[4];
    for (int i = threadIdx.x; i < 4096; i += 512) ((u32x4*)smem)[i] = seed[(blockIdx.x * 4096 + i) & 65535];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(threadIdx.x * 4 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(threadIdx.x * 4 + i + 2048) & 4095]);
    }
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (SHAPE == 16) {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (LDS) {  // 12 ds_read_b128 per 32 MFMAs (the NT kernel: 24 per stage of 64)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(lane + 64 * (i + 4 * h) + it) & 4095]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) b[i + 2 * h] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(lane + 64 * (i + 9) + it) & 4095]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[(h * 16 + i * 4 + j) & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[(h * 16 + i * 4 + j) & 31], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[(h * 16 + i * 4 + j) & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[(j + 1) & 3], a[i], acc[(h * 16 + i * 4 + j) & 31], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
        if (s == 1.2345e33f) sink[0] = s;
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {  // the same FLOPs per iteration: 32 MFMAs of twice the size
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (LDS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(lane + 64 * (i + 4 * h) + it) & 4095]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) b[i + 2 * h] = __builtin_bit_cast(bf16x8, ((const u32x4*)smem)[(lane + 64 * (i + 9) + it) & 4095]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(j + 2 * k) & 3], a[(i + k) & 3], acc[i * 2 + j], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
        if (s == 1.2345e33f) sink[0] = s;
    }
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int SHAPE, bool LDS>
static void run(const char* name, const u32x4* seed, int iters, float* sink, unsigned long long* clk, int wall_khz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {  // a few hundred ms of sustained load; the last repetition is reported
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) hipLaunchKernelGGL((mfma_stream<SHAPE, LDS>), dim3(256), dim3(512), 0, 0, seed, iters, sink, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, tk = 0;
    for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; tk += (double)h[2 * b + 1]; }
    const double flop = 20.0 * 256 * 8 * (double)iters * 64 * 16384.0;
    const double mhz = cyc / tk * wall_khz / 1e3;
    printf("%-34s %8.1f TFLOP/s   clock %6.0f MHz   -> %5.3f of the peak at that clock (256 CUs x 4096 FLOP/clk)\n", name, flop / (ms * 1e-3) / 1e12, mhz,
           flop / (ms * 1e-3) / (256.0 * 4096 * mhz * 1e6));
}

int main() {
    int wall_khz = 100000;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    u32x4* seed; float* sink; unsigned long long* clk;
    hipMalloc(&seed, 65536 * 16); hipMalloc(&sink, 4); hipMalloc(&clk, 512 * 8);
    unsigned* h = (unsigned*)malloc(65536 * 16);
    srand(1);
    for (int arm = 0; arm < 2; ++arm) {
        for (int i = 0; i < 65536 * 4; ++i) {
            // random bf16 pairs in [-2, 2): sign + exponent 126..128 + random mantissa (no NaN / Inf / denormals); second arm: zeros
            const unsigned lo = ((rand() & 1) << 15) | ((126 + rand() % 2) << 7) | (rand() & 127);
            const unsigned hi = ((rand() & 1) << 15) | ((126 + rand() % 2) << 7) | (rand() & 127);
            h[i] = arm == 0 ? (lo | (hi << 16)) : 0u;
        }
        hipMemcpy(seed, h, 65536 * 16, hipMemcpyHostToDevice);
        printf("--- operands: %s (constant-rate counter %d kHz)\n", arm == 0 ? "random bf16" : "zeros", wall_khz);
        const int iters = 4000;
        run<16, false>("16x16x32, MFMA only", seed, iters, sink, clk, wall_khz);
        run<32, false>("32x32x16, MFMA only", seed, iters, sink, clk, wall_khz);
        run<16, true>("16x16x32 + LDS fragment reads", seed, iters, sink, clk, wall_khz);
        run<32, true>("32x32x16 + LDS fragment reads", seed, iters, sink, clk, wall_khz);
    }
    return 0;
}
