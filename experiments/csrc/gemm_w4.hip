// EXPERIMENT (bench only, never loaded by the product): the 256x256 NT tile with FOUR waves -- one per SIMD, wave tile 128x128,
// 256 accumulator registers (AGPRs) + two fragment sets -- instead of the production kernel's eight waves of 128x64.  Two thirds of
// the LDS fragment bytes per MFMA, no second wave on the SIMD to hide a wave's own waits behind: what hipcc makes of a single
// instruction stream per SIMD decides it.  K loop only (plain bf16 stores, no fused epilogue): experiments/gemm_w4.py times it against
// the production kernel on long contractions.
#include "common.h"
#include "gemm_nt256.h"

struct StageOff4 { unsigned off[8]; };
__device__ __forceinline__ void stage_offsets4(StageOff4& o, int ld, int row0, int row_max, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int row = (t * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        o.off[t] = (unsigned)(grow - row0) * (unsigned)ld * 2u + (unsigned)chunk * 16u;
    }
}
__device__ __forceinline__ void stage_issue4(const StageOff4& o, const bf16* ubase_, char* lds_tile, int wave) {
    const char* ubase = uniform_ptr(ubase_);
#pragma unroll
    for (int t = 0; t < 8; ++t)
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))((const char*)ubase + o.off[t]),
                                         (LDS_PTR(void))(lds_tile + (t * 4 + wave) * 1024), 16, 0, 0);
}

__device__ __forceinline__ void stage_piece4(unsigned off, const char* ubase, char* lds_tile, int t, int wave) {
    __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(ubase + off), (LDS_PTR(void))(lds_tile + (t * 4 + wave) * 1024), 16, 0, 0);
}

template <int PIN>
__global__ __launch_bounds__(256, 1) void gemm_nt256w4_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32K | B 32K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;
    const int arow = wm * 128 + (lane & 15), brow = wn * 128 + (lane & 15), gq = lane >> 4;
#define SBW() do { if (PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int t = slot; t < range_n; t += per_xcd) {
        const int tile = range_lo + t;
        const int m0 = (tile / g.tiles_n) * 256, n0 = (tile % g.tiles_n) * 256;
        StageOff4 oa, ob;
        stage_offsets4(oa, g.lda, m0, g.M - 1, wave, lane);
        stage_offsets4(ob, g.ldb, n0, g.N - 1, wave, lane);
        const bf16* pa = g.A + (size_t)m0 * g.lda;
        const bf16* pb = g.B + (size_t)n0 * g.ldb;
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();  // the previous tile's reads are done
        stage_issue4(oa, pa, smem, wave);
        stage_issue4(ob, pb, smem + 32768, wave);
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        if (nk > 1) {
            stage_issue4(oa, pa + BK, smem + 65536, wave);
            stage_issue4(ob, pb + BK, smem + 65536 + 32768, wave);
        }
        f32x4 acc[8][8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 aF[2][8], bF[2][8];
#define LOADF(s, buf, ks)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) aF[s][i] = frag_rows128(buf, arow + i * 16, (ks) * 4 + gq);             \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) bF[s][j] = frag_rows128((buf) + 32768, brow + j * 16, (ks) * 4 + gq)
#define MFMA64(s)                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                          \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                      \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(bF[s][j]), "v"(aF[s][i]))
        LOADF(0, smem, 0);
        if constexpr (PIN == 2) {
            // interleaved: one fragment read of the OTHER set and (behind the barrier) one LDS-DMA piece of stage st + 2 per four MFMAs
#define READ1(s, buf, ks, q)                                                                                               \
    if ((q) < 8) aF[s][(q)] = frag_rows128(buf, arow + (q) * 16, (ks) * 4 + gq);                                          \
    else bF[s][(q) - 8] = frag_rows128((buf) + 32768, brow + ((q) - 8) * 16, (ks) * 4 + gq)
#define MFMA4(s, q)                                                                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                          \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(q) >> 1][((q) & 1) * 4 + e]) : "v"(bF[s][(q) >> 1]), "v"(aF[s][((q) & 1) * 4 + e]))
            for (int st = 0; st < nk; ++st) {
                char* cur = smem + (st & 1) * 65536;
                const char* nxt = smem + ((st + 1) & 1) * 65536;
                __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    READ1(1, cur, 1, qq);
                    __builtin_amdgcn_sched_barrier(0);
                    MFMA4(0, qq);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_waitcnt(0x0070);
                __builtin_amdgcn_s_barrier();
                const bool dma = st + 2 < nk, more = st + 1 < nk;
                const char* ua = uniform_ptr(pa + (size_t)(st + 2) * BK);
                const char* ub = uniform_ptr(pb + (size_t)(st + 2) * BK);
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    if (dma) {
                        if (qq < 8) stage_piece4(oa.off[qq], ua, cur, qq, wave);
                        else stage_piece4(ob.off[qq - 8], ub, cur + 32768, qq - 8, wave);
                    }
                    if (more) { READ1(0, nxt, 0, qq); }
                    __builtin_amdgcn_sched_barrier(0);
                    MFMA4(1, qq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef READ1
#undef MFMA4
        } else
        for (int st = 0; st < nk; ++st) {
            const char* cur = smem + (st & 1) * 65536;
            const char* nxt = smem + ((st + 1) & 1) * 65536;
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): set 0 was requested 64 MFMAs ago -- free, and it tells the compiler so
            LOADF(1, cur, 1);
            SBW();
            MFMA64(0);
            SBW();
            __builtin_amdgcn_s_waitcnt(0x0070);  // set 1 (and the next stage's DMA)
            __builtin_amdgcn_s_barrier();
            if (st + 2 < nk) {
                stage_issue4(oa, pa + (size_t)(st + 2) * BK, (char*)cur, wave);
                stage_issue4(ob, pb + (size_t)(st + 2) * BK, (char*)cur + 32768, wave);
            }
            if (st + 1 < nk) { LOADF(0, nxt, 0); }
            SBW();
            MFMA64(1);
            SBW();
        }
#undef LOADF
#undef MFMA64
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the stores read them (asm MFMAs: no compiler hazard handling)
        // plain bf16 stores: lane owns 4 consecutive columns of one row per accumulator tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + i * 16 + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + wn * 128 + j * 16 + gq * 4;
                if (n >= g.N) continue;
                const f32x4 v = acc[j][i];
                *(bf16x4*)((bf16*)g.out + (size_t)m * g.ldc + n) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            }
        }
    }
#undef SBW
}

extern "C" int tvts_exp_gemm_w4(int pin, const void* A, int lda, const void* B, int ldb, int M, int N, int K, void* out, int ldc,
                                hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK || N % 8 || lda % 8 || ldb % 8 || ldc % 4) return -22;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.bias = nullptr; g.residual = nullptr; g.ldr = 0; g.act = 0; g.preact = nullptr; g.ldp = 0; g.gate_h = nullptr; g.ldh = 0;
    g.gate_act = 0; g.out = out; g.ldc = ldc; g.out_f32 = 0; g.sa = nullptr; g.sb = nullptr; g.sa_rows = 0; g.gc = 0;
    g.tiles_n = (N + 255) / 256; g.tiles_m = (M + 255) / 256;
    const int total = g.tiles_m * g.tiles_n;
    const int grid = total < 256 ? ((total + 7) / 8) * 8 : 256;
    void (*kern)(GemmNT) = pin == 2 ? gemm_nt256w4_kernel<2> : pin ? gemm_nt256w4_kernel<1> : gemm_nt256w4_kernel<0>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 131072, stream, g);
    return (int)hipGetLastError();
}
