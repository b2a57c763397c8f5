// BENCH-ONLY experiment library (libtvts_exp.so): variants of the production 256x256 NT kernel, instantiated from the
// production header so that A/B timings compare like with like.  Nothing under tvts_amd/*.py loads this library;
// experiments/gemm_ab.py does.  A variant that wins moves into gemm_nt256.h / gemm.hip; one that does not stays here (or is deleted).
#include "gemm_nt256.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ExpArgs {
    int stagger_phases;  // > 1: blocks start in this many phases, phase p delayed by p * stagger_units * (s_sleep 127)
    int stagger_units;
};

__device__ __forceinline__ void start_stagger(const ExpArgs& x) {
    if (x.stagger_phases > 1) {
        const int phase = (blockIdx.x >> 3) % x.stagger_phases;
        for (int i = 0; i < phase * x.stagger_units; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// ---- variant 0: the production kernel with the block-start stagger in front (tile walk gc comes in through GemmNT)
template <int ACT, int GATE>
__global__ __launch_bounds__(512, 2) void gemm_nt256p_stag_kernel(GemmNT g, ExpArgs x);

// ------------------------------------------------------------------------------------------------
// variant M32: v_mfma_f32_32x32x16_bf16 instead of 16x16x32 (ubench ceiling 2382 vs 2075 TF, half the MFMA instructions
// per stage).  Wave tile 128(m) x 64(n) = 4 x 2 accumulator tiles of 32x32 (128 registers, as before); a 64-deep stage is
// 4 k-steps of 16, each 4 A + 2 B fragments (ds_read_b128) and 8 MFMAs; fragment sets double-buffered per k-step.
// LDS image: 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7) -- the 32-row fragment read (lane = row & 31, chunk by
// lane >> 5) is conflict-free with that swizzle and 2-way conflicted with the production one (c ^ (r & 7)).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_offsets256_m32(StageOff256& o, int ld, int row0, int row_max, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = (t * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        o.off[t] = (unsigned)(grow - row0) * (unsigned)ld * 2u + (unsigned)chunk * 16u;
    }
}
__device__ __forceinline__ bf16x8 frag_rows_m32(const char* lds_tile, int row, int chunk) {
    return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int ACT, int GATE>
__device__ __forceinline__ void epilogue256_patch_m32(const GemmNT& g, f32x16 (&acc)[2][4], int m0, int n0, int wm, int wn,
                                                      int lane, char* patch) {
    // accumulator tile (j, i): lane holds m = 32 i + (lane & 31), n = 32 j + 8 gi + 4 (lane >> 5) + (0..3) in regs 4 gi .. 4 gi + 3
    const int nb = n0 + wn * 64;
    const int r16 = lane & 15, half = (lane >> 4) & 1, hi = lane >> 5;
    // bias: 8 x 16 B per lane; re-read per slab from L1 instead of parking 32 registers next to the 128 accumulators
    const float* bias = g.bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (half == p) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        const int c = j * 8 + gi * 2 + hi;
                        const f32x4 v = {acc[j][i][4 * gi], acc[j][i][4 * gi + 1], acc[j][i][4 * gi + 2], acc[j][i][4 * gi + 3]};
                        const int n = nb + c * 4;
                        const f32x4 b4 = (bias && n < g.N) ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        *(f32x4*)(patch + r16 * 256 + ((c ^ r16) << 4)) = v + b4;
                    }
            }
            patch_readout<ACT, GATE>(g, patch, m0 + wm * 128 + i * 32 + p * 16, nb, lane, SideSlab());
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    }
}

template <int ACT, int GATE, int PRIO>
__global__ __launch_bounds__(512, 2) void gemm_nt256m32_kernel(GemmNT g, ExpArgs x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32K | B 32K] + 8 x 4K patches
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    char* patch = smem + 131072 + wave * 4096;

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;
    if (slot >= range_n) return;
    const int ntl = (range_n - slot + per_xcd - 1) / per_xcd;
    const int total_st = ntl * nk;
    const int gc = g.gc;
    start_stagger(x);

    int i_st = 0, i_kt = 0, i_tl = 0, i_m0, i_n0;
    StageOff256 oa, ob;
    {
        tile_origin256(g, range_lo + slot, gc, i_m0, i_n0);
        stage_offsets256_m32(oa, g.lda, i_m0, g.M - 1, wave, lane);
        stage_offsets256_m32(ob, g.ldb, i_n0, g.N - 1, wave, lane);
    }
    auto issue = [&]() {
        char* dst = smem + (i_st & 1) * 65536;
        stage_issue256<0>(oa, g.A + (size_t)i_m0 * g.lda + i_kt * BK, dst, wave);
        stage_issue256<0>(ob, g.B + (size_t)i_n0 * g.ldb + i_kt * BK, dst + 32768, wave);
        ++i_st;
        if (++i_kt == nk) {
            i_kt = 0; ++i_tl;
            tile_origin256(g, range_lo + slot + i_tl * per_xcd, gc, i_m0, i_n0);
            stage_offsets256_m32(oa, g.lda, i_m0, g.M - 1, wave, lane);
            stage_offsets256_m32(ob, g.ldb, i_n0, g.N - 1, wave, lane);
        }
    };
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER_P();
    if (i_st < total_st) issue();

    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    int kt = 0, tl = 0, m0, n0;
    tile_origin256(g, range_lo + slot, gc, m0, n0);
    const int arow = wm * 128 + (lane & 31), brow = wn * 64 + (lane & 31), hi = lane >> 5;
    bf16x8 aF[2][4], bF[2][2];
#define LOADF(s, buf, kk)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) bF[s][j] = frag_rows_m32((buf) + 32768, brow + j * 32, (kk) * 2 + hi); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) aF[s][i] = frag_rows_m32(buf, arow + i * 32, (kk) * 2 + hi)
#define MFMA8(s)                                                                                              \
    if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bF[s][j], aF[s][i], acc[j][i], 0, 0, 0);      \
    if (PRIO) __builtin_amdgcn_s_setprio(0)
    LOADF(0, smem, 0);

    for (int st = 0; st < total_st; ++st) {
        const char* cur = smem + (st & 1) * 65536;
        const char* nxt = smem + ((st + 1) & 1) * 65536;
        LOADF(1, cur, 1);
        MFMA8(0);
        LOADF(0, cur, 2);
        MFMA8(1);
        LOADF(1, cur, 3);
        MFMA8(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        RAW_BARRIER_P();
        if (i_st < total_st) issue();
        if (st + 1 < total_st) { LOADF(0, nxt, 0); }
        MFMA8(1);
        if (++kt == nk) {
            epilogue256_patch_m32<ACT, GATE>(g, acc, m0, n0, wm, wn, lane, patch);
            kt = 0; ++tl;
            tile_origin256(g, range_lo + slot + tl * per_xcd, gc, m0, n0);
        }
    }
#undef LOADF
#undef MFMA8
}

// variant 0 body: the production kernel cannot take the extra argument, so the stagger variant is a thin re-statement of its
// prologue is not possible without copying it -- instead variant 0 launches the production kernel itself (no stagger).

template <int PRIO>
static void (*pick_m32(int act, int gate_act, bool gated))(GemmNT, ExpArgs) {
    if (gated) return gate_act == ACT_QUICK_GELU ? gemm_nt256m32_kernel<0, 1, PRIO> : gate_act == ACT_GELU_ERF ? gemm_nt256m32_kernel<0, 2, PRIO> : nullptr;
    return act == ACT_NONE ? gemm_nt256m32_kernel<0, 0, PRIO> : act == ACT_QUICK_GELU ? gemm_nt256m32_kernel<1, 0, PRIO>
         : act == ACT_GELU_ERF ? gemm_nt256m32_kernel<2, 0, PRIO> : nullptr;
}

// register-path epilogue variants: the output kind (fp32 output, fp32 residual) is a template argument
template <int ABL>
static void (*pick_reg(int act, int gate_act, bool gated, int out_f32, bool res))(GemmNT) {
    if (gated) return out_f32 || res ? nullptr : gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, false, ABL, 0> : gemm_nt256p_kernel<0, 2, false, ABL, 0>;
    if (act != ACT_NONE) return out_f32 || res ? nullptr : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, false, ABL, 0> : gemm_nt256p_kernel<2, 0, false, ABL, 0>;
    return out_f32 ? (res ? gemm_nt256p_kernel<0, 0, false, ABL, 3> : gemm_nt256p_kernel<0, 0, false, ABL, 1>)
                   : (res ? gemm_nt256p_kernel<0, 0, false, ABL, 2> : gemm_nt256p_kernel<0, 0, false, ABL, 0>);
}

// hand-scheduled patch epilogue (bias inside the K loop, asm stores from scalar bases, asm side loads with counted waits)
template <int ABL>
static void (*pick_pasm(int act, int gate_act, bool gated, int out_f32, bool res))(GemmNT) {
    if (act != ACT_NONE) return (gated || out_f32 || res) ? nullptr : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, false, ABL, 0> : nullptr;
    if (gated) return out_f32 || res ? nullptr : gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, false, ABL, 0>
                      : gate_act == ACT_ADD_BF16 ? gemm_nt256p_kernel<0, 3, false, ABL, 0> : nullptr;
    return out_f32 ? (res ? gemm_nt256p_kernel<0, 0, false, ABL, 3> : gemm_nt256p_kernel<0, 0, false, ABL, 1>)
                   : (res ? gemm_nt256p_kernel<0, 0, false, ABL, 2> : gemm_nt256p_kernel<0, 0, false, ABL, 0>);
}

template <int ABL>
static void (*pick_abl(int act, int gate_act, bool gated))(GemmNT) {
    if (gated) return gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, false, ABL> : gate_act == ACT_ADD_BF16 ? gemm_nt256p_kernel<0, 3, false, ABL>
                      : gemm_nt256p_kernel<0, 2, false, ABL>;
    return act == ACT_NONE ? gemm_nt256p_kernel<0, 0, false, ABL> : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, false, ABL>
         : gemm_nt256p_kernel<2, 0, false, ABL>;
}

static void* g_trace = nullptr;  // [grid][32 tiles][6] u64 time stamps (variants with ABL & 2048)
extern "C" void tvts_exp_set_trace(void* p) { g_trace = p; }

// variant: 0 production kernel, 1 M32, 2 M32 + setprio, 10 + ABL: the production kernel with epilogue ablation ABL (1 none, 2 no side
// loads, 4 no stores, 6 neither).  gc < 0: the production column-group rule.
extern "C" int tvts_exp_gemm_nt(int variant, int gc, int stagger_phases, int stagger_units, const void* A, int lda, const void* B,
                                int ldb, int M, int N, int K, const float* bias, const float* residual, int ldr, int act,
                                void* preact, int ldp, const void* gate_h, int ldh, int gate_act, void* out, int ldc, int out_f32,
                                hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.preact = (bf16*)preact; g.ldp = ldp; g.gate_h = (const bf16*)gate_h; g.ldh = ldh; g.gate_act = gate_act;
    g.out = out; g.ldc = ldc; g.out_f32 = out_f32; g.sa = nullptr; g.sb = nullptr; g.sa_rows = 0;
    g.side_deriv = 1;  // the production forms of round 5: the side tensor IS act'(x)
    g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_tol = 0; g.q8 = nullptr; g.ldq8 = 0; g.q8_scale = nullptr; g.q8_amax = nullptr; g.clk = nullptr;
    g.tiles_n = ceil_div(N, 256); g.tiles_m = ceil_div(M, 256);
    if (gc < 0) { const int tn = g.tiles_n; gc = tn >= 10 ? (tn % 6 == 0 ? 6 : tn % 5 == 0 ? 5 : 0) : 0; }
    g.gc = gc;
    const int total_tiles = g.tiles_m * g.tiles_n;
    const int grid = total_tiles < 256 ? ((total_tiles + 7) / 8) * 8 : 256;
    ExpArgs x{stagger_phases, stagger_units};
    const bool gated = gate_h != nullptr;
    if (variant == 0 || variant >= 10) {
        void (*kern)(GemmNT) = nullptr;
        switch (variant) {
            case 0: kern = pick_abl<0>(act, gate_act, gated); break;
            case 11: kern = pick_abl<1>(act, gate_act, gated); break;
            case 12: kern = pick_abl<2>(act, gate_act, gated); break;
            case 14: kern = pick_abl<4>(act, gate_act, gated); break;
            case 16: kern = pick_abl<6>(act, gate_act, gated); break;
            case 18: kern = pick_abl<8>(act, gate_act, gated); break;    // 2-phase stagger
            case 26: kern = pick_abl<16>(act, gate_act, gated); break;   // 4-phase stagger
            case 42: kern = pick_abl<32>(act, gate_act, gated); break;   // sc1 stores
            case 74: kern = pick_abl<64>(act, gate_act, gated); break;   // plain stores (round 1)
            case 138: kern = pick_abl<128>(act, gate_act, gated); break; // nt stores + nt side loads
            case 266: kern = pick_abl<256>(act, gate_act, gated); break; // side inputs prefetched one slab ahead
            case 394: kern = pick_abl<384>(act, gate_act, gated); break; // prefetch + nt side loads
            case 2058: kern = pick_abl<2048>(act, gate_act, gated); g.sa = (const float*)g_trace; break;  // production + time stamps
            case 2074: kern = pick_abl<2064>(act, gate_act, gated); g.sa = (const float*)g_trace; break;  // 4-phase stagger + time stamps
            case 3594: kern = pick_reg<3584>(act, gate_act, gated, out_f32, residual != nullptr); g.sa = (const float*)g_trace; break;  // reg + cnt + stamps
            case 8202: kern = pick_pasm<8192>(act, gate_act, gated, out_f32, residual != nullptr); break;        // patch + counted side loads
            case 8714: kern = pick_pasm<8192 + 512>(act, gate_act, gated, out_f32, residual != nullptr); break;  // ... + counted vmcnt behind the epilogue
            case 32778: kern = pick_abl<32768>(act, gate_act, gated); break;                                             // generic epilogue + staggered DMA issue
            case 40970: kern = pick_pasm<8192 + 32768>(act, gate_act, gated, out_f32, residual != nullptr); break;    // hand-scheduled epilogue + staggered DMA issue
            case 32779: kern = pick_abl<32768 + 1>(act, gate_act, gated); break;                                        // no epilogue + staggered DMA issue
            case 163850: kern = pick_abl<32768 + 131072>(act, gate_act, gated); break;                                    // generic epilogue + stagger + pinned read / MFMA order
            case 172042: kern = pick_pasm<8192 + 32768 + 131072>(act, gate_act, gated, out_f32, residual != nullptr); break;  // hand-scheduled epilogue + stagger + pinned order
            case 163851: kern = pick_abl<32768 + 131072 + 1>(act, gate_act, gated); break;                                // no epilogue + stagger + pinned order
            case 8552458: kern = pick_abl<32768 + 131072 + 8388608>(act, gate_act, gated); break;                               // generic epilogue, bf16-first patch where it applies (round 6)
            case 8560650: kern = pick_pasm<8192 + 32768 + 131072 + 8388608>(act, gate_act, gated, out_f32, residual != nullptr); break;  // hand-scheduled epilogue, bf16-first patch where it applies
            case 25337866: kern = pick_pasm<8192 + 32768 + 131072 + 8388608 + 16777216>(act, gate_act, gated, out_f32, residual != nullptr); break;  // ... and for the bf16 side-input forms
            case 522: kern = pick_abl<512>(act, gate_act, gated); break;   // counted vmcnt behind the epilogue
            case 1034: kern = pick_reg<1024>(act, gate_act, gated, out_f32, residual != nullptr); break; // register-path epilogue
            case 1546: kern = pick_reg<1536>(act, gate_act, gated, out_f32, residual != nullptr); break; // ... + counted vmcnt
            case 1562: kern = pick_reg<1552>(act, gate_act, gated, out_f32, residual != nullptr); break; // ... + 4-phase stagger
            default: return TVTS_EINVAL;
        }
        if (!kern) return TVTS_EINVAL;
        if (variant >= 1034 && variant != 2058 && variant != 2074 && variant != 32778 && variant != 32779 && variant != 163850 && variant != 163851 && variant != 8552458 && K < 2 * BK) return TVTS_EINVAL;  // the register-path epilogue requests the bias two stages before the tile ends
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, stream, g);
    } else {
        void (*kern)(GemmNT, ExpArgs) = variant == 2 ? pick_m32<1>(act, gate_act, gated) : pick_m32<0>(act, gate_act, gated);
        if (!kern) return TVTS_EINVAL;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, stream, g, x);
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
