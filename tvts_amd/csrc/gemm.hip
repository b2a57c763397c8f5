// bf16 MFMA GEMMs for the TVTSv2 step on gfx950.
//
//   tvts_gemm_nt_bf16   : C[M,N] = epi(A[M,K] . B[N,K]^T)   (nn.Linear forward; dgrad with the [K,N] weight copy)
//   tvts_gemm_tn_bf16   : C[Na,Nb] (+)= P[M,Na]^T . Q[M,Nb]  (weight gradient, contraction over the token rows)
//   tvts_gemm_small_f32 : strided fp32 fallback for the handful of tiny matmuls (heads, projections of [B,E] rows)
//   tvts_colsum_bf16    : bias gradient, column sums of a bf16 [M,N] matrix accumulated into fp32
//
// Tiling (both MFMA kernels): 128x128 output tile, 64-deep stage, 256 threads = 4 waves in 2x2, each wave
// a 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 tiles.  Operands go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double buffered.  The LDS image is lane-linear as the
// DMA demands; bank conflicts are removed by XOR-swizzling the per-lane SOURCE address and applying the
// same involution on the ds_read side.
//
// MFMA roles are swapped (weights feed the A operand, activations the B operand) so that a lane ends up
// with 4 consecutive output columns of one output row: 8-byte bf16 / 16-byte fp32 epilogue accesses with
// bias, activation, activation-gradient gate and fp32 residual fused.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#include "gemm_nt256.h"

#define BM 128
#define BN 128
#define NTHREADS 256

// --- one [128 rows][64 k] bf16 tile: 16 KiB, rows of 128 B, 16-B chunk c of row r stored at chunk c^(r&7)
__device__ __forceinline__ void stage_rows128(const bf16* __restrict__ base, int ld, int row0, int row_max,
                                              int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r0 = (t * 4 + wave) * 8;
        const int row = r0 + (lane >> 3);
        const int slot = lane & 7;
        const int chunk = slot ^ (row & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        const bf16* src = base + (size_t)grow * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))src, (LDS_PTR(void))(lds_tile + r0 * 128), 16, 0, 0);
    }
}


template <int ACT, int GATE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // Persistent blocks.  XCD x (= blockIdx % 8, see xcd note in common.h) owns a contiguous range of tiles; its
    // blocks walk that range with stride (blocks per XCD), n fastest, so co-resident blocks of one XCD share the
    // A row-panels and the whole weight panel in that XCD's L2.
    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;

    int t = slot;  // index inside this XCD's range
    if (t >= range_n) return;
    int tile = range_lo + t;
    int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    stage_rows128(g.A, g.lda, m0, g.M - 1, 0, smem, wave, lane);
    stage_rows128(g.B, g.ldb, n0, g.N - 1, 0, smem + 16384, wave, lane);
    __syncthreads();
    int stage = 0;
    while (true) {
        f32x4 acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int t_next = t + per_xcd;
        const bool has_next = t_next < range_n;
        const int tile_n = range_lo + t_next;
        const int m0n = (tile_n / g.tiles_n) * BM, n0n = (tile_n % g.tiles_n) * BN;

        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + stage * 32768;
            char* nxt = smem + (stage ^ 1) * 32768;
            if (kt + 1 < nk) {
                stage_rows128(g.A, g.lda, m0, g.M - 1, (kt + 1) * BK, nxt, wave, lane);
                stage_rows128(g.B, g.ldb, n0, g.N - 1, (kt + 1) * BK, nxt + 16384, wave, lane);
            } else if (has_next) {  // first stage of the NEXT tile flies under this tile's last MFMAs + epilogue
                stage_rows128(g.A, g.lda, m0n, g.M - 1, 0, nxt, wave, lane);
                stage_rows128(g.B, g.ldb, n0n, g.N - 1, 0, nxt + 16384, wave, lane);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag_rows128(cur, wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bfr[j] = frag_rows128(cur + 16384, wn * 64 + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
            }
            stage ^= 1;
            if (kt + 1 < nk) __syncthreads();
        }

        // epilogue: lane owns row m (column of the swapped MFMA) and 4 consecutive n.  Side inputs are loaded UNCONDITIONALLY (rows /
        // columns past the matrix clamped to its last ones) and a row-tile's four at a time: with every load inside its own
        // `if (g.bias) / if (m < M)` block hipcc waited vmcnt(0) behind each one -- up to three dependent memory round trips for each of
        // the 16 sub-tiles, each of which also waited for the previous sub-tile's stores (round 5, from the ISA: the kernel of the
        // reference's 12 / 24-pair batches, 36 us per launch of which the epilogue was a third)
        f32x4 bias4[4];
        int ncol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            ncol[j] = n < g.N ? n : g.N - 4;
            bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (g.bias) {  // (block-uniform: one block of four loads, one wait)
#pragma unroll
            for (int j = 0; j < 4; ++j) bias4[j] = *(const f32x4*)(g.bias + ncol[j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            const size_t mc = (size_t)(m < g.M ? m : g.M - 1);
            bf16x4 h4[4];
            f32x4 r4[4];
            if (GATE != ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = *(const bf16x4*)(g.gate_h + mc * g.ldh + ncol[j]);
            }
            if (g.residual) {  // (block-uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = *(const f32x4*)(g.residual + mc * g.ldr + ncol[j]);
            }
            // every value of the row-tile first, then its stores: a store between two uses of loaded side inputs makes the second
            // use wait vmcnt(0), i.e. for that store's acknowledgement
            f32x4 vv[4], sd4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[j][i];
                if (g.bias) v += bias4[j];
                if (ACT != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float t; v[e] = act_fwd_side(v[e], ACT, g.side_deriv, t); sd4[j][e] = t; }
                }
                if (GATE != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gate_apply(v[e], (float)h4[j][e], GATE, g.side_deriv);
                }
                if (g.residual) v += r4[j];
                vv[j] = v;
            }
            // (pinned: left alone, hipcc sinks each value's arithmetic into the conditional block of its store, behind the previous
            // store, and the wait for the loaded side inputs at that block's entry is vmcnt(0) again)
            asm volatile("" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]));
            if (ACT != ACT_NONE) asm volatile("" : "+v"(sd4[0]), "+v"(sd4[1]), "+v"(sd4[2]), "+v"(sd4[3]));
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                const f32x4 v = vv[j];
                if (ACT != ACT_NONE && g.preact)
                    *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)sd4[j][0], (bf16)sd4[j][1], (bf16)sd4[j][2], (bf16)sd4[j][3]};
                if (g.out_f32) {
                    *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
                } else {
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *(bf16x4*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
                }
            }
        }
        if (!has_next) break;
        __syncthreads();  // next tile's first stage has landed (vmcnt(0)) and every wave is done with `cur`
        t = t_next; m0 = m0n; n0 = n0n;
    }
}

#include "gemm_nt_ring.h"

// ---- dispatch.  Tile choice: the pipelined 256x256 kernel when N is a multiple of 256 and the output has at least
// NT_MIN_TILES_256 of its tiles (the text tower of a 192-pair step has 192 of them, M = 24 576, N = 512, and runs ~10 % faster
// on it), else the persistent 128x128 kernel.  The per-call `opts` word (include/tvts_hip.h, TVTS_GEMM_*) overrides for benches and
// tests: the library keeps NO mutable process state (the entry points are called from the autograd worker thread and from
// communication hooks as well as from the main thread).
static const int NT_MIN_TILES_256 = 150;
static inline int opt_tile(int opts) { return (opts & 1) ? 128 : (opts & 2) ? 256 : 0; }
// persistent grid of the 256x256 NT kernel: at most this many blocks (one per CU; a multiple of 8, one share per XCD).  256 = the
// whole chip; fewer leaves CUs to kernels of other streams (a 160 KiB block shares its CU with nothing), and is how
// tools/gemm_cus.py measures what a CU's K loop waits for (per-CU rate against the number of CUs streaming)
static inline int opt_cus(int opts) {
    const int n = ((opts >> 8) & 63) * 8;
    return n == 0 ? 256 : n > 256 ? 256 : n;
}

static bool nt_use_256(int M, int N, int K, bool has_ws, int opts) {
    const int forced = opt_tile(opts);
    if (forced == 128) return false;
    if (N % 8) return false;
    if (forced == 256 || (opts & 32)) return true;
    if (N % 256) return false;
    (void)K; (void)has_ws;
    // ... or as soon as the 128x128 kernel would need a second round of its 512 co-resident blocks (129-149 tiles of 256: the text
    // tower at 24 pairs, M = 18 936 x N = 512: 54 vs 72 us at K = 2048, profiles/r03_gemm_tile_choice_small_m.txt)
    return (long)ceil_div(M, 256) * (N / 256) >= NT_MIN_TILES_256 || (long)ceil_div(M, 128) * (N / 128) > 512;
}

// Wide outputs (>= 10 tile columns: the MLP's 4x expansion): walk the tiles in column groups of 6 or 5.  An XCD's 32
// co-resident tiles then span 5-6 weight panels x 5-6 row panels instead of all 12-20 weight panels x 2-3 row panels
// -- FETCH_SIZE (L2 misses) of the fc1 / fc2-dgrad GEMMs falls by 40 %, time by 3-5 % (profiles/r01_pmc_gemm_traffic_by_shape.txt).
static int nt_column_group(int N) {
    const int tn = ceil_div(N, 256);
    if (tn >= 10) return tn % 6 == 0 ? 6 : tn % 5 == 0 ? 5 : 0;
    if (tn == 9) return 3;  // qkv (N = 2304): 553 -> 520 us at M = 150 720 (tools/gemm_ab.py with rotating buffers, round 2)
    return 0;
}

// ---- stream-K plan of the 256x256 NT kernel (gemm_nt256.h).  Workspace layout: [NT_SK_MAX_TILES arrival counters][grid x 2
// partial tiles of 256 x 256 fp32]; the counters must be zero on entry and are zero again when the launch has finished.
static const int NT_SK_MAX_TILES = 16384;
static inline size_t nt_sk_workspace_bytes(int grid) { return (size_t)NT_SK_MAX_TILES * 4 + (size_t)grid * 2 * 65536 * 4; }
extern "C" long tvts_gemm_nt_workspace_bytes(void) { return (long)nt_sk_workspace_bytes(256); }
// Measured (tools/gemm_sk.py, profiles/r04_gemm_streamk_nt.txt): the stage-granular walk LOSES to the tile walk on every shape
// of the step at 12 / 24 pairs (1.2 - 2.3x): a 256 x 256 fp32 partial is 256 KiB, every block leaves two of them, and the
// epilogue (6.7 of a tile's 20.5 us at K = 768) is not split by splitting K -- it is serialised in the block that arrives last.
// It therefore only runs when a call asks for it (TVTS_GEMM_STREAMK: parity tests, benches); what the small batches gain comes
// from running the weight gradients beside the input-gradient chain (tvts_amd/engine.py, wgrad stream).
static bool nt_sk_wanted(int opts) { return (opts & 32) != 0 && !(opts & 64); }

template <bool FP8>
static int launch_nt256(GemmNT& g, int act, int gate_act, bool gated, int opts, hipStream_t stream, void* workspace = nullptr,
                        long workspace_bytes = 0) {
    const int nt_cus = opt_cus(opts);
    const bool fp8_mx = !(opts & 4);
    g.tiles_n = ceil_div(g.N, 256);
    g.tiles_m = ceil_div(g.M, 256);
    g.gc = nt_column_group(g.N);
    const int total_tiles = g.tiles_m * g.tiles_n;
    int grid = total_tiles < nt_cus ? ((total_tiles + 7) / 8) * 8 : nt_cus;  // persistent: one block per CU, multiple of 8 (XCDs)
    g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_tol = 0;
    if constexpr (!FP8) {
        // stream-K: needs the caller's workspace, at least 3 stages per block on the emptiest XCD (no empty blocks: the
        // contributor lists are contiguous block ranges), and pays off where the tile walk wastes a good part of a round
        const int nk = g.K / BK;
        const long min_units = (long)(total_tiles / 8) * nk;  // the XCD with the fewest tiles
        int sk_grid = nt_cus;
        while (sk_grid > 8 && min_units / (sk_grid / 8) < 3) sk_grid -= 8;
        const bool can = workspace != nullptr && total_tiles >= 8 && total_tiles <= NT_SK_MAX_TILES && min_units / (sk_grid / 8) >= 3 &&
                         (size_t)workspace_bytes >= nt_sk_workspace_bytes(sk_grid) && nk >= 2;
        if ((opts & 32) && !can) return TVTS_EINVAL;  // forced, but the shape / workspace cannot take it
        if (can && nt_sk_wanted(opts)) {
            grid = sk_grid;
            g.sk_cnt = (int*)workspace;
            g.sk_ws = (float*)((char*)workspace + (size_t)NT_SK_MAX_TILES * 4);
            const int upb = (int)(min_units / (sk_grid / 8));
            const int tol = nk / 8 < (upb / 2 - 1) ? nk / 8 : (upb / 2 - 1);
            g.sk_tol = tol > 0 ? tol : 0;
        }
    }
    // every production instantiation staggers the LDS-DMA issue of the two waves of a SIMD (ABL 32768: +1-3 % on every shape of
    // the step, tools/gemm_ab.py; the same change is worth 6-10 % on the weight-gradient kernel)
    // ... and pins the written order of fragment reads and MFMA groups in the bf16 K loop (ABL 131072, round 3: hipcc otherwise
    // sinks every ds_read_b128 in front of its consumer and waits lgkmcnt(0) there; -1.2 % over the step's shapes with their
    // epilogues, tools/gemm_ab.py "pasm+pin" / "pin", profiles/r03_gemm_ab_pin.txt)
#ifndef TVTS_NT_SD
#define TVTS_NT_SD (32768 | 131072)   // overridable for A/B builds (tools/dbg/ab_flags.sh), e.g. | 262144: LDS-DMA from inline asm -- exact
                                      // lgkmcnt counts in the K loop instead of lgkmcnt(0), and no change in the step (145.5-146.0 ms either way)
#endif
    constexpr int SD = TVTS_NT_SD;
    const bool b16_patch = !(opts & 4194304);  // TVTS_GEMM_F32_PATCH: the fp32 patch of rounds 1 - 5 for the plain forms
    void (*kern)(GemmNT) = nullptr;
    if (gated) {
        if (act != ACT_NONE) return TVTS_EINVAL;
        if constexpr (FP8)  // e4m3 dgrad with the activation-gradient gate (tvts_gemm_nt_fp8_gate): the scaled-MFMA main loop only
            kern = gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, true, SD | 65536> : gate_act == ACT_GELU_ERF ? gemm_nt256p_kernel<0, 2, true, SD | 65536>
                 : gate_act == ACT_ADD_BF16 ? gemm_nt256p_kernel<0, 3, true, SD | 65536> : nullptr;
        else
            kern = gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, false, SD> : gate_act == ACT_GELU_ERF ? gemm_nt256p_kernel<0, 2, false, SD>
                 : gate_act == ACT_ADD_BF16 ? gemm_nt256p_kernel<0, 3, false, SD> : nullptr;
    } else {
        kern = act == ACT_NONE ? gemm_nt256p_kernel<0, 0, FP8, SD> : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, FP8, SD>
             : act == ACT_GELU_ERF ? gemm_nt256p_kernel<2, 0, FP8, SD> : nullptr;
        // round 6: plain results (no activation, no side input) through the bf16-FIRST patch (gemm_nt256.h, ABL 8388608: the tile is
        // rounded in the accumulator layout and crosses the LDS patch as bf16, two slabs per patch): same bits as the fp32 patch,
        // 187 -> 177 us at N = K = 768, 513 -> 498 us at N = 2304 (M = 150 720, profiles/r06_gemm_ab_b16_patch.txt).  The kernel takes
        // it where the call has a bf16 result and no residual and falls back to the fp32 patch inside otherwise; TVTS_GEMM_F32_PATCH
        // asks for the old kernel (parity test of the two).
        if constexpr (!FP8) {
            if (act == ACT_NONE && b16_patch) kern = gemm_nt256p_kernel<0, 0, false, SD | 8388608>;
        }
        if constexpr (FP8) {  // the K = 128 scaled-MFMA main loop (the fp8 issue rate); TVTS_GEMM_FP8_K32 selects the 16x16x32 form
            constexpr int MX = SD | 65536;
            if (fp8_mx) kern = act == ACT_NONE ? gemm_nt256p_kernel<0, 0, true, MX> : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, true, MX>
                               : act == ACT_GELU_ERF ? gemm_nt256p_kernel<2, 0, true, MX> : nullptr;
        }
    }
    // fp32 residual in the epilogue, no activation: the instantiation that requests the residual of slab i + 1 as soon as slab i
    // has consumed its registers (ABL 256: proj + residual 334 -> 294 us, fc2 + residual 749 -> 713 us at M = 150 720; it costs the
    // plain / activation / gate kernels 1-12 %, so they keep the in-place loads)
    if (!gated && act == ACT_NONE && g.residual) kern = gemm_nt256p_kernel<0, 0, FP8, 256 | SD>;
    if constexpr (FP8) {
        if (fp8_mx && act == ACT_NONE && g.residual) kern = gemm_nt256p_kernel<0, 0, true, 256 | SD | 65536>;
    }
    // The hand-scheduled patch epilogue (ABL 8192: bias added inside the K loop, stores by inline asm from scalar bases, side
    // inputs by inline asm two row-tiles ahead of their use and ahead of the stores, hand-counted vmcnt) where tools/gemm_ab.py
    // measures a gain at M = 150 720: QuickGELU gate 927 -> 817 us (fc2 dgrad), QuickGELU + pre-activation 908 -> 880 us (fc1),
    // bf16 + fp32 residual 260 -> 247 us, fp32 + residual 743 -> 712 us (K = 3072; = ABL 256 at K = 768), plain bf16 at N = 2304
    // 535 -> 524 us; plain N = 768 outputs lose 0-2 % and keep the generic epilogue, the erf-GELU forms spill and keep it too.
    // Needs whole tile columns, K of two stages or more and 32-bit byte offsets into the output / side matrices.
    if constexpr (!FP8) {
        const unsigned long long lim = 1ull << 32, Mu = (unsigned long long)g.M;
        const bool fits = g.N % 256 == 0 && g.K >= 2 * BK && Mu * (unsigned long long)g.ldc * (g.out_f32 ? 4ull : 2ull) < lim &&
                          (!g.residual || Mu * (unsigned long long)g.ldr * 4ull < lim) &&
                          (!g.preact || Mu * (unsigned long long)g.ldp * 2ull < lim) &&
                          (!gated || Mu * (unsigned long long)g.ldh * 2ull < lim);
        if (fits) {
            if (gated) {
                if (gate_act == ACT_QUICK_GELU && !g.out_f32 && !g.residual) kern = gemm_nt256p_kernel<0, 1, false, 8192 | SD, 0>;
                // bf16 residual added to a bf16 result (the residual stream of the space-time blocks): the gate form's traffic
                if (gate_act == ACT_ADD_BF16 && !g.out_f32 && !g.residual) kern = gemm_nt256p_kernel<0, 3, false, 8192 | SD, 0>;
            } else if (act == ACT_QUICK_GELU) {
                if (!g.out_f32 && !g.residual) kern = gemm_nt256p_kernel<1, 0, false, 8192 | SD, 0>;
            } else if (act == ACT_NONE) {
                if (g.residual) kern = g.out_f32 ? gemm_nt256p_kernel<0, 0, false, 8192 | SD, 3> : gemm_nt256p_kernel<0, 0, false, 8192 | SD, 2>;
                else if (!g.out_f32 && g.N >= 2304) kern = b16_patch ? gemm_nt256p_kernel<0, 0, false, 8192 | SD | 8388608, 0> : gemm_nt256p_kernel<0, 0, false, 8192 | SD, 0>;
            }
        }
    }
    if constexpr (FP8) {
        if (g.q8) {  // the epilogue also writes the per-tensor e4m3 copy of its bf16 result (activation / gate forms of the scaled-MFMA loop)
            constexpr int QF = SD | 65536 | 1048576;
            if (!fp8_mx || g.out_f32 || g.residual || (g.ldq8 % 8)) return TVTS_EINVAL;
            if (gated) kern = gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, true, QF> : gate_act == ACT_GELU_ERF ? gemm_nt256p_kernel<0, 2, true, QF> : nullptr;
            else kern = act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, true, QF> : act == ACT_GELU_ERF ? gemm_nt256p_kernel<2, 0, true, QF> : nullptr;
        }
    }
    if constexpr (!FP8) {
        if (g.sk_ws) {  // stream-K: the generic patch epilogue (bias and side inputs are read by the block that sums the pieces)
            constexpr int SKF = SD | 524288;
            if (gated) kern = gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1, false, SKF> : gate_act == ACT_GELU_ERF ? gemm_nt256p_kernel<0, 2, false, SKF>
                            : gemm_nt256p_kernel<0, 3, false, SKF>;
            else if (act != ACT_NONE) kern = act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, false, SKF> : gemm_nt256p_kernel<2, 0, false, SKF>;
            else kern = g.residual ? gemm_nt256p_kernel<0, 0, false, 256 | SKF> : gemm_nt256p_kernel<0, 0, false, SKF>;
        }
    }
    g.clk = nullptr;
    if constexpr (!FP8) {
        // TVTS_GEMM_CLOCK_SAMPLE: the same kernel with the two counter samples of block 0 compiled in (the production instantiations
        // carry nothing of it); the samples go to the last 32 bytes of the caller's workspace
        if ((opts & 2097152) && workspace && workspace_bytes >= 64 && !g.sk_ws && ((size_t)workspace + (size_t)workspace_bytes) % 8 == 0) {
            void (*ck)(GemmNT) = nullptr;
            if (kern == (void (*)(GemmNT))gemm_nt256p_kernel<0, 0, false, 8192 | SD | 8388608, 0>) ck = gemm_nt256p_kernel<0, 0, false, 8192 | SD | 8388608 | 4194304, 0>;
            else if (kern == (void (*)(GemmNT))gemm_nt256p_kernel<0, 0, false, SD | 8388608>) ck = gemm_nt256p_kernel<0, 0, false, SD | 8388608 | 4194304>;
            if (ck) {
                kern = ck;
                g.clk = (unsigned long long*)((char*)workspace + workspace_bytes - 32);
            }
        }
    }
    if (!kern) return TVTS_EINVAL;
    const int lds_bytes = 163840;  // 2 x 64 KiB stages + 8 x 4 KiB epilogue patches
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, g);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// The ring form of the 128-column kernel (gemm_nt_ring.h: one block per CU, 3 stages in flight, tile rows 128 / 192).  Measured
// per 64-deep stage with HBM-cold operands (tools/gemm_ring.py, profiles/r04_gemm_ring.txt): the double-buffered 128 kernel 1.25 us
// per round of its 512 block slots (one or two blocks per CU, one stage in flight each), the ring 0.58 us (128 rows, loader waves) / 0.88 us
// (192 rows) per round of 256 blocks.  Taken where that model says the ring is at least 10 % faster, i.e. where the 128 kernel would run
// about one block per CU: the text tower up to 24 pairs per GPU (fc2 43 -> 24 us, proj 15 -> 9.5 at 12 pairs), every GEMM of the
// 2-pair step.  TVTS_GEMM_RING forces it, TVTS_GEMM_NO_RING forbids.
static const double NT_RING_STAGE_US[5] = {0., 0., 0.58, 0.88, 1.32};
static int nt_ring_rm(int M, int N, int opts, double* t_out = nullptr) {
    const int forced = (opts >> 16) & 3;
    int best = 2;
    double best_t = 1e30;
    for (int rm = 2; rm <= 3; ++rm) {
        const long tiles = (long)ceil_div(M, 64 * rm) * ceil_div(N, BN);
        const double t = (double)((tiles + 255) / 256) * NT_RING_STAGE_US[rm];
        if (t < best_t - 1e-9) { best_t = t; best = rm; }
    }
    if (t_out) *t_out = best_t;
    return forced ? forced + 1 : best;
}
static bool nt_use_ring(int M, int N, int K, int opts) {
    (void)K;
    if (opts & 16384) return false;
    if (opts & 128) return true;
    if (opt_tile(opts)) return false;  // a forced tile size means that kernel
    double t_ring;
    nt_ring_rm(M, N, opts, &t_ring);
    const long tiles128 = (long)ceil_div(M, BM) * ceil_div(N, BN);
    const double t128 = (double)((tiles128 + 511) / 512) * 1.25;
    return t_ring < 0.9 * t128;
}
// which kernel a bf16 NT call of this shape takes: 256 (pipelined 256 x 256), 128 (double-buffered 128 x 128), or the ring kernel,
// reported as 1000 + its tile rows (1128 / 1192)
extern "C" int tvts_gemm_nt_select(int M, int N, int opts) {
    if (nt_use_256(M, N, 0, false, opts)) return 256;
    return nt_use_ring(M, N, 0, opts) ? 1000 + 64 * nt_ring_rm(M, N, opts) : 128;
}
// 32-bit byte offsets: operand rows from the tile's first row, side inputs from the matrix base; one side input at most
static bool nt_ring_fits(const GemmNT& g) {
    const unsigned long long lim = 1ull << 32, Mu = (unsigned long long)g.M;
    return (unsigned long long)g.lda * 2ull * 128ull < lim && (unsigned long long)g.ldb * 2ull * 128ull < lim &&
           !(g.gate_h && g.residual) && (!g.residual || Mu * (unsigned long long)g.ldr * 4ull < lim) &&
           (!g.gate_h || Mu * (unsigned long long)g.ldh * 2ull < lim) && (unsigned long long)g.N * 4ull < lim;
}
template <int RM, int NS>
static int launch_nt_ringl_rm(GemmNT& g, int act, int gate_act, bool gated, hipStream_t stream) {
    g.tiles_m = ceil_div(g.M, 64 * RM);
    g.tiles_n = ceil_div(g.N, BN);
    const int total_tiles = g.tiles_m * g.tiles_n;
    const int grid = total_tiles < 256 ? ((total_tiles + 7) / 8) * 8 : 256;
    void (*kern)(GemmNT) = nullptr;
    if (gated) {
        if (act != ACT_NONE) return TVTS_EINVAL;
        kern = gate_act == ACT_QUICK_GELU ? gemm_nt_ringl_kernel<0, 1, RM, NS> : gate_act == ACT_GELU_ERF ? gemm_nt_ringl_kernel<0, 2, RM, NS>
             : gate_act == ACT_ADD_BF16 ? gemm_nt_ringl_kernel<0, 3, RM, NS> : nullptr;
    } else {
        kern = act == ACT_NONE ? gemm_nt_ringl_kernel<0, 0, RM, NS> : act == ACT_QUICK_GELU ? gemm_nt_ringl_kernel<1, 0, RM, NS>
             : act == ACT_GELU_ERF ? gemm_nt_ringl_kernel<2, 0, RM, NS> : nullptr;
    }
    if (!kern) return TVTS_EINVAL;
    constexpr int lds_bytes = NS * (64 * RM * 128 + 16384);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds_bytes, stream, g);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
template <int RM, int NS>
static int launch_nt_ring_rm(GemmNT& g, int act, int gate_act, bool gated, hipStream_t stream) {
    g.tiles_m = ceil_div(g.M, 64 * RM);
    g.tiles_n = ceil_div(g.N, BN);
    const int total_tiles = g.tiles_m * g.tiles_n;
    const int grid = total_tiles < 256 ? ((total_tiles + 7) / 8) * 8 : 256;
    void (*kern)(GemmNT) = nullptr;
    if (gated) {
        if (act != ACT_NONE) return TVTS_EINVAL;
        kern = gate_act == ACT_QUICK_GELU ? gemm_nt_ring_kernel<0, 1, RM, NS> : gate_act == ACT_GELU_ERF ? gemm_nt_ring_kernel<0, 2, RM, NS>
             : gate_act == ACT_ADD_BF16 ? gemm_nt_ring_kernel<0, 3, RM, NS> : nullptr;
    } else {
        kern = act == ACT_NONE ? gemm_nt_ring_kernel<0, 0, RM, NS> : act == ACT_QUICK_GELU ? gemm_nt_ring_kernel<1, 0, RM, NS>
             : act == ACT_GELU_ERF ? gemm_nt_ring_kernel<2, 0, RM, NS> : nullptr;
    }
    if (!kern) return TVTS_EINVAL;
    constexpr int lds_bytes = NS * (64 * RM * 128 + 16384);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, g);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
static int launch_nt_ring(GemmNT& g, int act, int gate_act, bool gated, int opts, hipStream_t stream) {
    if (!nt_ring_fits(g)) return TVTS_EINVAL;
    switch (nt_ring_rm(g.M, g.N, opts)) {
        case 2: return launch_nt_ringl_rm<2, 4>(g, act, gate_act, gated, stream);  // 128 rows: 8 multiplying + 4 loader waves
        case 3: return launch_nt_ring_rm<3, 4>(g, act, gate_act, gated, stream);   // 192 rows: 8 waves that load for themselves (the 12-wave form would spill at 168 registers)
        default: return launch_nt_ring_rm<4, 3>(g, act, gate_act, gated, stream);
    }
}

extern "C" int tvts_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                                 const float* bias, const float* residual, int ldr, int act, void* preact,
                                 int ldp, const void* gate_h, int ldh, int gate_act, void* out, int ldc,
                                 int out_f32, void* workspace, long workspace_bytes, int opts, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TVTS_EINVAL;
    if (K % BK != 0 || N % 4 != 0 || lda % 8 != 0 || ldb % 8 != 0) return TVTS_EINVAL;
    if ((ldc % 4) || (residual && (ldr % 4)) || (preact && (ldp % 4)) || (gate_h && (ldh % 4))) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.preact = (bf16*)preact; g.ldp = ldp; g.gate_h = (const bf16*)gate_h; g.ldh = ldh; g.gate_act = gate_act;
    g.side_deriv = (opts >> 20) & 1;
    g.out = out; g.ldc = ldc; g.out_f32 = out_f32; g.sa = nullptr; g.sb = nullptr; g.sa_rows = 0; g.gc = 0;
    g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_tol = 0; g.q8 = nullptr; g.ldq8 = 0; g.q8_scale = nullptr; g.q8_amax = nullptr;
    if (nt_use_256(M, N, K, workspace != nullptr, opts)) {
        if (ldc % 8 || (preact && ldp % 8) || (gate_h && ldh % 8)) {
            if (opt_tile(opts) == 256) return TVTS_EINVAL;  // forced, but the 16-byte epilogue accesses do not fit
        } else {
            return launch_nt256<false>(g, act, gate_act, gate_h != nullptr, opts, stream, workspace, workspace_bytes);
        }
    }
    g.tiles_n = ceil_div(N, BN);
    g.tiles_m = ceil_div(M, BM);
    const int total_tiles = g.tiles_m * g.tiles_n;
    if (nt_use_ring(M, N, K, opts) && (nt_ring_fits(g) || (opts & 128))) return launch_nt_ring(g, act, gate_act, gate_h != nullptr, opts, stream);
    int tiles = total_tiles < 512 ? ((total_tiles + 7) / 8) * 8 : 512;  // persistent grid: 2 blocks x 256 CUs, multiple of 8
    void (*kern)(GemmNT) = nullptr;
    if (gate_h) {
        if (act != ACT_NONE) return TVTS_EINVAL;
        kern = gate_act == ACT_QUICK_GELU ? gemm_nt_kernel<0, 1> : gate_act == ACT_GELU_ERF ? gemm_nt_kernel<0, 2>
             : gate_act == ACT_ADD_BF16 ? gemm_nt_kernel<0, 3> : nullptr;
    } else {
        kern = act == ACT_NONE ? gemm_nt_kernel<0, 0> : act == ACT_QUICK_GELU ? gemm_nt_kernel<1, 0>
             : act == ACT_GELU_ERF ? gemm_nt_kernel<2, 0> : nullptr;
    }
    if (!kern) return TVTS_EINVAL;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), 65536, stream, g);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// fp8 (OCP e4m3) operands, fp32 accumulate: out[M,N] = scale_a * scale_b * (A[M,K] B[N,K]^T) + bias [+ residual], A / B row-major
// bytes, scales in device memory: one per tensor, or (scale_a_rows != 0) one per ROW of A -- the per-token activation scales
// of tvts_quant_fp8_rows (BASELINE config 4's weight/activation path).  Same pipelined 256x256 kernel: K % 128 == 0, lda / ldb % 16 == 0.
extern "C" int tvts_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* scale_a,
                                int scale_a_rows, const float* scale_b, const float* bias, const float* residual, int ldr, int act, void* preact,
                                int ldp, void* out, int ldc, int out_f32, void* q8out, int ldq8, const float* q8_scale, float* q8_amax,
                                int opts, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !scale_a || !scale_b || (q8out && !q8_scale) || (!out && !q8out)) return TVTS_EINVAL;
    if (K % 128 || N % 8 || lda % 16 || ldb % 16 || ldc % 8 || (residual && ldr % 4) || (preact && ldp % 8)) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda / 2; g.B = (const bf16*)B; g.ldb = ldb / 2;  // byte-identical bf16 view, half as wide
    g.M = M; g.N = N; g.K = K / 2; g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.preact = (bf16*)preact; g.ldp = ldp; g.gate_h = nullptr; g.ldh = 0; g.gate_act = ACT_NONE;
    g.side_deriv = (opts >> 20) & 1;
    g.out = out; g.ldc = ldc; g.out_f32 = out_f32; g.gc = 0; g.sa = scale_a; g.sb = scale_b; g.sa_rows = scale_a_rows ? 1 : 0;
    g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_tol = 0;
    g.q8 = (unsigned char*)q8out; g.ldq8 = ldq8; g.q8_scale = q8_scale; g.q8_amax = q8_amax;
    return launch_nt256<true>(g, act, ACT_NONE, false, opts, stream);
}

// the input-gradient form of the fp8 linear layer: out[M,N] (bf16) = gate'(h[M,N]) * (scale_a[m] * scale_b * (A[M,K] B[N,K]^T)) with
// A the e4m3 copy of the OUTPUT gradient (one scale per token, tvts_quant_fp8_rows), B the e4m3 copy of the TRANSPOSED weight
// and, optionally, the activation-gradient gate of the MLP's first layer (gate_h = its saved pre-activation)
extern "C" int tvts_gemm_nt_fp8_gate(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* scale_a,
                                     int scale_a_rows, const float* scale_b, const float* bias, const void* gate_h, int ldh,
                                     int gate_act, void* out, int ldc, void* q8out, int ldq8, const float* q8_scale, float* q8_amax,
                                     int opts, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !scale_a || !scale_b || !gate_h || (q8out && !q8_scale) || (!out && !q8out)) return TVTS_EINVAL;
    if (K % 128 || N % 8 || lda % 16 || ldb % 16 || ldc % 8 || ldh % 8) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda / 2; g.B = (const bf16*)B; g.ldb = ldb / 2;
    g.M = M; g.N = N; g.K = K / 2; g.bias = bias; g.residual = nullptr; g.ldr = 0; g.act = ACT_NONE;
    g.preact = nullptr; g.ldp = 0; g.gate_h = (const bf16*)gate_h; g.ldh = ldh; g.gate_act = gate_act;
    g.side_deriv = (opts >> 20) & 1;
    g.out = out; g.ldc = ldc; g.out_f32 = 0; g.gc = 0; g.sa = scale_a; g.sb = scale_b; g.sa_rows = scale_a_rows ? 1 : 0;
    g.sk_ws = nullptr; g.sk_cnt = nullptr; g.sk_tol = 0;
    g.q8 = (unsigned char*)q8out; g.ldq8 = ldq8; g.q8_scale = q8_scale; g.q8_amax = q8_amax;
    return launch_nt256<true>(g, ACT_NONE, gate_act, true, opts, stream);
}

// ------------------------------------------------------------------------------------------------
// TN: C[Na,Nb] (+)= sum_m P[m,Na] * Q[m,Nb].  Both operands are "contraction-strided", so the MFMA
// fragments come from LDS through the transposing read ds_read_b64_tr_b16 (4 rows x 16 columns per
// 16-lane group; lane (l&15) receives column l&15 of those 4 rows).  The MFMA k-slot <-> m mapping is
// the same permutation for both operands: slot (l>>4)*8+j <-> m = 32u + 16*(j>>2) + 4*(l>>4) + (j&3).
// LDS tile: [64 m][128 cols] bf16 = 16 KiB, rows of 256 B, 32-B chunk c of row r stored at c^(r&7).
// ------------------------------------------------------------------------------------------------
struct GemmTN {
    const bf16* P; int ldp;
    const bf16* Q; int ldq;
    int M, Na, Nb;
    float* out; int ldo;
    int tiles_b, tiles_ab, m_per_split, n_items;
    int early_dma;  // 128-tile kernel: issue the next stage's LDS-DMA (inline asm) before the current stage's fragment reads
    int tiles_a, a_fast;  // 128-tile kernel: walk the tiles of an m-range with the SHORTER tile dimension fastest
    int atomic;
    float* ws;      // split partials [splits][Na][Nb] (plain stores, reduced by tn_reduce_kernel) or nullptr -> fp32 atomics
    float* colsum;  // optional: colsum[a] += sum_m P[m,a]  (bias gradient fused into the weight gradient)
    float* cs_ws;   // split partials of the column sums [splits][Na] (plain stores, added in split order by tn_reduce_kernel);
                    // nullptr with one m-range: the single owner of a column adds its sum itself.  (fp32 atomics only remain
                    // for several m-ranges WITHOUT a workspace -- the engine always passes one: bias gradients are
                    // run-to-run reproducible.)
    int splits;
    int* cnt;        // fused reduce (round 4): one arrival counter per output tile, zero between launches; nullptr -> tn_reduce_kernel
    int accumulate;  // fused reduce: out += sum of the partials (else out = ...)
};
// column-sum partial of (m-range split, column a): exactly one wave of one block owns it
__device__ __forceinline__ void tn_colsum_out(const GemmTN& g, int split, int a, float v) {
    if (g.cs_ws) {
        if (g.cnt) __hip_atomic_store(g.cs_ws + (size_t)split * g.Na + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: read by the last block
        else g.cs_ws[(size_t)split * g.Na + a] = v;
    }
    else if (g.splits == 1) g.colsum[a] += v;
    else atomicAdd(g.colsum + a, v);
}

// ------------------------------------------------------------------------------------------------
// Fused reduce of the split partials (round 4).  The m-range partials of a tile used to meet in a second launch
// (tn_reduce_kernel: 88 launches of 8-19 us per step at the reference's 12 pairs per GPU, a twentieth of that step).  With a
// counter array from the caller every block stores its partial with agent-scope (sc1) stores and counts itself in at its
// tile; the block that arrives LAST adds the partials of all ranges IN RANGE ORDER on top of the output -- (out + p0) + p1 + ...,
// the very expression of tn_reduce_kernel, so the two paths give the same bits whoever arrives last -- and the column sums of
// the bias gradient likewise.  Nobody waits for anybody.  Taken while a tile's partials are small enough for one block to sum
// (splits x tile bytes <= 1.5 MiB); many ranges of a big tile keep the parallel reduce pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_sc1(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int NI, int NJ>
__device__ __forceinline__ void tn_fused_reduce(const GemmTN& g, f32x4 (&acc)[NI][NJ], int t, int split, int a_base, int b_base,
                                                int lane, int* flag) {
    const int ai = lane & 15, bq = (lane >> 4) * 4;
    float* mine = g.ws + (size_t)split * g.Na * g.Nb;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int a = a_base + i * 16 + ai;
        if (a >= g.Na) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int b = b_base + j * 16 + bq;
            if (b < g.Nb) st_sc1(mine + (size_t)a * g.Nb + b, acc[i][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial (and the column-sum partial before it) has left: that is the release
    __syncthreads();
    if (threadIdx.x == 0) *(volatile int*)flag = __hip_atomic_fetch_add(g.cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*(volatile int*)flag != g.splits - 1) return;
    if (threadIdx.x == 0) g.cnt[t] = 0;  // every range has arrived: zero again for the next launch
    const size_t plane = (size_t)g.Na * g.Nb;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int a = a_base + i * 16 + ai;
        if (a >= g.Na) continue;  // (a whole row-tile of lanes sits out: no wait below depends on it)
        f32x4 s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int b = b_base + j * 16 + bq;
            s[j] = (g.accumulate && b < g.Nb) ? *(const f32x4*)(g.out + (size_t)a * g.ldo + b) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int k = 0; k < g.splits; ++k) {
            const float* p = g.ws + (size_t)k * plane + (size_t)a * g.Nb;
            f32x4 v[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int b = b_base + j * 16 + bq;
                v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (b < g.Nb) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(p + b) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                asm volatile("" : "+v"(v[j]));  // the value is only defined behind the wait
                s[j] += v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int b = b_base + j * 16 + bq;
            if (b < g.Nb) *(f32x4*)(g.out + (size_t)a * g.ldo + b) = s[j];
        }
    }
}
// the bias gradient's share of the same: column a's sums of all ranges on top of colsum[a], in range order (called by the lanes
// that wrote the partial of column a, in the block that arrived last)
__device__ __forceinline__ void tn_fused_colsum(const GemmTN& g, int a) {
    float s = g.colsum[a];
    for (int k = 0; k < g.splits; ++k) s += __hip_atomic_load(g.cs_ws + (size_t)k * g.Na + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g.colsum[a] = s;
}

__device__ __forceinline__ void stage_cols128(const bf16* __restrict__ base, int ld, int m0, int m_max, int c0,
                                              int c_max, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r0 = (t * 4 + wave) * 4;  // 4 rows of 256 B per wave-issue
        const int row = r0 + (lane >> 4);
        const int s16 = lane & 15;
        const int chunk32 = (s16 >> 1) ^ (row & 7);
        int gm = m0 + row;
        gm = gm < m_max ? gm : m_max;
        int col = c0 + chunk32 * 16 + (s16 & 1) * 8;
        col = col < c_max ? col : c_max;
        const bf16* src = base + (size_t)gm * ld + col;
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))src, (LDS_PTR(void))(lds_tile + r0 * 256), 16, 0, 0);
    }
}

__device__ __forceinline__ void glds16_asm(unsigned voff, const char* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// the same 64-row stage issued from inline asm: hipcc parks an s_waitcnt vmcnt(0) in front of every transposing LDS read
// that follows the LDS-DMA builtin (it cannot tell the buffers apart), which forces "reads first, then the next stage's
// DMA"; issued this way the DMA can start BEFORE the reads of the current stage and flies under them as well.  Offsets are
// 32-bit from a wave-uniform base (the dispatcher checks the operand fits 4 GiB).
__device__ __forceinline__ void stage_cols128_asm(const bf16* __restrict__ base, int ld, int m0, int m_max, int c0, int c_max,
                                                  unsigned lds_tile, int wave, int lane) {
    const char* ub = uniform_ptr(base);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r0 = (t * 4 + wave) * 4;
        const int row = r0 + (lane >> 4);
        const int s16 = lane & 15;
        const int chunk32 = (s16 >> 1) ^ (row & 7);
        int gm = m0 + row;
        gm = gm < m_max ? gm : m_max;
        int col = c0 + chunk32 * 16 + (s16 & 1) * 8;
        col = col < c_max ? col : c_max;
        glds16_asm(((unsigned)gm * (unsigned)ld + (unsigned)col) * 2u, ub, lds_tile + (unsigned)r0 * 256u);
    }
}

// 8 k-slots (one MFMA k-step u) of column block ct (16 columns) for this lane
__device__ __forceinline__ bf16x8 frag_tr(const char* lds_tile, int u, int ct, int lane) {
    const int g = lane >> 4, i = lane & 15;
    s16x4 h[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = u * 32 + half * 16 + g * 4 + (i >> 2);
        const int chunk32 = ct ^ (row & 7);
        const char* p = lds_tile + row * 256 + chunk32 * 32 + (i & 3) * 8;
        h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// the body of the 128x128 weight-gradient kernel for work item `item` of problem g (shared by the single-problem kernel and the
// grouped one below)
__device__ __forceinline__ void tn128_item(const GemmTN& g, int item, char* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    if (item >= g.n_items) return;
    const int split = item / g.tiles_ab;
    const int t = item % g.tiles_ab;
    // the co-resident tiles of an XCD (64) then form a patch as square as the output allows: the panels of the wide operand
    // are shared by blocks running side by side, those of the narrow one are the ones re-read round after round
    const int ta = g.a_fast ? t % g.tiles_a : t / g.tiles_b, tb = g.a_fast ? t / g.tiles_a : t % g.tiles_b;
    const int a0 = ta * 128, b0 = tb * 128;
    const int m_begin = split * g.m_per_split;
    int m_end = m_begin + g.m_per_split;
    m_end = m_end < g.M ? m_end : g.M;
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin + 63) / 64;
    const bool do_cs = g.colsum != nullptr && tb == 0 && wb == 0;  // wave-uniform
    f32x4 cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // rows >= M are read clamped: the caller guarantees rows [M, round_up(M,64)) of P are ZERO or that
    // m_end is a multiple of 64 (see tvts_gemm_tn_bf16); we additionally zero the contribution by
    // clamping to row M-1 only when the caller says pad rows are valid (pad_ok), so here: plain clamp.
    stage_cols128(g.P, g.ldp, m_begin, g.M - 1, a0, g.Na - 8, smem, wave, lane);
    stage_cols128(g.Q, g.ldq, m_begin, g.M - 1, b0, g.Nb - 8, smem + 16384, wave, lane);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem;
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 32768;
        if (g.early_dma && kt + 1 < nk) {  // next stage's DMA first: it is in flight under the reads below as well
            const unsigned nxt = lds0 + (unsigned)((kt + 1) & 1) * 32768u;
            stage_cols128_asm(g.P, g.ldp, m_begin + (kt + 1) * 64, g.M - 1, a0, g.Na - 8, nxt, wave, lane);
            stage_cols128_asm(g.Q, g.ldq, m_begin + (kt + 1) * 64, g.M - 1, b0, g.Nb - 8, nxt + 16384u, wave, lane);
        }
        // (builtin path) all transposing reads of this stage first: hipcc drains vmcnt(0) in front of a ds_read_tr that
        // follows an LDS-DMA builtin, which would serialise the next stage's loads behind this stage's MFMAs.
        bf16x8 pf[2][4], qf[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[u][i] = frag_tr(cur, u, wa * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[u][j] = frag_tr(cur + 16384, u, wb * 4 + j, lane);
        }
        if (!g.early_dma && kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * 32768;
            stage_cols128(g.P, g.ldp, m_begin + (kt + 1) * 64, g.M - 1, a0, g.Na - 8, nxt, wave, lane);
            stage_cols128(g.Q, g.ldq, m_begin + (kt + 1) * 64, g.M - 1, b0, g.Nb - 8, nxt + 16384, wave, lane);
        }
        const int valid = m_end - (m_begin + kt * 64);  // tail rows beyond m_end must not contribute
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (valid < 64) {  // zero k-slots whose m is past the end (uniform branch, tail stage only)
                const int gq = lane >> 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = u * 32 + (e >> 2) * 16 + gq * 4 + (e & 3);
                    if (m >= valid) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) pf[u][i][e] = (bf16)0.f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[u][j], pf[u][i], acc[i][j], 0, 0, 0);
            if (do_cs) {  // every row of ones . P is the column sum of this stage
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[u][i], cs[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the asm-issued DMA is invisible to the compiler's own counting
        __syncthreads();
    }
    if (do_cs && lane < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = a0 + wa * 64 + i * 16 + lane;
            if (a < g.Na) tn_colsum_out(g, split, a, cs[i][0]);
        }
    }
    if (g.cnt) {  // fused reduce: partial + arrival; the last block of the tile sums the ranges in order
        tn_fused_reduce<4, 4>(g, acc, t, split, a0 + wa * 64, b0 + wb * 64, lane, (int*)smem);
        if (*(volatile int*)smem == g.splits - 1 && do_cs && g.cs_ws && lane < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int a = a0 + wa * 64 + i * 16 + lane;
                if (a < g.Na) tn_fused_colsum(g, a);
            }
        }
        return;
    }
    // acc[i][j]: MFMA A-operand = Q (rows = b within tile j), B-operand = P (cols = a within tile i)
    // lane: col = a = l&15, rows = b = (l>>4)*4 + r  -> 4 consecutive b for one a: 16-B fp32 access
    float* obase = g.ws ? g.ws + (size_t)split * g.Na * g.Nb : g.out;
    const int old_ = g.ws ? g.Nb : g.ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + wa * 64 + i * 16 + (lane & 15);
        if (a >= g.Na) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + wb * 64 + j * 16 + (lane >> 4) * 4;
            if (b >= g.Nb) continue;
            float* dst = obase + (size_t)a * old_ + b;
            if (g.atomic && !g.ws) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[i][j][e]);
            } else {
                *(f32x4*)dst = acc[i][j];
            }
        }
    }
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(GemmTN g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][P 16K | Q 16K]
    // Work items are (m-range, output tile) pairs in range-major order; XCD x (= blockIdx % 8) takes the x-th
    // contiguous eighth of that list, so the P/Q rows of an m-range are pulled into ONE XCD's L2 (two at a
    // boundary) and shared there by all output tiles, instead of being fetched by all eight L2s.
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
    tn128_item(g, xcd * per + jx, smem);
}

// ------------------------------------------------------------------------------------------------
// Grouped weight gradients (round 4, the reference's per-GPU batches): the six weight gradients of a ViT block in ONE launch
// and their split partials in ONE reduce launch.  At M = 9 420 a weight gradient is 100-430 work items for 512 block slots
// and ends after 0.2-0.85 of a round, twelve launches of 9-50 us per block of the model; grouped, the ~2 000 items of the six
// problems share the slots round after round (the XCD-contiguous walk runs over the concatenated item list) and ten launch
// ramps per block disappear.  The plan (one GemmTN per problem + the item prefix) lives in device memory the caller owns;
// every problem keeps its own contraction ranges, workspace slice and ordered reduce, so the results are the bits of the
// one-by-one launches.
// ------------------------------------------------------------------------------------------------
struct TnGroup { const GemmTN* tab; const int* prefix; int n; };
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_grouped_kernel(TnGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int item = xcd * per + jx;
    if (item >= grp.prefix[grp.n]) return;
    int p = 0;
    while (p + 1 < grp.n && item >= grp.prefix[p + 1]) ++p;
    const GemmTN g = grp.tab[p];  // block-uniform
    tn128_item(g, item - grp.prefix[p], smem);
}
// The element pass of the two reduce kernels: out[a, b] = (accumulate ? out[a, b] : 0) + range 0 + range 1 + ... (that order: the fused
// in-kernel reduce evaluates the same expression).  Partials of up to FOUR ranges are requested before the first is added (round 5:
// the plain `for k: s += ws[k]` loop was a load -> vmcnt(0) -> add chain per range, and the row / column of an element cost a 64-bit
// division, ~100 instructions).
__device__ __forceinline__ void tn_reduce_elements(const float* __restrict__ ws, int splits, int Na, int Nb, float* __restrict__ out,
                                                   int ldo, int accumulate) {
    const size_t plane = (size_t)Na * Nb;
    const size_t n4 = plane / 4;
    const bool dense = ldo == Nb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 4;
        size_t o = e;
        if (!dense) {  // (Na * Nb < 2^32: the workspace holds `splits` planes of it)
            const unsigned a = (unsigned)e / (unsigned)Nb;
            o = (size_t)a * ldo + ((unsigned)e - a * (unsigned)Nb);
        }
        const float* p = ws + e;
        // (the output is read also when it is overwritten -- it is valid memory, the value is dropped: no conditional load)
        const f32x4 prev = *(const f32x4*)(out + o);
        f32x4 s = accumulate ? prev : (f32x4){0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            const f32x4 v0 = *(const f32x4*)(p + (size_t)k * plane), v1 = *(const f32x4*)(p + (size_t)(k + 1) * plane);
            const f32x4 v2 = *(const f32x4*)(p + (size_t)(k + 2) * plane), v3 = *(const f32x4*)(p + (size_t)(k + 3) * plane);
            s += v0; s += v1; s += v2; s += v3;
        }
        if (k + 2 <= splits) {
            const f32x4 v0 = *(const f32x4*)(p + (size_t)k * plane), v1 = *(const f32x4*)(p + (size_t)(k + 1) * plane);
            s += v0; s += v1;
            k += 2;
        }
        if (k < splits) s += *(const f32x4*)(p + (size_t)k * plane);
        *(f32x4*)(out + o) = s;
    }
}
__global__ __launch_bounds__(256) void tn_reduce_grouped_kernel(TnGroup grp) {
    const GemmTN g = grp.tab[blockIdx.y];
    if (!g.ws) return;  // one range: the kernel wrote the output itself
    if (g.cs_ws) {
        for (int a = blockIdx.x * 256 + threadIdx.x; a < g.Na; a += gridDim.x * 256) {
            float s = g.colsum[a];
            for (int k = 0; k < g.splits; ++k) s += g.cs_ws[(size_t)k * g.Na + a];
            g.colsum[a] = s;
        }
    }
    tn_reduce_elements(g.ws, g.splits, g.Na, g.Nb, g.out, g.ldo, g.accumulate);
}

// out[a,b] = (accumulate ? out[a,b] : 0) + sum_s ws[s][a][b];  colsum[a] += sum_s cs_ws[s][a]  (both in split order)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int splits, int Na, int Nb,
                                                        float* __restrict__ out, int ldo, int accumulate,
                                                        const float* __restrict__ cs_ws, float* __restrict__ colsum) {
    if (cs_ws) {
        for (int a = blockIdx.x * 256 + threadIdx.x; a < Na; a += gridDim.x * 256) {
            float s = colsum[a];
            for (int k = 0; k < splits; ++k) s += cs_ws[(size_t)k * Na + a];
            colsum[a] = s;
        }
    }
    tn_reduce_elements(ws, splits, Na, Nb, out, ldo, accumulate);
}

#include "gemm_tn256.h"

// per-call options of the weight-gradient entry point (TVTS_GEMM_TILE_*, TVTS_TN_*, TVTS_TN_SPLITS in include/tvts_hip.h): test /
// bench hooks, 0 = automatic choice; no process state, no environment reads on the launch path
// Tile and range count: a small cost model of the launch (microseconds), fitted to tools/tn_ab.py (M = 150 720) and
// tools/tn_splits_small.py (the reference's per-GPU batches, M = 9 420 / 18 840; profiles/r03_tn_plan_small_m.txt):
//   rounds x rows per range x time per contraction row of one tile  (256: 26.5 ns, one block per CU; 128: 14.8 - 16.5 ns with two
//   blocks on a CU, 11 ns when the launch leaves every block a CU of its own)  +  8 bytes per output element and range for the partials
//   (written, read back by the reduce pass: ~8 TB/s, they mostly stay in the caches)  +  launch / prologue / reduce constants.
// The 256x256 kernel needs at most 15 % of its tiling's area wasted.  Rounds 1-2 picked "256 from M = 32 768, ranges to fill the
// round": right at 192 pairs, 15-25 % off at the reference's 24 pairs (128-tile kernel where the 256 one is faster) and at 12
// (14 ranges of partials where 4 are faster).
// (one range without the reduce pass adds into the output with fp32 atomics when it accumulates: ~12 us per million elements)
static double tn_model_us(int M, int Na, int Nb, int tile, int sp) {
    const long tiles = (long)ceil_div(Na, tile) * ceil_div(Nb, tile), items = tiles * sp;
    const long slots = tile == 256 ? 256 : 512;
    const long rounds = (items + slots - 1) / slots;
    const double rows = ceil_div(ceil_div(M, sp), 64) * 64.0;
    const double narrow = (Na <= 512 || Nb <= 512) ? 1.2 : 1.0;  // the 128-tile kernel on the text tower's / sort head's 512-wide operands
    const double per_row = tile == 256 ? 0.0265 : (items <= 256 ? 0.011 : (rounds == 1 ? 0.0148 : 0.0165) * narrow);
    const double partial = sp > 1 ? sp * (double)Na * (double)Nb * 1e-6 : 0.0;
    return rounds * rows * per_row + partial + (tile == 256 ? 8.0 : 5.0) + (sp > 1 ? 6.0 : 12e-6 * (double)Na * (double)Nb);
}
static int tn_best_splits(int M, int Na, int Nb, int tile, double* t_out) {
    int best = 1;
    double tb = 1e30;
    for (int sp = 1; sp <= 64; ++sp) {
        if (sp > 1 && M / sp < 768) break;
        const double t = tn_model_us(M, Na, Nb, tile, sp);
        if (t < tb) { tb = t; best = sp; }
    }
    if (t_out) *t_out = tb;
    return best;
}
static bool tn_use_256(int M, int Na, int Nb, int opts) {
    if (opt_tile(opts)) return opt_tile(opts) == 256;
    const double area = 65536.0 * ceil_div(Na, 256) * ceil_div(Nb, 256), elems = (double)Na * (double)Nb;
    if (area > 1.15 * elems) return false;
    double t128, t256;
    tn_best_splits(M, Na, Nb, 128, &t128);
    tn_best_splits(M, Na, Nb, 256, &t256);
    return t256 < t128;
}
extern "C" int tvts_gemm_tn_select(int M, int Na, int Nb, int opts) { return tn_use_256(M, Na, Nb, opts) ? 256 : 128; }

extern "C" int tvts_gemm_tn_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int Na, int Nb,
                                 float* out, int ldo, int accumulate, float* colsum, float* workspace,
                                 long workspace_elems, int* counters, int n_counters, int opts, hipStream_t stream) {
    if (M <= 0 || Na <= 0 || Nb <= 0) return TVTS_EINVAL;
    if (Na % 8 || Nb % 8 || ldp % 8 || ldq % 8 || ldo % 4) return TVTS_EINVAL;
    GemmTN g;
    g.ws = nullptr; g.cs_ws = nullptr; g.splits = 1; g.early_dma = 0; g.tiles_a = 0; g.a_fast = 0; g.cnt = nullptr; g.accumulate = accumulate;
    g.P = (const bf16*)P; g.ldp = ldp; g.Q = (const bf16*)Q; g.ldq = ldq; g.M = M; g.Na = Na; g.Nb = Nb;
    g.out = out; g.ldo = ldo; g.colsum = colsum;
    const bool t256 = tn_use_256(M, Na, Nb, opts);
    const int tile = t256 ? 256 : 128;
    const int tiles_a = ceil_div(Na, tile);
    g.tiles_b = ceil_div(Nb, tile);
    g.tiles_ab = tiles_a * g.tiles_b;
    g.tiles_a = tiles_a;
    g.a_fast = (opts & 8) ? 0 : (opts & 16) ? 1 : (tiles_a < g.tiles_b ? 1 : 0);  // walk the tiles of an m-range with the SHORTER tile dimension fastest
    // the early LDS-DMA issue addresses its operand with 32-bit offsets from a wave-uniform base: the operand must fit 4 GiB
    g.early_dma = ((unsigned long long)M * (unsigned long long)(ldp > ldq ? ldp : ldq) * 2ull < (1ull << 32)) ? 1 : 0;
    if (opts & 4) g.early_dma = 0;
    // split the contraction over M into S ranges (range s lives on XCD s % 8, see the kernel): the modeled optimum
    int splits = tn_best_splits(M, Na, Nb, tile, nullptr);
    if ((opts >> 8) > 0) splits = opts >> 8;
    // the partials must fit the caller's workspace: fewer, longer ranges beat the atomic fallback
    if (workspace != nullptr && splits > 1 && (long)splits * Na * Nb > workspace_elems) {
        const int fit = (int)(workspace_elems / ((long)Na * Nb));
        if (fit >= 2) splits = fit;
    }
    g.m_per_split = ceil_div(ceil_div(M, splits), 64) * 64;
    splits = ceil_div(M, g.m_per_split);
    // split partials: plain stores into the caller's workspace + one reduce pass (an fp32 atomic epilogue costs
    // ~190 us per launch whatever M is: 16 M scattered L2 atomics); atomics remain the fallback without workspace
    const bool use_ws = workspace != nullptr && splits > 1 && (long)splits * Na * Nb <= workspace_elems && Nb % 4 == 0;
    if (use_ws) g.ws = workspace;
    if (use_ws && colsum != nullptr && (long)splits * Na * Nb + (long)splits * Na <= workspace_elems)
        g.cs_ws = workspace + (size_t)splits * Na * Nb;
    g.splits = splits;
    // fused reduce: the last block of a tile sums its ranges -- only when the call asks for it (TVTS_GEMM_STREAMK).  Measured
    // (profiles/r04_tn_fused_reduce.txt): the ranges of a tile live on different XCDs, so the partials must travel at agent scope
    // (write-through stores, loads past the L2), and ONE block then reads splits x 64 KiB in dependent round trips where the
    // reduce pass spreads the same bytes over 2 048 blocks: 613 / 876 / 1 287 pairs/s at 12 / 24 / 192 pairs with it against
    // 693 / 958 / 1 332 with the separate pass.  Same bits either way (tests/test_kernels_gpu.py).
    const bool fused = use_ws && counters != nullptr && g.tiles_ab <= n_counters && (opts & 32) && !(opts & 64) &&
                       (colsum == nullptr || g.cs_ws != nullptr);
    if ((opts & 32) && splits > 1 && !fused) return TVTS_EINVAL;
    if (fused) g.cnt = counters;
    if (!use_ws && !accumulate && splits > 1) {
        hipError_t e = hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)Nb * 4, Na, stream);
        if (e != hipSuccess) return (int)e;
    }
    g.atomic = (accumulate || splits > 1) ? 1 : 0;
    g.n_items = g.tiles_ab * splits;
    const int grid = ceil_div(g.n_items, 8) * 8;
    if (t256) {
        hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e2 != hipSuccess) return (int)e2;
        hipLaunchKernelGGL(gemm_tn256_kernel, dim3(grid), dim3(512), 131072, stream, g);
    } else {
        hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e2 != hipSuccess) return (int)e2;
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(NTHREADS), 65536, stream, g);
    }
    if (use_ws && !fused) {
        const long n4 = (long)Na * Nb / 4;
        int rb = (int)((n4 + 255) / 256);
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(rb), dim3(256), 0, stream, workspace, splits, Na, Nb, out, ldo, accumulate,
                           g.cs_ws, colsum);
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---- grouped weight gradients (see gemm_tn_grouped_kernel).  problems: HOST array of n tvts_tn_problem; table_dev: device memory
// for the plan, n * sizeof(GemmTN) + (n + 1) * 4 bytes (tvts_gemm_tn_grouped_table_bytes).  upload != 0: the plan is written to
// table_dev (a synchronous copy: not inside a stream capture); upload == 0: table_dev already holds the plan of these very
// problems (same pointers, shapes, workspace) from an earlier call, only the launches are enqueued -- the form a captured step uses.
struct tvts_tn_problem_ { const void* P; int ldp; const void* Q; int ldq; int M, Na, Nb; float* out; int ldo; int accumulate; float* colsum; };
extern "C" long tvts_gemm_tn_grouped_table_bytes(int n) { return (long)n * (long)sizeof(GemmTN) + (long)(n + 1) * 4 + 64; }
extern "C" int tvts_gemm_tn_bf16_grouped(const void* problems, int n, void* table_dev, void* table_host, long table_bytes, int upload,
                                         float* workspace, long workspace_elems, int opts, hipStream_t stream) {
    if (n <= 0 || n > 64 || !problems || !table_dev || !workspace || table_bytes < tvts_gemm_tn_grouped_table_bytes(n)) return TVTS_EINVAL;
    if (upload && !table_host) return TVTS_EINVAL;
    const tvts_tn_problem_* pr = (const tvts_tn_problem_*)problems;
    // the plan is built in the CALLER's staging memory when it is uploaded (it has to outlive this call: the copy is asynchronous),
    // on the stack when only the launch geometry is needed
    GemmTN tab_local[64];
    int prefix_local[65];
    GemmTN* tab = upload ? (GemmTN*)table_host : tab_local;
    int* prefix = upload ? (int*)((char*)table_host + (size_t)n * sizeof(GemmTN)) : prefix_local;
    long tiles_total = 0, out_total = 0;
    int Mmax = 0;
    for (int i = 0; i < n; ++i) {
        if (pr[i].M <= 0 || pr[i].Na % 8 || pr[i].Nb % 8 || pr[i].ldp % 8 || pr[i].ldq % 8 || pr[i].ldo % 4 || pr[i].Nb % 4) return TVTS_EINVAL;
        tiles_total += (long)ceil_div(pr[i].Na, 128) * ceil_div(pr[i].Nb, 128);
        out_total += (long)pr[i].Na * pr[i].Nb;
        Mmax = pr[i].M > Mmax ? pr[i].M : Mmax;
    }
    // ONE range count for the group: rounds of the 512 block slots over ALL items x rows per range (0.0148 us per contraction row of a
    // 128-tile item, the single-problem model's figure) + the partials' bytes; the slots are filled by the other problems' items,
    // so a group takes fewer, longer ranges than its members would alone
    int sp_best = 1;
    if ((opts >> 8) > 0) {
        sp_best = opts >> 8;
    } else {
        double best = 1e30;
        for (int sp = 1; sp <= 32; ++sp) {
            if (sp > 1 && Mmax / sp < 512) break;
            if (sp > 1 && (double)sp * ((double)out_total + 4096.0 * n) > (double)workspace_elems) break;
            const double rounds = (double)((tiles_total * sp + 511) / 512);
            const double t = rounds * (ceil_div(ceil_div(Mmax, sp), 64) * 64.0) * 0.0148 + (sp > 1 ? sp * (double)out_total * 1e-6 : 0.0);
            if (t < best) { best = t; sp_best = sp; }
        }
    }
    float* ws = workspace;
    long used = 0;
    prefix[0] = 0;
    for (int i = 0; i < n; ++i) {
        GemmTN& g = tab[i];
        const tvts_tn_problem_& q = pr[i];
        g.P = (const bf16*)q.P; g.ldp = q.ldp; g.Q = (const bf16*)q.Q; g.ldq = q.ldq; g.M = q.M; g.Na = q.Na; g.Nb = q.Nb;
        g.out = q.out; g.ldo = q.ldo; g.colsum = q.colsum; g.accumulate = q.accumulate; g.cnt = nullptr;
        g.tiles_a = ceil_div(q.Na, 128); g.tiles_b = ceil_div(q.Nb, 128); g.tiles_ab = g.tiles_a * g.tiles_b;
        g.a_fast = g.tiles_a < g.tiles_b ? 1 : 0;
        g.early_dma = ((unsigned long long)q.M * (unsigned long long)(q.ldp > q.ldq ? q.ldp : q.ldq) * 2ull < (1ull << 32)) ? 1 : 0;
        int splits = sp_best;
        while (splits > 1 && q.M / splits < 256) --splits;
        g.m_per_split = ceil_div(ceil_div(q.M, splits), 64) * 64;
        splits = ceil_div(q.M, g.m_per_split);
        g.splits = splits;
        g.ws = nullptr; g.cs_ws = nullptr;
        if (splits > 1) {
            const long need = (long)splits * q.Na * q.Nb + (q.colsum ? (long)splits * q.Na : 0);
            if (used + need > workspace_elems) return TVTS_EINVAL;
            g.ws = ws + used;
            if (q.colsum) g.cs_ws = ws + used + (long)splits * q.Na * q.Nb;
            used += (need + 63) / 64 * 64;
        }
        g.atomic = (q.accumulate || splits > 1) ? 1 : 0;
        g.n_items = g.tiles_ab * splits;
        prefix[i + 1] = prefix[i] + g.n_items;
        if (splits == 1 && !q.accumulate) { /* the single owner of every element stores it */ }
    }
    char* td = (char*)table_dev;
    if (upload) {
        // ONE copy, ordered on the launch stream in front of the kernels that read the plan: no host-side wait, no legacy-stream
        // ordering assumption (round 4 used two synchronous hipMemcpy calls, ordered only against torch's default stream)
        hipError_t e = hipMemcpyAsync(td, table_host, (size_t)n * sizeof(GemmTN) + (size_t)(n + 1) * 4, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return (int)e;
    }
    TnGroup grp;
    grp.tab = (const GemmTN*)td; grp.prefix = (const int*)(td + (size_t)n * sizeof(GemmTN)); grp.n = n;
    const int grid = ceil_div(prefix[n], 8) * 8;
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e2 != hipSuccess) return (int)e2;
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(grid), dim3(NTHREADS), 65536, stream, grp);
    bool any_ws = false;
    for (int i = 0; i < n; ++i) any_ws = any_ws || tab[i].ws != nullptr;
    if (any_ws) hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3(384, n), dim3(256), 0, stream, grp);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

#include "gemm_tn8.h"

// e4m3 weight gradient (include/tvts_hip.h): the 256x256 scaled-MFMA kernel of gemm_tn8.h over (m-range, tile) items, partials
// through the caller's workspace + the ordered reduce pass like the bf16 entry point.  Na, Nb multiples of 16, ldp / ldq of 16 (bytes).
extern "C" int tvts_gemm_tn_fp8(const void* P8, int ldp, const void* Q8, int ldq, int M, int Na, int Nb, const float* scale_p,
                                const float* scale_q, float* out, int ldo, int accumulate, float* colsum, float* workspace,
                                long workspace_elems, int opts, hipStream_t stream) {
    if (M <= 0 || Na <= 0 || Nb <= 0 || !scale_p || !scale_q) return TVTS_EINVAL;
    if (Na % 16 || Nb % 16 || ldp % 16 || ldq % 16 || ldo % 4) return TVTS_EINVAL;
    if ((unsigned long long)M * (unsigned long long)(ldp > ldq ? ldp : ldq) >= (1ull << 32)) return TVTS_EINVAL;  // 32-bit DMA offsets
    GemmTN8 g;
    g.P = (const unsigned char*)P8; g.ldp = ldp; g.Q = (const unsigned char*)Q8; g.ldq = ldq; g.M = M; g.Na = Na; g.Nb = Nb;
    g.out = out; g.ldo = ldo; g.sp = scale_p; g.sq = scale_q; g.accumulate = accumulate; g.ws = nullptr;
    g.colsum = colsum; g.cs_ws = nullptr;
    g.tiles_a = ceil_div(Na, 256); g.tiles_b = ceil_div(Nb, 256); g.tiles_ab = g.tiles_a * g.tiles_b;
    g.a_fast = g.tiles_a < g.tiles_b ? 1 : 0;
    // contraction ranges by the cost model of the bf16 entry point: rounds of the 256 CUs x stages per range x ~1.9 us per 128-row
    // stage (tools/tn_fp8_bench.py) + 8 bytes per output element and range for the partials; ranges of at least 1 024 token rows
    int splits = 1;
    if ((opts >> 8) > 0) {
        splits = opts >> 8;
    } else {
        double best = 1e30;
        for (int sp = 1; sp <= 64; ++sp) {
            if (sp > 1 && M / sp < 1024) break;
            const double rounds = (double)ceil_div(g.tiles_ab * sp, 256);
            const double t = rounds * ceil_div(ceil_div(M, sp), 128) * 1.9 + (sp > 1 ? sp * (double)Na * (double)Nb * 1e-6 + 6.0 : 0.0);
            if (t < best) { best = t; splits = sp; }
        }
    }
    if (workspace == nullptr) splits = 1;
    else if ((long)splits * (Na * (long)Nb + Na) > workspace_elems) {
        const long fit = workspace_elems / (Na * (long)Nb + Na);
        splits = fit >= 2 ? (int)fit : 1;
    }
    g.m_per_split = ceil_div(ceil_div(M, splits), 128) * 128;
    splits = ceil_div(M, g.m_per_split);
    if (splits > 1) {
        g.ws = workspace;
        if (colsum) g.cs_ws = workspace + (size_t)splits * Na * Nb;
    }
    g.n_items = g.tiles_ab * splits;
    const int grid = ceil_div(g.n_items, 8) * 8;
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gemm_tn8_kernel, dim3(grid), dim3(512), 131072, stream, g);
    if (splits > 1) {
        const long n4 = (long)Na * Nb / 4;
        int rb = (int)((n4 + 255) / 256);
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(rb), dim3(256), 0, stream, workspace, splits, Na, Nb, out, ldo, accumulate,
                           (const float*)g.cs_ws, colsum);
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ------------------------------------------------------------------------------------------------
// tiny strided fp32 matmul: C[i,j] (+)= alpha * sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj]  (16x16 LDS tiles)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_small_kernel(const float* __restrict__ A, long sai, long sak,
                                                         const float* __restrict__ B, long sbk, long sbj,
                                                         int M, int N, int K, float alpha, const float* bias,
                                                         float* C, long ldc, int accumulate) {
    __shared__ float As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int ka = k0 + tx, kb = k0 + ty;
        As[ty][tx] = (i < M && ka < K) ? A[i * sai + ka * sak] : 0.f;
        Bs[ty][tx] = (kb < K && j < N) ? B[kb * sbk + j * sbj] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += As[ty][k] * Bs[k][tx];
        __syncthreads();
    }
    if (i < M && j < N) {
        float v = alpha * acc + (bias ? bias[j] : 0.f);
        if (accumulate) C[i * ldc + j] += v; else C[i * ldc + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same strided fp32 product on the matrix cores for outputs of at least one 64x64 tile: the G x G x E similarity of
// the gathered embeddings and its two backward products (G = world x pairs: 1536 at 8 x 192), the text / pooled
// projections.  v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at the fp32 vector rate (157 TF) that leaves the VALU
// free; 64x64 block tile, 4 waves in 2 x 2 with a 32x32 accumulator each, 16-deep stages staged k-major in LDS
// (As[k][i], Bs[k][j]: the fragment reads are unit-stride over lanes whatever the operand strides were).
// AK1 / BJ1: the operand's unit-stride direction (A: k or i; B: j or k) -- it fixes the thread -> element mapping of
// the staging loads so that they are coalesced for row-major and for transposed views alike.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// One operand's 64 x 32 stage (64 rows / columns of the output tile x 32 k) held in registers between its global loads and its
// LDS stores: 8 floats per thread.  U1 = the operand is unit-stride along k; VEC = 16-byte loads are legal (alignment and
// extents checked by the dispatcher), else scalar loads with bounds checks.
template <bool U1, bool VEC>
struct StageF32 {
    float v[8];
    // element e of this thread: (x = output row / column inside the tile, k inside the stage)
    static __device__ __forceinline__ void coord(int tid, int e, int& x, int& k) {
        if (U1) { x = (tid >> 3) + 32 * (e >> 2); k = (tid & 7) * 4 + (e & 3); }   // float4 along k, two row groups
        else    { k = (tid >> 4) + 16 * (e >> 2); x = (tid & 15) * 4 + (e & 3); }  // float4 along x, two k groups
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, long sx, long sk, int x0, int k0, int X, int Kd, int tid) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int x, k;
            coord(tid, 4 * h, x, k);
            const int gx = x0 + x, gk = k0 + k;
            if (VEC) {
                const bool ok = U1 ? (gx < X && gk < Kd) : (gk < Kd && gx < X);  // extents are multiples of 4 along the vector
                const f32x4 t = ok ? *(const f32x4*)(base + gx * sx + gk * sk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * h + e] = t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int xe, ke;
                    coord(tid, 4 * h + e, xe, ke);
                    v[4 * h + e] = (x0 + xe < X && k0 + ke < Kd) ? base[(x0 + xe) * sx + (k0 + ke) * sk] : 0.f;
                }
            }
        }
    }
    __device__ __forceinline__ void store(float (*S)[68], int tid) const {  // S[k][x]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int x, k;
            coord(tid, 4 * h, x, k);
            if (U1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) S[k + e][x] = v[4 * h + e];
            } else {
                *(f32x4*)&S[k][x] = (f32x4){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
            }
        }
    }
};

template <bool AK1, bool BJ1, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A, long sai, long sak,
                                                            const float* __restrict__ B, long sbk, long sbj, int M, int N,
                                                            int K, float alpha, const float* __restrict__ bias,
                                                            float* __restrict__ C, long ldc, int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[32][68], Bs[32][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    StageF32<AK1, VEC> ra;
    StageF32<!BJ1, VEC> rb;  // B is indexed (k, j): unit stride along k iff NOT along j
    ra.load(A, sai, sak, i0, 0, M, K, tid);
    rb.load(B, sbj, sbk, j0, 0, N, K, tid);
    for (int k0 = 0; k0 < K; k0 += 32) {
        ra.store(As, tid);
        rb.store(Bs, tid);
        __syncthreads();
        if (k0 + 32 < K) {  // the next stage's global loads fly under this stage's MFMAs
            ra.load(A, sai, sak, i0, k0 + 32, M, K, tid);
            rb.load(B, sbj, sbk, j0, k0 + 32, N, K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int j = j0 + wn * 32 + (lane & 31);
    if (j >= N) return;
    const float bj = bias ? bias[j] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (i >= M) continue;
        const float v = alpha * acc[e] + bj;
        float* c = C + (size_t)i * ldc + j;
        *c = accumulate ? *c + v : v;
    }
}

extern "C" int tvts_gemm_small_f32(const float* A, long sai, long sak, const float* B, long sbk, long sbj,
                                   int M, int N, int K, float alpha, const float* bias, float* C, long ldc,
                                   int accumulate, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TVTS_EINVAL;
    if (M >= 32 && N >= 32 && K >= 16) {  // worth a 64x64 MFMA tile; the tiny products (4-way heads, [B,E] rows) stay below
        const dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
        const bool ak1 = sak == 1, bj1 = sbj == 1;
        // 16-byte loads: unit stride in one direction, the other stride and the extent along the vector multiples of 4, bases aligned
        const bool va = ak1 ? (sai % 4 == 0 && K % 4 == 0) : (sai == 1 && sak % 4 == 0 && M % 4 == 0);
        const bool vb = bj1 ? (sbk % 4 == 0 && N % 4 == 0) : (sbk == 1 && sbj % 4 == 0 && K % 4 == 0);
        const bool vec = va && vb && ((size_t)A % 16 == 0) && ((size_t)B % 16 == 0);
        void (*kern)(const float*, long, long, const float*, long, long, int, int, int, float, const float*, float*, long, int);
        if (vec) kern = ak1 ? (bj1 ? gemm_f32_mfma_kernel<true, true, true> : gemm_f32_mfma_kernel<true, false, true>)
                            : (bj1 ? gemm_f32_mfma_kernel<false, true, true> : gemm_f32_mfma_kernel<false, false, true>);
        else kern = ak1 ? (bj1 ? gemm_f32_mfma_kernel<true, true, false> : gemm_f32_mfma_kernel<true, false, false>)
                        : (bj1 ? gemm_f32_mfma_kernel<false, true, false> : gemm_f32_mfma_kernel<false, false, false>);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, A, sai, sak, B, sbk, sbj, M, N, K, alpha, bias, C, ldc, accumulate);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    hipLaunchKernelGGL(gemm_small_kernel, dim3(ceil_div(N, 16), ceil_div(M, 16)), dim3(256), 0, stream, A, sai,
                       sak, B, sbk, sbj, M, N, K, alpha, bias, C, ldc, accumulate);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ------------------------------------------------------------------------------------------------
// A FEW rows through a linear layer with fp32 residual: out[r, n] = residual[r, n] + bias[n] + sum_k A[r * lda + k] * W[n * ldw + k]
// (A, W bf16; out / residual fp32) -- the CLS rows of the hybrid residual stream (one row per clip, row stride S * K in the block's
// [B * S, K] operand).  R is the per-GPU batch (2 ... 192): the tiled kernels would run it as ONE 128- or 256-row tile per output
// column block -- 6 blocks walking K = 3072 in 48 dependent stages, 12-35 us per launch, 24 launches per step.  Here a block owns a
// 16 x 16 output tile and its four waves split the contraction (wave w takes the 32-deep steps w, w + 4, ...), operands straight
// from L2 into the MFMA fragments (a lane's 16 bytes are contiguous in both), partials summed through LDS in wave order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_linear_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ W, int ldw,
                                                          int R, int N, int K, const float* __restrict__ bias,
                                                          const float* __restrict__ residual, int ldr, float* __restrict__ out, int ldo) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, gq = lane >> 4;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
    const int row = r0 + li < R ? r0 + li : R - 1;
    const bf16* ap = A + (size_t)row * lda + gq * 8;
    const bf16* wp = W + (size_t)(n0 + li) * ldw + gq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = wave * 32;
    for (; k + 3 * 128 < K; k += 4 * 128) {  // four steps' loads in flight before their MFMAs
        bf16x8 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = *(const bf16x8*)(ap + k + u * 128); w[u] = *(const bf16x8*)(wp + k + u * 128); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], w[u], acc, 0, 0, 0);
    }
    for (; k < K; k += 128) {
        const bf16x8 a = *(const bf16x8*)(ap + k), w = *(const bf16x8*)(wp + k);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w, acc, 0, 0, 0);
    }
    // lane holds D[m = 4 gq + e][n = li]
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][gq * 4 + e][li] = acc[e];
    __syncthreads();
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
    if (r0 + m < R) {
        float v = ((red[0][m][n] + red[1][m][n]) + red[2][m][n]) + red[3][m][n];
        if (bias) v += bias[n0 + n];
        if (residual) v += residual[(size_t)(r0 + m) * ldr + n0 + n];
        out[(size_t)(r0 + m) * ldo + n0 + n] = v;
    }
}

extern "C" int tvts_rows_linear_bf16(const void* A, long lda, const void* W, int ldw, int R, int N, int K, const float* bias,
                                     const float* residual, int ldr, float* out, int ldo, hipStream_t stream) {
    if (R <= 0 || N <= 0 || K <= 0 || !A || !W || !out) return TVTS_EINVAL;
    if (N % 16 || K % 32 || lda % 8 || ldw % 8) return TVTS_EINVAL;
    hipLaunchKernelGGL(rows_linear_kernel, dim3(N / 16, ceil_div(R, 16)), dim3(256), 0, stream, (const bf16*)A, lda, (const bf16*)W, ldw,
                       R, N, K, bias, residual, ldr, out, ldo);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ------------------------------------------------------------------------------------------------
// bias gradient: out[n] += sum_m X[m,n]  (bf16 in).  A block owns 64 columns and one of gridDim.y row ranges: 8 column threads
// (8 columns each) x 32 row lanes striding the range, merged through LDS in row-lane order.  With a workspace the row ranges'
// sums go to partials [range][N] that colsum_ranges_kernel adds in range order -- no atomics, run-to-run reproducible, and
// (rows / 4096) x (N / 64) blocks instead of N / 64 (12 blocks for a 768-wide bias at 150 k rows); without one a single range.
// (Only used where a bias trains under a frozen weight; the trained layers get their bias gradient inside the weight-gradient kernel.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int ld, int M, int N, int rows_per_range,
                                                     float* __restrict__ out, float* __restrict__ part) {
    __shared__ float lds[32][65];
    const int ct = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col = blockIdx.x * 64 + ct * 8;
    const int r0 = blockIdx.y * rows_per_range;
    const int r1 = r0 + rows_per_range < M ? r0 + rows_per_range : M;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const bf16x8 v = *(const bf16x8*)(X + (size_t)r * ld + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[rl][ct * 8 + e] = s[e];
    __syncthreads();
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && n < N) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += lds[r][threadIdx.x];
        if (part) part[(size_t)blockIdx.y * N + n] = t;
        else out[n] += t;
    }
}
__global__ __launch_bounds__(256) void colsum_ranges_kernel(const float* __restrict__ part, int ranges, int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float t = out[n];
    for (int y = 0; y < ranges; ++y) t += part[(size_t)y * N + n];
    out[n] = t;
}

extern "C" int tvts_colsum_bf16(const void* X, int ld, int M, int N, float* out, float* workspace, long workspace_elems,
                                hipStream_t stream) {
    if (M <= 0 || N <= 0 || N % 8 || ld % 8) return TVTS_EINVAL;
    int ranges = ceil_div(M, 4096);
    if (ranges > 256) ranges = 256;
    if (workspace == nullptr || (long)ranges * N > workspace_elems) ranges = 1;
    const int rows = ceil_div(M, ranges);
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), ranges), dim3(256), 0, stream, (const bf16*)X, ld, M, N, rows, out,
                       ranges > 1 ? workspace : nullptr);
    if (ranges > 1) hipLaunchKernelGGL(colsum_ranges_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, workspace, ranges, N, out);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
