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

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

struct GemmNT {
    const bf16* A; int lda;
    const bf16* B; int ldb;
    int M, N, K;
    const float* bias;
    const float* residual; int ldr;
    int act;
    bf16* preact; int ldp;
    const bf16* gate_h; int ldh; int gate_act;
    void* out; int ldc; int out_f32;
    int tiles_m, tiles_n;
    int sa_rows; // fp8: scale_a holds one scale per row of A (per-token activation scales) instead of one for the tensor
    int gc;      // 256x256 pipelined kernel: tile columns per column group (0 = plain row-major tile order)
    int ablate;  // experiment knob TVTS_NT_ABLATE: 1 skip MFMA, 2 skip DMA in the K loop, 4 skip fragment reads, 8 skip epilogue
    int swz;  // XOR mask of the LDS chunk swizzle (7; 0 = linear image, experiment knob TVTS_NT_SWZ)
    const float* sa; const float* sb;  // fp8 operands: per-tensor scales (device scalars), out = sa*sb * (A B^T) + ...
};

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

__device__ __forceinline__ bf16x8 frag_rows128(const char* lds_tile, int row, int chunk, int swz = 7) {
    return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & swz)) << 4));
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

        // epilogue: lane owns row m (column of the swapped MFMA) and 4 consecutive n
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= g.N) continue;
                f32x4 v = acc[j][i];
                if (g.bias) {
                    const f32x4 b = *(const f32x4*)(g.bias + n);
                    v += b;
                }
                if (ACT != ACT_NONE) {
                    if (g.preact) {
                        bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = h;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACT);
                }
                if (GATE != ACT_NONE) {
                    const bf16x4 h = *(const bf16x4*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_bwd((float)h[e], GATE);
                }
                if (g.residual) {
                    const f32x4 r = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n);
                    v += r;
                }
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

// ------------------------------------------------------------------------------------------------
// 256x256 tile variant: 512 threads = 8 waves as 2(M) x 4(N), each wave a 128x64 sub-tile (8x4 MFMA tiles,
// 128 accumulator registers).  Per 64-deep stage a wave issues the same 8 LDS-DMA pieces and 24 ds_read_b128
// as in the 128x128 kernel but feeds 64 MFMAs instead of 32, which is what lifts the MFMA duty cycle.
// LDS: 2 stages x (A 32 KiB + B 32 KiB) = 128 KiB, one block (2 waves per SIMD) per CU; persistent over tiles.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_rows256(const bf16* __restrict__ base, int ld, int row0, int row_max,
                                              int k0, char* lds_tile, int wave, int lane, int swz = 7) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r0 = (t * 8 + wave) * 8;
        const int row = r0 + (lane >> 3);
        const int slot = lane & 7;
        const int chunk = slot ^ (row & swz);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        const bf16* src = base + (size_t)grow * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))src, (LDS_PTR(void))(lds_tile + r0 * 128), 16, 0, 0);
    }
}

// Epilogue of a 128x64 wave sub-tile through a wave-private LDS patch: the MFMA layout (16 rows x 8 B per
// store instruction = sixteen 32-byte fragments of sixteen cache lines) is re-read as whole rows, so every store
// / residual load / gate load instruction covers full 128-B (bf16) or 256-B (fp32) row segments.
// patch: 16 rows x 272 B (64 fp32 + 16 B pad), one 16-row slab (MFMA tile row i) per pass.
template <int ACT, int GATE>
__device__ __forceinline__ void epilogue256_lds(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn,
                                                int lane, char* patch) {
    const int nb = n0 + wn * 64;
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nb + j * 16 + (lane >> 4) * 4;
        bias4[j] = (g.bias && n < g.N) ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)(patch + (lane & 15) * 272 + (j * 16 + (lane >> 4) * 4) * 4) = acc[j][i] + bias4[j];
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (g.out_f32) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
                f32x4 v = *(const f32x4*)(patch + r * 272 + c16 * 16);
                const int m = m0 + wm * 128 + i * 16 + r, n = nb + c16 * 4;
                if (m >= g.M || n >= g.N) continue;
                if (ACT != ACT_NONE) {
                    if (g.preact) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACT);
                }
                if (GATE != ACT_NONE) {
                    const bf16x4 h = *(const bf16x4*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_bwd((float)h[e], GATE);
                }
                if (g.residual) v += *(const f32x4*)(g.residual + (size_t)m * g.ldr + n);
                *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
            }
        } else {  // bf16 out: 8 columns (16 bytes) per lane
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                const f32x4 v0 = *(const f32x4*)(patch + r * 272 + c8 * 32), v1 = *(const f32x4*)(patch + r * 272 + c8 * 32 + 16);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int m = m0 + wm * 128 + i * 16 + r, n = nb + c8 * 8;
                if (m >= g.M || n >= g.N) continue;
                if (ACT != ACT_NONE) {
                    if (g.preact) {
                        bf16x8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (bf16)v[e];
                        *(bf16x8*)(g.preact + (size_t)m * g.ldp + n) = h;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e], ACT);
                }
                if (GATE != ACT_NONE) {
                    const bf16x8 h = *(const bf16x8*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= act_bwd((float)h[e], GATE);
                }
                if (g.residual) {
                    const f32x4 r0 = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n), r1 = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                *(bf16x8*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
            }
        }
    }
}

struct StageOff256 { unsigned off[4]; };
// byte offset (from the tile's first row, k = 0) of the 16-B chunk this lane fetches in DMA piece t; rows past
// the matrix end are clamped to its last row.  Invariant along k, so the K loop only bumps a scalar base pointer.
__device__ __forceinline__ void stage_offsets256(StageOff256& o, int ld, int row0, int row_max, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = (t * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        o.off[t] = (unsigned)(grow - row0) * (unsigned)ld * 2u + (unsigned)chunk * 16u;
    }
}
__device__ __forceinline__ const char* uniform_ptr(const void* p) {  // make wave-uniformity provable: SGPR base
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
template <int AUX = 0>
__device__ __forceinline__ void stage_issue256(const StageOff256& o, const bf16* ubase_, char* lds_tile, int wave) {
    const char* ubase = uniform_ptr(ubase_);
#pragma unroll
    for (int t = 0; t < 4; ++t)
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))((const char*)ubase + o.off[t]),
                                         (LDS_PTR(void))(lds_tile + (t * 8 + wave) * 1024), 16, 0, AUX);
}
// tile index -> tile origin.  gc == 0: row-major over (m, n).  gc > 0: column groups of gc tile columns, row-major
// inside a group, so the tiles an XCD works on at one time span gc weight panels instead of all of them.
__device__ __forceinline__ void tile_origin256(const GemmNT& g, int t, int gc, int& m0, int& n0) {
    if (gc <= 0) { m0 = (t / g.tiles_n) * 256; n0 = (t % g.tiles_n) * 256; return; }
    const int per_group = g.tiles_m * gc;
    const int grp = t / per_group, r = t - grp * per_group;
    const int c0 = grp * gc;
    const int w = (g.tiles_n - c0) < gc ? (g.tiles_n - c0) : gc;
    m0 = (r / w) * 256; n0 = (c0 + r % w) * 256;
}

template <int ACT, int GATE>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 32K | B 32K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;

    int t = slot;
    if (t >= range_n) return;
    int tile = range_lo + t;
    int m0 = (tile / g.tiles_n) * 256, n0 = (tile % g.tiles_n) * 256;
    StageOff256 oa, ob;  // per-lane byte offsets of this tile's DMA pieces (k-invariant)
    stage_offsets256(oa, g.lda, m0, g.M - 1, wave, lane);
    stage_offsets256(ob, g.ldb, n0, g.N - 1, wave, lane);
    stage_issue256(oa, g.A + (size_t)m0 * g.lda, smem, wave);
    stage_issue256(ob, g.B + (size_t)n0 * g.ldb, smem + 32768, wave);
    __syncthreads();
    int stage = 0;
    while (true) {
        f32x4 acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int t_next = t + per_xcd;
        const bool has_next = t_next < range_n;
        const int tile_n = range_lo + t_next;
        const int m0n = (tile_n / g.tiles_n) * 256, n0n = (tile_n % g.tiles_n) * 256;

        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + stage * 65536;
            char* nxt = smem + (stage ^ 1) * 65536;
            if (g.ablate & 2) {
            } else if (kt + 1 < nk) {
                stage_issue256(oa, g.A + (size_t)m0 * g.lda + (kt + 1) * BK, nxt, wave);
                stage_issue256(ob, g.B + (size_t)n0 * g.ldb + (kt + 1) * BK, nxt + 32768, wave);
            } else if (has_next) {
                stage_offsets256(oa, g.lda, m0n, g.M - 1, wave, lane);
                stage_offsets256(ob, g.ldb, n0n, g.N - 1, wave, lane);
                stage_issue256(oa, g.A + (size_t)m0n * g.lda, nxt, wave);
                stage_issue256(ob, g.B + (size_t)n0n * g.ldb, nxt + 32768, wave);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[8], bfr[4];
                if (g.ablate & 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (bf16)(float)(lane + j); bfr[j] = z; }
#pragma unroll
                    for (int i = 0; i < 8; ++i) { bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (bf16)(float)(lane + i); af[i] = z; }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bfr[j] = frag_rows128(cur + 32768, wn * 64 + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
                    for (int i = 0; i < 8; ++i) af[i] = frag_rows128(cur, wm * 128 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
                }
                if (g.ablate & 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(bfr[j]));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
                }
            }
            stage ^= 1;
            if (kt + 1 < nk) __syncthreads();
        }

        // all waves are done reading the last stage (its buffer hosts the epilogue patches); a raw barrier keeps the
        // next tile's first-stage DMA in flight (a __syncthreads would drain it here)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(g.ablate & 8)) epilogue256_lds<ACT, GATE>(g, acc, m0, n0, wm, wn, lane, smem + (stage ^ 1) * 65536 + wave * 8192);
        else if (lane == 0 && m0 == 123457) *(float*)g.out = acc[0][0][0] + acc[3][7][3];
        if (!has_next) break;
        __syncthreads();
        t = t_next; m0 = m0n; n0 = n0n;
    }
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, ANTI-PHASE wave groups.  The 8 waves form two groups (wm = 0 / 1); a SIMD hosts one wave of
// each.  Group 1 runs one s_barrier behind group 0, so in every barrier-to-barrier interval one group issues its
// LDS reads / LDS-DMA while the other issues 32 MFMAs: the matrix pipe of each SIMD always has a wave feeding it
// instead of all waves loading, then all waves multiplying.  Per 64-deep K-tile and wave:
//   I0  12 ds_read_b128 (k 0..31) + the 8 LDS-DMA pieces of the NEXT K-tile | barrier
//   I1  32 MFMA                                                             | barrier
//   I2  12 ds_read_b128 (k 32..63)                 [group 1: vmcnt(0)]      | barrier
//   I3  32 MFMA   [group 0: vmcnt(0)]  [last K-tile of an output tile: epilogue] | barrier
// The DMA of K-tile t+1 is issued 3 intervals before its first reader; raw s_barrier + hand-placed waitcnts keep
// it in flight across barriers (a __syncthreads would drain it).  K-tiles form one flat sequence across the
// block's persistent list of output tiles, so the next tile's first K-tile is fetched under the epilogue.
// ------------------------------------------------------------------------------------------------
#define RAW_BARRIER()                         \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_s_barrier();         \
        asm volatile("" ::: "memory");        \
    } while (0)

template <int ACT, int GATE>
__device__ __forceinline__ void epilogue256(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            f32x4 v = acc[j][i];
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m >= g.M || n >= g.N) continue;
            if (g.bias) {
                const f32x4 b = *(const f32x4*)(g.bias + n);
                v += b;
            }
            if (ACT != ACT_NONE) {
                if (g.preact) {
                    bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = h;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACT);
            }
            if (GATE != ACT_NONE) {
                const bf16x4 h = *(const bf16x4*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= act_bwd((float)h[e], GATE);
            }
            if (g.residual) {
                const f32x4 r = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n);
                v += r;
            }
            if (g.out_f32) {
                *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
            } else {
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
            }
        }
    }
}

template <int ACT, int GATE>
__global__ __launch_bounds__(512, 2) void gemm_nt256s_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 K-tiles][A 32K | B 32K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;
    if (slot >= range_n) return;  // whole block
    const int ntl = (range_n - slot + per_xcd - 1) / per_xcd;
    const int total_it = ntl * nk;

    int tile = range_lo + slot;
    int m0 = (tile / g.tiles_n) * 256, n0 = (tile % g.tiles_n) * 256;
    stage_rows256(g.A, g.lda, m0, g.M - 1, 0, smem, wave, lane);
    stage_rows256(g.B, g.ldb, n0, g.N - 1, 0, smem + 32768, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();
    if (wm == 1) RAW_BARRIER();  // group 1 trails by one interval from here on

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kt = 0, tl = 0;
    for (int it = 0; it < total_it; ++it) {
        char* cur = smem + (it & 1) * 65536;
        char* nxt = smem + ((it + 1) & 1) * 65536;
        bf16x8 af[8], bfr[4];
        // ---- I0: fragments of k 0..31, DMA of the next K-tile
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_rows128(cur + 32768, wn * 64 + j * 16 + (lane & 15), (lane >> 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = frag_rows128(cur, wm * 128 + i * 16 + (lane & 15), (lane >> 4));
        if (it + 1 < total_it) {
            int nm0 = m0, nn0 = n0, nk0 = (kt + 1) * BK;
            if (kt + 1 == nk) {
                const int tn = range_lo + slot + (tl + 1) * per_xcd;
                nm0 = (tn / g.tiles_n) * 256; nn0 = (tn % g.tiles_n) * 256; nk0 = 0;
            }
            stage_rows256(g.A, g.lda, nm0, g.M - 1, nk0, nxt, wave, lane);
            stage_rows256(g.B, g.ldb, nn0, g.N - 1, nk0, nxt + 32768, wave, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RAW_BARRIER();
        // ---- I1
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        RAW_BARRIER();
        // ---- I2: fragments of k 32..63
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_rows128(cur + 32768, wn * 64 + j * 16 + (lane & 15), 4 + (lane >> 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = frag_rows128(cur, wm * 128 + i * 16 + (lane & 15), 4 + (lane >> 4));
        if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RAW_BARRIER();
        // ---- I3
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kt + 1 == nk) {
            epilogue256<ACT, GATE>(g, acc, m0, n0, wm, wn, lane);
            kt = 0; ++tl;
            const int tn = range_lo + slot + tl * per_xcd;
            m0 = (tn / g.tiles_n) * 256; n0 = (tn % g.tiles_n) * 256;
        } else {
            ++kt;
        }
        RAW_BARRIER();
    }
    if (wm == 0) RAW_BARRIER();
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, SOFTWARE-PIPELINED fragments (the production NT kernel).  An ablation of the plain 256x256
// kernel (TVTS_NT_ABLATE) showed its three phases -- LDS-DMA wait, 24 ds_read_b128 per wave, 64 MFMAs per
// wave -- running back to back: MFMA alone 105 us, DMA + reads alone 95 us, epilogue 53 us, together 237 us
// (M 50240, N 2304, K 768).  Here the fragment registers are double-buffered so that the reads of the next
// half K-step are in flight while the matrix pipe works on the current one, and the DMA of stage s+2 is issued
// right after the barrier that frees its buffer, a full stage ahead of its consumer:
//     F1 <- ds_read k 32..63 (cur) | MFMA(F0) | vmcnt(0) lgkmcnt(0) barrier | DMA(s+2 -> cur) |
//     F0 <- ds_read k 0..31 (nxt)  | MFMA(F1) | [tile end: epilogue]
// Stages form one flat sequence over the block's persistent tile list.  LDS: 2 x 64 KiB stages + 8 x 4 KiB
// XOR-swizzled epilogue patches = 160 KiB exactly.
// ------------------------------------------------------------------------------------------------
template <int ACT, int GATE>
__device__ __forceinline__ void epilogue256_patch(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn,
                                                  int lane, char* patch, float scale = 1.0f,
                                                  const float* row_scale = nullptr) {
    // patch: 16 rows x 256 B (64 fp32), 16-B chunk c of row r stored at chunk c ^ r
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nb + j * 16 + gq * 4;
        bias4[j] = (g.bias && n < g.N) ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float sc = scale;
        if (row_scale) {  // per-row (token) scale of the fp8 A operand; `scale` then holds the weight's tensor scale
            const int m = m0 + wm * 128 + i * 16 + li;
            sc *= row_scale[m < g.M ? m : g.M - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)(patch + li * 256 + (((j * 4 + gq) ^ li) << 4)) = acc[j][i] * sc + bias4[j];
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (g.out_f32) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
                f32x4 v = *(const f32x4*)(patch + r * 256 + ((c16 ^ r) << 4));
                const int m = m0 + wm * 128 + i * 16 + r, n = nb + c16 * 4;
                if (m >= g.M || n >= g.N) continue;
                if (ACT != ACT_NONE) {
                    if (g.preact) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACT);
                }
                if (GATE != ACT_NONE) {
                    const bf16x4 h = *(const bf16x4*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_bwd((float)h[e], GATE);
                }
                if (g.residual) v += *(const f32x4*)(g.residual + (size_t)m * g.ldr + n);
                *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                const f32x4 v0 = *(const f32x4*)(patch + r * 256 + (((2 * c8) ^ r) << 4));
                const f32x4 v1 = *(const f32x4*)(patch + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int m = m0 + wm * 128 + i * 16 + r, n = nb + c8 * 8;
                if (m >= g.M || n >= g.N) continue;
                if (ACT != ACT_NONE) {
                    if (g.preact) {
                        bf16x8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (bf16)v[e];
                        *(bf16x8*)(g.preact + (size_t)m * g.ldp + n) = h;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e], ACT);
                }
                if (GATE != ACT_NONE) {
                    const bf16x8 h = *(const bf16x8*)(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= act_bwd((float)h[e], GATE);
                }
                if (g.residual) {
                    const f32x4 r0 = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n), r1 = *(const f32x4*)(g.residual + (size_t)m * g.ldr + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                *(bf16x8*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
            }
        }
    }
}

#define RAW_BARRIER_P()                       \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_s_barrier();         \
        asm volatile("" ::: "memory");        \
    } while (0)

template <int ACT, int GATE, bool FP8 = false>
__global__ __launch_bounds__(512, 2) void gemm_nt256p_kernel(GemmNT g) {
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

    // DMA cursor
    int i_st = 0, i_kt = 0, i_tl = 0, i_m0, i_n0;
    StageOff256 oa, ob;
    {
        tile_origin256(g, range_lo + slot, gc, i_m0, i_n0);
        stage_offsets256(oa, g.lda, i_m0, g.M - 1, wave, lane);
        stage_offsets256(ob, g.ldb, i_n0, g.N - 1, wave, lane);
    }
    auto issue = [&]() {
        char* dst = smem + (i_st & 1) * 65536;
        if (g.ablate & 16) stage_issue256<2>(oa, g.A + (size_t)i_m0 * g.lda + i_kt * BK, dst, wave);
        else stage_issue256<0>(oa, g.A + (size_t)i_m0 * g.lda + i_kt * BK, dst, wave);
        if (g.ablate & 32) stage_issue256<2>(ob, g.B + (size_t)i_n0 * g.ldb + i_kt * BK, dst + 32768, wave);
        else stage_issue256<0>(ob, g.B + (size_t)i_n0 * g.ldb + i_kt * BK, dst + 32768, wave);
        ++i_st;
        if (++i_kt == nk) {
            i_kt = 0; ++i_tl;
            tile_origin256(g, range_lo + slot + i_tl * per_xcd, gc, i_m0, i_n0);
            stage_offsets256(oa, g.lda, i_m0, g.M - 1, wave, lane);
            stage_offsets256(ob, g.ldb, i_n0, g.N - 1, wave, lane);
        }
    };
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER_P();
    if (i_st < total_st) issue();

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kt = 0, tl = 0, m0, n0;
    tile_origin256(g, range_lo + slot, gc, m0, n0);
    const int arow = wm * 128 + (lane & 15), brow = wn * 64 + (lane & 15), gq = lane >> 4;
    // fragment registers: two A half-sets (4 MFMA row-tiles each) and two B sets, refilled while the matrix pipe
    // works on the other one
    bf16x8 aF[2][4], bF[2][4];
#define LOAD_A(dst, buf, ks, h)                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) dst[i] = frag_rows128(buf, arow + ((h) * 4 + i) * 16, (ks) * 4 + gq)
#define LOAD_B(dst, buf, ks)                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) dst[j] = frag_rows128((buf) + 32768, brow + j * 16, (ks) * 4 + gq)
    // FP8: the operands are e4m3 matrices addressed as bf16 matrices of half the width (the staging and the LDS image are
    // byte-identical); a 16-byte fragment then holds 16 k-values of its row and feeds two 16x16x32 fp8 MFMAs (its low and
    // its high 8 bytes -- A and B use the same split, so every k meets its partner).
    typedef __attribute__((ext_vector_type(2))) long i64x2;
#define MFMA16(av, bv, h)                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
            if (FP8) {                                                                                 \
                const i64x2 a8 = __builtin_bit_cast(i64x2, av[i]), b8 = __builtin_bit_cast(i64x2, bv[j]); \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b8[0], a8[0], acc[j][(h) * 4 + i], 0, 0, 0); \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b8[1], a8[1], acc[j][(h) * 4 + i], 0, 0, 0); \
            } else {                                                                                   \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[j], av[i], acc[j][(h) * 4 + i], 0, 0, 0); \
            }                                                                                          \
        }
    LOAD_B(bF[0], smem, 0);
    LOAD_A(aF[0], smem, 0, 0);

    for (int st = 0; st < total_st; ++st) {
        const char* cur = smem + (st & 1) * 65536;
        const char* nxt = smem + ((st + 1) & 1) * 65536;
        LOAD_A(aF[1], cur, 0, 1);
        MFMA16(aF[0], bF[0], 0);
        LOAD_B(bF[1], cur, 1);
        LOAD_A(aF[0], cur, 1, 0);
        MFMA16(aF[1], bF[0], 1);
        LOAD_A(aF[1], cur, 1, 1);
        MFMA16(aF[0], bF[1], 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        RAW_BARRIER_P();
        if (i_st < total_st) issue();  // stage st+2 into the buffer every wave has just finished reading
        if (st + 1 < total_st) {
            LOAD_B(bF[0], nxt, 0);
            LOAD_A(aF[0], nxt, 0, 0);
        }
        MFMA16(aF[1], bF[1], 1);
        if (++kt == nk) {
            if (FP8 && g.sa_rows) epilogue256_patch<ACT, GATE>(g, acc, m0, n0, wm, wn, lane, patch, g.sb[0], g.sa);
            else epilogue256_patch<ACT, GATE>(g, acc, m0, n0, wm, wn, lane, patch, FP8 ? g.sa[0] * g.sb[0] : 1.0f);
            kt = 0; ++tl;
            tile_origin256(g, range_lo + slot + tl * per_xcd, gc, m0, n0);
        }
    }
#undef LOAD_A
#undef LOAD_B
#undef MFMA16
}

// ------------------------------------------------------------------------------------------------
// 256x256 tile, DEEP RING: 32-deep stages in an NS-slot LDS ring (NS x 32 KiB), NS-1 stages of LDS-DMA in
// flight at all times.  PMC on the 2-stage kernels shows the matrix pipe 36 % busy and the waves parked on
// vmcnt/barrier 41 % of the time: with one K-tile of prefetch the K-tile time stretches to the L2/MALL
// latency under load (~13 B/clk/CU delivered).  More bytes in flight per CU is the lever, so the stage is
// halved (BK 32) and the ring deepened; waits are COUNTED (s_waitcnt vmcnt(4*(NS-2))) with raw s_barrier so
// the younger stages stay in flight across the barrier.
// Stage layout: A [256 rows][32 k] bf16 = rows of 64 B (16 KiB) | B the same; 16-B chunk c of row r is stored
// at chunk c ^ ((r >> 2) & 3)  (conflict-free ds_read_b128 for 16 consecutive rows at one chunk).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_rows256_k32(const bf16* __restrict__ base, int ld, int row0, int row_max,
                                                  int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r0 = (t * 8 + wave) * 16;
        const int row = r0 + (lane >> 2);
        const int slot = lane & 3;
        const int chunk = slot ^ ((row >> 2) & 3);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        const bf16* src = base + (size_t)grow * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const GLB_PTR(void))src, (LDS_PTR(void))(lds_tile + r0 * 64), 16, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 frag_rows_k32(const char* lds_tile, int row, int chunk) {
    return *(const bf16x8*)(lds_tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int ACT, int GATE, int NS>
__global__ __launch_bounds__(512, 2) void gemm_nt256r_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS][A 16K | B 16K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / 32;
    if (slot >= range_n) return;
    const int ntl = (range_n - slot + per_xcd - 1) / per_xcd;
    const int total_st = ntl * nk;

    // issue cursor (stage to be loaded next) and compute cursor
    int i_st = 0, i_kt = 0, i_tl = 0;
    int i_m0, i_n0;
    {
        const int tile = range_lo + slot;
        i_m0 = (tile / g.tiles_n) * 256; i_n0 = (tile % g.tiles_n) * 256;
    }
    auto issue = [&]() {
        char* dst = smem + (i_st % NS) * 32768;
        stage_rows256_k32(g.A, g.lda, i_m0, g.M - 1, i_kt * 32, dst, wave, lane);
        stage_rows256_k32(g.B, g.ldb, i_n0, g.N - 1, i_kt * 32, dst + 16384, wave, lane);
        ++i_st;
        if (++i_kt == nk) {
            i_kt = 0; ++i_tl;
            const int tile = range_lo + slot + i_tl * per_xcd;
            i_m0 = (tile / g.tiles_n) * 256; i_n0 = (tile % g.tiles_n) * 256;
        }
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (i_st < total_st) issue();

    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kt = 0, tl = 0;
    int m0, n0;
    {
        const int tile = range_lo + slot;
        m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
    }
    for (int st = 0; st < total_st; ++st) {
        // stage `st` must have landed: stages issued after it may stay in flight (4 DMA pieces each)
        const int younger = i_st - st - 1;  // 0 .. NS-2
        if (younger >= NS - 2) wait_vmcnt<4 * (NS - 2)>();
        else if (younger == 2) wait_vmcnt<8>();
        else if (younger == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        RAW_BARRIER();
        if (i_st < total_st) issue();  // into the slot whose last reader finished before the barrier
        const char* cur = smem + (st % NS) * 32768;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_rows_k32(cur + 16384, wn * 64 + j * 16 + (lane & 15), (lane >> 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = frag_rows_k32(cur, wm * 128 + i * 16 + (lane & 15), (lane >> 4));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (++kt == nk) {
            epilogue256<ACT, GATE>(g, acc, m0, n0, wm, wn, lane);
            kt = 0; ++tl;
            const int tile = range_lo + slot + tl * per_xcd;
            m0 = (tile / g.tiles_n) * 256; n0 = (tile % g.tiles_n) * 256;
        }
    }
}

#include <stdlib.h>
static int nt_tile_env() { const char* e = getenv("TVTS_NT_TILE"); return e ? atoi(e) : 0; }
static int g_nt_tile = nt_tile_env();  // 0 = auto, 128 / 256 / 512 (= 256 anti-phase) forced (tools/gemm_bench.py)
extern "C" void tvts_gemm_set_nt_tile(int t) { g_nt_tile = t; }

// fewest 256x256 tiles for which the pipelined 256-tile kernel is chosen over the 128-tile one (dev knob TVTS_NT_MIN_TILES):
// the text tower of a 192-pair step has 192 of them (M = 24 576, N = 512) and runs ~10 % faster on the 256-tile kernel
static int nt_min_tiles() {
    static const int v = []() { const char* e = getenv("TVTS_NT_MIN_TILES"); return e ? atoi(e) : 150; }();
    return v;
}

extern "C" int tvts_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                                 const float* bias, const float* residual, int ldr, int act, void* preact,
                                 int ldp, const void* gate_h, int ldh, int gate_act, void* out, int ldc,
                                 int out_f32, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TVTS_EINVAL;
    if (K % BK != 0 || N % 4 != 0 || lda % 8 != 0 || ldb % 8 != 0) return TVTS_EINVAL;
    if ((ldc % 4) || (residual && (ldr % 4)) || (preact && (ldp % 4)) || (gate_h && (ldh % 4))) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.preact = (bf16*)preact; g.ldp = ldp; g.gate_h = (const bf16*)gate_h; g.ldh = ldh; g.gate_act = gate_act;
    g.out = out; g.ldc = ldc; g.out_f32 = out_f32; g.sa = nullptr; g.sb = nullptr; g.sa_rows = 0;
    { static const char* e = getenv("TVTS_NT_SWZ"); g.swz = e ? atoi(e) : 7; }
    { const char* e = getenv("TVTS_NT_ABLATE"); g.ablate = e ? atoi(e) : 0; }
    // Wide outputs (>= 10 tile columns: the MLP's 4x expansion): walk the tiles in column groups of 6 or 5.  An XCD's 32
    // co-resident tiles then span 5-6 weight panels x 5-6 row panels instead of all 12-20 weight panels x 2-3 row panels
    // -- FETCH_SIZE (L2 misses) of the fc1 / fc2-dgrad GEMMs falls by 40 % (tools/gemm_l2.py, tools/l2run.sh), time by 3-5 %.
    g.gc = 0;
    {
        const int tn = ceil_div(N, 256);
        if (tn >= 10) g.gc = tn % 6 == 0 ? 6 : tn % 5 == 0 ? 5 : 0;
        if (g.ablate >> 8) g.gc = (g.ablate >> 8) & 15;   // dev override (15 = force 0)
        if (g.gc == 15) g.gc = 0;
    }
    const bool pipe = g_nt_tile == 768 || (g_nt_tile == 0);
    const bool stag = g_nt_tile == 512;
    const int ring = g_nt_tile == 1024 ? 4 : g_nt_tile == 1280 ? 5 : 0;
    if ((g_nt_tile >= 256) && (N % 8 || ldc % 8 || (preact && ldp % 8) || (gate_h && ldh % 8))) return TVTS_EINVAL;
    const bool big = stag || ring || g_nt_tile == 768 || (g_nt_tile == 256) || (g_nt_tile == 0 && N % 256 == 0 && (long)ceil_div(M, 256) * (N / 256) >= nt_min_tiles());
    if (big) {
        g.tiles_n = ceil_div(N, 256);
        g.tiles_m = ceil_div(M, 256);
        const int total_tiles = g.tiles_m * g.tiles_n;
        const int grid = total_tiles < 256 ? ((total_tiles + 7) / 8) * 8 : 256;
        void (*kern)(GemmNT) = nullptr;
        if (gate_h) {
            if (act != ACT_NONE) return TVTS_EINVAL;
            kern = gate_act == ACT_QUICK_GELU ? (stag ? gemm_nt256s_kernel<0, 1> : gemm_nt256_kernel<0, 1>)
                 : gate_act == ACT_GELU_ERF ? (stag ? gemm_nt256s_kernel<0, 2> : gemm_nt256_kernel<0, 2>) : nullptr;
        } else {
            kern = act == ACT_NONE ? (stag ? gemm_nt256s_kernel<0, 0> : gemm_nt256_kernel<0, 0>)
                 : act == ACT_QUICK_GELU ? (stag ? gemm_nt256s_kernel<1, 0> : gemm_nt256_kernel<1, 0>)
                 : act == ACT_GELU_ERF ? (stag ? gemm_nt256s_kernel<2, 0> : gemm_nt256_kernel<2, 0>) : nullptr;
        }
        int lds_bytes = 131072;
        if (pipe && !stag && !ring && g_nt_tile != 256) {
            lds_bytes = 163840;
            if (gate_h) kern = gate_act == ACT_QUICK_GELU ? gemm_nt256p_kernel<0, 1> : gemm_nt256p_kernel<0, 2>;
            else kern = act == ACT_NONE ? gemm_nt256p_kernel<0, 0> : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0> : gemm_nt256p_kernel<2, 0>;
        }
        if (ring) {
            if (K % 32) return TVTS_EINVAL;
            lds_bytes = ring * 32768;
            if (gate_h) kern = gate_act == ACT_QUICK_GELU ? (ring == 4 ? gemm_nt256r_kernel<0, 1, 4> : gemm_nt256r_kernel<0, 1, 5>)
                             : (ring == 4 ? gemm_nt256r_kernel<0, 2, 4> : gemm_nt256r_kernel<0, 2, 5>);
            else kern = act == ACT_NONE ? (ring == 4 ? gemm_nt256r_kernel<0, 0, 4> : gemm_nt256r_kernel<0, 0, 5>)
                      : act == ACT_QUICK_GELU ? (ring == 4 ? gemm_nt256r_kernel<1, 0, 4> : gemm_nt256r_kernel<1, 0, 5>)
                      : (ring == 4 ? gemm_nt256r_kernel<2, 0, 4> : gemm_nt256r_kernel<2, 0, 5>);
        }
        if (!kern) return TVTS_EINVAL;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, g);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    g.tiles_n = ceil_div(N, BN);
    g.tiles_m = ceil_div(M, BM);
    const int total_tiles = g.tiles_m * g.tiles_n;
    int tiles = total_tiles < 512 ? ((total_tiles + 7) / 8) * 8 : 512;  // persistent grid: 2 blocks x 256 CUs, multiple of 8
    void (*kern)(GemmNT) = nullptr;
    if (gate_h) {
        if (act != ACT_NONE) return TVTS_EINVAL;
        kern = gate_act == ACT_QUICK_GELU ? gemm_nt_kernel<0, 1> : gate_act == ACT_GELU_ERF ? gemm_nt_kernel<0, 2> : nullptr;
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
                                int ldp, void* out, int ldc, int out_f32, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !scale_a || !scale_b) return TVTS_EINVAL;
    if (K % 128 || N % 8 || lda % 16 || ldb % 16 || ldc % 8 || (residual && ldr % 4) || (preact && ldp % 8)) return TVTS_EINVAL;
    GemmNT g;
    g.A = (const bf16*)A; g.lda = lda / 2; g.B = (const bf16*)B; g.ldb = ldb / 2;  // byte-identical bf16 view, half as wide
    g.M = M; g.N = N; g.K = K / 2; g.bias = bias; g.residual = residual; g.ldr = ldr; g.act = act;
    g.preact = (bf16*)preact; g.ldp = ldp; g.gate_h = nullptr; g.ldh = 0; g.gate_act = ACT_NONE;
    g.out = out; g.ldc = ldc; g.out_f32 = out_f32; g.swz = 7; g.ablate = 0; g.gc = 0; g.sa = scale_a; g.sb = scale_b; g.sa_rows = scale_a_rows ? 1 : 0;
    {   // same column-group walk as the bf16 dispatch (N = 3840: 415 -> 398 us)
        const char* e = getenv("TVTS_NT_ABLATE");
        const int ab = e ? atoi(e) : 0;
        const int tn = ceil_div(N, 256);
        if (tn >= 10) g.gc = tn % 6 == 0 ? 6 : tn % 5 == 0 ? 5 : 0;
        if (ab >> 8) g.gc = (ab >> 8) & 15;   // dev override (15 = force 0)
        if (g.gc == 15) g.gc = 0;
    }
    g.tiles_n = ceil_div(N, 256);
    g.tiles_m = ceil_div(M, 256);
    const int total_tiles = g.tiles_m * g.tiles_n;
    const int grid = total_tiles < 256 ? ((total_tiles + 7) / 8) * 8 : 256;
    void (*kern)(GemmNT) = act == ACT_NONE ? gemm_nt256p_kernel<0, 0, true> : act == ACT_QUICK_GELU ? gemm_nt256p_kernel<1, 0, true>
                         : act == ACT_GELU_ERF ? gemm_nt256p_kernel<2, 0, true> : nullptr;
    if (!kern) return TVTS_EINVAL;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, stream, g);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
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
    int ablate;     // experiment knob TVTS_TN_ABLATE: 1 skip MFMA, 2 skip DMA after the prologue, 4 skip fragment reads, 8 skip epilogue
};

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

__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(GemmTN g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][P 16K | Q 16K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;

    // Work items are (m-range, output tile) pairs in range-major order; XCD x (= blockIdx % 8) takes the x-th
    // contiguous eighth of that list, so the P/Q rows of an m-range are pulled into ONE XCD's L2 (two at a
    // boundary) and shared there by all output tiles, instead of being fetched by all eight L2s.
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int item = xcd * per + jx;
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
            if (a < g.Na) atomicAdd(g.colsum + a, cs[i][0]);
        }
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

// out[a,b] = (accumulate ? out[a,b] : 0) + sum_s ws[s][a][b]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int splits, int Na, int Nb,
                                                        float* __restrict__ out, int ldo, int accumulate) {
    const size_t n4 = (size_t)Na * Nb / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const size_t e = i * 4;
        const int a = (int)(e / Nb), b = (int)(e % Nb);
        f32x4 s = accumulate ? *(const f32x4*)(out + (size_t)a * ldo + b) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += *(const f32x4*)(ws + (size_t)k * Na * Nb + e);
        *(f32x4*)(out + (size_t)a * ldo + b) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// TN, 256x256 tile, software-pipelined (the production weight-gradient kernel for large outputs).
// Same skeleton as gemm_nt256p_kernel: 8 waves (2 x 4), wave tile 128(a) x 64(b), two 64-row stages in LDS,
// fragment registers double-buffered in four chunks per stage, DMA of stage s+2 issued right after the barrier
// that frees its buffer.  Differences: the contraction runs over token rows, so stages advance along m and the
// fragments are gathered with ds_read_b64_tr_b16 from [64 m][256 cols] tiles (512-B rows, 32-B chunk c of row r
// at c ^ (r & 7)).  hipcc parks an s_waitcnt vmcnt(0) in front of every transposing read that follows an LDS-DMA
// builtin, which would drain the prefetch, so the DMA is issued from inline asm (M0 saved/restored) and waited
// for by hand.  The bias gradient (column sums of P) is accumulated on the VALU from the P fragments.
// Work items = (m-range, tile) in range-major order, one contiguous eighth per XCD, one round of 256 blocks.
// ------------------------------------------------------------------------------------------------
struct StageOffTN { unsigned off[4]; };
__device__ __forceinline__ void tn_offsets(StageOffTN& o, int ld, int rows_valid, int c_max, int c0, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = 2 * (t * 8 + wave) + (lane >> 5);
        const int s16 = lane & 31;
        const int c32 = (s16 >> 1) ^ (row & 7);
        int col = c0 + c32 * 16 + (s16 & 1) * 8;
        col = col < c_max ? col : c_max;
        const int r = row < rows_valid ? row : rows_valid - 1;
        o.off[t] = ((unsigned)r * (unsigned)ld + (unsigned)col) * 2u;
    }
}
__device__ __forceinline__ void tn_issue(const StageOffTN& o, const bf16* ubase_, unsigned lds_tile, int wave) {
    const char* ubase = uniform_ptr(ubase_);
#pragma unroll
    for (int t = 0; t < 4; ++t) glds16_asm(o.off[t], ubase, lds_tile + (unsigned)(t * 8 + wave) * 1024u);
}
__device__ __forceinline__ bf16x8 frag_tr512(const char* lds_tile, int u, int ct, int lane) {
    const int g = lane >> 4, i = lane & 15;
    s16x4 h[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = u * 32 + half * 16 + g * 4 + (i >> 2);
        const char* p = lds_tile + row * 512 + ((ct ^ (row & 7)) << 5) + (i & 3) * 8;
        h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// DMA piece offsets for a 32-row stage: piece t of wave w covers rows 2*(t*8+w), +1  (t = 0,1)
struct StageOffTN32 { unsigned off[2]; };
// 32-row stage of one operand = 4 column blocks of [32 rows][64 cols] (128-B rows, 4 KiB each).  A DMA piece is
// 8 rows x 128 B of one column block (the request shape of the NT kernels: eight different rows per wave
// instruction); piece p = cb*4 + rg, wave w issues pieces w and w+8.  32-B chunk c of row r sits at c ^ ((r>>1)&3).
__device__ __forceinline__ void tn_offsets32(StageOffTN32& o, int ld, int rows_valid, int c_max, int c0, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int piece = t * 8 + wave, cb = piece >> 2, rg = piece & 3;
        const int row = rg * 8 + (lane >> 3);
        const int s16 = lane & 7;
        const int c4 = (s16 >> 1) ^ ((row >> 1) & 3);
        int col = c0 + cb * 64 + c4 * 16 + (s16 & 1) * 8;
        col = col < c_max ? col : c_max;
        const int r = row < rows_valid ? row : rows_valid - 1;
        o.off[t] = ((unsigned)r * (unsigned)ld + (unsigned)col) * 2u;
    }
}
__device__ __forceinline__ void tn_issue32(const StageOffTN32& o, const bf16* ubase_, unsigned lds_tile, int wave, bool use_builtin = false) {
    const char* ubase = uniform_ptr(ubase_);
    if (use_builtin) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            __builtin_amdgcn_global_load_lds((const GLB_PTR(void))(ubase + o.off[t]), (LDS_PTR(void))(size_t)(lds_tile + (unsigned)(t * 8 + wave) * 1024u), 16, 0, 0);
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) glds16_asm(o.off[t], ubase, lds_tile + (unsigned)(t * 8 + wave) * 1024u);
}
// fragment of 16-column block ct (0..15) over the 32 rows of the stage
__device__ __forceinline__ bf16x8 frag_tr_cb(const char* lds_tile, int ct, int lane) {
    const int g = lane >> 4, i = lane & 15;
    s16x4 h[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = half * 16 + g * 4 + (i >> 2);
        const char* p = lds_tile + (ct >> 2) * 4096 + row * 128 + (((ct & 3) ^ ((row >> 1) & 3)) << 5) + (i & 3) * 8;
        h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}
template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// Stages are 32 token rows ([32][256] bf16 per operand, 16 KiB each) in an NS-slot ring: NS-1 stages of LDS-DMA
// stay in flight (the weight-gradient operands are streamed once from HBM with no reuse along the contraction,
// so the loop is bound by miss latency x bytes in flight, not by L2 bandwidth).
template <bool CS, int NS>
__device__ __forceinline__ void tn256_body(const GemmTN& g, char* smem, unsigned lds0, int wave, int lane, int a0, int b0,
                                           int m_begin, int m_end) {
    const int wa = wave >> 2, wb = wave & 3;
    const int nst = (m_end - m_begin + 31) / 32;
    const int tail_rows = (m_end - m_begin) - (nst - 1) * 32;  // 1..32
    const int nfull = tail_rows == 32 ? nst : nst - 1;

    const int rot = (g.ablate & 64) ? 0 : (nfull > 0 ? (int)(((unsigned)(a0 / 256) * 7u + (unsigned)(b0 / 256) * 3u) % (unsigned)nfull) : 0);
    StageOffTN32 op, oq;
    tn_offsets32(op, g.ldp, 32, g.Na - 8, a0, wave, lane);
    tn_offsets32(oq, g.ldq, 32, g.Nb - 8, b0, wave, lane);
    int i_st = 0;
    auto issue = [&]() {
        const unsigned dst = lds0 + (unsigned)(i_st % NS) * 32768u;
        // blocks that share operand panels start at different stages (rotation over the full stages), so the
        // eight-or-so CUs reading one line do not all ask the same L2 channel in the same microsecond
        const int ph = (i_st < nfull) ? (i_st + rot) % nfull : i_st;
        const size_t mrow = (size_t)(m_begin + ph * 32);
        if (i_st == nst - 1 && tail_rows < 32) {
            StageOffTN32 opt, oqt;
            tn_offsets32(opt, g.ldp, tail_rows, g.Na - 8, a0, wave, lane);
            tn_offsets32(oqt, g.ldq, tail_rows, g.Nb - 8, b0, wave, lane);
            tn_issue32(opt, g.P + mrow * g.ldp, dst, wave);
            tn_issue32(oqt, g.Q + mrow * g.ldq, dst + 16384u, wave);
        } else {
            if (!(g.ablate & 16)) tn_issue32(op, g.P + mrow * g.ldp, dst, wave, (g.ablate & 128) != 0);
            if (!(g.ablate & 32)) tn_issue32(oq, g.Q + mrow * g.ldq, dst + 16384u, wave, (g.ablate & 128) != 0);
        }
        ++i_st;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (i_st < nst && !((g.ablate & 2) && p > 0)) issue();
    auto wait_stage = [&](int st) {  // stage `st` landed; younger stages may stay in flight (4 pieces each)
        const int younger = i_st - st - 1;
        if (younger >= 3) wait_vm_lgkm<12>();
        else if (younger == 2) wait_vm_lgkm<8>();
        else if (younger == 1) wait_vm_lgkm<4>();
        else wait_vm_lgkm<0>();
    };
    wait_stage(0);
    RAW_BARRIER_P();

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 csacc = {0.f, 0.f, 0.f, 0.f};  // bias gradient: row i of this tile = column sums of a-tile i (selector MFMA)

    bf16x8 pF[2][4], qF[2][4];
    if (g.ablate & 4) { for (int x = 0; x < 2; ++x) for (int y = 0; y < 4; ++y) for (int e = 0; e < 8; ++e) { pF[x][y][e] = (bf16)(float)(lane + y); qF[x][y][e] = (bf16)(float)(lane - y); } }
#define TN_LOAD_P(dst, buf, h) if (!(g.ablate & 4)) { _Pragma("unroll") for (int i = 0; i < 4; ++i) dst[i] = frag_tr_cb(buf, wa * 8 + (h) * 4 + i, lane); }
#define TN_LOAD_Q(dst, buf) if (!(g.ablate & 4)) { _Pragma("unroll") for (int j = 0; j < 4; ++j) dst[j] = frag_tr_cb((buf) + 16384, wb * 4 + j, lane); }
#define TN_MFMA16(pv, qv, h)                                                                              \
    if (g.ablate & 1) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(pv[i])); asm volatile("" :: "v"(qv[i])); } } else \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
            acc[(h) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qv[j], pv[i], acc[(h) * 4 + i][j], 0, 0, 0); \
        if (CS) {                                                                                         \
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;                                   \
            const unsigned pat = ((lane & 15) == (h) * 4 + i) ? 0x3f803f80u : 0u;                          \
            const bf16x8 sel = __builtin_bit_cast(bf16x8, (u32x4){pat, pat, pat, pat});                    \
            csacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, pv[i], csacc, 0, 0, 0);                  \
        }                                                                                                 \
    }
    if (nfull > 0) {
        TN_LOAD_Q(qF[0], smem);
        TN_LOAD_P(pF[0], smem, 0);
    }
    // stage st uses qF[st & 1]; the loop is unrolled by two so the fragment sets are indexed statically
    auto step = [&](int st, auto parity) {
        constexpr int PQ = decltype(parity)::value;
        const char* cur = smem + (st % NS) * 32768;
        const char* nxt = smem + ((st + 1) % NS) * 32768;
        TN_LOAD_P(pF[1], cur, 1);
        TN_MFMA16(pF[0], qF[PQ], 0);
        if (st + 1 < nst) wait_stage(st + 1); else wait_vm_lgkm<0>();
        RAW_BARRIER_P();
        if (i_st < nst && !(g.ablate & 2)) issue();  // into the slot of stage st, which every wave has finished reading
        if (st + 1 < nfull) {
            TN_LOAD_Q(qF[PQ ^ 1], nxt);
            TN_LOAD_P(pF[0], nxt, 0);
        }
        TN_MFMA16(pF[1], qF[PQ], 1);
    };
    int st = 0;
    for (; st + 1 < nfull; st += 2) {
        step(st, std::integral_constant<int, 0>{});
        step(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < nfull) { step(st, std::integral_constant<int, 0>{}); ++st; }
    if (nfull < nst) {  // short last stage (landed: the last step waited for it, or the prologue did)
        const char* cur = smem + (nfull % NS) * 32768;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            TN_LOAD_Q(qF[0], cur);
            TN_LOAD_P(pF[0], cur, h);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = (e >> 2) * 16 + (lane >> 4) * 4 + (e & 3) < tail_rows;
#pragma unroll
                for (int i = 0; i < 4; ++i) pF[0][i][e] = ok ? pF[0][i][e] : (bf16)0.f;
            }
            if (h == 0) { TN_MFMA16(pF[0], qF[0], 0); } else { TN_MFMA16(pF[0], qF[0], 1); }
        }
    }
#undef TN_LOAD_P
#undef TN_LOAD_Q
#undef TN_MFMA16
    if (CS && wb == 0 && lane < 32) {  // csacc[r] in lane (gq, li): a-tile i = gq*4 + r, column li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = a0 + wa * 128 + ((lane >> 4) * 4 + r) * 16 + (lane & 15);
            if (a < g.Na) atomicAdd(g.colsum + a, csacc[r]);
        }
    }
    if (g.ablate & 8) { if (lane == 0 && a0 == 123457) g.out[0] = acc[0][0][0] + acc[7][3][3]; return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int a = a0 + wa * 128 + i * 16 + (lane & 15);
        if (a >= g.Na) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + wb * 64 + j * 16 + (lane >> 4) * 4;
            if (b >= g.Nb) continue;
            float* dst = g.out + (size_t)a * g.ldo + b;
            if (g.atomic) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, acc[i][j][e]);
            } else {
                *(f32x4*)dst = acc[i][j];
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void gemm_tn256p_kernel(GemmTN g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ring of [P 16K | Q 16K] stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int item = xcd * per + jx;
    if (item >= g.n_items) return;
    const int split = item / g.tiles_ab, t = item % g.tiles_ab;
    const int a0 = (t / g.tiles_b) * 256, b0 = (t % g.tiles_b) * 256;
    const int m_begin = split * g.m_per_split;
    int m_end = m_begin + g.m_per_split;
    m_end = m_end < g.M ? m_end : g.M;
    if (m_begin >= m_end) return;
    if (g.colsum != nullptr && (t % g.tiles_b) == 0) tn256_body<true, 5>(g, smem, lds0, wave, lane, a0, b0, m_begin, m_end);
    else tn256_body<false, 5>(g, smem, lds0, wave, lane, a0, b0, m_begin, m_end);
}

static int tn_tile_env() { const char* e = getenv("TVTS_TN_TILE"); return e ? atoi(e) : 0; }
static int g_tn_tile = tn_tile_env();  // 0 auto, 128, 256

extern "C" int tvts_gemm_tn_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int Na, int Nb,
                                 float* out, int ldo, int accumulate, float* colsum, float* workspace,
                                 long workspace_elems, hipStream_t stream) {
    if (M <= 0 || Na <= 0 || Nb <= 0) return TVTS_EINVAL;
    if (Na % 8 || Nb % 8 || ldp % 8 || ldq % 8 || ldo % 4) return TVTS_EINVAL;
    GemmTN g;
    g.ws = nullptr; g.early_dma = 0; g.tiles_a = 0; g.a_fast = 0;
    g.P = (const bf16*)P; g.ldp = ldp; g.Q = (const bf16*)Q; g.ldq = ldq; g.M = M; g.Na = Na; g.Nb = Nb;
    g.out = out; g.ldo = ldo; g.colsum = colsum;
    { static const char* e = getenv("TVTS_TN_ABLATE"); g.ablate = e ? atoi(e) : 0; }
    // the 256x256 ring variant is correct but not faster than the 128x128 kernel yet (its LDS-DMA stream delivers
    // ~4.5 TB/s whatever the ring depth): opt-in via TVTS_TN_TILE=256
    const bool big = Na % 256 == 0 && Nb % 256 == 0 && g_tn_tile == 256;
    if (big) {
        g.tiles_b = Nb / 256;
        g.tiles_ab = (Na / 256) * g.tiles_b;
        int splits = 256 / g.tiles_ab;  // one round of 256 blocks (1 per CU)
        if (splits < 1) splits = 1;
        while (splits > 1 && M / splits < 512) --splits;
        g.m_per_split = ceil_div(ceil_div(M, splits), 32) * 32;
        splits = ceil_div(M, g.m_per_split);
        if (!accumulate && splits > 1) {
            hipError_t e = hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)Nb * 4, Na, stream);
            if (e != hipSuccess) return (int)e;
        }
        g.atomic = (accumulate || splits > 1) ? 1 : 0;
        g.n_items = g.tiles_ab * splits;
        const int grid = ceil_div(g.n_items, 8) * 8;
        hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tn256p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e2 != hipSuccess) return (int)e2;
        hipLaunchKernelGGL(gemm_tn256p_kernel, dim3(grid), dim3(512), 163840, stream, g);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const int tiles_a = ceil_div(Na, 128);
    g.tiles_b = ceil_div(Nb, 128);
    g.tiles_ab = tiles_a * g.tiles_b;
    g.tiles_a = tiles_a;
    { const char* e = getenv("TVTS_TN_AFAST"); g.a_fast = e ? atoi(e) : (tiles_a < g.tiles_b ? 1 : 0); }
    {
        const char* e = getenv("TVTS_TN_EARLY");
        const bool fits = (unsigned long long)M * (unsigned long long)(ldp > ldq ? ldp : ldq) * 2ull < (1ull << 32);
        g.early_dma = (fits && !(e && atoi(e) == 0)) ? 1 : 0;
    }
    // split the contraction over M into S ranges (range s lives on XCD s % 8, see the kernel).  S is chosen for
    // whole rounds of 2 blocks x 256 CUs: the smallest S reaching >= 93 % round efficiency, else the best one.
    int splits = 1;
    {
        double best = 0.0;
        for (int sp = 1; sp <= 64; ++sp) {
            if (sp > 1 && M / sp < 768) break;
            const long blocks = (long)g.tiles_ab * sp;
            const double eff = (double)blocks / (512.0 * (double)((blocks + 511) / 512));
            if (eff > best + 1e-9) { best = eff; splits = sp; }
            if (eff >= 0.93) { splits = sp; break; }
        }
    }
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
    if (!use_ws && !accumulate && splits > 1) {
        hipError_t e = hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)Nb * 4, Na, stream);
        if (e != hipSuccess) return (int)e;
    }
    g.atomic = (accumulate || splits > 1) ? 1 : 0;
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e2 != hipSuccess) return (int)e2;
    g.n_items = g.tiles_ab * splits;
    const int grid = ceil_div(g.n_items, 8) * 8;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(NTHREADS), 65536, stream, g);
    if (use_ws) {
        const long n4 = (long)Na * Nb / 4;
        int rb = (int)((n4 + 255) / 256);
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(rb), dim3(256), 0, stream, workspace, splits, Na, Nb, out, ldo, accumulate);
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

extern "C" int tvts_gemm_small_f32(const float* A, long sai, long sak, const float* B, long sbk, long sbj,
                                   int M, int N, int K, float alpha, const float* bias, float* C, long ldc,
                                   int accumulate, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(gemm_small_kernel, dim3(ceil_div(N, 16), ceil_div(M, 16)), dim3(256), 0, stream, A, sai,
                       sak, B, sbk, sbj, M, N, K, alpha, bias, C, ldc, accumulate);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ------------------------------------------------------------------------------------------------
// bias gradient: out[n] += sum_m X[m,n]  (bf16 in, fp32 atomics; 8 columns per lane, rows strided by blocks)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int ld, int M, int N,
                                                     float* __restrict__ out, int rows_per_block) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (col >= N) return;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    r1 = r1 < M ? r1 : M;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0; r < r1; ++r) {
        const bf16x8 v = *(const bf16x8*)(X + (size_t)r * ld + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(out + col + e, s[e]);
}

extern "C" int tvts_colsum_bf16(const void* X, int ld, int M, int N, float* out, hipStream_t stream) {
    if (M <= 0 || N <= 0 || N % 8 || ld % 8) return TVTS_EINVAL;
    const int rows_per_block = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 2048), ceil_div(M, rows_per_block)), dim3(256), 0, stream,
                       (const bf16*)X, ld, M, N, out, rows_per_block);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
