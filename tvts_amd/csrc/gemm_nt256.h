// 256x256-tile software-pipelined NT GEMM kernel (the production forward / dgrad kernel of the TVTSv2 step) as a header,
// so that the bench-only experiment library (csrc/exp/) can instantiate variants of the SAME source without any
// run-time switches living in the production kernel.  Included by gemm.hip.
#pragma once
#include "common.h"

#define BK 64

struct GemmNT {
    const bf16* A; int lda;
    const bf16* B; int ldb;
    int M, N, K;
    const float* bias;
    const float* residual; int ldr;
    int act;
    bf16* preact; int ldp;
    const bf16* gate_h; int ldh; int gate_act;
    int side_deriv;  // TVTS_GEMM_SIDE_DERIV: preact receives act'(x) instead of x / gate_h holds act'(x) and is multiplied as is
    void* out; int ldc; int out_f32;
    int tiles_m, tiles_n;
    int sa_rows; // fp8: scale_a holds one scale per row of A (per-token activation scales) instead of one for the tensor
    int gc;      // 256x256 pipelined kernel: tile columns per column group (0 = plain row-major tile order)
    const float* sa; const float* sb;  // fp8 operands: per-tensor scales (device scalars), out = sa*sb * (A B^T) + ...
    // stream-K (ABL & 524288): fp32 partial tiles [grid][2][256 x 256] and one arrival counter per output tile (zero between launches)
    float* sk_ws; int* sk_cnt; int sk_tol;
    // e4m3 copy of the (bf16) result under one scale per tensor, written by the epilogue (ABL & 1048576; BASELINE config 5: the GELU
    // output is the operand of the next layer's forward AND weight-gradient GEMMs, the gated input gradient that of the previous
    // layer's): q8[m, n] = e4m3(out[m, n] / q8_scale[0]), q8_amax = max(q8_amax, max |out|)
    unsigned char* q8; int ldq8; const float* q8_scale; float* q8_amax;
    // ABL & 4194304 (TVTS_GEMM_CLOCK_SAMPLE): block 0 stores {s_memtime, s_memrealtime} at its start and end here (4 x u64)
    unsigned long long* clk;
};

// ------------------------------------------------------------------------------------------------
// Stream-K (round 4): at the reference's own per-GPU batches (12 / 24 pairs: M = 9 420 / 18 840) an output of 111 ... 444
// tiles of 256 x 256 leaves the 256 persistent blocks 0.43 ... 0.87 of a round -- tile quantisation, not the K loop, is what
// the step loses there.  With ABL & 524288 the unit of work is a K STAGE, not a tile: XCD x still owns a contiguous range of
// tiles (its share of the weight / activation panels stays in that XCD's L2), and its P blocks split the range's
// range_n * nk stages evenly -- block b takes stages [sk_bound(b), sk_bound(b + 1)), i.e. the tail of one tile, whole tiles,
// and the head of another.  A piece that does not cover its tile's whole K leaves as an fp32 partial in the accumulator
// layout (slot 1 of the block if the piece starts the tile, slot 0 otherwise: a block has at most one of each), then the
// block counts itself in at the tile's arrival counter; the LAST block to arrive adds the partials IN CONTRIBUTOR ORDER
// (k ascending, every one re-read from the workspace, its own included) and runs the ordinary fused epilogue -- the order is
// a function of the shape and the grid only, so results are bit-reproducible whoever arrives last, and nobody ever waits
// for another block (no co-residency assumption, no deadlock with kernels of other streams).  All contributors of a tile
// sit on one XCD; the fences are agent-scope all the same, so correctness does not rest on the block -> XCD mapping.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int sk_bound(int U, int P, int b, int nk, int tol) {
    if (b <= 0) return 0;
    if (b >= P) return U;
    int u = (int)(((unsigned)U * (unsigned)b) / (unsigned)P);  // U * P < 2^31 (dispatcher)
    const int r = u % nk;  // a boundary within `tol` stages of a tile edge moves onto it: no partial for a sliver
    if (r <= tol) u -= r;
    else if (nk - r <= tol) u += nk - r;
    return u;
}
// the largest block index whose first stage is <= u
__device__ __forceinline__ int sk_owner(int U, int P, int nk, int tol, int u) {
    int b = (int)(((unsigned)u * (unsigned)P) / (unsigned)U);
    b = b < P - 1 ? b : P - 1;
    while (b + 1 < P && sk_bound(U, P, b + 1, nk, tol) <= u) ++b;
    while (b > 0 && sk_bound(U, P, b, nk, tol) > u) --b;
    return b;
}
__device__ __forceinline__ bf16x8 frag_rows128(const char* lds_tile, int row, int chunk) {
    return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
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
// the same four pieces from inline asm (ABL & 262144).  The builtin is a FLAT-class instruction that writes LDS: hipcc's wait
// bookkeeping marks it as touching both the vector-memory and the LDS counter ("pending flat"), after which every wait it
// inserts for an LDS read is a full lgkmcnt(0) -- in the K loop two of them per stage sit right behind a burst of four fragment
// reads whose data is only needed a whole MFMA group later.  Issued this way the compiler counts its own reads exactly
// (lgkmcnt(4)) and the waits for the DMA itself are the explicit ones at the stage barrier.  M0 carries the LDS destination
// (8 KiB between pieces); the offsets are 32-bit from a wave-uniform base.
__device__ __forceinline__ void stage_issue256_asm(const StageOff256& o, const bf16* ubase_, unsigned lds_addr) {
    const char* ubase = uniform_ptr(ubase_);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o.off[0]), "v"(o.off[1]), "v"(o.off[2]), "v"(o.off[3]), "s"(ubase), "s"(lds_addr) : "memory", "scc");
}
// A piece of tile `tile` ends here without covering the tile's whole K: it leaves as a partial, and the block counts itself in.
// Returns true (block-uniform) in the block that arrived last -- that block sums the pieces and runs the epilogue once its own
// stage range is done (sk_gather, behind the main loop: the accumulators of the loop are dead there, so the ordered sum does not
// compete with them for registers; inside the loop the very same code made hipcc spill 125 registers, fragments included).
__device__ __forceinline__ bool sk_publish(const GemmNT& g, const f32x4 (&acc)[4][8], int tile, int n_pieces, bool starts_tile,
                                           int wave, int lane, int* flag) {
    // Agent-scope (sc1) stores and loads on the partials themselves instead of a device-wide fence: a release fence is
    // buffer_wbl2 -- a write-back of the XCD's whole L2 -- and 512 of them per launch cost 150-200 us (the first build of this
    // path: 217 us for a 47 us GEMM).  An sc1 store is written through, an sc1 load is served coherently at agent scope
    // (the accesses LLVM's gfx942 memory model uses for agent-scope atomics), so `vmcnt(0)` behind the stores is the release.
    const char* mine = uniform_ptr(g.sk_ws + ((size_t)blockIdx.x * 2 + (starts_tile ? 1 : 0)) * 65536 + (size_t)wave * 8192);
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "v"(acc[j][i]), "s"(mine + (j * 8 + i) * 1024) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *(volatile int*)flag = __hip_atomic_fetch_add(g.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *(volatile int*)flag == n_pieces - 1;
}
// the ordered sum of the n pieces of a tile (k ascending; piece 0 starts the tile: slot 1 of its block, the others slot 0)
__device__ __forceinline__ void sk_gather(const GemmNT& g, f32x4 (&acc)[4][8], int first, int n, int xcd, int wave, int lane) {
    const unsigned voff = (unsigned)lane * 16u;
    for (int c = 0; c < n; ++c) {
        const char* p = uniform_ptr(g.sk_ws + ((size_t)((first + c) * 8 + xcd) * 2 + (c == 0 ? 1 : 0)) * 65536 + (size_t)wave * 8192);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[i]) : "v"(voff), "s"(p + (j * 8 + i) * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = c == 0 ? v[i] : acc[j][i] + v[i];
        }
    }
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

// ------------------------------------------------------------------------------------------------
// 256x256 tile, 512 threads = 8 waves as 2(M) x 4(N), wave tile 128x64, SOFTWARE-PIPELINED fragments.  A plain
// load-then-multiply 256x256 kernel runs its three phases -- LDS-DMA wait, 24 ds_read_b128 per wave, 64 MFMAs per
// wave -- back to back (round 1: MFMA alone 105 us, DMA + reads alone 95 us, epilogue 53 us, together 237 us at
// M 50240, N 2304, K 768).  Here the fragment registers are double-buffered so that the reads of the next
// half K-step are in flight while the matrix pipe works on the current one, and the DMA of stage s+2 is issued
// right after the barrier that frees its buffer, a full stage ahead of its consumer:
//     F1 <- ds_read k 32..63 (cur) | MFMA(F0) | vmcnt(0) lgkmcnt(0) barrier | DMA(s+2 -> cur) |
//     F0 <- ds_read k 0..31 (nxt)  | MFMA(F1) | [tile end: epilogue]
// Stages form one flat sequence over the block's persistent tile list.  LDS: 2 x 64 KiB stages + 8 x 4 KiB
// XOR-swizzled epilogue patches = 160 KiB exactly.
// ------------------------------------------------------------------------------------------------
// read-out half of the epilogue: the wave-private patch holds a 16-row x 64-column fp32 slab of the output tile (rows
// m_base .. m_base + 15, columns nb .. nb + 63; 16-B chunk c of row r stored at chunk c ^ r, bias already added); it leaves as
// whole row segments -- every store / residual load / gate load instruction covers full 128-B (bf16) or 256-B (fp32) pieces
// of output rows -- with the activation (+ pre-activation side output), the activation-gradient gate and the fp32 residual.
// 16-byte global store of a result that this kernel never reads again: non-temporal, so that the 32 MB of tile results the 256
// CUs write per round leave the L2s before the operand panels do.  Worth 1.5 % over plain stores on the step's shapes when the
// operands are NOT cache-resident from a previous launch (tools/gemm_ab.py rotates three buffer sets; with one set the same
// change reads as +20 % because the 231 MB activation operand then survives in the Infinity Cache between iterations -- an
// artefact the training step never sees).  sc1 (write-through) stores and staggered block starts change nothing: the store tail
// of a tile is bound by the CU's own store path (~13 B/clk/CU, 128 KB in ~4.8 us), not by chip-wide HBM bandwidth.
// ABL & 64 = plain stores, ABL & 32 = sc1 (experiment library only).
template <int ABL, typename V16>
__device__ __forceinline__ void store16(void* p, const V16& v) {
    static_assert(sizeof(V16) == 16, "16-byte vector");
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    if constexpr ((ABL & 32) != 0) {
        const u32x4 d = __builtin_bit_cast(u32x4, v);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    } else if constexpr ((ABL & 64) != 0) {
        *(V16*)p = v;
    } else {
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), (u32x4*)p);
    }
}
// side input of the epilogue (fp32 residual, bf16 pre-activation of the gate): read once; ABL & 128 = non-temporal load
template <int ABL, typename V>
__device__ __forceinline__ V side_load(const void* p) {
    if constexpr ((ABL & 128) != 0) return __builtin_nontemporal_load((const V*)p);
    else return *(const V*)p;
}

// side inputs of one 16-row slab in this lane's read-out layout (fp32 out: 4 x (row idx>>4, 4 columns); bf16 out: 2 x (row idx>>3,
// 8 columns)): fetched ONE SLAB AHEAD of their use so that their latency hides under the previous slab's LDS round trip and
// stores instead of serialising eight load -> compute -> store chains per tile
struct SideSlab {
    f32x4 res[4];   // fp32 residual
    bf16x8 gate[2]; // bf16 pre-activation of the gate (fp32-out path uses the halves)
};
template <int GATE, int ABL>
__device__ __forceinline__ void side_prefetch(const GemmNT& g, int m_base, int nb, int lane, SideSlab& s) {
    if (ABL & 2) return;
    if (g.out_f32) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
            int m = m_base + r;
            const int n = nb + c16 * 4;
            m = m < g.M ? m : g.M - 1;
            if (n >= g.N) continue;
            if (GATE == ACT_NONE && g.residual) s.res[t] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
            if (GATE != ACT_NONE) {
                const bf16x4 h = side_load<ABL, bf16x4>(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) s.gate[t >> 1][(t & 1) * 4 + e] = h[e];
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
            int m = m_base + r;
            const int n = nb + c8 * 8;
            m = m < g.M ? m : g.M - 1;
            if (n >= g.N) continue;
            if (GATE == ACT_NONE && g.residual) {
                s.res[2 * t] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
                s.res[2 * t + 1] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n + 4);
            }
            if (GATE != ACT_NONE) s.gate[t] = side_load<ABL, bf16x8>(g.gate_h + (size_t)m * g.ldh + n);
        }
    }
}

// ABL (compile-time, 0 in every production instantiation; csrc/exp instantiates the others to price the epilogue's parts):
// 2 = side inputs (residual / gate) are not loaded, 4 = results are not stored (one never-taken store keeps them live).
// ABL & 256 = side inputs come from `side` (prefetched one slab ahead) instead of being loaded here.
template <int ACT, int GATE, int ABL = 0>
__device__ __forceinline__ void patch_readout(const GemmNT& g, const char* patch, int m_base, int nb, int lane, const SideSlab& side,
                                              float q8inv = 0.f, float* q8am = nullptr) {
    constexpr bool PF = (ABL & 256) != 0;
    constexpr bool Q8OUT = (ABL & 1048576) != 0;
    if (g.out_f32) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
            f32x4 v = *(const f32x4*)(patch + r * 256 + ((c16 ^ r) << 4));
            const int m = m_base + r, n = nb + c16 * 4;
            if (m >= g.M || n >= g.N) continue;
            if (ACT != ACT_NONE) {
                f32x4 sd;
#pragma unroll
                for (int e = 0; e < 4; ++e) { float t; v[e] = act_fwd_side(v[e], ACT, g.side_deriv, t); sd[e] = t; }
                if (g.preact && (!(ABL & 4) || v[0] == 1.2345e33f)) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)sd[0], (bf16)sd[1], (bf16)sd[2], (bf16)sd[3]};
            }
            if (GATE != ACT_NONE) {
                const bf16x4 h = (ABL & 2) ? (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}
                               : PF ? (bf16x4){side.gate[t >> 1][(t & 1) * 4], side.gate[t >> 1][(t & 1) * 4 + 1], side.gate[t >> 1][(t & 1) * 4 + 2], side.gate[t >> 1][(t & 1) * 4 + 3]}
                                    : side_load<ABL, bf16x4>(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gate_apply(v[e], (float)h[e], GATE, g.side_deriv);
            }
            if (g.residual && !(ABL & 2)) v += (PF && GATE == ACT_NONE) ? side.res[t] : side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
            if (!(ABL & 4) || v[0] == 1.2345e33f) store16<ABL>((float*)g.out + (size_t)m * g.ldc + n, v);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
            const f32x4 v0 = *(const f32x4*)(patch + r * 256 + (((2 * c8) ^ r) << 4));
            const f32x4 v1 = *(const f32x4*)(patch + r * 256 + (((2 * c8 + 1) ^ r) << 4));
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int m = m_base + r, n = nb + c8 * 8;
            if (m >= g.M || n >= g.N) continue;
            if (ACT != ACT_NONE) {
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {  // pairs on the packed fp32 pipe
                    f32x2_ t;
                    const f32x2_ y = act_fwd_side_pk((f32x2_){v[e], v[e + 1]}, ACT, g.side_deriv, t);
                    v[e] = y[0]; v[e + 1] = y[1];
                    h[e] = (bf16)t[0]; h[e + 1] = (bf16)t[1];
                }
                if (g.preact && (!(ABL & 4) || v[0] == 1.2345e33f)) store16<ABL>(g.preact + (size_t)m * g.ldp + n, h);
            }
            if (GATE != ACT_NONE) {
                bf16x8 h;
                if (ABL & 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (bf16)v[e];
                } else {
                    h = PF ? side.gate[t] : side_load<ABL, bf16x8>(g.gate_h + (size_t)m * g.ldh + n);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gate_apply(v[e], (float)h[e], GATE, g.side_deriv);
            }
            if (g.residual && !(ABL & 2)) {
                const f32x4 r0 = (PF && GATE == ACT_NONE) ? side.res[2 * t] : side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
                const f32x4 r1 = (PF && GATE == ACT_NONE) ? side.res[2 * t + 1] : side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
            // (q8 forms: out == nullptr -> the e4m3 copy is the ONLY result -- in the per-tensor regime every consumer of the MLP's
            // GELU output / gated gradient multiplies the e4m3 bytes, the bf16 tensor would be written and never read)
            const bool q8_alone = Q8OUT && g.out == nullptr;
            if (!q8_alone) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = (bf16)v[e]; if (Q8OUT) v[e] = (float)o[e]; }  // (the copy quantises what the bf16 consumers see)
                if (!(ABL & 4) || v[0] == 1.2345e33f) store16<ABL>((bf16*)g.out + (size_t)m * g.ldc + n, o);
            }
            if constexpr (Q8OUT) {  // the 8 values as e4m3 bytes under the tensor's scale; alone, straight from the fp32 results
                float f[8];
                float am = *q8am;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float w = v[e];
                    am = fmaxf(am, fabsf(w));
                    f[e] = __builtin_amdgcn_fmed3f(w * q8inv, -448.0f, 448.0f);
                }
                *q8am = am;
                int p0 = 0, p1 = 0;
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], p0, false);
                p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p0, true);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], p1, false);
                p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], p1, true);
                typedef __attribute__((ext_vector_type(2))) int i32x2_;
                *(i32x2_*)(g.q8 + (size_t)m * g.ldq8 + n) = (i32x2_){p0, p1};
            }
        }
    }
}

// epilogue of a 128x64 wave sub-tile held as 8 x 4 accumulator tiles of v_mfma_f32_16x16x32 (lane: row m = lane & 15 of MFMA
// row-tile i, 4 consecutive n at 16 j + 4 (lane >> 4)): one 16-row slab per pass through the patch
template <int ACT, int GATE, int ABL = 0>
__device__ __forceinline__ void epilogue256_patch(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn,
                                                  int lane, char* patch, float scale = 1.0f,
                                                  const float* row_scale = nullptr) {
    if (ABL & 1) {  // no epilogue at all: the accumulators are consumed by a never-taken store and cleared
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s += acc[j][i][0] + acc[j][i][3]; acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (s == 1.2345e33f) *(float*)g.out = s;
        return;
    }
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nb + j * 16 + gq * 4;
        bias4[j] = (g.bias && n < g.N) ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    constexpr bool PF = (ABL & 256) != 0;
    constexpr bool Q8OUT = (ABL & 1048576) != 0;
    float q8inv = 0.f, q8am = 0.f;
    if constexpr (Q8OUT) {
        const float s8 = g.q8_scale[0];
        q8inv = s8 > 0.f ? 1.0f / s8 : 1.0f;
    }
    SideSlab side;  // ONE buffer: slab i + 1 is requested as soon as slab i has consumed it (a second buffer spills)
    const bool has_side = (g.residual != nullptr) || GATE != ACT_NONE;
    if (PF && has_side) side_prefetch<GATE, ABL>(g, m0 + wm * 128, nb, lane, side);
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
        // (an opaque copy of the lane id per slab: the read-out's lane-derived LDS / global offsets do not depend on the slab, so the
        // compiler hoists them out of this loop -- and in the register-heavy forms (e4m3 operands, GELU, e4m3 copies) SPILLS them; every
        // reload then waits vmcnt(0), i.e. for the previous slab's stores to be acknowledged: 8 store round trips per tile in series)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        patch_readout<ACT, GATE, ABL>(g, patch, m0 + wm * 128 + i * 16, nb, ln, side, q8inv, &q8am);
        if (PF && has_side && i + 1 < 8) side_prefetch<GATE, ABL>(g, m0 + wm * 128 + (i + 1) * 16, nb, lane, side);
    }
    if constexpr (Q8OUT) amax_publish(g.q8_amax, wave_max(q8am), lane);  // one conditional atomic per wave and tile
}

// ------------------------------------------------------------------------------------------------
// Register-path epilogue (ABL & 1024): no LDS round trip.  fp32 results leave in the accumulator layout itself (a store
// instruction covers 16 rows x 64 contiguous bytes).  bf16 results: v_permlane16_swap_b32 between the accumulator tiles of
// column pairs (j, j + 1) hands every lane 8 consecutive columns of its row -- lane (li, gq) ends up with columns
// 16 (2p + (gq & 1)) + 8 (gq >> 1) .. + 7 of pair p -- so that a store instruction again covers 16 rows x 64 contiguous bytes.
// Side inputs (residual / gate) are requested TWO row-tiles ahead and always BEFORE the stores of the current one: the
// memory counter retires in order, so a side load issued behind a store cannot be consumed before that store is acknowledged.
// hipcc waits vmcnt(0) for every ordinary load it tracks while an LDS-DMA is in flight (which is always, here), so on wave
// tiles that lie wholly inside the matrix (FULL) the side loads are inline asm and their waits are counted by hand:
//   issue order  L0 L1 | L2 S0 | L3 S1 | ... | L7 S5 | S6 | S7      (Lk / Sk = the NL loads / NS stores of row-tile k)
//   wait before consuming Lk:  k = 0: NL,  k = 1: NL + NS,  2 <= k <= 6: NL + 2 NS,  k = 7: 2 NS  operations may stay outstanding.
// The bias is already in the accumulators: the K loop requests it two stages before the tile ends and adds it during the last one.
// ------------------------------------------------------------------------------------------------
struct SideReg {
    f32x4 res[4];   // fp32 out: residual of column tile j; bf16 out: the two halves of pair p at [2p], [2p + 1]
    bf16x8 gate[2]; // bf16 out: pre-activation of pair p; fp32 out: column tiles 2p, 2p + 1 in the halves
};
// 16-byte load the compiler does not track: uniform base (SGPR pair) + per-lane 32-bit byte offset + immediate
template <int ABL, int IMM, typename V>
__device__ __forceinline__ void asm_load16(V& d, const void* base, unsigned voff) {
    static_assert(sizeof(V) == 16, "16-byte vector");
    if constexpr ((ABL & 128) != 0) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(d) : "v"(voff), "s"(base), "i"(IMM) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(base), "i"(IMM) : "memory");
}
// 16-byte non-temporal store the compiler does not track: uniform base + per-lane 32-bit byte offset (s_nop 1: the data registers
// of a 4-dword store may not be overwritten by the very next instruction)
template <typename V>
__device__ __forceinline__ void asm_store16(const V& d, void* base, unsigned voff) {
    static_assert(sizeof(V) == 16, "16-byte vector");
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(d), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vm(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vm(bf16x8& a, bf16x8& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

// side inputs of one row-tile, FULL wave tiles: inline-asm loads at byte offset voff (this lane's row and first column) from the
// side matrix's base; the column tiles / pairs are immediates
template <int GATE, int ABL, int CFG>
__device__ __forceinline__ void side_load_asm(const GemmNT& g, unsigned voff, SideReg& s) {
    constexpr bool out_f32 = (CFG & 1) != 0;
    if constexpr (GATE != ACT_NONE) {
        static_assert(!out_f32, "gate + fp32 output takes the tracked loads");
        asm_load16<ABL, 0>(s.gate[0], g.gate_h, voff);
        asm_load16<ABL, 64>(s.gate[1], g.gate_h, voff);
    } else if constexpr (out_f32) {
        asm_load16<ABL, 0>(s.res[0], g.residual, voff);
        asm_load16<ABL, 64>(s.res[1], g.residual, voff);
        asm_load16<ABL, 128>(s.res[2], g.residual, voff);
        asm_load16<ABL, 192>(s.res[3], g.residual, voff);
    } else {
        asm_load16<ABL, 0>(s.res[0], g.residual, voff);
        asm_load16<ABL, 16>(s.res[1], g.residual, voff);
        asm_load16<ABL, 128>(s.res[2], g.residual, voff);
        asm_load16<ABL, 144>(s.res[3], g.residual, voff);
    }
}
// any wave tile: compiler-tracked loads, rows / columns clamped
template <int GATE, int ABL, int CFG>
__device__ __forceinline__ void side_load_reg(const GemmNT& g, int m, int nb, int gq, SideReg& s) {
    constexpr bool out_f32 = (CFG & 1) != 0, has_res = (CFG & 2) != 0;
    m = m < g.M ? m : g.M - 1;
    if (out_f32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = nb + 16 * j + 4 * gq;
            n = n < g.N ? n : g.N - 4;  // clamped, not skipped: a predicated load would make every later use wait for it
            if (GATE == ACT_NONE && has_res) s.res[j] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
            if (GATE != ACT_NONE) {
                const bf16x4 h = side_load<ABL, bf16x4>(g.gate_h + (size_t)m * g.ldh + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) s.gate[j >> 1][(j & 1) * 4 + e] = h[e];
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int n = nb + 16 * (2 * p + (gq & 1)) + 8 * (gq >> 1);
            n = n < g.N ? n : g.N - 8;
            if (GATE == ACT_NONE && has_res) {
                s.res[2 * p] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n);
                s.res[2 * p + 1] = side_load<ABL, f32x4>(g.residual + (size_t)m * g.ldr + n + 4);
            }
            if (GATE != ACT_NONE) s.gate[p] = side_load<ABL, bf16x8>(g.gate_h + (size_t)m * g.ldh + n);
        }
    }
}

template <int I> struct IC { static constexpr int value = I; };

// FULL: the wave tile (128 rows x 64 columns) lies inside the matrix -- no clamps, no predicates, counted waits.
// CFG bit 0: fp32 output, bit 1: fp32 residual present.
template <int ACT, int GATE, int ABL, int CFG, bool FULL>
__device__ __forceinline__ void epilogue256_reg(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn, int lane) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    constexpr bool out_f32 = (CFG & 1) != 0, has_res = (CFG & 2) != 0;
    constexpr bool GATED = GATE != ACT_NONE;
    constexpr bool has_side = has_res || GATED;
    // asm loads + counted waits exist for: fp32 out + residual, bf16 out + residual, bf16 out + gate
    constexpr bool ASM = FULL && has_side && !(out_f32 && GATED);
    constexpr int NL = GATED ? 2 : 4;
    constexpr int NS = out_f32 ? 4 : 2;
    static_assert(!(has_side && ACT != ACT_NONE), "activation epilogues take no side input");
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    const int mb = m0 + wm * 128 + li;
    SideReg sA, sB;
    // ASM: byte offset of this lane's first side element (its row, its first column) and the step of 16 rows; < 4 GiB (dispatch)
    const int col0 = out_f32 ? 4 * gq : 16 * (gq & 1) + 8 * (gq >> 1);
    const unsigned esz = GATED ? 2u : 4u, lds_ = GATED ? (unsigned)g.ldh : (unsigned)g.ldr;
    unsigned voff = ((unsigned)mb * lds_ + (unsigned)(nb + col0)) * esz;
    const unsigned vstep = 16u * lds_ * esz;
    if (has_side) {
        if constexpr (ASM) {
            side_load_asm<GATE, ABL, CFG>(g, voff, sA);
            side_load_asm<GATE, ABL, CFG>(g, voff + vstep, sB);
            voff += 2 * vstep;
        }
    }
    auto slab = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = mb + i * 16;
        SideReg& s = (i & 1) ? sB : sA;
        if (has_side && !ASM) side_load_reg<GATE, ABL, CFG>(g, m, nb, gq, s);  // edge tiles: fetched in place (few tiles, few registers)
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[j][i];  // bias included (added inside the K loop)
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (ASM) {
            constexpr int N = i == 0 ? NL : i == 1 ? NL + NS : i == 7 ? 2 * NS : NL + 2 * NS;
            if constexpr (GATED) asm_wait_vm<N>(s.gate[0], s.gate[1]);
            else asm_wait_vm<N>(s.res[0], s.res[1], s.res[2], s.res[3]);
        }
        if constexpr (out_f32) {
            bf16x4 pre[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ACT != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        f32x2_ t;
                        const f32x2_ y = act_fwd_side_pk((f32x2_){v[j][e], v[j][e + 1]}, ACT, g.side_deriv, t);
                        v[j][e] = y[0]; v[j][e + 1] = y[1];
                        pre[j][e] = (bf16)t[0]; pre[j][e + 1] = (bf16)t[1];
                    }
                }
                if (GATED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = gate_apply(v[j][e], (float)s.gate[j >> 1][(j & 1) * 4 + e], GATE, g.side_deriv);
                }
                if (!GATED && has_res) v[j] += s.res[j];
            }
            if constexpr (ASM && i + 2 < 8) { side_load_asm<GATE, ABL, CFG>(g, voff, s); voff += vstep; }
            if (FULL || m < g.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nb + 16 * j + 4 * gq;
                    if (!FULL && n >= g.N) continue;
                    if (ACT != ACT_NONE && g.preact) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = pre[j];
                    store16<ABL>((float*)g.out + (size_t)m * g.ldc + n, v[j]);
                }
            }
        } else {
            // hand every lane 8 consecutive columns: swap row 1 / 3 (gq odd) of tile 2p with row 0 / 2 (gq even) of tile 2p + 1
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (element copies first: __builtin_bit_cast applied to a vector element reads element 0 whatever e is)
                    const float xa = v[2 * p][e], xb = v[2 * p + 1][e];
                    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    v[2 * p][e] = __uint_as_float(r0);
                    v[2 * p + 1][e] = __uint_as_float(r1);
                }
            bf16x8 pre[2], o[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float w[8] = {v[2 * p][0], v[2 * p][1], v[2 * p][2], v[2 * p][3], v[2 * p + 1][0], v[2 * p + 1][1], v[2 * p + 1][2], v[2 * p + 1][3]};
                if (ACT != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {  // pairs on the packed fp32 pipe
                        f32x2_ t;
                        const f32x2_ y = act_fwd_side_pk((f32x2_){w[e], w[e + 1]}, ACT, g.side_deriv, t);
                        w[e] = y[0]; w[e + 1] = y[1];
                        pre[p][e] = (bf16)t[0]; pre[p][e + 1] = (bf16)t[1];
                    }
                }
                if (GATED) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = gate_apply(w[e], (float)s.gate[p][e], GATE, g.side_deriv);
                }
                if (!GATED && has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w[e] += s.res[2 * p][e]; w[4 + e] += s.res[2 * p + 1][e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[p][e] = (bf16)w[e];
            }
            if constexpr (ASM && i + 2 < 8) { side_load_asm<GATE, ABL, CFG>(g, voff, s); voff += vstep; }
            if (FULL || m < g.M) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int n = nb + 16 * (2 * p + (gq & 1)) + 8 * (gq >> 1);
                    if (!FULL && n >= g.N) continue;
                    if (ACT != ACT_NONE && g.preact) store16<ABL>(g.preact + (size_t)m * g.ldp + n, pre[p]);
                    store16<ABL>((bf16*)g.out + (size_t)m * g.ldc + n, o[p]);
                }
            }
        }
    };
    slab(IC<0>{}); slab(IC<1>{}); slab(IC<2>{}); slab(IC<3>{});
    slab(IC<4>{}); slab(IC<5>{}); slab(IC<6>{}); slab(IC<7>{});
}

// ------------------------------------------------------------------------------------------------
// Patch epilogue with hand-counted side loads (ABL & 8192; FULL wave tiles of the residual / gate kernels).  The LDS patch keeps
// the stores and side loads at whole 128-B (bf16) / 256-B (fp32) row segments -- the register path's 64-B segments cost more
// than its missing LDS round trip returns -- and the side inputs of row-tile k + 2 are requested (inline asm: hipcc would wait
// vmcnt(0) for a tracked load while an LDS-DMA is in flight) BEFORE the stores of row-tile k, so that consuming them never waits
// for a store younger than two row-tiles (issue order and wait counts as in the register path: L0 L1 | L2 S0 | L3 S1 | ...).
// The bias is already in the accumulators.  CFG bit 0: fp32 output, bit 1: fp32 residual.
// ------------------------------------------------------------------------------------------------
template <int GATE, int ABL, int CFG, bool FULLT>
__device__ __forceinline__ void side_load_patch_asm(const GemmNT& g, int m_base, int r0, unsigned col_off, unsigned row_step, SideSlab& s) {
    constexpr bool out_f32 = (CFG & 1) != 0;
    // read-out layout: fp32 out t < 4: rows t * 4 + (lane >> 4), 4 columns; bf16 out t < 2: rows t * 8 + (lane >> 3), 8 columns.
    // FULLT: ONE per-lane offset, the row groups ride in the scalar base; edge tiles: every row clamped to the matrix on its own
    if constexpr (FULLT) {
        const unsigned voff = (unsigned)r0 * row_step + col_off;
        const char* b = (const char*)(GATE != ACT_NONE ? (const void*)g.gate_h : (const void*)g.residual) + (size_t)m_base * row_step;
        if constexpr (GATE != ACT_NONE) {
            asm_load16<ABL, 0>(s.gate[0], b, voff);
            asm_load16<ABL, 0>(s.gate[1], b + (size_t)8 * row_step, voff);
        } else if constexpr (out_f32) {
            asm_load16<ABL, 0>(s.res[0], b, voff);
            asm_load16<ABL, 0>(s.res[1], b + (size_t)4 * row_step, voff);
            asm_load16<ABL, 0>(s.res[2], b + (size_t)8 * row_step, voff);
            asm_load16<ABL, 0>(s.res[3], b + (size_t)12 * row_step, voff);
        } else {
            asm_load16<ABL, 0>(s.res[0], b, voff);
            asm_load16<ABL, 16>(s.res[1], b, voff);
            asm_load16<ABL, 0>(s.res[2], b + (size_t)8 * row_step, voff);
            asm_load16<ABL, 16>(s.res[3], b + (size_t)8 * row_step, voff);
        }
        return;
    }
    auto off = [&](int dr) -> unsigned {
        int m = m_base + r0 + dr;
        m = m < g.M ? m : g.M - 1;
        return (unsigned)m * row_step + col_off;
    };
    if constexpr (GATE != ACT_NONE) {
        static_assert(!out_f32, "gate + fp32 output takes the tracked loads");
        asm_load16<ABL, 0>(s.gate[0], g.gate_h, off(0));
        asm_load16<ABL, 0>(s.gate[1], g.gate_h, off(8));
    } else if constexpr (out_f32) {
        asm_load16<ABL, 0>(s.res[0], g.residual, off(0));
        asm_load16<ABL, 0>(s.res[1], g.residual, off(4));
        asm_load16<ABL, 0>(s.res[2], g.residual, off(8));
        asm_load16<ABL, 0>(s.res[3], g.residual, off(12));
    } else {
        const unsigned o0 = off(0), o8 = off(8);
        asm_load16<ABL, 0>(s.res[0], g.residual, o0);
        asm_load16<ABL, 16>(s.res[1], g.residual, o0);
        asm_load16<ABL, 0>(s.res[2], g.residual, o8);
        asm_load16<ABL, 16>(s.res[3], g.residual, o8);
    }
}
// FULLT: the wave tile lies inside the matrix (counted waits, no predicates); else rows clamped / stores predicated and every
// wait is vmcnt(0) (predicated stores may not issue, which breaks the counts)
template <int ACT, int GATE, int ABL, int CFG, bool FULLT>
__device__ __forceinline__ void epilogue256_patch_asm(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn, int lane,
                                                      char* patch) {
    constexpr bool out_f32 = (CFG & 1) != 0, has_res = (CFG & 2) != 0;
    constexpr bool GATED = GATE != ACT_NONE;
    constexpr bool SIDE = has_res || GATED;
    static_assert(!(has_res && GATED) && !(SIDE && ACT != ACT_NONE) && !(ACT != ACT_NONE && out_f32), "one side input at most; activation: bf16 out");
    constexpr int NL = GATED ? 2 : 4;
    constexpr int NS = out_f32 ? 4 : 2;
    // the lane-derived constants below are recomputed from an opaque copy of the lane id: sharing them with the K loop's would keep
    // them live across it, and the register allocator then spills them and reloads them here behind an s_waitcnt vmcnt(0)
    asm volatile("" : "+v"(lane));
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    const int mw = m0 + wm * 128;
    const unsigned esz = GATED ? 2u : 4u, lds_ = GATED ? (unsigned)g.ldh : (unsigned)g.ldr;
    const unsigned row_step = lds_ * esz;  // bytes per row of the side matrix (the matrix fits 4 GiB: dispatcher)
    const int r0 = out_f32 ? (lane >> 4) : (lane >> 3), c0 = out_f32 ? (lane & 15) * 4 : (lane & 7) * 8;
    const unsigned col_off = (unsigned)(nb + c0) * esz;
    const unsigned out_step = (unsigned)g.ldc * (out_f32 ? 4u : 2u);  // bytes per output row
    const unsigned out_voff = (unsigned)r0 * out_step + (unsigned)(nb + c0) * (out_f32 ? 4u : 2u);
    SideSlab sA, sB;
    if constexpr (FULLT && SIDE) {
        side_load_patch_asm<GATE, ABL, CFG, FULLT>(g, mw, r0, col_off, row_step, sA);
        side_load_patch_asm<GATE, ABL, CFG, FULLT>(g, mw + 16, r0, col_off, row_step, sB);
    }
    const unsigned pre_step = (unsigned)g.ldp * 2u;  // activation kernels: bytes per row of the pre-activation side output
    const unsigned pre_voff = (unsigned)r0 * pre_step + (unsigned)(nb + c0) * 2u;
    auto slab = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        SideSlab& s = (FULLT && (i & 1)) ? sB : sA;
        if constexpr (!FULLT && SIDE) side_load_patch_asm<GATE, ABL, CFG, FULLT>(g, mw + i * 16, r0, col_off, row_step, s);  // edge tiles: in place
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)(patch + li * 256 + (((j * 4 + gq) ^ li) << 4)) = acc[j][i];
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        constexpr int N = !FULLT ? 0 : i == 0 ? NL : i == 1 ? NL + NS : i == 7 ? 2 * NS : NL + 2 * NS;
        if constexpr (GATED) asm_wait_vm<N>(s.gate[0], s.gate[1]);
        else if constexpr (has_res) asm_wait_vm<N>(s.res[0], s.res[1], s.res[2], s.res[3]);
        const int m_base = mw + i * 16;
        if constexpr (out_f32) {
            f32x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
                v[t] = *(const f32x4*)(patch + r * 256 + ((c16 ^ r) << 4));
                if constexpr (has_res) v[t] += s.res[t];
            }
            if constexpr (FULLT && SIDE && i + 2 < 8) side_load_patch_asm<GATE, ABL, CFG, FULLT>(g, m_base + 32, r0, col_off, row_step, s);
            if constexpr (FULLT) {
                char* ob = (char*)g.out + (size_t)m_base * out_step;
#pragma unroll
                for (int t = 0; t < 4; ++t) asm_store16(v[t], ob + (size_t)(4 * t) * out_step, out_voff);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int idx = lane + 64 * t, r = idx >> 4, c16 = idx & 15;
                    if (m_base + r < g.M) store16<ABL>((float*)g.out + (size_t)(m_base + r) * g.ldc + nb + c16 * 4, v[t]);
                }
            }
        } else {
            bf16x8 o[2], pre[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                const f32x4 v0 = *(const f32x4*)(patch + r * 256 + (((2 * c8) ^ r) << 4));
                const f32x4 v1 = *(const f32x4*)(patch + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                float w[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if constexpr (ACT != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {  // pairs on the packed fp32 pipe
                        f32x2_ sd_;
                        const f32x2_ y = act_fwd_side_pk((f32x2_){w[e], w[e + 1]}, ACT, g.side_deriv, sd_);
                        w[e] = y[0]; w[e + 1] = y[1];
                        pre[t][e] = (bf16)sd_[0]; pre[t][e + 1] = (bf16)sd_[1];
                    }
                }
                if constexpr (GATED) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = gate_apply(w[e], (float)s.gate[t][e], GATE, g.side_deriv);
                } else if constexpr (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w[e] += s.res[2 * t][e]; w[4 + e] += s.res[2 * t + 1][e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[t][e] = (bf16)w[e];
            }
            if constexpr (FULLT && SIDE && i + 2 < 8) side_load_patch_asm<GATE, ABL, CFG, FULLT>(g, m_base + 32, r0, col_off, row_step, s);
            if constexpr (FULLT) {
                char* ob = (char*)g.out + (size_t)m_base * out_step;
                if (ACT != ACT_NONE && g.preact) {
                    char* pb = (char*)g.preact + (size_t)m_base * pre_step;
#pragma unroll
                    for (int t = 0; t < 2; ++t) asm_store16(pre[t], pb + (size_t)(8 * t) * pre_step, pre_voff);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) asm_store16(o[t], ob + (size_t)(8 * t) * out_step, out_voff);
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int idx = lane + 64 * t, r = idx >> 3, c8 = idx & 7;
                    if (m_base + r < g.M) {
                        if (ACT != ACT_NONE && g.preact) store16<ABL>(g.preact + (size_t)(m_base + r) * g.ldp + nb + c8 * 8, pre[t]);
                        store16<ABL>((bf16*)g.out + (size_t)(m_base + r) * g.ldc + nb + c8 * 8, o[t]);
                    }
                }
            }
        }
    };
    slab(IC<0>{}); slab(IC<1>{}); slab(IC<2>{}); slab(IC<3>{});
    slab(IC<4>{}); slab(IC<5>{}); slab(IC<6>{}); slab(IC<7>{});
}

// ------------------------------------------------------------------------------------------------
// bf16-FIRST patch epilogue (ABL & 8388608; round 6 experiment): for bf16 results with no side input (plain, activation + side output)
// the accumulators are finished -- activation, conversion -- in the accumulator layout, where a lane owns 4 consecutive columns, and go
// through the LDS patch as bf16: a 16-row x 64-column slab is 2 KiB instead of 4, so the 4 KiB patch holds TWO (plain: slab i + 1 is
// written while slab i is read out -- the write -> wait -> read -> wait chain of the fp32 patch runs eight times in series per wave;
// activation forms: result and side output of one slab), the write is one ds_write_b64 per accumulator tile instead of a b128, the
// read-out one ds_read_b128 per 16-byte store instead of two, and the activation of slab i + 1 does not depend on the patch at all.
// Same values, same single rounding as the fp32 patch.  8-byte granule g of row r sits at granule g ^ (r & 14): a lane's 16-byte
// read (granules 2c, 2c + 1 of its row) stays one aligned piece at chunk c ^ ((r & 14) >> 1).
// ------------------------------------------------------------------------------------------------
template <int ACT, int ABL, bool FULLT>
__device__ __forceinline__ void epilogue256_patch_b16(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn, int lane,
                                                      char* patch, const f32x4* bias4 /* nullptr: the bias is in the accumulators */) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    asm volatile("" : "+v"(lane));
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    const int mw = m0 + wm * 128;
    const int r0 = lane >> 3, c0 = (lane & 7) * 8;
    const unsigned out_step = (unsigned)g.ldc * 2u, pre_step = (unsigned)g.ldp * 2u;
    const unsigned out_voff = (unsigned)r0 * out_step + (unsigned)(nb + c0) * 2u;
    const unsigned pre_voff = (unsigned)r0 * pre_step + (unsigned)(nb + c0) * 2u;
    const unsigned wr_off = (unsigned)li * 128u;                       // + granule position * 8
    const unsigned swz = (unsigned)(li & 14);
    constexpr bool HAS_PRE = ACT != ACT_NONE;
    auto slab = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        char* half = patch + (HAS_PRE ? 0 : (i & 1) * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j][i];
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias4) v += bias4[j];
            const unsigned pos = wr_off + ((((unsigned)(j * 4 + gq)) ^ swz) << 3);
            if constexpr (HAS_PRE) {
                f32x2_ s0, s1;
                const f32x2_ y0 = act_fwd_side_pk((f32x2_){v[0], v[1]}, ACT, g.side_deriv, s0);
                const f32x2_ y1 = act_fwd_side_pk((f32x2_){v[2], v[3]}, ACT, g.side_deriv, s1);
                *(bf16x4_*)(half + pos) = (bf16x4_){(bf16)y0[0], (bf16)y0[1], (bf16)y1[0], (bf16)y1[1]};
                *(bf16x4_*)(half + 2048 + pos) = (bf16x4_){(bf16)s0[0], (bf16)s0[1], (bf16)s1[0], (bf16)s1[1]};
            } else {
                *(bf16x4_*)(half + pos) = (bf16x4_){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            }
        }
        const int m_base = mw + i * 16;
        bf16x8 o[2], pre[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = t * 8 + r0;
            const unsigned rd = (unsigned)r * 128u + ((((unsigned)(lane & 7)) ^ ((unsigned)(r & 14) >> 1)) << 4);
            o[t] = *(const bf16x8*)(half + rd);
            if constexpr (HAS_PRE) pre[t] = *(const bf16x8*)(half + 2048 + rd);
        }
        if constexpr (FULLT) {
            char* ob = (char*)g.out + (size_t)m_base * out_step;
            if (HAS_PRE && g.preact) {
                char* pb = (char*)g.preact + (size_t)m_base * pre_step;
#pragma unroll
                for (int t = 0; t < 2; ++t) asm_store16(pre[t], pb + (size_t)(8 * t) * pre_step, pre_voff);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) asm_store16(o[t], ob + (size_t)(8 * t) * out_step, out_voff);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = t * 8 + r0, n = nb + c0;
                if (m_base + r < g.M && n < g.N) {
                    if (HAS_PRE && g.preact) store16<ABL>(g.preact + (size_t)(m_base + r) * g.ldp + n, pre[t]);
                    store16<ABL>((bf16*)g.out + (size_t)(m_base + r) * g.ldc + n, o[t]);
                }
            }
        }
    };
    slab(IC<0>{}); slab(IC<1>{}); slab(IC<2>{}); slab(IC<3>{});
    slab(IC<4>{}); slab(IC<5>{}); slab(IC<6>{}); slab(IC<7>{});
}

// The bf16-first patch for the forms with a bf16 SIDE input (ABL & 16777216 on top of 8388608; GATE = bf16 residual add / activation-
// gradient gate): the side values are needed where the tile is rounded, i.e. in the ACCUMULATOR layout -- lane (li, gq) of accumulator
// tile (j, i) owns row 16 i + li, columns 16 j + 4 gq .. + 3: one 8-byte load, a wave instruction covers 16 rows x 32 bytes, the four
// column tiles of a slab the rows' whole 128-byte lines (the first one fetches them into L1).  Inline-asm loads two slabs ahead of their
// use and ahead of the stores of the slab in between, counted waits (issue order L0 L1 | L2 S0 | L3 S1 | ...), as in the fp32-patch path.
// The arithmetic is the fp32 patch's (gate_apply on the fp32 accumulator value, ONE rounding): the same bits.
// MEASURED SLOWER, NOT ADOPTED (experiment library only, profiles/r06_gemm_ab_b16_side_input_forms.txt, M = 150 720): bf16 residual add at
// N = K = 768 212 -> 227 us, at K = 3072 627 -> 640 us, the activation-gradient gate 788 -> 843 us -- four quarter-line 8-byte loads per
// slab cost more than the halved LDS round trip returns; these forms keep the fp32 patch with 16-byte row-segment side loads.
struct SideAcc { unsigned long long v[4]; };  // 4 x (4 bf16): the side values of column tiles j = 0..3 of one slab
template <int IMM>
__device__ __forceinline__ void asm_load8(unsigned long long& d, const void* base, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(base), "i"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vm8(SideAcc& s) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]) : "i"(N) : "memory");
}
template <int GATE, int ABL, bool FULLT>
__device__ __forceinline__ void epilogue256_patch_b16_side(const GemmNT& g, f32x4 (&acc)[4][8], int m0, int n0, int wm, int wn, int lane,
                                                           char* patch) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
    static_assert(GATE != ACT_NONE, "side-input form");
    asm volatile("" : "+v"(lane));
    const int nb = n0 + wn * 64;
    const int li = lane & 15, gq = lane >> 4;
    const int mw = m0 + wm * 128;
    const int r0 = lane >> 3, c0 = (lane & 7) * 8;
    const unsigned out_step = (unsigned)g.ldc * 2u, side_step = (unsigned)g.ldh * 2u;
    const unsigned out_voff = (unsigned)r0 * out_step + (unsigned)(nb + c0) * 2u;
    const unsigned side_voff = (unsigned)li * side_step + (unsigned)(nb + 4 * gq) * 2u;   // this lane's row and first column inside a slab
    const unsigned wr_off = (unsigned)li * 128u, swz = (unsigned)(li & 14);
    constexpr int NL = 4, NS = 2;
    SideAcc sA, sB;
    auto side_issue = [&](int slab_i, SideAcc& sd) {
        const char* b = (const char*)g.gate_h + (size_t)(mw + slab_i * 16) * side_step;
        if constexpr (FULLT) {
            asm_load8<0>(sd.v[0], b, side_voff); asm_load8<32>(sd.v[1], b, side_voff);
            asm_load8<64>(sd.v[2], b, side_voff); asm_load8<96>(sd.v[3], b, side_voff);
        } else {  // edge tiles: the row clamped to the matrix
            int m = mw + slab_i * 16 + li;
            m = m < g.M ? m : g.M - 1;
            const unsigned vo = (unsigned)(nb + 4 * gq) * 2u;
            const char* rb = (const char*)g.gate_h + (size_t)m * side_step;
            // (per-lane row base: the offset register carries it, the scalar base is the matrix)
            const unsigned long long d = (unsigned long long)(rb - (const char*)g.gate_h);
            const unsigned voff2 = (unsigned)d + vo;
            asm_load8<0>(sd.v[0], g.gate_h, voff2); asm_load8<32>(sd.v[1], g.gate_h, voff2);
            asm_load8<64>(sd.v[2], g.gate_h, voff2); asm_load8<96>(sd.v[3], g.gate_h, voff2);
        }
    };
    side_issue(0, sA);
    side_issue(1, sB);
    auto slab = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        SideAcc& sd = (i & 1) ? sB : sA;
        constexpr int N = !FULLT ? 0 : i == 0 ? NL : i == 1 ? NL + NS : i == 7 ? 2 * NS : NL + 2 * NS;
        asm_wait_vm8<N>(sd);
        char* half = patch + (i & 1) * 2048;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[j][i];  // bias included (added inside the K loop)
            acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16x4_ h = __builtin_bit_cast(bf16x4_, sd.v[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gate_apply(v[e], (float)h[e], GATE, g.side_deriv);
            *(bf16x4_*)(half + wr_off + ((((unsigned)(j * 4 + gq)) ^ swz) << 3)) = (bf16x4_){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
        if constexpr (i + 2 < 8) side_issue(i + 2, sd);   // the slab after next, ahead of this slab's stores
        const int m_base = mw + i * 16;
        bf16x8 o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = t * 8 + r0;
            o[t] = *(const bf16x8*)(half + (unsigned)r * 128u + ((((unsigned)(lane & 7)) ^ ((unsigned)(r & 14) >> 1)) << 4));
        }
        if constexpr (FULLT) {
            char* ob = (char*)g.out + (size_t)m_base * out_step;
#pragma unroll
            for (int t = 0; t < 2; ++t) asm_store16(o[t], ob + (size_t)(8 * t) * out_step, out_voff);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = t * 8 + r0;
                if (m_base + r < g.M) store16<ABL>((bf16*)g.out + (size_t)(m_base + r) * g.ldc + nb + c0, o[t]);
            }
        }
    };
    slab(IC<0>{}); slab(IC<1>{}); slab(IC<2>{}); slab(IC<3>{});
    slab(IC<4>{}); slab(IC<5>{}); slab(IC<6>{}); slab(IC<7>{});
    if constexpr (!FULLT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// timing ablations of the K loop for tools/gemm_cus.py (experiment builds only, -DTVTS_LOOP_ABL=n; results are wrong by construction):
// 1 no LDS-DMA behind the prologue, 2 no fragment reads behind the first stage, 4 no MFMAs (and with them no reads), 8 no barriers
#ifndef TVTS_LOOP_ABL
#define TVTS_LOOP_ABL 0
#endif
#define RAW_BARRIER_P()                       \
    do {                                      \
        asm volatile("" ::: "memory");        \
        if (!(TVTS_LOOP_ABL & 8)) __builtin_amdgcn_s_barrier(); \
        asm volatile("" ::: "memory");        \
    } while (0)

// CFG >= 0 (register-path epilogue): the output kind is compiled in -- bit 0 fp32 output, bit 1 fp32 residual -- so that the
// plain instantiations do not carry the residual's registers; CFG < 0: read from the arguments
template <int ACT, int GATE, bool FP8 = false, int ABL = 0, int CFG = -1>
__global__ __launch_bounds__(512, 2) void gemm_nt256p_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32K | B 32K] + 8 x 4K patches
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    char* patch = smem + 131072 + wave * 4096;

    if constexpr ((ABL & 4194304) != 0) {  // clock sample: shader cycles and constant-rate ticks at the block's first instruction
        if (blockIdx.x == 0 && tid == 0 && g.clk) { g.clk[0] = __builtin_amdgcn_s_memtime(); g.clk[1] = __builtin_amdgcn_s_memrealtime(); }
    }
    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    const int nk = g.K / BK;
    constexpr bool SK = (ABL & 524288) != 0;  // stream-K: this block owns stages [u0, u1) of the XCD's range_n * nk
    static_assert(!SK || (!FP8 && (ABL & (1024 | 8192)) == 0), "stream-K: bf16 operands, generic patch epilogue");
    int u0 = 0, total_st;
    if constexpr (SK) {
        u0 = sk_bound(range_n * nk, per_xcd, slot, nk, g.sk_tol);
        total_st = sk_bound(range_n * nk, per_xcd, slot + 1, nk, g.sk_tol) - u0;
        if (total_st <= 0) return;
    } else {
        if (slot >= range_n) return;
        total_st = ((range_n - slot + per_xcd - 1) / per_xcd) * nk;
    }
    // the tl-th tile of this block: stream-K walks the XCD's range contiguously, the tile-granular walk strides by the blocks per XCD
    auto tile_of = [&](int tl_) -> int { return SK ? range_lo + tl_ : range_lo + slot + tl_ * per_xcd; };
    const int gc = g.gc;
    if constexpr ((ABL & 24) != 0) {  // experiment: blocks start in 2 (8) / 4 (16) phases spread over one tile time (~1 us per stage)
        const int phases = (ABL & 16) ? 4 : 2;
        const int ph = slot % phases;
        for (int i = 0; i < ph * nk / (2 * phases); ++i) __builtin_amdgcn_s_sleep(127);  // 127 * 64 cycles ~ 3.9 us
    }

    // DMA cursor
    int i_st = 0, i_kt = SK ? u0 % nk : 0, i_tl = SK ? u0 / nk : 0, i_m0, i_n0;
    StageOff256 oa, ob;
    {
        tile_origin256(g, tile_of(i_tl), gc, i_m0, i_n0);
        stage_offsets256(oa, g.lda, i_m0, g.M - 1, wave, lane);
        stage_offsets256(ob, g.ldb, i_n0, g.N - 1, wave, lane);
    }
    constexpr bool MX = FP8 && (ABL & 65536) != 0;  // the K = 128 scaled-MFMA main loop below
    // MX: the lane id the once-per-tile code needs is re-read (mbcnt) instead of kept: lane-derived constants that stay live across
    // the K loop are the first thing the allocator spills in this 256-register kernel, and their reload sits behind an
    // s_waitcnt vmcnt(0) that waits for the LDS-DMA just issued
    auto fresh_lane = [&]() -> int {
        if constexpr (MX) {
            int l;  // (volatile asm: as builtins the two mbcnt are hoisted to the kernel's prologue and the RESULT is spilled)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            return l;
        } else {
            return lane;
        }
    };
    auto issue = [&]() {
        char* dst = smem + (i_st & 1) * 65536;
        if constexpr ((ABL & 262144) != 0) {
            const unsigned d = (unsigned)(size_t)(LDS_PTR(char))smem + (unsigned)(i_st & 1) * 65536u + (unsigned)wave * 1024u;
            stage_issue256_asm(oa, g.A + (size_t)i_m0 * g.lda + i_kt * BK, d);
            stage_issue256_asm(ob, g.B + (size_t)i_n0 * g.ldb + i_kt * BK, d + 32768u);
        } else {
            stage_issue256<0>(oa, g.A + (size_t)i_m0 * g.lda + i_kt * BK, dst, wave);
            stage_issue256<0>(ob, g.B + (size_t)i_n0 * g.ldb + i_kt * BK, dst + 32768, wave);
        }
        ++i_st;
        if (++i_kt == nk) {
            i_kt = 0; ++i_tl;
            tile_origin256(g, tile_of(i_tl), gc, i_m0, i_n0);
            const int ln = fresh_lane();
            stage_offsets256(oa, g.lda, i_m0, g.M - 1, wave, ln);
            stage_offsets256(ob, g.ldb, i_n0, g.N - 1, wave, ln);
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
    int kt = SK ? u0 % nk : 0, tl = SK ? u0 / nk : 0, m0, n0;
    int seg_k0 = kt;  // stream-K: the K stage the current piece began at
    int fin_tl[2] = {-1, -1}, fin_first[2] = {0, 0}, fin_n[2] = {0, 0};  // tiles (at most two) this block arrived last at
    int young_stores = 0;
    bool trace_next = false;
    f32x4 bias4[4];  // register-path epilogue only
    tile_origin256(g, tile_of(tl), gc, m0, n0);
    const int arow = wm * 128 + (lane & 15), brow = wn * 64 + (lane & 15), gq = lane >> 4;
    // fragment registers: two A half-sets (4 MFMA row-tiles each) and two B sets, refilled while the matrix pipe
    // works on the other one
    bf16x8 aF[2][4], bF[2][4];
    bool frag_rd = true;  // TVTS_LOOP_ABL & 2: fragment reads only up to the first stage
#define LOAD_A(dst, buf, ks, h)                                                                        \
    if (frag_rd) _Pragma("unroll") for (int i = 0; i < 4; ++i) dst[i] = frag_rows128(buf, arow + ((h) * 4 + i) * 16, (ks) * 4 + gq)
#define LOAD_B(dst, buf, ks)                                                                           \
    if (frag_rd) _Pragma("unroll") for (int j = 0; j < 4; ++j) dst[j] = frag_rows128((buf) + 32768, brow + j * 16, (ks) * 4 + gq)
    // FP8: the operands are e4m3 matrices addressed as bf16 matrices of half the width (the staging and the LDS image are
    // byte-identical); a 16-byte fragment then holds 16 k-values of its row and feeds two 16x16x32 fp8 MFMAs (its low and
    // its high 8 bytes -- A and B use the same split, so every k meets its partner).
    typedef __attribute__((ext_vector_type(2))) long i64x2;
#define MFMA16(av, bv, h)                                                                              \
    if (!(TVTS_LOOP_ABL & 4)) _Pragma("unroll") for (int i = 0; i < 4; ++i)                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
            if (FP8) {                                                                                 \
                const i64x2 a8 = __builtin_bit_cast(i64x2, av[i]), b8 = __builtin_bit_cast(i64x2, bv[j]); \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b8[0], a8[0], acc[j][(h) * 4 + i], 0, 0, 0); \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b8[1], a8[1], acc[j][(h) * 4 + i], 0, 0, 0); \
            } else {                                                                                   \
                acc[j][(h) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[j], av[i], acc[j][(h) * 4 + i], 0, 0, 0); \
            }                                                                                          \
        }
    // FP8 at the fp8 rate (ABL & 65536): v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales -- the only fp8 MFMA form of
    // gfx950 that issues at twice the bf16 rate.  A lane's operand is 32 bytes of K: exactly the two 16-byte pieces of its row
    // that the two k-steps of a 128-byte stage read (A and B take the same two pieces, so every k meets its partner).  One MFMA
    // then covers a whole stage for a (row-tile, column-tile) pair: 32 MFMAs per stage in four chunks of 2 row-tiles x 4
    // column-tiles; the A pairs are double-buffered across chunks, the four B operands are re-read IN PLACE for the next stage
    // behind the last MFMAs that use them (column-tile-major order in the last chunk) -- 64 fragment registers like the bf16 loop.
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    i32x8 aQ[2][2], bQ[4];
#define LDQ(buf, row)                                                                                   \
    __builtin_shufflevector(__builtin_bit_cast(i32x4, frag_rows128(buf, row, gq)),                      \
                            __builtin_bit_cast(i32x4, frag_rows128(buf, row, 4 + gq)), 0, 1, 2, 3, 4, 5, 6, 7)
#define LOAD_A2(dst, buf, p)                                                                            \
    if (frag_rd) _Pragma("unroll") for (int t = 0; t < 2; ++t) dst[t] = LDQ(buf, arow + ((p) * 2 + t) * 16)
    // the MFMA as volatile inline asm: left to the compiler (the builtin), every MFMA of the stage is sunk behind the stage's last
    // branch -- their results are only read by the epilogue -- which makes all eight A operands live at once (189 spilled
    // registers).  cbsz = blgp = 0: both operands e4m3; the scale register holds E8M0 127 (= 1.0) in every byte.  The compiler
    // does not know these are MFMAs: the wait between the tile's last MFMA and the epilogue's reads is explicit below.
    int mx_one = 0x7F7F7F7F;
    asm volatile("" : "+v"(mx_one));
#define MXMFMA(bv, av, c)                                                                               \
    if (!(TVTS_LOOP_ABL & 4)) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(bv), "v"(av), "v"(mx_one))
#define MFMA8(av, p)                                                                                    \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) MXMFMA(bQ[j], av[t], acc[j][(p) * 2 + t])
    if constexpr (MX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bQ[j] = LDQ(smem + 32768, brow + j * 16);
        LOAD_A2(aQ[0], smem, 0);
    } else {
        LOAD_B(bF[0], smem, 0);
        LOAD_A(aF[0], smem, 0, 0);
    }

    for (int st = 0; st < total_st; ++st) {
        if ((TVTS_LOOP_ABL & 2) && st == 1) frag_rd = false;
        const char* cur = smem + (st & 1) * 65536;
        const char* nxt = smem + ((st + 1) & 1) * 65536;
        if constexpr (MX) {
            LOAD_A2(aQ[1], cur, 1);
            MFMA8(aQ[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            LOAD_A2(aQ[0], cur, 2);
            MFMA8(aQ[1], 1);
            __builtin_amdgcn_sched_barrier(0);
            LOAD_A2(aQ[1], cur, 3);
            MFMA8(aQ[0], 2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            RAW_BARRIER_P();
            const bool do_issue = i_st < total_st && !(TVTS_LOOP_ABL & 1);
            if (do_issue && wave < 4) issue();
            // the next stage's first fragments: under this stage's last MFMAs, or -- across a tile boundary -- behind the epilogue
            // (48 registers that would otherwise stay live across it)
            const bool more = st + 1 < total_st;
            const bool pre = more && kt + 1 != nk;
            if (pre) LOAD_A2(aQ[0], nxt, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                MXMFMA(bQ[j], aQ[1][0], acc[j][6]);
                MXMFMA(bQ[j], aQ[1][1], acc[j][7]);
                __builtin_amdgcn_sched_barrier(0);
                if (pre && frag_rd) bQ[j] = LDQ(nxt + 32768, brow + j * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_issue && wave >= 4) issue();
            if (++kt == nk) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (16 passes) before the epilogue reads them
                const int le = fresh_lane();
                // (ONE call: the epilogue is inlined, and two calls doubled the kernel's code -- the per-row and the per-tensor scale
                // regimes differ in two scalars)
                epilogue256_patch<ACT, GATE, ABL>(g, acc, m0, n0, wm, wn, le, patch, g.sa_rows ? g.sb[0] : g.sa[0] * g.sb[0],
                                                  g.sa_rows ? g.sa : nullptr);
                kt = 0; ++tl;
                tile_origin256(g, tile_of(tl), gc, m0, n0);
                if (more) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (frag_rd) bQ[j] = LDQ(nxt + 32768, brow + j * 16);
                    LOAD_A2(aQ[0], nxt, 0);
                }
            }
            continue;
        }
        // ABL & 131072: pin the written order of fragment reads and MFMA groups.  Left to itself hipcc sinks every ds_read_b128 down
        // to just in front of the MFMA group that consumes it and waits lgkmcnt(0) there (round 3, from the ISA of the round-2 build:
        // four read bursts + full waits per stage with the matrix pipe idle behind each) -- the fragment double-buffering written
        // here only exists in the source.  With the order pinned the reads of group G + 1 are in flight under group G's MFMAs and
        // the compiler's own counted lgkmcnt(4 / 8) in front of each group is all that is left.
#define SB() do { if constexpr ((ABL & 131072) != 0) __builtin_amdgcn_sched_barrier(0); } while (0)
        LOAD_A(aF[1], cur, 0, 1);
        SB();
        MFMA16(aF[0], bF[0], 0);
        SB();
        LOAD_B(bF[1], cur, 1);
        LOAD_A(aF[0], cur, 1, 0);
        SB();
        MFMA16(aF[1], bF[0], 1);
        SB();
        LOAD_A(aF[1], cur, 1, 1);
        SB();
        MFMA16(aF[0], bF[1], 0);
        SB();
        if constexpr ((ABL & 512) != 0) {
            // the stage this wait is for was requested BEFORE the previous tile's epilogue; the counter retires in order, so
            // leaving as many operations outstanding as the epilogue's tail issued stores still proves the stage has landed
            if (young_stores == 32) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
            else if (young_stores == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (young_stores == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (young_stores == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            young_stores = 0;
        } else {
            if constexpr ((ABL & 262144) != 0) __builtin_amdgcn_s_waitcnt(0x0070);
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        RAW_BARRIER_P();
        if constexpr ((ABL & 2048) != 0) {  // first barrier behind an epilogue: every wave of the block has finished its stores' issue
            if (trace_next) {
                if (wave == 0 && lane == 0 && tl - 1 < 32) {
                    unsigned long long* tr = (unsigned long long*)g.sa + ((size_t)blockIdx.x * 32 + tl - 1) * 6;
                    tr[4] = __builtin_amdgcn_s_memrealtime(); tr[5] = __builtin_amdgcn_s_memtime();
                }
                trace_next = false;
            }
        }
        if constexpr ((ABL & (1024 | 8192)) != 0) {
            // the tile's bias slice (this lane's 4 x 4 columns), requested two stages before the tile ends and AHEAD of this
            // step's DMA, so that the next step's vmcnt wait retires it and the epilogue starts without a memory wait
            if (kt == nk - 2) {
                if (g.bias) {
                    // (the lane's column group from a fresh mbcnt: kept live across the K loop it is spilled by the register-heavy
                    // epilogue forms, and its reload here waits vmcnt(0) -- for the LDS-DMA in flight)
                    int lb;  // (volatile asm: as builtins the two mbcnt are hoisted to the kernel's prologue and the RESULT is spilled)
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lb));
                    const int gqb = lb >> 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int n = n0 + wn * 64 + j * 16 + gqb * 4;
                        n = n < g.N ? n : g.N - 4;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(g.bias + n) : "memory");
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            // ... and added into the accumulators one stage later (it has landed: the vmcnt wait above), under the tile's last
            // MFMAs -- the matrix pipe keeps accumulating on top, and the epilogue holds no bias registers
            if (kt == nk - 1) {
                asm volatile("" : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[j][i] += bias4[j];
            }
        }
        // stage st+2 goes into the buffer every wave has just finished reading.  ABL & 32768: the two waves of a SIMD (w, w + 4) do
        // not issue their 8 LDS-DMA pieces at the same time (both would sit in ~100 cycles of VMEM issue per piece with the matrix
        // pipe idle): waves 0..3 issue here, waves 4..7 behind the stage's last MFMA group
        const bool do_issue = i_st < total_st && !(TVTS_LOOP_ABL & 1);
        if (do_issue && ((ABL & 32768) == 0 || wave < 4)) issue();
        // register-path epilogue: the next tile's first fragments are read behind the epilogue instead of across it (32 registers)
        const bool defer_frag = ((ABL & (1024 | 8192)) != 0 && kt + 1 == nk) || (SK && (kt + 1 == nk || st + 1 == total_st));
        if (st + 1 < total_st && !defer_frag) {
            LOAD_B(bF[0], nxt, 0);
            LOAD_A(aF[0], nxt, 0, 0);
        }
        SB();
        MFMA16(aF[1], bF[1], 1);
        SB();
        if (do_issue && (ABL & 32768) != 0 && wave >= 4) issue();
        ++kt;
        bool run_epi = kt == nk;
        if constexpr (SK) {
            // a piece ends with its tile or with the block's stage range; one that does not cover the tile's whole K leaves as a
            // partial -- the block that arrives last at a tile notes it and finishes it behind the loop
            const bool seg_end = kt == nk || st + 1 == total_st;
            if (seg_end && (seg_k0 != 0 || kt != nk)) {
                run_epi = false;
                const int U = range_n * nk;
                const int first = sk_owner(U, per_xcd, nk, g.sk_tol, tl * nk), last = sk_owner(U, per_xcd, nk, g.sk_tol, (tl + 1) * nk - 1);
                if (sk_publish(g, acc, tile_of(tl), last - first + 1, seg_k0 == 0, wave, lane, (int*)(smem + 131072))) {
                    if (seg_k0 != 0) { fin_tl[0] = tl; fin_first[0] = first; fin_n[0] = last - first + 1; }
                    else { fin_tl[1] = tl; fin_first[1] = first; fin_n[1] = last - first + 1; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                kt = 0; ++tl;
                tile_origin256(g, tile_of(tl), gc, m0, n0);
                if (st + 1 < total_st) {
                    LOAD_B(bF[0], nxt, 0);
                    LOAD_A(aF[0], nxt, 0, 0);
                }
            }
            if (seg_end) seg_k0 = 0;
        }
        if (run_epi) {
            if constexpr ((ABL & 2048) != 0) {  // experiment library: per-tile time stamps of wave 0 (g.sa carries the trace buffer)
                if (wave == 0 && lane == 0 && tl < 32) {
                    unsigned long long* tr = (unsigned long long*)g.sa + ((size_t)blockIdx.x * 32 + tl) * 6;
                    tr[0] = __builtin_amdgcn_s_memrealtime(); tr[1] = __builtin_amdgcn_s_memtime();
                }
            }
            if constexpr ((ABL & 8192) != 0) {
                static_assert(!FP8 && CFG >= 0, "hand-scheduled patch epilogue: bf16 operands, output kind compiled in");
                // (the 256-tile dispatch guarantees N % 256 == 0 for this instantiation: only rows can stick out)
                const bool full = m0 + wm * 128 + 128 <= g.M;
                if constexpr ((ABL & 8388608) != 0 && CFG == 0 && GATE == ACT_NONE) {  // bf16 result, no side input: bf16-first patch
                    if (full) epilogue256_patch_b16<ACT, ABL, true>(g, acc, m0, n0, wm, wn, lane, patch, nullptr);
                    else epilogue256_patch_b16<ACT, ABL, false>(g, acc, m0, n0, wm, wn, lane, patch, nullptr);
                } else if constexpr ((ABL & 16777216) != 0 && CFG == 0 && GATE != ACT_NONE && ACT == ACT_NONE) {  // ... with a bf16 side input
                    if (full) epilogue256_patch_b16_side<GATE, ABL, true>(g, acc, m0, n0, wm, wn, lane, patch);
                    else epilogue256_patch_b16_side<GATE, ABL, false>(g, acc, m0, n0, wm, wn, lane, patch);
                } else if (full) epilogue256_patch_asm<ACT, GATE, ABL, CFG, true>(g, acc, m0, n0, wm, wn, lane, patch);
                else epilogue256_patch_asm<ACT, GATE, ABL, CFG, false>(g, acc, m0, n0, wm, wn, lane, patch);
            } else if constexpr ((ABL & 1024) != 0) {
                static_assert(!FP8 && CFG >= 0, "register-path epilogue: bf16 operands, output kind compiled in");
                const bool full = (m0 + wm * 128 + 128 <= g.M) && (n0 + wn * 64 + 64 <= g.N);
                if (full) epilogue256_reg<ACT, GATE, ABL, CFG, true>(g, acc, m0, n0, wm, wn, lane);
                else epilogue256_reg<ACT, GATE, ABL, CFG, false>(g, acc, m0, n0, wm, wn, lane);
            } else {
                if constexpr (FP8) epilogue256_patch<ACT, GATE, ABL>(g, acc, m0, n0, wm, wn, lane, patch, g.sa_rows ? g.sb[0] : g.sa[0] * g.sb[0],
                                                                     g.sa_rows ? g.sa : nullptr);
                else if constexpr ((ABL & 8388608) != 0 && GATE == ACT_NONE && !SK) {
                    if (!g.out_f32 && !g.residual && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (!g.preact || (g.ldp & 7) == 0)) {
                        const int nbw = n0 + wn * 64, gqw = lane >> 4;
                        f32x4 b4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int n = nbw + j * 16 + gqw * 4;
                            b4[j] = (g.bias && n < g.N) ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        const bool full = (m0 + wm * 128 + 128 <= g.M) && (nbw + 64 <= g.N);
                        if (full) epilogue256_patch_b16<ACT, ABL, true>(g, acc, m0, n0, wm, wn, lane, patch, b4);
                        else epilogue256_patch_b16<ACT, ABL, false>(g, acc, m0, n0, wm, wn, lane, patch, b4);
                    } else {
                        epilogue256_patch<ACT, GATE, ABL>(g, acc, m0, n0, wm, wn, lane, patch, 1.0f);
                    }
                }
                else epilogue256_patch<ACT, GATE, ABL>(g, acc, m0, n0, wm, wn, lane, patch, 1.0f);
            }
            if constexpr ((ABL & 2048) != 0) {
                if (wave == 0 && lane == 0 && tl < 32) {
                    unsigned long long* tr = (unsigned long long*)g.sa + ((size_t)blockIdx.x * 32 + tl) * 6;
                    tr[2] = __builtin_amdgcn_s_memrealtime(); tr[3] = __builtin_amdgcn_s_memtime();
                }
                trace_next = true;
            }
            if constexpr ((ABL & 512) != 0) {
                // a lower bound of the memory operations the epilogue issued: its stores (every row / column of the wave tile inside
                // the matrix, else rows were skipped and the bound does not hold)
                const bool full = (m0 + wm * 128 + 128 <= g.M) && (n0 + wn * 64 + 64 <= g.N);
                const int per_slab = (g.out_f32 ? 4 : 2) * ((ACT != ACT_NONE && g.preact) ? 2 : 1);
                young_stores = full ? (per_slab >= 4 ? 32 : 16) : 0;
            }
            kt = 0; ++tl;
            tile_origin256(g, tile_of(tl), gc, m0, n0);
            if (((ABL & (1024 | 8192)) != 0 || SK) && st + 1 < total_st) {
                LOAD_B(bF[0], nxt, 0);
                LOAD_A(aF[0], nxt, 0, 0);
            }
        }
    }
    if constexpr ((ABL & 4194304) != 0) {  // ... and behind its last tile
        if (blockIdx.x == 0 && tid == 0 && g.clk) { g.clk[2] = __builtin_amdgcn_s_memtime(); g.clk[3] = __builtin_amdgcn_s_memrealtime(); }
    }
    if constexpr (SK) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            if (fin_tl[w] < 0) continue;  // block-uniform
            if (threadIdx.x == 0) g.sk_cnt[tile_of(fin_tl[w])] = 0;  // every piece has arrived: zero again for the next launch
            sk_gather(g, acc, fin_first[w], fin_n[w], xcd, wave, lane);
            tile_origin256(g, tile_of(fin_tl[w]), gc, m0, n0);
            epilogue256_patch<ACT, GATE, ABL>(g, acc, m0, n0, wm, wn, lane, patch, 1.0f);
        }
    }
#undef SB
#undef LOAD_A
#undef LOAD_B
#undef MFMA16
#undef LDQ
#undef LOAD_A2
#undef MXMFMA
#undef MFMA8
}
