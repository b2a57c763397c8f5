// The SMALL-BATCH form of the 128-column NT kernel (round 4): one block of 8 waves per CU, a ring of four 64-deep stages in the LDS.
//
// Why: at the reference's 2 - 24 pairs per GPU many GEMMs of the step have fewer 128 x 128 tiles than the chip has CUs (the whole text
// tower: 48 - 190 tiles), and the double-buffered 128 kernel (gemm_nt_kernel: ONE stage in flight per block, built for two blocks
// per CU) then runs one block per CU at the memory latency, 1.1 - 1.3 us per 64-deep stage where the stage's MFMAs need 0.21
// (profiles/r04_base_timeline_b12.txt: fc2 of the text tower, M = 1536, K = 2048, 41 us = 32 stages x 1.2 us + ramp).  Here three
// stages are in flight per block: counted vmcnt waits (the LDS-DMA leaves from inline asm, so hipcc does not park a vmcnt(0) in
// front of the fragment reads), one barrier per stage, fragments double-buffered across the two k-steps of a stage, two waves per
// SIMD whose DMA issue is staggered, the stages of the block's tiles as ONE sequence (the next tile's first stages fly under the
// epilogue), and an epilogue whose bias / residual / gate inputs are requested 3 stages ahead by inline asm and retired by count -- a
// tracked load would wait for every DMA in flight.  Tile rows 128 or 192 (RM = 2 / 3), whichever needs fewer one-block-per-CU rounds.
//
// What it reaches, and why not more (profiles/r04_gemm_ring.txt, profiles/r04_cu_intake_probe.txt): 0.65 - 0.70 us per stage of a
// 128 x 128 tile, hot or cold, three or four stages in flight, 0.55 us with the MFMAs switched off.  A CU alone can take 64 B/clk from
// L2 with this very access pattern, and the bare stage loop (pieces, counted waits, barrier) runs 0.27 - 0.36 us per stage on 48 CUs:
// the kernel's stage is the fragment reads + MFMAs (660 - 800 clocks per wave) plus the ISSUE of the wave's four LDS-DMA pieces
// (90 - 140 clocks each beside the partner wave's MFMAs), which a wave cannot overlap with its own MFMAs.  With all 256 CUs on cold
// panels the chip delivers 14.5 TB/s past L2 (56 GB/s per CU): 0.55 - 0.6 us per 32 KiB stage whatever the kernel does.  So the ring
// brings ONE block per CU to the rate the old kernel needs TWO co-resident blocks for, which is what the small batches lack; where
// the 128 kernel fills its 512 slots (the N = 768 GEMMs of the video tower at 12 pairs: 444 tiles) or the 256 kernel its 256 (24
// pairs), they stay.  Dedicated loader waves would be the next step.
//
// Same tile walk (XCD x owns a contiguous tile range, n fastest), same k order and the same epilogue arithmetic as
// gemm_nt_kernel: bit-identical results in every epilogue form (tests/test_kernels_gpu.py::test_gemm_nt_ring_*; the erf-GELU forms since
// their fp contraction is pinned by hand in common.h -- left to hipcc, the two kernels disagreed by one bf16 ulp in a few elements per million).
#pragma once

template <int RM> struct RingOff { unsigned a[RM], b[2]; };
// byte offsets of the two 16-B chunks this lane fetches per operand and stage, from (first tile row, k0): piece t covers rows
// (8t + wave) * 8 .. + 7, 16-B chunk c of row r lands at chunk c ^ (r & 7) (frag_rows128 undoes it); rows past the end are clamped
template <int RM>
__device__ __forceinline__ void ring_offsets(RingOff<RM>& o, const GemmNT& g, int m0, int n0, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < RM; ++t) {
        const int row = (t * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        int ga = m0 + row;
        ga = (ga < g.M ? ga : g.M - 1) - m0;
        o.a[t] = (unsigned)ga * (unsigned)g.lda * 2u + (unsigned)chunk * 16u;
        if (t < 2) {
            int gb = n0 + row;
            gb = (gb < g.N ? gb : g.N - 1) - n0;
            o.b[t] = (unsigned)gb * (unsigned)g.ldb * 2u + (unsigned)chunk * 16u;
        }
    }
}
// 2 / 3 / 4 LDS-DMA pieces of one operand (8 KiB apart in the LDS: eight waves x 1 KiB per piece), M0 = the LDS destination
#define RING_PIECE(n) "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %" #n ", %[b]\n\t"
__device__ __forceinline__ void ring_issue(const unsigned (&off)[2], const char* ubase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %[b]\n\t" RING_PIECE(2) "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off[0]), "v"(off[1]), [b] "s"(ubase), [d] "s"(lds_addr) : "memory", "scc");
}
__device__ __forceinline__ void ring_issue(const unsigned (&off)[3], const char* ubase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %[b]\n\t" RING_PIECE(2) RING_PIECE(3) "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off[0]), "v"(off[1]), "v"(off[2]), [b] "s"(ubase), [d] "s"(lds_addr) : "memory", "scc");
}
__device__ __forceinline__ void ring_issue(const unsigned (&off)[4], const char* ubase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %[b]\n\t" RING_PIECE(2) RING_PIECE(3) RING_PIECE(4) "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), [b] "s"(ubase), [d] "s"(lds_addr) : "memory", "scc");
}
#undef RING_PIECE
// at most n vector-memory operations of this wave outstanding (the counter retires in order)
__device__ __forceinline__ void ring_wait(int n) {
#define RW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : n) {
        RW(0) RW(1) RW(2) RW(3) RW(4) RW(5) RW(6) RW(7) RW(8) RW(9) RW(10) RW(11) RW(12) RW(13) RW(14) RW(15) RW(16) RW(17) RW(18) RW(19)
        RW(20) RW(21) RW(22) RW(23) RW(24) RW(25) RW(26) RW(27) RW(28) RW(29) RW(30) RW(31) RW(32) RW(33) RW(34) RW(35) RW(36) RW(37) RW(38) RW(39)
        default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    }
#undef RW
}
__device__ __forceinline__ void ring_load8(bf16x4& d, const void* base, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void ring_load16(f32x4& d, const void* base, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}

// RM: 16-row tiles per wave -> the block's tile is (64 RM) rows x 128 columns: 128 / 192 / 256 rows, whichever wastes least of the
// one-block-per-CU rounds (launch_nt_ring)
template <int ACT, int GATE, int RM, int NS>
__global__ __launch_bounds__(512, 1) void gemm_nt_ring_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS stages][A 8 RM K | B 16K]
    constexpr int PW = RM + 2;  // DMA pieces per wave and stage
    constexpr int RBM = 64 * RM, A_BYTES = RBM * 128, ST_BYTES = A_BYTES + 16384;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // wave tile: 16 RM rows x 64 columns

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    if (slot >= range_n) return;
    const int n_mine = (range_n - slot + per_xcd - 1) / per_xcd;
    const int nk = g.K / BK;
    const int total_st = n_mine * nk;
    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem + (unsigned)wave * 1024u;
    auto tile_org = [&](int tl, int& m0, int& n0) {
        const int tile = range_lo + slot + tl * per_xcd;
        m0 = (tile / g.tiles_n) * RBM;
        n0 = (tile % g.tiles_n) * BN;
    };

    // ---- the issue side: its own cursor, NS - 1 stages ahead of the MFMAs
    int i_st = 0, i_kt = 0, i_tl = 0, i_slot = 0, i_m0, i_n0;
    RingOff<RM> io;
    tile_org(0, i_m0, i_n0);
    ring_offsets(io, g, i_m0, i_n0, wave, lane);
    auto issue = [&]() {
        const unsigned d = lds0 + (unsigned)i_slot * (unsigned)ST_BYTES;
        ring_issue(io.a, uniform_ptr(g.A + (size_t)i_m0 * g.lda + i_kt * BK), d);
        ring_issue(io.b, uniform_ptr(g.B + (size_t)i_n0 * g.ldb + i_kt * BK), d + (unsigned)A_BYTES);
        ++i_st;
        if (++i_slot == NS) i_slot = 0;
        if (++i_kt == nk) {
            i_kt = 0;
            ++i_tl;
            if (i_st < total_st) {
                tile_org(i_tl, i_m0, i_n0);
                ring_offsets(io, g, i_m0, i_n0, wave, lane);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (i_st < total_st) issue();

    const int arow = wm * (16 * RM) + (lane & 15), brow = wn * 64 + (lane & 15), gq = lane >> 4;
    bf16x8 aF[2][RM], bF[2][4];
#define RING_LOAD(buf, h)                                                                                               \
    do {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < RM; ++i) aF[h][i] = frag_rows128(buf, arow + i * 16, (h) * 4 + gq);        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bF[h][j] = frag_rows128((buf) + A_BYTES, brow + j * 16, (h) * 4 + gq); \
    } while (0)
#define RING_MFMA(h)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        _Pragma("unroll") for (int i = 0; i < RM; ++i)                                                  \
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bF[h][j], aF[h][i], acc[j][i], 0, 0, 0)
#define RING_SB() __builtin_amdgcn_sched_barrier(0)
#define RING_BARRIER()                           \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_s_barrier();            \
        asm volatile("" ::: "memory");           \
    } while (0)

    ring_wait((i_st - 1) * PW);
    RING_BARRIER();
    RING_LOAD(smem, 0);

    f32x4 acc[4][RM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < RM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kt = 0, tl = 0, c_slot = 0, m0, n0;
    tile_org(0, m0, n0);

    // ---- epilogue inputs in this lane's accumulator layout (row m = .. + i * 16 + (lane & 15), 4 columns n = .. + j * 16 + gq * 4),
    // requested SIDE_LEAD stages before the tile ends
    constexpr bool GATED = GATE != ACT_NONE;
    constexpr int SIDE_LEAD = 3;
    f32x4 bias_r[4], res_r[4][RM];
    bf16x4 gate_r[4][RM];
    const bool has_bias = g.bias != nullptr, has_res = !GATED && g.residual != nullptr;
    const bool has_side = has_bias || has_res || GATED;
    const int side_loads = (has_bias ? 4 : 0) + ((has_res || GATED) ? 4 * RM : 0);
    int side_mark = 0;      // the side loads are younger than the DMA of every stage below this one ...
    int side_young = 0;     // ... while they are in flight
    int young_stores = 0;   // stores of the last epilogue (exact for interior tiles, else 0: a conservative wait) ...
    int store_mark = 0;     // ... which are younger than the DMA of every stage below this one
    auto side_request = [&]() {
        const int mb = m0 + wm * (16 * RM) + (lane & 15), nb = n0 + wn * 64 + gq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = nb + j * 16;
            n = n < g.N ? n : g.N - 4;
            if (has_bias) ring_load16(bias_r[j], g.bias, (unsigned)n * 4u);
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                int m = mb + i * 16;
                m = m < g.M ? m : g.M - 1;
                if (GATED) ring_load8(gate_r[j][i], g.gate_h, ((unsigned)m * (unsigned)g.ldh + (unsigned)n) * 2u);
                else if (has_res) ring_load16(res_r[j][i], g.residual, ((unsigned)m * (unsigned)g.ldr + (unsigned)n) * 4u);
            }
        }
        side_mark = i_st;
        side_young = side_loads;
    };
    const int side_kt = nk > SIDE_LEAD ? nk - SIDE_LEAD : 0;

    for (int st = 0; st < total_st; ++st) {
        const char* cur = smem + c_slot * ST_BYTES;
        if (++c_slot == NS) c_slot = 0;
        const char* nxt = smem + c_slot * ST_BYTES;
        RING_LOAD(cur, 1);
        RING_SB();
        RING_MFMA(0);
        RING_SB();
        const bool more = st + 1 < total_st;
        // stage st + 1 has landed (this wave's pieces): younger are the DMA of the stages behind it and -- while stage st + 1 left
        // before them -- the last epilogue's stores and the side loads in flight
        if (more)
            ring_wait((i_st - (st + 2)) * PW + (st + 1 < store_mark ? young_stores : 0) + (st + 1 < side_mark ? side_young : 0));
        RING_BARRIER();  // ... everybody's have, and everybody is done with the slot of stage st - 1
        if (kt == side_kt && has_side) side_request();
        const bool do_issue = i_st < total_st;
        if (do_issue && wave < 4) issue();  // stage st + NS - 1 -> that slot
        RING_LOAD(nxt, 0);  // (behind the last stage: a slot nobody needs; unconditional keeps hipcc's lgkmcnt counts exact)
        RING_SB();
        RING_MFMA(1);
        RING_SB();
        if (do_issue && wave >= 4) issue();  // the second wave of each SIMD issues under the first one's MFMAs
        if (++kt == nk) {
            if (has_side) {  // the side inputs have landed: younger are only the DMA pieces issued since
                ring_wait((i_st - side_mark) * PW);
                side_young = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (has_bias) asm volatile("" : "+v"(bias_r[j]));
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        if (GATED) asm volatile("" : "+v"(gate_r[j][i]));
                        else if (has_res) asm volatile("" : "+v"(res_r[j][i]));
                    }
                }
            }
            const bool interior = m0 + RBM <= g.M && n0 + BN <= g.N;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int m = m0 + wm * (16 * RM) + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + gq * 4;
                    const bool ok = m < g.M && n < g.N;
                    f32x4 v = acc[j][i];
                    if (has_bias) v += bias_r[j];
                    if (ACT != ACT_NONE) {
                        f32x4 sd;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float t; v[e] = act_fwd_side(v[e], ACT, g.side_deriv, t); sd[e] = t; }
                        if (g.preact && ok) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)sd[0], (bf16)sd[1], (bf16)sd[2], (bf16)sd[3]};
                    }
                    if (GATED) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gate_apply(v[e], (float)gate_r[j][i][e], GATE, g.side_deriv);
                    }
                    if (has_res) v += res_r[j][i];
                    if (ok) {
                        if (g.out_f32) {
                            *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
                        } else {
                            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                            *(bf16x4*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
                        }
                    }
                    acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            young_stores = interior ? (ACT != ACT_NONE && g.preact ? 8 * RM : 4 * RM) : 0;
            store_mark = i_st;
            kt = 0;
            ++tl;
            if (more) tile_org(tl, m0, n0);
        }
    }
#undef RING_LOAD
#undef RING_MFMA
#undef RING_SB
#undef RING_BARRIER
}

// ------------------------------------------------------------------------------------------------
// The same ring with DEDICATED LOADER WAVES (12 waves: 8 multiply, 4 issue every LDS-DMA piece of a stage and do the counted waits).
// The s_memtime trace of the kernel above shows a wave spending 350 - 570 clocks per stage inside the issue of its four pieces (the
// CU's texture path takes 16 clocks per 1 KiB piece, 32 pieces per stage, and the issuing wave stalls in that queue) -- time it
// cannot give to its MFMAs.  Here the multiplying waves never touch the vector-memory queue inside the K loop (their vmcnt counts
// only their own epilogue loads and stores), the loaders stall in it harmlessly.  Same LDS image, slots, barriers (one per stage, all
// 12 waves), k order and epilogue: the same bits.  0.58 us per stage against 0.68 (text fc2 at 12 pairs 27.4 -> 23.6 us, text qkv
// 11.0 -> 9.2; profiles/r04_gemm_ring.txt); what is left is LDS traffic -- 96 KiB of fragment reads + 32 KiB of DMA writes per stage
// of a 128 x 128 tile on eight 32 x 64 wave tiles.  Taken for 128 tile rows; at 192 rows the multiplying waves need more than the 168
// registers a 12-wave block leaves them (hipcc spills an inline-asm load's destination BEFORE its data arrives: wrong results, seen
// in the bench's bit comparison), so that height keeps the 8-wave kernel above.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ring_issue1(unsigned off, const char* ubase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ubase), "s"(lds_addr) : "memory");
}
template <int ACT, int GATE, int RM, int NS>
__global__ __launch_bounds__(768, 1) void gemm_nt_ringl_kernel(GemmNT g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NS stages][A 8 RM K | B 16K]
    constexpr int RBM = 64 * RM, A_BYTES = RBM * 128, ST_BYTES = A_BYTES + 16384;
    constexpr int PA = RBM / 32, PL = PA + 4;  // A pieces / all pieces per loader wave and stage (a piece = 8 rows = 1 KiB)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = total >> 3, rem = total & 7;
    const int range_lo = xcd * q + (xcd < rem ? xcd : rem);
    const int range_n = q + (xcd < rem ? 1 : 0);
    if (slot >= range_n) return;
    const int n_mine = (range_n - slot + per_xcd - 1) / per_xcd;
    const int nk = g.K / BK;
    const int total_st = n_mine * nk;
    auto tile_org = [&](int tl, int& m0, int& n0) {
        const int tile = range_lo + slot + tl * per_xcd;
        m0 = (tile / g.tiles_n) * RBM;
        n0 = (tile % g.tiles_n) * BN;
    };
#define RING_BARRIER()                           \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_s_barrier();            \
        asm volatile("" ::: "memory");           \
    } while (0)

    if (wave >= 8) {
        // ---------------- loader wave l: pieces l, l + 4, l + 8, ... of A and of B, NS - 1 stages ahead
        const int l = wave - 8;
        const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem + (unsigned)l * 1024u;
        int i_st = 0, i_kt = 0, i_tl = 0, i_slot = 0, i_m0, i_n0;
        unsigned oa[PA], ob[4];
        auto offsets = [&]() {
#pragma unroll
            for (int t = 0; t < PA; ++t) {
                const int row = (t * 4 + l) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ (row & 7);
                int ga = i_m0 + row;
                ga = (ga < g.M ? ga : g.M - 1) - i_m0;
                oa[t] = (unsigned)ga * (unsigned)g.lda * 2u + (unsigned)chunk * 16u;
                if (t < 4) {
                    int gb = i_n0 + row;
                    gb = (gb < g.N ? gb : g.N - 1) - i_n0;
                    ob[t] = (unsigned)gb * (unsigned)g.ldb * 2u + (unsigned)chunk * 16u;
                }
            }
        };
        tile_org(0, i_m0, i_n0);
        offsets();
        auto issue = [&]() {
            const unsigned d = lds0 + (unsigned)i_slot * (unsigned)ST_BYTES;
            const char* ua = uniform_ptr(g.A + (size_t)i_m0 * g.lda + i_kt * BK);
            const char* ub = uniform_ptr(g.B + (size_t)i_n0 * g.ldb + i_kt * BK);
#pragma unroll
            for (int t = 0; t < PA; ++t) ring_issue1(oa[t], ua, d + (unsigned)t * 4096u);
#pragma unroll
            for (int t = 0; t < 4; ++t) ring_issue1(ob[t], ub, d + (unsigned)A_BYTES + (unsigned)t * 4096u);
            ++i_st;
            if (++i_slot == NS) i_slot = 0;
            if (++i_kt == nk) {
                i_kt = 0;
                ++i_tl;
                if (i_st < total_st) {
                    tile_org(i_tl, i_m0, i_n0);
                    offsets();
                }
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (i_st < total_st) issue();
        ring_wait((i_st - 1) * PL);
        RING_BARRIER();
        for (int st = 0; st < total_st; ++st) {
            if (st + 1 < total_st) ring_wait((i_st - (st + 2)) * PL);  // stage st + 1 has landed (this wave's pieces)
            RING_BARRIER();                                             // ... everybody's have; the slot of stage st - 1 is free
            if (i_st < total_st) issue();                               // stage st + NS - 1 -> that slot
        }
        return;
    }

    // ---------------- multiplying waves
    const int wm = wave >> 1, wn = wave & 1;  // wave tile: 16 RM rows x 64 columns
    const int arow = wm * (16 * RM) + (lane & 15), brow = wn * 64 + (lane & 15), gq = lane >> 4;
    bf16x8 aF[2][RM], bF[2][4];
#define RING_LOAD(buf, h)                                                                                               \
    do {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < RM; ++i) aF[h][i] = frag_rows128(buf, arow + i * 16, (h) * 4 + gq);        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bF[h][j] = frag_rows128((buf) + A_BYTES, brow + j * 16, (h) * 4 + gq); \
    } while (0)
#define RING_MFMA(h)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        _Pragma("unroll") for (int i = 0; i < RM; ++i)                                                  \
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bF[h][j], aF[h][i], acc[j][i], 0, 0, 0)
#define RING_SB() __builtin_amdgcn_sched_barrier(0)

    RING_BARRIER();
    RING_LOAD(smem, 0);

    f32x4 acc[4][RM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < RM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kt = 0, tl = 0, c_slot = 0, m0, n0;
    tile_org(0, m0, n0);

    // epilogue inputs in this lane's accumulator layout, requested SIDE_LEAD stages before the tile ends (inline asm: the compiler
    // would park its wait right behind a tracked load, in front of the stages' MFMAs)
    constexpr bool GATED = GATE != ACT_NONE;
    constexpr int SIDE_LEAD = 3;
    f32x4 bias_r[4], res_r[4][RM];
    bf16x4 gate_r[4][RM];
    const bool has_bias = g.bias != nullptr, has_res = !GATED && g.residual != nullptr;
    const bool has_side = has_bias || has_res || GATED;
    auto side_request = [&]() {
        const int mb = m0 + wm * (16 * RM) + (lane & 15), nb = n0 + wn * 64 + gq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = nb + j * 16;
            n = n < g.N ? n : g.N - 4;
            if (has_bias) ring_load16(bias_r[j], g.bias, (unsigned)n * 4u);
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                int m = mb + i * 16;
                m = m < g.M ? m : g.M - 1;
                if (GATED) ring_load8(gate_r[j][i], g.gate_h, ((unsigned)m * (unsigned)g.ldh + (unsigned)n) * 2u);
                else if (has_res) ring_load16(res_r[j][i], g.residual, ((unsigned)m * (unsigned)g.ldr + (unsigned)n) * 4u);
            }
        }
    };
    const int side_kt = nk > SIDE_LEAD ? nk - SIDE_LEAD : 0;

    for (int st = 0; st < total_st; ++st) {
        const char* cur = smem + c_slot * ST_BYTES;
        if (++c_slot == NS) c_slot = 0;
        const char* nxt = smem + c_slot * ST_BYTES;
        RING_LOAD(cur, 1);
        RING_SB();
        RING_MFMA(0);
        RING_SB();
        RING_BARRIER();  // stage st + 1 has landed (the loaders waited for it), and everybody is done with the slot of stage st - 1
        if (kt == side_kt && has_side) side_request();
        RING_LOAD(nxt, 0);  // (behind the last stage: a slot nobody needs; unconditional keeps hipcc's lgkmcnt counts exact)
        RING_SB();
        RING_MFMA(1);
        RING_SB();
        if (++kt == nk) {
            if (has_side) {  // this wave's vector-memory queue holds its side loads (and older stores) only
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (has_bias) asm volatile("" : "+v"(bias_r[j]));
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        if (GATED) asm volatile("" : "+v"(gate_r[j][i]));
                        else if (has_res) asm volatile("" : "+v"(res_r[j][i]));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int m = m0 + wm * (16 * RM) + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + gq * 4;
                    const bool ok = m < g.M && n < g.N;
                    f32x4 v = acc[j][i];
                    if (has_bias) v += bias_r[j];
                    if (ACT != ACT_NONE) {
                        f32x4 sd;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float t; v[e] = act_fwd_side(v[e], ACT, g.side_deriv, t); sd[e] = t; }
                        if (g.preact && ok) *(bf16x4*)(g.preact + (size_t)m * g.ldp + n) = (bf16x4){(bf16)sd[0], (bf16)sd[1], (bf16)sd[2], (bf16)sd[3]};
                    }
                    if (GATED) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gate_apply(v[e], (float)gate_r[j][i][e], GATE, g.side_deriv);
                    }
                    if (has_res) v += res_r[j][i];
                    if (ok) {
                        if (g.out_f32) {
                            *(f32x4*)((float*)g.out + (size_t)m * g.ldc + n) = v;
                        } else {
                            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                            *(bf16x4*)((bf16*)g.out + (size_t)m * g.ldc + n) = o;
                        }
                    }
                    acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            kt = 0;
            ++tl;
            if (st + 1 < total_st) tile_org(tl, m0, n0);
        }
    }
#undef RING_LOAD
#undef RING_MFMA
#undef RING_SB
#undef RING_BARRIER
}
