// 256x256-tile TN kernel (weight gradients of the TVTSv2 step: C[Na,Nb] = sum_m P[m,Na] Q[m,Nb]) with the NT kernel's
// pipeline: 512 threads = 8 waves as 2 (a) x 4 (b), wave tile 128 (a) x 64 (b) = 8 x 4 accumulator tiles of
// v_mfma_f32_16x16x32_bf16, 64-row (m) stages of 64 KiB [P 2 x 16 KiB | Q 2 x 16 KiB] in two LDS buffers, fragment registers
// double-buffered so that the transposing reads (ds_read_b64_tr_b16) of the next 16-MFMA group are in flight under the current
// one, and the LDS-DMA of stage s+2 issued right behind the barrier that frees its buffer.  Against the 128x128 kernel
// (gemm.hip) a block moves half the operand bytes per flop through L2 -> LDS and reads half the LDS bytes per MFMA.
// The DMA is issued from inline asm: hipcc parks an s_waitcnt vmcnt(0) in front of every transposing read that follows the
// LDS-DMA builtin.  One (m-range, tile) work item per block; the partials go to the workspace (or fp32 atomics).
// Included by gemm.hip (GemmTN, frag layout and the dispatcher live there).
#pragma once

// per-lane byte offset (inside a [64 m][128 col] LDS tile, u = 0, half = 0) of the transposing read of column block ct
__device__ __forceinline__ unsigned tn_frag_off(int ct, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int row = g * 4 + (i >> 2);
    return (unsigned)(row * 256 + ((ct ^ (row & 7)) << 5) + (i & 3) * 8);
}
// the 8 k-slots of MFMA k-step u (32 rows of m) for the column block whose lane offset is `off`: rows u*32 + half*16 + ...
#ifndef TN_STAGGER_DMA
#define TN_STAGGER_DMA 1
#endif
#ifndef TN_ABL
#define TN_ABL 0   // timing ablations, tools/dbg/tn_abl.sh (wrong results): 1 = no LDS-DMA behind the prologue, 2 = one ds_read_b128 per fragment, 4 = L2-hot DMA (every stage re-reads the first)
#endif
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, unsigned off, int u) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    if (TN_ABL & 2) return *(const bf16x8*)(tile + ((off + u * 8192) & ~15u));
    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off + u * 8192));
    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off + u * 8192 + 4096));
    const s16x8 both = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}
// zero the k-slots of a fragment whose m (inside the stage) is >= valid (tail stage of an m-range)
__device__ __forceinline__ void tn_mask(bf16x8& f, int u, int valid, int lane) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int gq = lane >> 4;
    u32x4 w = __builtin_bit_cast(u32x4, f);
#pragma unroll
    for (int d = 0; d < 4; ++d) {  // dword d = k-slots 2d, 2d + 1: m = u*32 + (d >> 1)*16 + gq*4 + (d & 1)*2 + {0, 1}
        const int m = u * 32 + (d >> 1) * 16 + gq * 4 + (d & 1) * 2;
        const unsigned keep = (m < valid ? 0x0000ffffu : 0u) | (m + 1 < valid ? 0xffff0000u : 0u);
        w[d] &= keep;
    }
    f = __builtin_bit_cast(bf16x8, w);
}

// LDS-DMA piece q of a stage (q = 0..7): LDS tile q >> 1 (0, 1: the two 128-column halves of P; 2, 3: of Q), rows
// (q & 1) * 32 + wave * 4 .. + 3 (4 rows x 256 B per wave-issue).  16-B piece s16 of LDS row `row` holds the source columns
// chunk32 * 16 + (s16 & 1) * 8 .. with chunk32 = (s16 >> 1) ^ (row & 7); row & 7 does not depend on q, so inside the matrix
// the 8 source addresses of a lane differ by constants: 32 rows (folded into the scalar base) and 128 columns (an immediate).
// (the instruction's immediate offset moves the LDS destination as well as the source: M0 carries the destination minus it)
#ifndef TN_IMM_COMP
#define TN_IMM_COMP 1
#endif
template <int IMM>
__device__ __forceinline__ void glds16_asm_imm(unsigned voff, const char* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr - (unsigned)(TN_IMM_COMP * IMM)), "i"(IMM) : "memory");
}
// general form (tiles / stages that stick out of the matrix): every address clamped on its own
__device__ __forceinline__ void tn_stage_issue_clamped(const GemmTN& g, int a0, int b0, int m_stage, const char* baseP, const char* baseQ,
                                                       unsigned lds_stage, int wave, int lane) {
    asm volatile("" : "+v"(lane));  // opaque: this rare path recomputes its lane constants instead of keeping them live in the K loop
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int tile = q >> 1;
        const int r0 = ((q & 1) * 8 + wave) * 4;
        const int row = r0 + (lane >> 4);
        const int s16 = lane & 15;
        const int chunk32 = (s16 >> 1) ^ (row & 7);
        const bool isP = tile < 2;
        const int c0 = (isP ? a0 : b0) + (tile & 1) * 128;
        const int cmax = (isP ? g.Na : g.Nb) - 8;
        int col = c0 + chunk32 * 16 + (s16 & 1) * 8;
        col = col < cmax ? col : cmax;
        int gm = m_stage + row;
        gm = gm < g.M - 1 ? gm : g.M - 1;  // rows past the matrix are read clamped and masked out of the product
        const unsigned off = ((unsigned)(gm - m_stage) * (unsigned)(isP ? g.ldp : g.ldq) + (unsigned)col) * 2u;
        glds16_asm(off, isP ? baseP : baseQ, lds_stage + (unsigned)tile * 16384u + (unsigned)r0 * 256u);
    }
}

__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(GemmTN g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][P 2 x 16K | Q 2 x 16K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 2, wb = wave & 3;

    // work items = (m-range, output tile) in range-major order; XCD x takes the x-th contiguous eighth (one m-range's rows
    // of P / Q are pulled into one L2 and shared there by the tiles of that range)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int item = xcd * per + jx;
    if (item >= g.n_items) return;
    const int split = item / g.tiles_ab;
    const int t = item % g.tiles_ab;
    const int ta = g.a_fast ? t % g.tiles_a : t / g.tiles_b, tb = g.a_fast ? t / g.tiles_a : t % g.tiles_b;
    const int a0 = ta * 256, b0 = tb * 256;
    const int m_begin = split * g.m_per_split;
    int m_end = m_begin + g.m_per_split;
    m_end = m_end < g.M ? m_end : g.M;
    if (m_begin >= m_end) return;
    const int nk = (m_end - m_begin + 63) / 64;
    const bool do_cs = g.colsum != nullptr && tb == 0;  // block-uniform: every wave sums 2 of its 8 a-tiles' columns

    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem;
    const char* baseP = (const char*)g.P + (size_t)m_begin * g.ldp * 2;
    const char* baseQ = (const char*)g.Q + (size_t)m_begin * g.ldq * 2;
    const size_t stepP = (size_t)64 * g.ldp * 2, stepQ = (size_t)64 * g.ldq * 2;
    // inside the matrix: one offset register per operand (this lane's row wave * 4 + (lane >> 4), its permuted column chunk)
    const int row_l = wave * 4 + (lane >> 4), s16_l = lane & 15;
    const int col_l = (((s16_l >> 1) ^ (row_l & 7)) << 4) + (s16_l & 1) * 8;
    const unsigned offP = ((unsigned)row_l * (unsigned)g.ldp + (unsigned)(a0 + col_l)) * 2u;
    const unsigned offQ = ((unsigned)row_l * (unsigned)g.ldq + (unsigned)(b0 + col_l)) * 2u;
    const bool cols_inside = a0 + 256 <= g.Na && b0 + 256 <= g.Nb;
    const size_t halfP = (size_t)32 * g.ldp * 2, halfQ = (size_t)32 * g.ldq * 2;
    int i_st = 0;
    auto issue = [&]() {
        const int m_stage = m_begin + i_st * 64;
        const unsigned dst = lds0 + (unsigned)(i_st & 1) * 65536u + (unsigned)wave * 1024u;
        const char* bp = baseP + ((TN_ABL & 4) ? 0 : (size_t)i_st * stepP);  // TN_ABL 4: every stage re-reads the first one (L2-hot DMA)
        const char* bq = baseQ + ((TN_ABL & 4) ? 0 : (size_t)i_st * stepQ);
        if (cols_inside && m_stage + 64 <= g.M) {
            glds16_asm_imm<0>(offP, bp, dst);
            glds16_asm_imm<0>(offP, bp + halfP, dst + 8192u);
            glds16_asm_imm<256>(offP, bp, dst + 16384u);
            glds16_asm_imm<256>(offP, bp + halfP, dst + 16384u + 8192u);
            glds16_asm_imm<0>(offQ, bq, dst + 32768u);
            glds16_asm_imm<0>(offQ, bq + halfQ, dst + 32768u + 8192u);
            glds16_asm_imm<256>(offQ, bq, dst + 49152u);
            glds16_asm_imm<256>(offQ, bq + halfQ, dst + 49152u + 8192u);
        } else {
            tn_stage_issue_clamped(g, a0, b0, m_stage, bp, bq, lds0 + (unsigned)(i_st & 1) * 65536u, wave, lane);
        }
        ++i_st;
    };
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER_P();
    if (i_st < nk) issue();

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs0 = 0.f, cs1 = 0.f;

    // lane offset of column block ct inside a [64][128] LDS tile = offset of block 0 XOR (ct << 5): one register per operand
    const unsigned off0 = tn_frag_off(0, lane);
    const unsigned pBase = (unsigned)wa * 16384u, qBase = 32768u + (unsigned)(wb >> 1) * 16384u;
    const unsigned qOff0 = off0 ^ ((unsigned)(wb & 1) << 7);
#define pOff(ct) (pBase + (off0 ^ ((unsigned)(ct) << 5)))
#define qOff(j) (qBase + (qOff0 ^ ((unsigned)(j) << 5)))

    bf16x8 pF[2][4], qF[2][4];
    // TAIL: the m-range ends inside this stage or the next one -- the P fragments of rows past the end are zeroed
#define TN_LOAD_P(dst, buf, u, h, valid)                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
        dst[i] = tn_frag(buf, pOff((h) * 4 + i), u);                                                   \
        if (TAIL && (valid) < 64) tn_mask(dst[i], u, valid, lane);                                     \
    }
#define TN_LOAD_Q(dst, buf, u) _Pragma("unroll") for (int j = 0; j < 4; ++j) dst[j] = tn_frag(buf, qOff(j), u)
#define TN_MFMA16(pv, qv, h)                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
            acc[(h) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qv[j], pv[i], acc[(h) * 4 + i][j], 0, 0, 0)
    // bias gradient: wave wb sums the columns of a-tiles 2 wb, 2 wb + 1 of its 128.  A lane's fragment holds 8 rows (k-slots) of ONE
    // column: v_dot2c_f32_bf16 against (1, 1) adds them into a per-lane fp32 partial (two registers in all; the ones . P MFMA form
    // of the 128x128 kernel costs twelve, which this kernel does not have); the four lane groups meet after the loop
#define TN_CS1(acc_, frag_)                                                                            \
    {                                                                                                  \
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;                                    \
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;                                   \
        const u32x4_ w_ = __builtin_bit_cast(u32x4_, frag_);                                           \
        const bf16x2_ one_ = {(bf16)1.0f, (bf16)1.0f};                                                 \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                                \
            const unsigned wd_ = w_[d];                                                                \
            acc_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_, wd_), one_, acc_, false); \
        }                                                                                              \
    }
#define TN_CS(pv, h)                                                                                   \
    if (CS && (wb >> 1) == (h)) {                                                                      \
        if (wb & 1) { TN_CS1(cs0, pv[2]) TN_CS1(cs1, pv[3]) }                                          \
        else        { TN_CS1(cs0, pv[0]) TN_CS1(cs1, pv[1]) }                                          \
    }
    int valid = m_end - m_begin;  // rows of the current stage that belong to the m-range (>= 64: all)
    {
        constexpr bool TAIL = true;
        TN_LOAD_Q(qF[0], smem, 0);
        TN_LOAD_P(pF[0], smem, 0, 0, valid);
    }
#define TN_STAGE()                                                                                     \
    {                                                                                                  \
        const char* cur = smem + (st & 1) * 65536;                                                     \
        const char* nxt = smem + ((st + 1) & 1) * 65536;                                               \
        const int valid_n = valid - 64;                                                                \
        TN_LOAD_P(pF[1], cur, 0, 1, valid);                                                            \
        TN_MFMA16(pF[0], qF[0], 0);                                                                    \
        TN_CS(pF[0], 0);                                                                               \
        TN_LOAD_Q(qF[1], cur, 1);                                                                      \
        TN_LOAD_P(pF[0], cur, 1, 0, valid);                                                            \
        TN_MFMA16(pF[1], qF[0], 1);                                                                    \
        TN_CS(pF[1], 1);                                                                               \
        TN_LOAD_P(pF[1], cur, 1, 1, valid);                                                            \
        TN_MFMA16(pF[0], qF[1], 0);                                                                    \
        TN_CS(pF[0], 0);                                                                               \
        __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) expcnt(7); the builtin, not inline asm: the compiler's own wait */ \
        /* bookkeeping sees it (with the asm form it re-waits for the same reads behind the barrier); +1-2 %, tools/tn_ab.py */ \
        RAW_BARRIER_P();                                                                               \
        /* stage st+2 goes into the buffer every wave has just finished reading.  The two waves of a SIMD (w, w + 4) do not issue */ \
        /* their 8 LDS-DMA pieces at the same time (a piece costs ~100+ cycles of VMEM issue and both would sit in it with the */ \
        /* matrix pipe idle): waves 0..3 issue here, waves 4..7 behind the stage's last MFMA group */ \
        const bool do_issue = i_st < nk && !(TN_ABL & 1);                                              \
        if (do_issue && (!TN_STAGGER_DMA || wave < 4)) issue();                                        \
        if (!TAIL || st + 1 < nk) { /* the mask-free copy only runs while a next stage exists */       \
            TN_LOAD_Q(qF[0], nxt, 0);                                                                  \
            TN_LOAD_P(pF[0], nxt, 0, 0, valid_n);                                                      \
        }                                                                                              \
        TN_MFMA16(pF[1], qF[1], 1);                                                                    \
        TN_CS(pF[1], 1);                                                                               \
        if (do_issue && TN_STAGGER_DMA && wave >= 4) issue();                                          \
        valid = valid_n;                                                                               \
    }
    int st = 0;
    // four copies of the stage (bias-gradient MFMAs or not, masks or not), each straight-line: a branch inside the stage would end
    // the scheduling region the compiler interleaves the transposing reads and the MFMAs in
    if (do_cs) {
        constexpr bool CS = true;
        { constexpr bool TAIL = false; for (; st < nk && valid >= 128; ++st) TN_STAGE() }
        { constexpr bool TAIL = true; for (; st < nk; ++st) TN_STAGE() }
    } else {
        constexpr bool CS = false;
        { constexpr bool TAIL = false; for (; st < nk && valid >= 128; ++st) TN_STAGE() }  // no row of this stage or the next is past the m-range
        { constexpr bool TAIL = true; for (; st < nk; ++st) TN_STAGE() }                   // the last one or two stages
    }
#undef TN_STAGE
#undef pOff
#undef qOff
#undef TN_LOAD_P
#undef TN_LOAD_Q
#undef TN_MFMA16
#undef TN_CS
#undef TN_CS1
    if (do_cs) {
        cs0 += __shfl_xor(cs0, 16, 64); cs0 += __shfl_xor(cs0, 32, 64);
        cs1 += __shfl_xor(cs1, 16, 64); cs1 += __shfl_xor(cs1, 32, 64);
        if (lane < 16) {
            const int a = a0 + wa * 128 + 2 * wb * 16 + lane;
            if (a < g.Na) tn_colsum_out(g, split, a, cs0);
            if (a + 16 < g.Na) tn_colsum_out(g, split, a + 16, cs1);
        }
    }
    if (g.cnt) {  // fused reduce (gemm.hip): partial + arrival; the last block of the tile sums the ranges in order
        tn_fused_reduce<8, 4>(g, acc, t, split, a0 + wa * 128, b0 + wb * 64, lane, (int*)smem);
        if (*(volatile int*)smem == g.splits - 1 && do_cs && g.cs_ws && lane < 16) {
            const int a = a0 + wa * 128 + 2 * wb * 16 + lane;
            if (a < g.Na) tn_fused_colsum(g, a);
            if (a + 16 < g.Na) tn_fused_colsum(g, a + 16);
        }
        return;
    }
    // acc[i][j]: lane holds a = a-tile i column (lane & 15), b = b-tile j rows (lane >> 4) * 4 .. + 3 -> one 16-B fp32 access
    float* obase = g.ws ? g.ws + (size_t)split * g.Na * g.Nb : g.out;
    const int old_ = g.ws ? g.Nb : g.ldo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int a = a0 + wa * 128 + i * 16 + (lane & 15);
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
