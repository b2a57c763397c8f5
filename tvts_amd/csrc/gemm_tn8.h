// e4m3 weight-gradient kernel (BASELINE config 5: the fp8 MFMA weight/activation path, round 4):
//   C[Na,Nb] (+)= sp * sq * sum_m P8[m,Na] * Q8[m,Nb],  P8 / Q8 row-major e4m3 bytes under ONE scale per tensor.
// The contraction runs over the token rows, so both operands are contraction-strided: the fragments of the K = 128 scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, unit E8M0 scales -- the only fp8 form of gfx950 that issues at twice the bf16 rate) come
// from LDS through ds_read_b64_tr_b8: a 16-lane group hands the hardware an [8 tokens][16 columns] block (lane l supplies the
// 8 bytes at token l >> 1, half l & 1 of the 16 columns) and lane i receives column i of those 8 tokens
// (experiments/probes/tr8_probe.hip measured the mapping).  Four such reads make a lane's 32-byte operand; lane group g takes
// tokens 32 g .. 32 g + 31 of the 128-token stage for P and Q alike, so every token meets its partner.
//
// Shape of the kernel = gemm_tn256.h's: 256 x 256 output tile, 512 threads = 8 waves as 2 (a) x 4 (b), wave tile 128 x 64 = 8 x 4
// accumulator tiles; a stage is 128 token rows = [P 2 x 16 KiB | Q 2 x 16 KiB] (sub-tile = 64 rows x 256 B, the byte geometry of
// the bf16 kernel's [64][128] tile, 32-B chunk c of row r at c ^ (r & 7), and the two 16-B halves of a chunk swapped in rows
// 32-63: a transposing read of one 16-lane group touches one half of each of the 8 chunks = 32 of the 64 banks, and lane groups
// 0 / 1, which read the same columns 32 rows apart, then take opposite halves (no measurable difference against the unswapped
// image: the reads are not what the stage waits for), two stages in LDS, stage s + 2's LDS-DMA issued behind
// the barrier that frees its buffer, staggered between the two waves of a SIMD.  Per stage a wave issues 32 MFMAs in four chunks
// of 2 a-tiles x 4 b-tiles; the P pairs are double-buffered across chunks, the four Q operands are re-read in place for the next
// stage behind the last MFMAs that use them: 64 fragment registers + 128 accumulators.
// One (m-range, tile) work item per block; partials to the workspace + tn_reduce_kernel (ordered, deterministic), or the output
// itself with one range.  Bias gradient (optional): the blocks of the first tile column also multiply a ones operand (e4m3 1.0 in
// every byte) with two of the wave's eight P operands per stage -- every row of that product is the column sum of the stage -- so
// the four b-waves of an a-half cover its 8 column blocks with 2 extra MFMAs (+6 %) and 8 registers each; the sums leave as
// per-range partials behind the output partials and are added in range order by tn_reduce_kernel.  They are the sums of the
// QUANTISED output gradient (scale_p * sum of the e4m3 values), consistent with the weight gradient made of the same bytes.
// Included by gemm.hip.
#pragma once

typedef __attribute__((ext_vector_type(2))) int tn8_i32x2;
typedef __attribute__((ext_vector_type(8))) int tn8_i32x8;

struct GemmTN8 {
    const unsigned char* P; int ldp;  // bytes
    const unsigned char* Q; int ldq;
    int M, Na, Nb;
    float* out; int ldo;
    const float* sp; const float* sq;  // device scalars
    int tiles_a, tiles_b, tiles_ab, m_per_split, n_items, a_fast;
    int accumulate;  // one range, no workspace: out += (else =)
    float* ws;       // split partials [splits][Na][Nb]
    float* colsum;   // optional bias gradient: colsum[a] += sp * sum_m P8[m,a]
    float* cs_ws;    // its split partials [splits][Na] (nullptr with one range: the owner adds its sum itself)
};

// one 32-byte operand: tokens 32 g + 8 j + (0..7), j = 0..3, of the 16-column block whose lane offset is `off`
__device__ __forceinline__ tn8_i32x8 tn8_frag(const char* stage_base, unsigned off) {
    tn8_i32x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const tn8_i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_PTR(tn8_i32x2))(stage_base + off + j * 2048));
        r[2 * j] = v[0];
        r[2 * j + 1] = v[1];
    }
    return r;
}
// zero the bytes of an operand whose token (inside the stage) is >= valid: dword d holds tokens 32 g + 8 (d >> 1) + 4 (d & 1) + (0..3)
__device__ __forceinline__ void tn8_mask(tn8_i32x8& f, int valid, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int t0 = 32 * g + 8 * (d >> 1) + 4 * (d & 1);
        const int keep = valid - t0;  // bytes 0 .. keep - 1 stay
        const unsigned m = keep >= 4 ? 0xffffffffu : keep <= 0 ? 0u : (0xffffffffu >> (8 * (4 - keep)));
        f[d] &= (int)m;
    }
}

__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(GemmTN8 g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][P rows 0-63 | P rows 64-127 | Q 0-63 | Q 64-127], 16 KiB each
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 2, wb = wave & 3;
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
    const int nk = (m_end - m_begin + 127) / 128;

    const unsigned lds0 = (unsigned)(size_t)(LDS_PTR(char))smem;
    const char* baseP = (const char*)g.P + (size_t)m_begin * g.ldp;
    const char* baseQ = (const char*)g.Q + (size_t)m_begin * g.ldq;
    // LDS-DMA: piece q of a wave = sub-tile q >> 1, rows (q & 1) * 32 + wave * 4 + (lane >> 4), 16-byte piece lane & 15 of the row
    const int row_l = wave * 4 + (lane >> 4), s16_l = lane & 15;
    const int col_l = (((s16_l >> 1) ^ (row_l & 7)) << 5) + (s16_l & 1) * 16;   // rows 0-31 of a sub-tile (pieces q even)
    const int col_h = (((s16_l >> 1) ^ (row_l & 7)) << 5) + ((s16_l & 1) ^ 1) * 16;  // rows 32-63 (q odd): halves swapped
    const unsigned offP = (unsigned)row_l * (unsigned)g.ldp + (unsigned)(a0 + col_l);
    const unsigned offQ = (unsigned)row_l * (unsigned)g.ldq + (unsigned)(b0 + col_l);
    const unsigned offPh = (unsigned)row_l * (unsigned)g.ldp + (unsigned)(a0 + col_h);
    const unsigned offQh = (unsigned)row_l * (unsigned)g.ldq + (unsigned)(b0 + col_h);
    const bool cols_inside = a0 + 256 <= g.Na && b0 + 256 <= g.Nb;
    int i_st = 0;
    auto issue = [&]() {
        const int m_stage = m_begin + i_st * 128;
        const unsigned dst = lds0 + (unsigned)(i_st & 1) * 65536u + (unsigned)wave * 1024u;
        const char* bp = baseP + (size_t)i_st * 128 * g.ldp;
        const char* bq = baseQ + (size_t)i_st * 128 * g.ldq;
        if (cols_inside && m_stage + 128 <= g.M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                glds16_asm((q & 1) ? offPh : offP, bp + (size_t)(q * 32) * g.ldp, dst + (unsigned)q * 8192u);
                glds16_asm((q & 1) ? offQh : offQ, bq + (size_t)(q * 32) * g.ldq, dst + 32768u + (unsigned)q * 8192u);
            }
        } else {  // rows past the matrix / columns past it: every address clamped on its own (finite bytes; the rows are masked out of P)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool isP = q < 4;
                const int r = (q & 3) * 32 + row_l;
                int gm = m_stage + r;
                gm = gm < g.M - 1 ? gm : g.M - 1;
                int col = (isP ? a0 : b0) + ((q & 1) ? col_h : col_l);
                const int cmax = (isP ? g.Na : g.Nb) - 16;
                col = col < cmax ? col : cmax;
                const unsigned off = (unsigned)(gm - m_stage) * (unsigned)(isP ? g.ldp : g.ldq) + (unsigned)col;
                glds16_asm(off, isP ? bp : bq, dst + (isP ? 0u : 32768u) + (unsigned)(q & 3) * 8192u);
            }
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

    // fragment offsets: lane (g = lane >> 4, i = lane & 15) supplies token 32 g + 8 j + (i >> 1), byte half i & 1; 32 g + .. < 64 lives in
    // sub-tile 0, the rest in sub-tile 1.  Column block ct (16 bytes) sits in 32-B chunk (ct >> 1) ^ (token & 7) = (ct >> 1) ^ (i >> 1).
    const int fg = lane >> 4, fi = lane & 15;
    // ... and in 16-B half (ct & 1) ^ (g & 1) of that chunk (tokens 32-63 of a sub-tile keep their halves swapped)
    const unsigned fbase = (unsigned)(fg >> 1) * 16384u + (unsigned)((fg & 1) * 32 + (fi >> 1)) * 256u + (unsigned)(fi & 1) * 8u;
    const unsigned fx = ((unsigned)(fi >> 1) << 5) | ((unsigned)(fg & 1) << 4);
#define TN8_OFF(ct) (fbase + (fx ^ (((unsigned)((ct) >> 1) << 5) | ((unsigned)((ct) & 1) << 4))))
    int mx_one = 0x7F7F7F7F;  // E8M0 127 = 1.0 in every byte of the scale operand
    asm volatile("" : "+v"(mx_one));
#define TN8_MFMA(bv, av, c) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(bv), "v"(av), "v"(mx_one))
    tn8_i32x8 pQ[2][2], qQ[4];
    // bias gradient: wave (wa, wb) of a first-column block sums a-tiles 2 wb, 2 wb + 1 of its half = pair wb of the stage's four P pairs
    const bool do_cs = g.colsum != nullptr && tb == 0;  // block-uniform
    f32x4 cs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    tn8_i32x8 ones;
#pragma unroll
    for (int d = 0; d < 8; ++d) ones[d] = 0x38383838;  // e4m3 1.0
    asm volatile("" : "+v"(ones));
#define TN8_CS(pv, pr) if (do_cs && wb == (pr)) { TN8_MFMA(ones, pv[0], cs[0]); TN8_MFMA(ones, pv[1], cs[1]); }
    int valid = m_end - m_begin;  // token rows of the current stage that belong to the m-range (>= 128: all)
#define TN8_LOAD_P(dst, buf, pr, vld)                                                                   \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                                  \
        dst[t2] = tn8_frag(buf, TN8_OFF(wa * 8 + (pr) * 2 + t2));                                       \
        if ((vld) < 128) tn8_mask(dst[t2], vld, lane);                                                  \
    }
#define TN8_LOAD_Q(j, buf) qQ[j] = tn8_frag((buf) + 32768, TN8_OFF(wb * 4 + (j)))
#define TN8_MFMA8(pv, pr)                                                                               \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) TN8_MFMA(qQ[j], pv[t2], acc[(pr) * 2 + t2][j])
#pragma unroll
    for (int j = 0; j < 4; ++j) TN8_LOAD_Q(j, smem);
    TN8_LOAD_P(pQ[0], smem, 0, valid);
    for (int st = 0; st < nk; ++st) {
        const char* cur = smem + (st & 1) * 65536;
        const char* nxt = smem + ((st + 1) & 1) * 65536;
        const int valid_n = valid - 128;
        TN8_LOAD_P(pQ[1], cur, 1, valid);
        TN8_MFMA8(pQ[0], 0);
        TN8_CS(pQ[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        TN8_LOAD_P(pQ[0], cur, 2, valid);
        TN8_MFMA8(pQ[1], 1);
        TN8_CS(pQ[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        TN8_LOAD_P(pQ[1], cur, 3, valid);
        TN8_MFMA8(pQ[0], 2);
        TN8_CS(pQ[0], 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        RAW_BARRIER_P();
        const bool do_issue = i_st < nk;
        if (do_issue && wave < 4) issue();
        const bool more = st + 1 < nk;
        TN8_CS(pQ[1], 3);  // (before pQ[0] is reloaded: the compiler keeps the order of the volatile asm)
        if (more) TN8_LOAD_P(pQ[0], nxt, 0, valid_n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            TN8_MFMA(qQ[j], pQ[1][0], acc[6][j]);
            TN8_MFMA(qQ[j], pQ[1][1], acc[7][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) TN8_LOAD_Q(j, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (do_issue && wave >= 4) issue();
        valid = valid_n;
    }
#undef TN8_CS
#undef TN8_OFF
#undef TN8_MFMA
#undef TN8_LOAD_P
#undef TN8_LOAD_Q
#undef TN8_MFMA8
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (asm: the compiler does not know they are MFMAs)
    if (do_cs && lane < 16) {  // the ones operand makes every row of cs[] the column sum: row 0 = element 0 of lanes 0 .. 15
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int a = a0 + wa * 128 + (wb * 2 + t2) * 16 + lane;
            if (a < g.Na) {
                const float v = cs[t2][0] * g.sp[0];
                if (g.cs_ws) g.cs_ws[(size_t)split * g.Na + a] = v;
                else g.colsum[a] += v;
            }
        }
    }
    const float scale = g.sp[0] * g.sq[0];
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
            f32x4 v = acc[i][j] * scale;
            if (!g.ws && g.accumulate) v += *(const f32x4*)dst;
            *(f32x4*)dst = v;
        }
    }
}
