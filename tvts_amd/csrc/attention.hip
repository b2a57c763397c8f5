// Fused softmax attention (forward, dQ, dK/dV) for every attention site of the TVTSv2 step, head dim 64.
//
// One kernel family, four token-index geometries over a packed qkv buffer [rows, 3*W] (q | k | v, heads
// contiguous inside each third -- the layout of VarAttention.qkv, nn.MultiheadAttention.in_proj and
// SelfAttention.qkv alike):
//   FULL  : group (b,h); queries = keys = rows b*S .. b*S+S-1; optional causal mask
//           (CLIP text tower v2/CLIP/clip/model.py:185-187,330-336; sort head v2/model/sort_transformer.py:45-53)
//   SPACE : group (b,h,f); queries = the n patch tokens of frame f; keys = CLS token + those n tokens
//   TIME  : group (b,h,p); queries = the T tokens at patch slot p; keys = CLS token + those T tokens
//   CLS   : group (b,h); query = CLS token; keys = all S tokens
//           (divided space-time attention, v2/model/video_encoder_ViT_B_16.py:38-76: q scaled by dh^-0.5
//            before the CLS split :45-51, CLS k/v prepended to every group :56-60)
// Nothing is regrouped in memory: the einops rearrange / repeat / cat of the reference become index math.
//
// MFMA formulation (wave64, v_mfma_f32_16x16x32_bf16): scores are computed TRANSPOSED, S^T = K Q^T, so
// a lane holds 4 consecutive keys of ONE query (C layout: col = query = lane&15, rows = keys).  The
// softmax reduction over keys is then in-lane + 2 cross-lane steps, and exp(S^T) converts in-lane into
// the B operand of O^T = V^T P^T by choosing the MFMA k-slot <-> key permutation
//   slot (l>>4)*8 + j  <->  key 32u + 16*(j>>2) + 4*(l>>4) + (j&3)
// on both operands.  The matching V^T A-operand is gathered from an LDS image of the V tile with the
// transposing read ds_read_b64_tr_b16.  Q/K/V/dO fragments that are k-contiguous are loaded straight from
// global memory (16 B per lane) -- no LDS round trip, no P matrix in memory.
// Backward recomputes P from the saved log-sum-exp (flash-attention style): one pass over key tiles for
// dQ, one pass over query tiles for dK/dV.  In SPACE/TIME geometry the dK/dV pass also folds in the CLS
// query (which attends every token), so each dqkv element is written exactly once; the CLS key/value
// gradient is the only cross-group sum and goes through fp32 atomics + a tiny finalize kernel.
#include "common.h"
#include <stdlib.h>

enum { MODE_FULL = 0, MODE_SPACE = 1, MODE_TIME = 2, MODE_CLS = 3 };
// Head dimension is a compile-time constant of this translation unit: the file is compiled twice
// (-DTVTS_DH=64 -> tvts_attn_*, -DTVTS_DH=80 -> tvts_attn80_*; ViT-H/14 has 1280/16 = 80).
#ifndef TVTS_DH
#define TVTS_DH 64
#endif
#define DH TVTS_DH
#if TVTS_DH == 64
#define NS_DH dh64
#define ABI(name) tvts_attn_##name
#define VSTRIDE 160  // bytes per LDS row of a [rows][DH] bf16 tile (128 + 32: conflict-free b64_tr reads)
#else
#define NS_DH dh80
#define ABI(name) tvts_attn80_##name
#define VSTRIDE 192
#endif
namespace NS_DH {
constexpr int DT = DH / 16;         // 16-wide d tiles of the output
constexpr int KS = (DH + 31) / 32;  // MFMA k-steps over d (the last one is half empty for DH = 80)
constexpr int NCH = DH / 8;         // 16-byte chunks per head row

struct AttnGeom {
    int B, heads, S, T, n;  // S tokens per sample (FULL: sequence length); T,n only for SPACE/TIME
    int causal;
    int ld;                 // qkv row stride (elements)
    int W;                  // heads * 64
    float scale2;           // dh^-0.5 * log2(e)
    float scale;            // dh^-0.5
    int q8_only;            // fused divided kernels with a q8 copy: 1 = the patch rows' bf16 values are NOT stored (opts bit 7; the CLS
                            // row, which the merge / finalize kernels write, keeps both forms)
    int ablate;             // dev knob (opts bits 4..6 of tvts_attn_bwd): 1 skip phase A, 2 skip phase B, 4 skip global loads
    const int* kv_len;      // FULL only, optional: valid keys per sequence (keys >= kv_len[b] are padding and masked)
    int cls_nq, cls_q0;     // CLS geometry generalised: cls_nq (<= 16) queries at tokens cls_q0 .. of the sequence (1, 0 = the CLS token)
    const int* cls_qpos;    // ... or ONE query per sequence at token cls_qpos[b] that sees the keys 0 .. cls_qpos[b] (causal)
    int cls_parts;          // fused backward: > 0 = cls_acc holds one partial per block ([B, heads, cls_parts, 3, dh]); 0 = atomics
    // dropout on the attention probabilities (FULL geometry with kv_len: the DistilBERT text tower of v1 in training mode,
    // transformers MultiHeadSelfAttention: weights = dropout(softmax(scores))).  Counter-based: element (b, h, q, k) is kept iff
    // the upper 32 bits of splitmix64(seed + index) are >= drop_thr = p * 2^32; kept probabilities are scaled by drop_inv =
    // 1 / (1 - p).  The backward regenerates the same mask from the same (seed, index).
    unsigned drop_thr;
    float drop_inv;
    const unsigned long long* drop_seed;  // device memory (a captured graph draws new masks on every replay); + drop_site per call
    unsigned long long drop_site;
    // per-tensor e4m3 copy of the kernel's bf16 result (fused divided kernels; BASELINE config 5): q8[r, c] = e4m3(res[r, c] / q8_scale[0])
    // for every element the kernel stores at q8_ref + r * ld + c, q8 has the same leading dimension (in bytes); q8_amax = max |res|
    const bf16* q8_ref; unsigned char* q8; const float* q8_scale; float* q8_amax;
};
// per-thread state of that copy
struct Q8Out {
    const bf16* ref; unsigned char* q8; float inv; float am;
    bool only;  // the e4m3 copy is the ONLY output of the fused kernels' patch rows (g.q8_only): no bf16 row is stored
    __device__ __forceinline__ void init(const AttnGeom& g) {
        ref = g.q8_ref; q8 = g.q8; am = 0.f; inv = 0.f; only = g.q8 && g.q8_only;
        if (q8) { const float sc = g.q8_scale[0]; inv = sc > 0.f ? 1.0f / sc : 1.0f; }
    }
    __device__ __forceinline__ void emit8(const bf16* dst8, const bf16x8& w) {  // the 8 values stored at dst8, as 8 e4m3 bytes
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)w[e];
            am = fmaxf(am, fabsf(x));
            f[e] = fminf(fmaxf(x * inv, -448.0f), 448.0f);
        }
        int p0 = 0, p1 = 0;
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], p0, false);
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p0, true);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], p1, false);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], p1, true);
        typedef __attribute__((ext_vector_type(2))) int i32x2_;
        *(i32x2_*)(q8 + (dst8 - ref)) = (i32x2_){p0, p1};
    }
    __device__ __forceinline__ void finish(const AttnGeom& g, int lane) {
        if (q8) amax_publish(g.q8_amax, wave_max(am), lane);
    }
};
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned long long idx, unsigned thr) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thr;
}

struct Grp { int b, h, sub, nq, nk; };

template <int MODE>
__device__ __forceinline__ int n_groups(const AttnGeom& g) {
    if (MODE == MODE_SPACE) return g.B * g.heads * g.T;
    if (MODE == MODE_TIME) return g.B * g.heads * g.n;
    return g.B * g.heads;
}
template <int MODE>
__device__ __forceinline__ Grp decode(const AttnGeom& g, int gid) {
    // head index fastest: neighbouring work items (waves of a block, consecutive blocks) touch neighbouring
    // 128-byte head slices of the SAME token rows, so a row is opened once per pass instead of once per head
    Grp r;
    r.h = gid % g.heads;
    gid /= g.heads;
    if (MODE == MODE_SPACE) { r.sub = gid % g.T; r.b = gid / g.T; r.nq = g.n; r.nk = g.n + 1; }
    else if (MODE == MODE_TIME) { r.sub = gid % g.n; r.b = gid / g.n; r.nq = g.T; r.nk = g.T + 1; }
    else if (MODE == MODE_CLS) {
        r.sub = 0; r.b = gid; r.nq = g.cls_nq; r.nk = g.S;
        if (g.cls_qpos) { r.nq = 1; r.nk = g.cls_qpos[gid] + 1; }
    }
    else {
        r.sub = 0; r.b = gid; r.nq = g.S; r.nk = g.S;
        if (g.kv_len) { const int l = g.kv_len[gid]; r.nk = l < 1 ? 1 : (l < g.S ? l : g.S); }
    }
    return r;
}
// token row of query i / key j of a group
template <int MODE>
__device__ __forceinline__ int q_row(const AttnGeom& g, const Grp& r, int i) {
    const int base = r.b * g.S;
    if (MODE == MODE_SPACE) return base + 1 + r.sub * g.n + i;
    if (MODE == MODE_TIME) return base + 1 + i * g.n + r.sub;
    if (MODE == MODE_CLS) return base + (g.cls_qpos ? g.cls_qpos[r.b] : g.cls_q0) + i;
    return base + i;
}
template <int MODE>
__device__ __forceinline__ int k_row(const AttnGeom& g, const Grp& r, int j) {
    const int base = r.b * g.S;
    if (MODE == MODE_SPACE) return j == 0 ? base : base + r.sub * g.n + j;
    if (MODE == MODE_TIME) return j == 0 ? base : base + 1 + (j - 1) * g.n + r.sub;
    return base + j;
}
// queries with the CLS query prepended (SPACE/TIME dK/dV pass)
template <int MODE>
__device__ __forceinline__ int qx_row(const AttnGeom& g, const Grp& r, int i) {
    if (MODE == MODE_SPACE || MODE == MODE_TIME) return i == 0 ? r.b * g.S : q_row<MODE>(g, r, i - 1);
    return q_row<MODE>(g, r, i);
}

__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
    return z;
}
// A tile row that may lie past the group's last token: the LOAD is unconditional (row 0 of the group stands in -- always a valid
// address) and the value is zeroed afterwards.  Written as `ok ? ldg8(p) : zero8()`, hipcc puts every such load into a conditional
// block of its own and, the loaded register being merged with the zeros of the other arm, waits vmcnt(0) right behind it: loads
// meant to be in flight together left one or two at a time (round 5, from the ISA of the fused SPACE kernels, which load a group and
// stage it at once.  NOT for loads that are consumed an iteration later -- the TIME / sequence kernels' prefetch of the next group:
// the selects then sit right behind the loads and the wait for them makes the prefetch synchronous; their conditional form is fine).
__device__ __forceinline__ bf16x8 sel8(bool ok, const bf16x8& v) { return ok ? v : zero8(); }
// KS k-contiguous fragments of one head row (d = ks*32 + gq*8 ..); slots past DH are zero
__device__ __forceinline__ void ld_frags(const bf16* p, int gq, bf16x8 (&f)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 32 + gq * 8;
        f[ks] = (d0 + 8 <= DH) ? ldg8(p + d0) : zero8();
    }
}

// k-contiguous fragment of tile row `row` (d = ks*32 + gq*8 ..); slots past DH are zero
// rs: rows the tile actually stores (rows >= rs read as zeros without touching LDS: the TIME kernels of ViT-H/14 keep 20 rows of
// their 32-row MFMA geometry, 17 of them real, and fit two blocks on a CU instead of one)
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row, int ks, int gq, int rs = 1 << 30) {
    const int d0 = ks * 32 + gq * 8;
    return (d0 + 8 <= DH && row < rs) ? *(const bf16x8*)(tile + row * VSTRIDE + d0 * 2) : zero8();
}

// stage `nrows` (<= 64, multiple of 16) rows x 64 bf16 (rows given by a functor) into a wave-private LDS tile;
// rows [nrows, round_up(nrows, 32)) are ZERO-filled: the MFMA k-step that covers them multiplies by P = 0, and
// stale LDS bits could be NaN.
template <typename RowFn>
__device__ __forceinline__ void stage_tile(char* tile, int nrows, int lane, const bf16* base, int ld, int col0, RowFn rowfn) {
    for (int c = lane; c < nrows * NCH; c += 64) {
        const int r = c / NCH, ch = c % NCH;
        const bf16x8 v = ldg8(base + (size_t)rowfn(r) * ld + col0 + ch * 8);
        *(bf16x8*)(tile + r * VSTRIDE + ch * 16) = v;
    }
    if (nrows & 16) {
        const bf16x8 z = zero8();
        for (int c = lane; c < 16 * NCH; c += 64) *(bf16x8*)(tile + (nrows + c / NCH) * VSTRIDE + (c % NCH) * 16) = z;
    }
}

// A-operand fragment of X^T (rows = 16 columns d of block dt, k-slots = tile rows of k-step u) from a
// row-major LDS tile: two transposing reads, each a [4 rows][16 cols] block per 16-lane group.
template <bool TR>
__device__ __forceinline__ bf16x8 frag_T(const char* tile, int u, int dt, int lane) {
    const int gq = lane >> 4, i = lane & 15;
    bf16x8 out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row0 = u * 32 + half * 16 + gq * 4;
        if (TR) {
            const char* p = tile + (row0 + (i >> 2)) * VSTRIDE + dt * 32 + (i & 3) * 8;
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
            const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[half * 4 + e] = tb[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[half * 4 + e] = *(const bf16*)(tile + (row0 + e) * VSTRIDE + (dt * 16 + i) * 2);
        }
    }
    return out;
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups holding one query/key column
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Coalesced store of one 16-row x DH output tile held in the MFMA C layout (lane (li = row, gq) holds columns
// dt*16 + gq*4 .. +3 of its row): written straight from registers that is one 8-byte piece per lane and instruction,
// 32 bytes per row -- measured at half the fused TIME backward's run time.  Two dt-columns (a 64-byte row segment) at a
// time go through a 1 KiB wave-private LDS patch (16-byte chunks XOR-swizzled by row) so that every lane stores 16
// contiguous bytes and four neighbouring lanes complete the segment.  rowfn(row) -> destination of that row's first
// head column, or nullptr for a masked row.
template <typename RowFn>
__device__ __forceinline__ void store_tile_rows(char* patch, const f32x4 (&v)[DT], int lane, RowFn rowfn, Q8Out* q8o = nullptr) {
    const int li = lane & 15, gq = lane >> 4;
    const int rrow = lane >> 2, rc = lane & 3;
    bf16* dst = rowfn(rrow);
#pragma unroll
    for (int d2 = 0; d2 < (DT + 1) / 2; ++d2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int dt = 2 * d2 + h;
            if (dt < DT) {
                const int pidx = h * 4 + gq;  // 8-byte piece of the 64-byte row segment
                *(bf16x4*)(patch + li * 64 + (((pidx >> 1) ^ (li & 3)) << 4) + (pidx & 1) * 8) =
                    (bf16x4){(bf16)v[dt][0], (bf16)v[dt][1], (bf16)v[dt][2], (bf16)v[dt][3]};
            }
        }
        // the 8-byte writes and the 16-byte read below use different vector types: keep the compiler from reordering them
        // on type-based alias grounds (the LDS itself executes a wave's accesses in order)
        asm volatile("" ::: "memory");
        const bf16x8 w = *(const bf16x8*)(patch + rrow * 64 + ((rc ^ (rrow & 3)) << 4));
        asm volatile("" ::: "memory");
        if (dst && 2 * d2 + (rc >> 1) < DT) {
            if (!(q8o && q8o->only)) *(bf16x8*)(dst + d2 * 32 + rc * 8) = w;
            if (q8o && q8o->q8) q8o->emit8(dst + d2 * 32 + rc * 8, w);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int MODE, bool TR>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnGeom g, const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                       int ldo, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * VSTRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* vt = smem + wave * 64 * VSTRIDE;
    const int nq_max = (MODE == MODE_SPACE) ? g.n : (MODE == MODE_TIME) ? g.T : (MODE == MODE_CLS) ? 1 : g.S;
    const int qtiles = (nq_max + 15) >> 4;
    constexpr bool SPLIT = (MODE == MODE_CLS);    // one group per block, key tiles dealt round-robin to the waves
    const int item = SPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
    if (item >= n_groups<MODE>(g) * qtiles) return;
    const Grp r = decode<MODE>(g, item / qtiles);
    const int q0 = (item % qtiles) * 16;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;

    const int qi = q0 + li;                       // this lane's query (column of S^T)
    const int qi_c = qi < r.nq ? qi : r.nq - 1;
    const bf16* qp = qkv + (size_t)q_row<MODE>(g, r, qi_c) * g.ld + hcol;
    bf16x8 qf[KS];
    ld_frags(qp, gq, qf);

    float m_run = SPLIT ? -1e30f : -INFINITY, l_run = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};

    int nk_eff = r.nk;
    if (MODE == MODE_FULL && g.causal) { const int lim = q0 + 16; nk_eff = lim < r.nk ? lim : r.nk; }

    for (int kt0 = SPLIT ? wave * 64 : 0; kt0 < nk_eff; kt0 += SPLIT ? 256 : 64) {
        const int rows = (nk_eff - kt0) < 64 ? (nk_eff - kt0) : 64;
        stage_tile(vt, (rows + 15) & ~15, lane, qkv, g.ld, 2 * g.W + hcol, [&](int rr) {
            int j = kt0 + rr; j = j < r.nk ? j : r.nk - 1; return k_row<MODE>(g, r, j); });
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t] = (f32x4){0, 0, 0, 0};
            if (t * 16 < rows) {
                int kj = kt0 + t * 16 + li; kj = kj < r.nk ? kj : r.nk - 1;
                const bf16* kp = qkv + (size_t)k_row<MODE>(g, r, kj) * g.ld + g.W + hcol;
                bf16x8 kf[KS];
                ld_frags(kp, gq, kf);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], st[t], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt0 + t * 16 + gq * 4 + e;
                const bool ok = key < r.nk && !(MODE == MODE_FULL && g.causal && key > qi);
                const float s = ok ? st[t][e] * g.scale2 : -INFINITY;
                st[t][e] = s;
                mx = fmaxf(mx, s);
            }
        mx = group_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __builtin_amdgcn_exp2f(st[t][e] - m_new);
                st[t][e] = p;
                rs += p;
            }
        rs = group_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u * 32 < rows) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[2 * u + (j >> 2)][j & 3];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = frag_T<TR>(vt, u, dt, lane);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
    }
    if (SPLIT) {  // merge the 4 partial softmax states (m, l, O) through LDS; wave 0 writes the result
        __syncthreads();
        float* red = (float*)smem;  // [wave][2 + DT*4][64]
        red[(wave * (2 + DT * 4) + 0) * 64 + lane] = m_run;
        red[(wave * (2 + DT * 4) + 1) * 64 + lane] = l_run;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * (2 + DT * 4) + 2 + dt * 4 + e) * 64 + lane] = o[dt][e];
        __syncthreads();
        if (wave != 0) return;
        float mm = m_run;
#pragma unroll
        for (int w = 1; w < 4; ++w) mm = fmaxf(mm, red[(w * (2 + DT * 4) + 0) * 64 + lane]);
        float ll = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __builtin_amdgcn_exp2f(red[(w * (2 + DT * 4) + 0) * 64 + lane] - mm);
            ll += red[(w * (2 + DT * 4) + 1) * 64 + lane] * sc;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[dt][e] += red[(w * (2 + DT * 4) + 2 + dt * 4 + e) * 64 + lane] * sc;
        }
        m_run = mm;
        l_run = ll;
    }
    if (qi < r.nq) {
        const float inv = 1.0f / l_run;
        const int row = q_row<MODE>(g, r, qi);
        bf16* op = out + (size_t)row * ldo + hcol;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 v = {(bf16)(o[dt][0] * inv), (bf16)(o[dt][1] * inv), (bf16)(o[dt][2] * inv), (bf16)(o[dt][3] * inv)};
            *(bf16x4*)(op + dt * 16 + gq * 4) = v;
        }
        if (gq == 0) lse2[(size_t)row * g.heads + r.h] = m_run + log2f(l_run);
    }
}

// ------------------------------------------------------------------------------------------------ D = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ dO, int lddo, const bf16* __restrict__ O,
                                                         int ldo, int rows, int row_step, int heads, float* __restrict__ delta) {
    long rh = (long)blockIdx.x * 256 + threadIdx.x;  // (row, head)
    if (rh >= (long)rows * heads) return;
    const int row = (int)(rh / heads) * row_step, h = (int)(rh % heads);  // row_step = S: the CLS rows only
    rh = (long)row * heads + h;
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const bf16x8 a = ldg8(dO + (size_t)row * lddo + h * DH + ch * 8);
        const bf16x8 b = ldg8(O + (size_t)row * ldo + h * DH + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)b[e];
    }
    delta[rh] = s;
}

// ------------------------------------------------------------------------------------------------ dQ
template <int MODE, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                          const bf16* __restrict__ dO, int lddo,
                                                          const float* __restrict__ lse2, const float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, int lddq) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * VSTRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* kt_lds = smem + wave * 64 * VSTRIDE;
    const int nq_max = (MODE == MODE_SPACE) ? g.n : (MODE == MODE_TIME) ? g.T : (MODE == MODE_CLS) ? 1 : g.S;
    const int qtiles = (nq_max + 15) >> 4;
    constexpr bool SPLIT = (MODE == MODE_CLS);
    const int item = SPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
    if (item >= n_groups<MODE>(g) * qtiles) return;
    const Grp r = decode<MODE>(g, item / qtiles);
    const int q0 = (item % qtiles) * 16;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;

    const int qi = q0 + li;
    const int qi_c = qi < r.nq ? qi : r.nq - 1;
    const int qrow = q_row<MODE>(g, r, qi_c);
    const bf16* qp = qkv + (size_t)qrow * g.ld + hcol;
    const bf16* dop = dO + (size_t)qrow * lddo + hcol;
    bf16x8 qf[KS], dof[KS];
    ld_frags(qp, gq, qf);
    ld_frags(dop, gq, dof);
    const float lse = lse2[(size_t)qrow * g.heads + r.h];
    const float dlt = delta[(size_t)qrow * g.heads + r.h];

    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0, 0, 0, 0};
    int nk_eff = r.nk;
    if (MODE == MODE_FULL && g.causal) { const int lim = q0 + 16; nk_eff = lim < r.nk ? lim : r.nk; }

    for (int kt0 = SPLIT ? wave * 64 : 0; kt0 < nk_eff; kt0 += SPLIT ? 256 : 64) {
        const int rows = (nk_eff - kt0) < 64 ? (nk_eff - kt0) : 64;
        stage_tile(kt_lds, (rows + 15) & ~15, lane, qkv, g.ld, g.W + hcol, [&](int rr) {
            int j = kt0 + rr; j = j < r.nk ? j : r.nk - 1; return k_row<MODE>(g, r, j); });
        f32x4 ds[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ds[t] = (f32x4){0, 0, 0, 0};
            if (t * 16 < rows) {
                int kj = kt0 + t * 16 + li; kj = kj < r.nk ? kj : r.nk - 1;
                const size_t krow = (size_t)k_row<MODE>(g, r, kj) * g.ld;
                const bf16* kp = qkv + krow + g.W + hcol;
                const bf16* vp = qkv + krow + 2 * g.W + hcol;
                f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                bf16x8 kf[KS], vf[KS];
                ld_frags(kp, gq, kf);
                ld_frags(vp, gq, vf);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], dof[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kt0 + t * 16 + gq * 4 + e;
                    const bool ok = key < r.nk && !(MODE == MODE_FULL && g.causal && key > qi);
                    const float p = ok ? __builtin_amdgcn_exp2f(s[e] * g.scale2 - lse) : 0.f;
                    ds[t][e] = p * (dp[e] - dlt) * g.scale;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u * 32 < rows) {
                bf16x8 dsf;
#pragma unroll
                for (int j = 0; j < 8; ++j) dsf[j] = (bf16)ds[2 * u + (j >> 2)][j & 3];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(kt_lds, u, dt, lane), dsf, acc[dt], 0, 0, 0);
            }
        }
    }
    if (SPLIT) {  // sum the 4 partial dQ accumulators through LDS
        __syncthreads();
        float* red = (float*)smem;  // [wave][16][64]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(wave * (DT * 4) + dt * 4 + e) * 64 + lane] = acc[dt][e];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[dt][e] += red[(w * (DT * 4) + dt * 4 + e) * 64 + lane];
    }
    if (qi < r.nq) {
        bf16* dq = dqkv + (size_t)qrow * lddq + hcol;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 v = {(bf16)acc[dt][0], (bf16)acc[dt][1], (bf16)acc[dt][2], (bf16)acc[dt][3]};
            *(bf16x4*)(dq + dt * 16 + gq * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int MODE, bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                           const bf16* __restrict__ dO, int lddo,
                                                           const float* __restrict__ lse2, const float* __restrict__ delta,
                                                           bf16* __restrict__ dqkv, int lddq, float* __restrict__ cls_acc) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * 32 * VSTRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* q_lds = smem + wave * 2 * 32 * VSTRIDE;
    char* do_lds = q_lds + 32 * VSTRIDE;
    constexpr bool EXT = (MODE == MODE_SPACE || MODE == MODE_TIME);
    const int nk_max = (MODE == MODE_SPACE) ? g.n + 1 : (MODE == MODE_TIME) ? g.T + 1 : g.S;
    const int ktiles = (nk_max + 15) >> 4;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_groups<MODE>(g) * ktiles) return;
    const Grp r = decode<MODE>(g, item / ktiles);
    const int k0 = (item % ktiles) * 16;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int nqx = EXT ? r.nq + 1 : r.nq;

    const int kj = k0 + li;                      // this lane's key (column)
    const int kj_c = kj < r.nk ? kj : r.nk - 1;
    const int krow = k_row<MODE>(g, r, kj_c);
    const bf16* kp = qkv + (size_t)krow * g.ld + g.W + hcol;
    const bf16* vp = qkv + (size_t)krow * g.ld + 2 * g.W + hcol;
    bf16x8 kb[KS], vb[KS];
    ld_frags(kp, gq, kb);
    ld_frags(vp, gq, vb);

    f32x4 dv[DT], dk[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }

    int q_begin = 0;
    if (MODE == MODE_FULL && g.causal) q_begin = k0 & ~31;  // queries before the first key of the tile see none of it

    for (int qt0 = q_begin; qt0 < nqx; qt0 += 32) {
        auto rowfn = [&](int rr) { int i = qt0 + rr; i = i < nqx ? i : nqx - 1; return qx_row<MODE>(g, r, i); };
        const int qrows = (nqx - qt0) > 16 ? 32 : 16;
        stage_tile(q_lds, qrows, lane, qkv, g.ld, hcol, rowfn);
        stage_tile(do_lds, qrows, lane, dO, lddo, hcol, rowfn);
        bf16x8 pf, dsf;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
            if (t * 16 < qrows) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(q_lds, t * 16 + li, ks, gq), kb[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(do_lds, t * 16 + li, ks, gq), vb[ks], dp, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qi = qt0 + t * 16 + gq * 4 + e;  // (extended) query index of this row
                bool ok = qi < nqx && kj < r.nk;
                if (MODE == MODE_FULL && g.causal) ok = ok && kj <= qi;
                if (EXT) ok = ok && !(qi == 0 && kj == 0 && r.sub != 0);  // CLS query x CLS key counted once
                float p = 0.f, d = 0.f;
                if (ok) {
                    const int qrow = qx_row<MODE>(g, r, qi);
                    const float lse = lse2[(size_t)qrow * g.heads + r.h];
                    const float dlt = delta[(size_t)qrow * g.heads + r.h];
                    p = __builtin_amdgcn_exp2f(s[e] * g.scale2 - lse);
                    d = p * (dp[e] - dlt) * g.scale;
                }
                pf[t * 4 + e] = (bf16)p;
                dsf[t * 4 + e] = (bf16)d;
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(do_lds, 0, dt, lane), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(q_lds, 0, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
    }
    if (kj < r.nk) {
        if (EXT && kj == 0) {  // CLS key/value: summed over all groups of (b,h)
            float* a = cls_acc + ((size_t)(r.b * g.heads + r.h) * 3) * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(a + dt * 16 + gq * 4 + e, dk[dt][e]);
                    atomicAdd(a + DH + dt * 16 + gq * 4 + e, dv[dt][e]);
                }
        } else {
            bf16* dkp = dqkv + (size_t)krow * lddq + g.W + hcol;
            bf16* dvp = dqkv + (size_t)krow * lddq + 2 * g.W + hcol;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                *(bf16x4*)(dkp + dt * 16 + gq * 4) = (bf16x4){(bf16)dk[dt][0], (bf16)dk[dt][1], (bf16)dk[dt][2], (bf16)dk[dt][3]};
                *(bf16x4*)(dvp + dt * 16 + gq * 4) = (bf16x4){(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
            }
        }
    }
}

// ================================================================================================
// BLOCK-SHARED variants for the geometries whose groups span several 16-row tiles (FULL, SPACE): the four
// waves of a block take four consecutive query (or key) tiles of ONE group and share the K/V (or Q/dO) tiles,
// which are staged once per block -- global -> registers at the top of an iteration, registers -> LDS at its
// bottom (async-stage split), double-buffered, one __syncthreads per tile.  K/Q/dO fragments then come from
// LDS (ds_read_b128) instead of four redundant global fetches.
// ================================================================================================
constexpr int ST64 = (64 * NCH + 255) / 256;  // 16-byte chunks per thread for a 64-row tile (2 for DH 64, 3 for DH 80)
constexpr int ST32 = (32 * NCH + 255) / 256;  // ... for a 32-row tile
struct Stage2 { bf16x8 v[ST64]; };
template <typename RowFn>
__device__ __forceinline__ void stage_load64(Stage2& r, int nrows, int tid, const bf16* base, int ld, int col0, RowFn rowfn) {
#pragma unroll
    for (int i = 0; i < ST64; ++i) {
        const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
        r.v[i] = (row < nrows) ? ldg8(base + (size_t)rowfn(row) * ld + col0 + ch * 8) : zero8();
    }
}
__device__ __forceinline__ void stage_store64(const Stage2& r, char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < ST64; ++i) {
        const int c = tid + 256 * i;
        if (c < 64 * NCH) *(bf16x8*)(tile + (c / NCH) * VSTRIDE + (c % NCH) * 16) = r.v[i];
    }
}
struct Stage1 { bf16x8 v[ST32]; };
template <typename RowFn>
__device__ __forceinline__ Stage1 stage_load32(int nrows, int tid, const bf16* base, int ld, int col0, RowFn rowfn) {
    Stage1 r;
#pragma unroll
    for (int i = 0; i < ST32; ++i) {
        const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
        r.v[i] = (row < nrows) ? ldg8(base + (size_t)rowfn(row) * ld + col0 + ch * 8) : zero8();
    }
    return r;
}
__device__ __forceinline__ void stage_store32(const Stage1& r, char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < ST32; ++i) {
        const int c = tid + 256 * i;
        if (c < 32 * NCH) *(bf16x8*)(tile + (c / NCH) * VSTRIDE + (c % NCH) * 16) = r.v[i];
    }
}
#define TILE_B (64 * VSTRIDE)

template <int MODE, bool TR, bool DROP = false>
__global__ __launch_bounds__(256) void attn_fwd_shared_kernel(AttnGeom g, const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                              int ldo, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];  // [buf][K | V]
    const unsigned long long dseed = DROP ? g.drop_seed[0] + g.drop_site : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq_max = (MODE == MODE_SPACE) ? g.n : g.S;
    const int qblocks = (((nq_max + 15) >> 4) + 3) >> 2;
    const Grp r = decode<MODE>(g, blockIdx.x / qblocks);
    const int qb = blockIdx.x % qblocks;
    const int q0 = (qb * 4 + wave) * 16;
    const bool active = q0 < r.nq;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int qi = q0 + li;
    const int qi_c = qi < r.nq ? qi : r.nq - 1;
    const bf16* qp = qkv + (size_t)q_row<MODE>(g, r, qi_c) * g.ld + hcol;
    bf16x8 qf[KS];
    ld_frags(qp, gq, qf);

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
    const bool causal = (MODE == MODE_FULL) && g.causal;
    int nk_blk = r.nk, nk_w = r.nk;
    if (causal) {
        const int limb = (qb * 4 + 4) * 16, limw = q0 + 16;
        nk_blk = limb < r.nk ? limb : r.nk;
        nk_w = limw < r.nk ? limw : r.nk;
    }
    auto krow = [&](int kt0) { return [&, kt0](int rr) { return k_row<MODE>(g, r, kt0 + rr); }; };
    Stage2 sk, sv;
    {
        const int rows = nk_blk < 64 ? nk_blk : 64;
        stage_load64(sk, rows, tid, qkv, g.ld, g.W + hcol, krow(0));
        stage_load64(sv, rows, tid, qkv, g.ld, 2 * g.W + hcol, krow(0));
        stage_store64(sk, smem, tid);
        stage_store64(sv, smem + TILE_B, tid);
    }
    int buf = 0;
    for (int kt0 = 0; kt0 < nk_blk; kt0 += 64, buf ^= 1) {
        __syncthreads();
        const char* kt_ = smem + buf * 2 * TILE_B;
        const char* vt = kt_ + TILE_B;
        const bool more = kt0 + 64 < nk_blk;
        if (more) {
            const int rows = (nk_blk - kt0 - 64) < 64 ? (nk_blk - kt0 - 64) : 64;
            stage_load64(sk, rows, tid, qkv, g.ld, g.W + hcol, krow(kt0 + 64));
            stage_load64(sv, rows, tid, qkv, g.ld, 2 * g.W + hcol, krow(kt0 + 64));
        }
        if (active && kt0 < nk_w) {
            const int rows = (nk_w - kt0) < 64 ? (nk_w - kt0) : 64;
            f32x4 st[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                st[t] = (f32x4){0, 0, 0, 0};
                if (t * 16 < rows) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(kt_, t * 16 + li, ks, gq), qf[ks], st[t], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
            if (!causal && kt0 + 64 <= r.nk) {  // interior tile (wave-uniform): every key is valid, no mask arithmetic
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st[t][e] *= g.scale2;
                        mx = fmaxf(mx, st[t][e]);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = kt0 + t * 16 + gq * 4 + e;
                        const bool ok = key < r.nk && !(causal && key > qi);
                        const float sv_ = ok ? st[t][e] * g.scale2 : -INFINITY;
                        st[t][e] = sv_;
                        mx = fmaxf(mx, sv_);
                    }
            }
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(st[t][e] - m_new);
                    st[t][e] = p;
                    rs += p;
                }
            rs = group_sum(rs);
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u * 32 < rows) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float pv = st[2 * u + (j >> 2)][j & 3];
                        if (DROP) {  // the row sum above is of the UNdropped probabilities (softmax first, dropout second)
                            const int key = kt0 + (2 * u + (j >> 2)) * 16 + gq * 4 + (j & 3);
                            const unsigned long long idx = ((unsigned long long)(r.b * g.heads + r.h) * g.S + qi) * g.S + key;
                            pv = drop_keep(dseed, idx, g.drop_thr) ? pv * g.drop_inv : 0.f;
                        }
                        pf[j] = (bf16)pv;
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(vt, u, dt, lane), pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (more) {
            char* nb = smem + (buf ^ 1) * 2 * TILE_B;
            stage_store64(sk, nb, tid);
            stage_store64(sv, nb + TILE_B, tid);
        }
    }
    __syncthreads();  // every wave is done with the K/V tiles: their LDS now serves as the output-staging patches
    {
        const float inv = 1.0f / l_run;
        f32x4 on[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) on[dt] = o[dt] * inv;
        store_tile_rows(smem + wave * 1024, on, lane, [&](int rr) -> bf16* {
            const int j = q0 + rr;
            return (active && j < r.nq) ? out + (size_t)q_row<MODE>(g, r, j) * ldo + hcol : nullptr; });
    }
    if (active && qi < r.nq && gq == 0) lse2[(size_t)q_row<MODE>(g, r, qi) * g.heads + r.h] = m_run + log2f(l_run);
}

template <int MODE, bool TR, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_shared_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                 const bf16* __restrict__ dO, int lddo,
                                                                 const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                 bf16* __restrict__ dqkv, int lddq) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];  // [buf][K | V]
    const unsigned long long dseed = DROP ? g.drop_seed[0] + g.drop_site : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq_max = (MODE == MODE_SPACE) ? g.n : g.S;
    const int qblocks = (((nq_max + 15) >> 4) + 3) >> 2;
    const Grp r = decode<MODE>(g, blockIdx.x / qblocks);
    const int qb = blockIdx.x % qblocks;
    const int q0 = (qb * 4 + wave) * 16;
    const bool active = q0 < r.nq;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int qi = q0 + li;
    const int qi_c = qi < r.nq ? qi : r.nq - 1;
    const int qrow = q_row<MODE>(g, r, qi_c);
    const bf16* qp = qkv + (size_t)qrow * g.ld + hcol;
    const bf16* dop = dO + (size_t)qrow * lddo + hcol;
    bf16x8 qf[KS], dof[KS];
    ld_frags(qp, gq, qf);
    ld_frags(dop, gq, dof);
    const float lse = lse2[(size_t)qrow * g.heads + r.h];
    const float dlt = delta[(size_t)qrow * g.heads + r.h];
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0, 0, 0, 0};
    const bool causal = (MODE == MODE_FULL) && g.causal;
    int nk_blk = r.nk, nk_w = r.nk;
    if (causal) {
        const int limb = (qb * 4 + 4) * 16, limw = q0 + 16;
        nk_blk = limb < r.nk ? limb : r.nk;
        nk_w = limw < r.nk ? limw : r.nk;
    }
    auto krow = [&](int kt0) { return [&, kt0](int rr) { return k_row<MODE>(g, r, kt0 + rr); }; };
    Stage2 sk, sv;
    {
        const int rows = nk_blk < 64 ? nk_blk : 64;
        stage_load64(sk, rows, tid, qkv, g.ld, g.W + hcol, krow(0));
        stage_load64(sv, rows, tid, qkv, g.ld, 2 * g.W + hcol, krow(0));
        stage_store64(sk, smem, tid);
        stage_store64(sv, smem + TILE_B, tid);
    }
    int buf = 0;
    for (int kt0 = 0; kt0 < nk_blk; kt0 += 64, buf ^= 1) {
        __syncthreads();
        const char* kt_ = smem + buf * 2 * TILE_B;
        const char* vt = kt_ + TILE_B;
        const bool more = kt0 + 64 < nk_blk;
        if (more) {
            const int rows = (nk_blk - kt0 - 64) < 64 ? (nk_blk - kt0 - 64) : 64;
            stage_load64(sk, rows, tid, qkv, g.ld, g.W + hcol, krow(kt0 + 64));
            stage_load64(sv, rows, tid, qkv, g.ld, 2 * g.W + hcol, krow(kt0 + 64));
        }
        if (active && kt0 < nk_w) {
            const int rows = (nk_w - kt0) < 64 ? (nk_w - kt0) : 64;
            f32x4 ds[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ds[t] = (f32x4){0, 0, 0, 0};
                if (t * 16 < rows) {
                    f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(kt_, t * 16 + li, ks, gq), qf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(vt, t * 16 + li, ks, gq), dof[ks], dp, 0, 0, 0);
                    }
                    if (DROP) {  // dP of the dropped probabilities: dS = P o (m / (1 - p) o dP - D)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = kt0 + t * 16 + gq * 4 + e;
                            const unsigned long long idx = ((unsigned long long)(r.b * g.heads + r.h) * g.S + qi) * g.S + key;
                            dp[e] = drop_keep(dseed, idx, g.drop_thr) ? dp[e] * g.drop_inv : 0.f;
                        }
                    }
                    if (!causal && kt0 + 64 <= r.nk) {  // interior tile (wave-uniform): no mask arithmetic
#pragma unroll
                        for (int e = 0; e < 4; ++e) ds[t][e] = __builtin_amdgcn_exp2f(s[e] * g.scale2 - lse) * (dp[e] - dlt) * g.scale;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = kt0 + t * 16 + gq * 4 + e;
                            const bool ok = key < r.nk && !(causal && key > qi);
                            const float p = ok ? __builtin_amdgcn_exp2f(s[e] * g.scale2 - lse) : 0.f;
                            ds[t][e] = p * (dp[e] - dlt) * g.scale;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u * 32 < rows) {
                    bf16x8 dsf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dsf[j] = (bf16)ds[2 * u + (j >> 2)][j & 3];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(kt_, u, dt, lane), dsf, acc[dt], 0, 0, 0);
                }
            }
        }
        if (more) {
            char* nb = smem + (buf ^ 1) * 2 * TILE_B;
            stage_store64(sk, nb, tid);
            stage_store64(sv, nb + TILE_B, tid);
        }
    }
    __syncthreads();  // the K/V tiles are dead: reuse their LDS as output-staging patches
    store_tile_rows(smem + wave * 1024, acc, lane, [&](int rr) -> bf16* {
        const int j = q0 + rr;
        return (active && j < r.nq) ? dqkv + (size_t)q_row<MODE>(g, r, j) * lddq + hcol : nullptr; });
}

template <int MODE, bool TR, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_shared_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                  const bf16* __restrict__ dO, int lddo,
                                                                  const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                  bf16* __restrict__ dqkv, int lddq, float* __restrict__ cls_acc) {
    // 64 queries per stage (was 32): half the block barriers per key tile -- this kernel ran at half the rate of its dQ twin
    constexpr int QT = 64;
    __shared__ __attribute__((aligned(16))) char smem[4 * QT * VSTRIDE];  // [buf][Q | dO] of QT rows
    __shared__ __attribute__((aligned(16))) float stat[2][2][QT];         // [buf][lse2 | delta][query]
    const unsigned long long dseed = DROP ? g.drop_seed[0] + g.drop_site : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool EXT = (MODE == MODE_SPACE);
    constexpr int HT = QT * VSTRIDE;
    const int nk_max = (MODE == MODE_SPACE) ? g.n + 1 : g.S;
    const int kblocks = (((nk_max + 15) >> 4) + 3) >> 2;
    const Grp r = decode<MODE>(g, blockIdx.x / kblocks);
    const int kb_ = blockIdx.x % kblocks;
    const int k0 = (kb_ * 4 + wave) * 16;
    const bool active = k0 < r.nk;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int nqx = EXT ? r.nq + 1 : r.nq;
    const int kj = k0 + li;
    const int kj_c = kj < r.nk ? kj : r.nk - 1;
    const int krow_ = k_row<MODE>(g, r, kj_c);
    const bf16* kp = qkv + (size_t)krow_ * g.ld + g.W + hcol;
    const bf16* vp = qkv + (size_t)krow_ * g.ld + 2 * g.W + hcol;
    bf16x8 kb[KS], vb[KS];
    ld_frags(kp, gq, kb);
    ld_frags(vp, gq, vb);
    f32x4 dv[DT], dk[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
    const bool causal = (MODE == MODE_FULL) && g.causal;
    const int q_begin = causal ? ((kb_ * 64) & ~(QT - 1)) : 0;  // queries before the block's first key see none of its keys
    auto qrowf = [&](int qt0) { return [&, qt0](int rr) { return qx_row<MODE>(g, r, qt0 + rr); }; };
    Stage2 sq, sd;
    {
        const int rows = (nqx - q_begin) < QT ? (nqx - q_begin) : QT;
        stage_load64(sq, rows, tid, qkv, g.ld, hcol, qrowf(q_begin));
        stage_load64(sd, rows, tid, dO, lddo, hcol, qrowf(q_begin));
        stage_store64(sq, smem, tid);
        stage_store64(sd, smem + HT, tid);
        if (tid < 2 * QT) {
            const int qi = q_begin + (tid % QT);
            const size_t o = (size_t)qx_row<MODE>(g, r, qi < nqx ? qi : nqx - 1) * g.heads + r.h;
            stat[0][tid / QT][tid % QT] = (tid < QT) ? lse2[o] : delta[o];
        }
    }
    float sstat = 0.f;
    int buf = 0;
    for (int qt0 = q_begin; qt0 < nqx; qt0 += QT, buf ^= 1) {
        __syncthreads();
        const char* q_lds = smem + buf * 2 * HT;
        const char* do_lds = q_lds + HT;
        const bool more = qt0 + QT < nqx;
        if (more) {
            const int rows = (nqx - qt0 - QT) < QT ? (nqx - qt0 - QT) : QT;
            stage_load64(sq, rows, tid, qkv, g.ld, hcol, qrowf(qt0 + QT));
            stage_load64(sd, rows, tid, dO, lddo, hcol, qrowf(qt0 + QT));
            if (tid < 2 * QT) {
                const int qi = qt0 + QT + (tid % QT);
                const size_t o = (size_t)qx_row<MODE>(g, r, qi < nqx ? qi : nqx - 1) * g.heads + r.h;
                sstat = (tid < QT) ? lse2[o] : delta[o];
            }
        }
        if (active && !(causal && qt0 + QT <= k0)) {
            const int left = nqx - qt0;
            const int qrows = left >= QT ? QT : ((left + 15) & ~15);
#pragma unroll
            for (int u = 0; u < QT / 32; ++u) {
                if (u * 32 < qrows) {
                    bf16x8 pf, dsf;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int tile = 2 * u + t;
                        f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                        if (tile * 16 < qrows) {
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(q_lds, tile * 16 + li, ks, gq), kb[ks], s, 0, 0, 0);
                                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(do_lds, tile * 16 + li, ks, gq), vb[ks], dp, 0, 0, 0);
                            }
                        }
                        // the lane's 4 consecutive queries: their stats in two 16-byte LDS reads (were 8 scalar ones)
                        const f32x4 l4 = *(const f32x4*)&stat[buf][0][tile * 16 + gq * 4];
                        const f32x4 d4 = *(const f32x4*)&stat[buf][1][tile * 16 + gq * 4];
                        float dm[4] = {1.f, 1.f, 1.f, 1.f};  // dropout factor m / (1 - p) of (query, key) for the lane's 4 queries
                        if (DROP) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int qi = qt0 + tile * 16 + gq * 4 + e;
                                const unsigned long long idx = ((unsigned long long)(r.b * g.heads + r.h) * g.S + qi) * g.S + kj;
                                dm[e] = drop_keep(dseed, idx, g.drop_thr) ? g.drop_inv : 0.f;
                            }
                        }
                        if (!EXT && !causal && qt0 + QT <= nqx && k0 + 16 <= r.nk) {  // interior (wave-uniform): no mask arithmetic
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float p = __builtin_amdgcn_exp2f(s[e] * g.scale2 - l4[e]);
                                pf[t * 4 + e] = (bf16)(DROP ? p * dm[e] : p);
                                dsf[t * 4 + e] = (bf16)(p * ((DROP ? dp[e] * dm[e] : dp[e]) - d4[e]) * g.scale);
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int qi = qt0 + tile * 16 + gq * 4 + e;
                                bool ok = qi < nqx && kj < r.nk;
                                if (causal) ok = ok && kj <= qi;
                                if (EXT) ok = ok && !(qi == 0 && kj == 0 && r.sub != 0);
                                const float p = ok ? __builtin_amdgcn_exp2f(s[e] * g.scale2 - l4[e]) : 0.f;
                                pf[t * 4 + e] = (bf16)(DROP ? p * dm[e] : p);
                                dsf[t * 4 + e] = (bf16)(p * ((DROP ? dp[e] * dm[e] : dp[e]) - d4[e]) * g.scale);
                            }
                        }
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(do_lds, u, dt, lane), pf, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(q_lds, u, dt, lane), dsf, dk[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            char* nb = smem + (buf ^ 1) * 2 * HT;
            stage_store64(sq, nb, tid);
            stage_store64(sd, nb + HT, tid);
            if (tid < 2 * QT) stat[buf ^ 1][tid / QT][tid % QT] = sstat;
        }
    }
    if (active && kj < r.nk && EXT && kj == 0) {
        float* a = cls_acc + ((size_t)(r.b * g.heads + r.h) * 3) * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(a + dt * 16 + gq * 4 + e, dk[dt][e]);
                atomicAdd(a + DH + dt * 16 + gq * 4 + e, dv[dt][e]);
            }
    }
    __syncthreads();  // the Q/dO tiles are dead: reuse their LDS as output-staging patches
    auto krowp = [&](int third) {
        return [&, third](int rr) -> bf16* {
            const int j = k0 + rr;
            return (active && j < r.nk && !(EXT && j == 0)) ? dqkv + (size_t)k_row<MODE>(g, r, j) * lddq + third * g.W + hcol : nullptr; };
    };
    store_tile_rows(smem + wave * 1024, dk, lane, krowp(1));
    store_tile_rows(smem + wave * 1024, dv, lane, krowp(2));
}

// TIME geometry dK/dV: the groups are tiny (T queries x T+1 keys) and very many (B*h*n).  A block takes one
// (b,h) and a CHUNK of patch slots; each wave walks its share of the slots, keeps the CLS key/value gradient of
// all of them in registers, the four waves combine through LDS and ONE set of fp32 atomics per block goes out
// (n/CHUNK per address instead of n).
#define TIME_CHUNK 28
template <bool TR>
__global__ __launch_bounds__(256) void attn_bwd_dkv_time_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                const bf16* __restrict__ dO, int lddo,
                                                                const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                bf16* __restrict__ dqkv, int lddq, float* __restrict__ cls_acc) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * 32 * VSTRIDE];
    __shared__ float stat[4][2][32];
    __shared__ float red[4][2][DH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* q_lds = smem + wave * 2 * 32 * VSTRIDE;
    char* do_lds = q_lds + 32 * VSTRIDE;
    const int chunks = (g.n + TIME_CHUNK - 1) / TIME_CHUNK;
    Grp r;  // block order (b, chunk, h): the heads of one token range run next to each other
    r.h = blockIdx.x % g.heads;
    const int c = (blockIdx.x / g.heads) % chunks;
    r.b = blockIdx.x / (g.heads * chunks); r.nq = g.T; r.nk = g.T + 1;
    const int p_end = (c + 1) * TIME_CHUNK < g.n ? (c + 1) * TIME_CHUNK : g.n;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int nqx = r.nq + 1;
    const int ktiles = (r.nk + 15) >> 4;
    f32x4 cdk[DT], cdv[DT];  // CLS key/value gradient (valid in the lanes whose key column is 0)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { cdk[dt] = (f32x4){0, 0, 0, 0}; cdv[dt] = (f32x4){0, 0, 0, 0}; }

    for (int p = c * TIME_CHUNK + wave; p < p_end; p += 4) {
        r.sub = p;
        for (int kt = 0; kt < ktiles; ++kt) {
            const int k0 = kt * 16;
            const int kj = k0 + li;
            const int kj_c = kj < r.nk ? kj : r.nk - 1;
            const int krow = k_row<MODE_TIME>(g, r, kj_c);
            const bf16* kp = qkv + (size_t)krow * g.ld + g.W + hcol;
            const bf16* vp = qkv + (size_t)krow * g.ld + 2 * g.W + hcol;
            bf16x8 kb[KS], vb[KS];
            ld_frags(kp, gq, kb);
            ld_frags(vp, gq, vb);
            f32x4 dv[DT], dk[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
            for (int qt0 = 0; qt0 < nqx; qt0 += 32) {
                auto rowfn = [&](int rr) { int i = qt0 + rr; i = i < nqx ? i : nqx - 1; return qx_row<MODE_TIME>(g, r, i); };
                const int qrows = (nqx - qt0) > 16 ? 32 : 16;
                stage_tile(q_lds, qrows, lane, qkv, g.ld, hcol, rowfn);
                stage_tile(do_lds, qrows, lane, dO, lddo, hcol, rowfn);
                {
                    const size_t o = (size_t)rowfn(lane & 31) * g.heads + r.h;
                    stat[wave][lane >> 5][lane & 31] = (lane < 32) ? lse2[o] : delta[o];
                }
                bf16x8 pf, dsf;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                    if (t * 16 < qrows) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(q_lds, t * 16 + li, ks, gq), kb[ks], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(do_lds, t * 16 + li, ks, gq), vb[ks], dp, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ql = t * 16 + gq * 4 + e, qi = qt0 + ql;
                        const bool ok = qi < nqx && kj < r.nk && !(qi == 0 && kj == 0 && p != 0);  // CLS x CLS once
                        float pp = 0.f, d = 0.f;
                        if (ok) {
                            pp = __builtin_amdgcn_exp2f(s[e] * g.scale2 - stat[wave][0][ql]);
                            d = pp * (dp[e] - stat[wave][1][ql]) * g.scale;
                        }
                        pf[t * 4 + e] = (bf16)pp;
                        dsf[t * 4 + e] = (bf16)d;
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(do_lds, 0, dt, lane), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T<TR>(q_lds, 0, dt, lane), dsf, dk[dt], 0, 0, 0);
                }
            }
            if (kj < r.nk) {
                if (kj == 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { cdk[dt] += dk[dt]; cdv[dt] += dv[dt]; }
                } else {
                    bf16* dkp = dqkv + (size_t)krow * lddq + g.W + hcol;
                    bf16* dvp = dqkv + (size_t)krow * lddq + 2 * g.W + hcol;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        *(bf16x4*)(dkp + dt * 16 + gq * 4) = (bf16x4){(bf16)dk[dt][0], (bf16)dk[dt][1], (bf16)dk[dt][2], (bf16)dk[dt][3]};
                        *(bf16x4*)(dvp + dt * 16 + gq * 4) = (bf16x4){(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
                    }
                }
            }
        }
    }
    // combine the four waves' CLS partials (held by lanes with li == 0: d = dt*16 + gq*4 + e)
    if (li == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[wave][0][dt * 16 + gq * 4 + e] = cdk[dt][e];
                red[wave][1][dt * 16 + gq * 4 + e] = cdv[dt][e];
            }
    }
    __syncthreads();
    if (threadIdx.x < 2 * DH) {
        const int kv = threadIdx.x / DH, d = threadIdx.x % DH;
        const float v = red[0][kv][d] + red[1][kv][d] + red[2][kv][d] + red[3][kv][d];
        atomicAdd(cls_acc + ((size_t)(r.b * g.heads + r.h) * 3 + kv) * DH + d, v);
    }
}

// ================================================================================================
// FUSED SPACE-geometry backward: one block per (b, frame, head) group does dQ, dK and dV in one launch.
// The group's token set X = {CLS, the n kept patches of the frame} serves as the extended query set AND the key
// set, so its Q / K / V / dO head slices are staged into LDS exactly once (the split dQ + dK/dV passes read them
// twice and once more for D = rowsum(dO*O)).  Phase A: each wave owns query tiles, forms S^T and dP^T against every
// key tile from LDS, gets D_q = sum_k P*dP in registers (exact: all keys of a patch query live in this group), writes
// dQ and leaves D_q in LDS.  Phase B: each wave owns key tiles and accumulates dK / dV over all (extended) queries
// with the stats from LDS; the CLS query takes lse / D from global (its softmax spans every frame), the CLS key /
// value gradient goes through the fp32 cls_acc atomics as before.  Needs n + 1 <= 112 (7 row tiles).
// ================================================================================================
#define FUSED_MAX_TILES 7
#define FUSED_THREADS 512  // 8 waves: one query tile and one key tile each; 2 blocks per CU (LDS) = 4 waves per SIMD
template <bool TR>
__device__ __forceinline__ bf16x8 frag_T_lim(const char* tile, int u, int dt, int lane, int lim, int rs = 1 << 30) {
    const int gq = lane >> 4, i = lane & 15;
    bf16x8 out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row_real = u * 32 + half * 16 + gq * 4;
        // rows >= rs (a multiple of 4) are not stored: the lane group reads a stored 4-row group instead and zeroes what it got
        const int row0 = row_real < rs ? row_real : rs - 4;
        if (u * 32 + half * 16 < lim) {  // wave-uniform: tiles past `lim` rows are not allocated, their P / dS are zero
            if (TR) {
                const char* p = tile + (row0 + (i >> 2)) * VSTRIDE + dt * 32 + (i & 3) * 8;
                const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
                const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[half * 4 + e] = row_real < rs ? tb[e] : (bf16)0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    out[half * 4 + e] = row_real < rs ? *(const bf16*)(tile + (row0 + e) * VSTRIDE + (dt * 16 + i) * 2) : (bf16)0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[half * 4 + e] = (bf16)0.f;
        }
    }
    return out;
}

template <int MT, bool TR>
__global__ __launch_bounds__(FUSED_THREADS, 4) void attn_bwd_space_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                              const bf16* __restrict__ dO, int lddo,
                                                                              const float* __restrict__ lse2,
                                                                              const float* __restrict__ delta,
                                                                              bf16* __restrict__ dqkv, int lddq,
                                                                              float* __restrict__ cls_acc) {
    // MT (row tiles of the group) is a template parameter so that every loop below is straight-line code: the LDS
    // reads of a phase are then issued back to back instead of one dependent round trip per branch.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RA = MT * 16, TB = RA * VSTRIDE, NU = (MT + 1) / 2, NW = FUSED_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Grp r = decode<MODE_SPACE>(g, blockIdx.x);
    const int m = g.n + 1;
    char* Qs = smem;
    char* Ks = Qs + TB;
    char* Vs = Ks + TB;
    char* Ds = Vs + TB;
    float* st_lse = (float*)(Ds + TB);
    float* st_dl = st_lse + RA;
    char* opatch = (char*)(st_dl + RA) + wave * 1024;  // this wave's output-staging patch
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    Q8Out q8o;
    q8o.init(g);

    // ---- stage Q | K | V | dO of the group's tokens (rows m..RA-1 zero), all loads in flight before the first store
    {
        constexpr int PER = (RA * NCH + FUSED_THREADS - 1) / FUSED_THREADS;
        bf16x8 stg[4][PER];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16* base = t == 3 ? dO + hcol : qkv + t * g.W + hcol;
            const int ld = t == 3 ? lddo : g.ld;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = tid + FUSED_THREADS * i, row = c / NCH, ch = c % NCH;
                stg[t][i] = sel8(row < m && !(g.ablate & 4), ldg8(base + (size_t)k_row<MODE_SPACE>(g, r, row < m ? row : 0) * ld + ch * 8));
            }
        }
        {   // (both statistics loaded unconditionally, by every thread, from a row that exists: no wait between them)
            const size_t o = (size_t)k_row<MODE_SPACE>(g, r, tid < m ? tid : 0) * g.heads + r.h;
            const float lv = lse2[o], dv = delta[o];
            if (tid < RA) {
                st_lse[tid] = tid < m ? lv : 0.f;
                st_dl[tid] = tid == 0 ? dv : 0.f;  // CLS query: D from the delta pass over the CLS rows
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = tid + FUSED_THREADS * i, row = c / NCH, ch = c % NCH;
                if (row < RA) *(bf16x8*)(smem + t * TB + row * VSTRIDE + ch * 16) = stg[t][i];
            }
    }
    __syncthreads();

    // ---- phase A: dQ of the query tiles this wave owns (+ D_q into LDS)
    if (!(g.ablate & 1))
    for (int qt = wave; qt < MT; qt += NW) {
        const int qj = qt * 16 + li;
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag_row(Qs, qj, ks, gq); dof[ks] = frag_row(Ds, qj, ks, gq); }
        const float lse = st_lse[qj];
        f32x4 P[2 * NU], dP[2 * NU];
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NU; ++t) {
            P[t] = (f32x4){0, 0, 0, 0};
            dP[t] = (f32x4){0, 0, 0, 0};
            if (t < MT) {
                f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, t * 16 + li, ks, gq), qf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Vs, t * 16 + li, ks, gq), dof[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = t * 16 + gq * 4 + e;
                    const float pe = __builtin_amdgcn_exp2f(sc[e] * g.scale2 - lse);
                    float p = key < m ? pe : 0.f;
                    if (t == 0 && e == 0) p = (gq == 0 && qj == 0 && r.sub != 0) ? 0.f : p;  // CLS x CLS: frame 0 only
                    P[t][e] = p;
                    dP[t][e] = dp[e];
                    part += p * dp[e];
                }
            }
        }
        float dlt = group_sum(part);
        if (qj == 0) dlt = st_dl[0];  // CLS query: its softmax spans every frame, D comes from the delta pass
        f32x4 acc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bf16x8 dsf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 2 * u + (j >> 2), e = j & 3;
                dsf[j] = (bf16)(P[t][e] * (dP[t][e] - dlt) * g.scale);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ks, u, dt, lane, RA), dsf, acc[dt], 0, 0, 0);
        }
        store_tile_rows(opatch, acc, lane, [&](int rr) -> bf16* {
            const int j = qt * 16 + rr;
            return (j >= 1 && j < m) ? dqkv + (size_t)k_row<MODE_SPACE>(g, r, j) * lddq + hcol : nullptr; }, &q8o);
        if (qj == 0) {  // this frame's share of the CLS query gradient
            if (g.cls_parts) {  // its own slot of the per-frame partials (plain stores; summed in frame order by the finalize kernel)
                float* a = cls_acc + (((size_t)(r.b * g.heads + r.h) * g.cls_parts + r.sub) * 3 + 2) * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) *(f32x4*)(a + dt * 16 + gq * 4) = acc[dt];
            } else {
                float* a = cls_acc + ((size_t)(r.b * g.heads + r.h) * 3 + 2) * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(a + dt * 16 + gq * 4 + e, acc[dt][e]);
            }
        }
        if (gq == 0 && qj >= 1) st_dl[qj] = dlt;
    }
    __syncthreads();

    // ---- phase B: dK / dV of the key tiles this wave owns, over all extended queries
    if (!(g.ablate & 2))
    for (int kt = wave; kt < MT; kt += NW) {
        const int kj = kt * 16 + li;
        bf16x8 kb[KS], vb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kb[ks] = frag_row(Ks, kj, ks, gq); vb[ks] = frag_row(Vs, kj, ks, gq); }
        f32x4 dv[DT], dk[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
        const bool cls_dup = kj == 0 && r.sub != 0;  // CLS query x CLS key is counted by frame 0 only
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bf16x8 pf, dsf;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tile = 2 * u + t;
                if (tile < MT) {  // compile-time
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Qs, tile * 16 + li, ks, gq), kb[ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ds, tile * 16 + li, ks, gq), vb[ks], dp, 0, 0, 0);
                    }
                    const f32x4 l4 = *(const f32x4*)(st_lse + tile * 16 + gq * 4);
                    const f32x4 d4 = *(const f32x4*)(st_dl + tile * 16 + gq * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qi = tile * 16 + gq * 4 + e;
                        const bool ok = qi < m && kj < m && !(qi == 0 && cls_dup);
                        const float pe = __builtin_amdgcn_exp2f(sc[e] * g.scale2 - l4[e]);
                        const float pp = ok ? pe : 0.f;
                        pf[t * 4 + e] = (bf16)pp;
                        dsf[t * 4 + e] = (bf16)(pp * (dp[e] - d4[e]) * g.scale);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pf[t * 4 + e] = (bf16)0.f; dsf[t * 4 + e] = (bf16)0.f; }
                }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ds, u, dt, lane, RA), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Qs, u, dt, lane, RA), dsf, dk[dt], 0, 0, 0);
            }
        }
        if (kj == 0) {  // CLS key/value: summed over the frames of (b,h)
            if (g.cls_parts) {
                float* a = cls_acc + (((size_t)(r.b * g.heads + r.h) * g.cls_parts + r.sub) * 3) * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    *(f32x4*)(a + dt * 16 + gq * 4) = dk[dt];
                    *(f32x4*)(a + DH + dt * 16 + gq * 4) = dv[dt];
                }
            } else {
                float* a = cls_acc + ((size_t)(r.b * g.heads + r.h) * 3) * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        atomicAdd(a + dt * 16 + gq * 4 + e, dk[dt][e]);
                        atomicAdd(a + DH + dt * 16 + gq * 4 + e, dv[dt][e]);
                    }
            }
        }
        auto krowp = [&](int third) {
            return [&, third](int rr) -> bf16* {
                const int j = kt * 16 + rr;
                return (j >= 1 && j < m) ? dqkv + (size_t)k_row<MODE_SPACE>(g, r, j) * lddq + third * g.W + hcol : nullptr; };
        };
        store_tile_rows(opatch, dk, lane, krowp(1), &q8o);
        store_tile_rows(opatch, dv, lane, krowp(2), &q8o);
    }
    q8o.finish(g, lane);
}

// FUSED TIME-geometry backward.  The groups (b, patch slot, head) are tiny (T + 1 tokens) and very many, so a WAVE owns
// a group: its Q / K / V / dO rows live in wave-private LDS tiles (the next group's rows are already in flight in
// registers while the current one is computed), phase A and phase B run back to back in the same wave, and the three
// CLS-token sums (dK, dV, dQ) stay in registers over the block's chunk of patch slots -> one set of atomics per block.
template <int MT, bool TR, int RS_ = 0>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 2) void attn_bwd_time_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                  const bf16* __restrict__ dO, int lddo,
                                                                  const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                  bf16* __restrict__ dqkv, int lddq, float* __restrict__ cls_acc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RA = MT * 16, RS = RS_ ? RS_ : RA, TB = RS * VSTRIDE, NU = (MT + 1) / 2;  // RS rows of the RA-row geometry are stored
    constexpr int WB = 4 * TB + 2 * RA * 4 + 3 * DH * 4 + 1024;  // bytes of one wave's region: tiles, stats, CLS sums, output patch
    constexpr int PT = (RS * NCH + 63) / 64;         // 16-byte chunks per lane per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = smem + wave * WB;
    char* Qs = base;
    char* Ks = Qs + TB;
    char* Vs = Ks + TB;
    char* Ds = Vs + TB;
    float* st_lse = (float*)(Ds + TB);
    float* st_dl = st_lse + RA;
    float* csum = st_dl + RA;                        // [dK | dV | dQ][DH] of the CLS token, summed over this wave's groups
    char* opatch = (char*)(csum + 3 * DH);           // output-staging patch
    const int chunks = (g.n + TIME_CHUNK - 1) / TIME_CHUNK;
    Grp r;  // block order (b, chunk, h): the heads of one token range run next to each other
    r.h = blockIdx.x % g.heads;
    const int c = (blockIdx.x / g.heads) % chunks;
    r.b = blockIdx.x / (g.heads * chunks); r.nq = g.T; r.nk = g.T + 1;
    const int p_end = (c + 1) * TIME_CHUNK < g.n ? (c + 1) * TIME_CHUNK : g.n;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int m = g.T + 1;
    const size_t ocls = (size_t)(r.b * g.S) * g.heads + r.h;
    const float lse_c = lse2[ocls], dl_c = delta[ocls];
    for (int i = lane; i < 3 * DH; i += 64) csum[i] = 0.f;
    Q8Out q8o;
    q8o.init(g);

    bf16x8 stg[4][PT];
    float lse_pf = 0.f;
    auto issue = [&](int p) {
        r.sub = p;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16* src = t == 3 ? dO + hcol : qkv + t * g.W + hcol;
            const int ld = t == 3 ? lddo : g.ld;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
                stg[t][i] = row < m ? ldg8(src + (size_t)k_row<MODE_TIME>(g, r, row) * ld + ch * 8) : zero8();
            }
        }
        if (lane >= 1 && lane < m) lse_pf = lse2[(size_t)k_row<MODE_TIME>(g, r, lane) * g.heads + r.h];
    };
    int p = c * TIME_CHUNK + wave;
    if (p < p_end) issue(p);
    for (; p < p_end; p += 4) {
        // registers -> this wave's LDS tiles (rows m..RA-1 arrive as zeros), stats of the extended query rows
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
                if (row < RS) *(bf16x8*)(base + t * TB + row * VSTRIDE + ch * 16) = stg[t][i];
            }
        if (lane < RA) {
            st_lse[lane] = lane == 0 ? lse_c : (lane < m ? lse_pf : 0.f);
            st_dl[lane] = lane == 0 ? dl_c : 0.f;
        }
        const bool first = p == 0;  // the CLS query x CLS key term is counted by patch slot 0 only
        if (p + 4 < p_end) issue(p + 4);
        r.sub = p;

        // ---- phase A: dQ (+ D_q)
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) {
            const int qj = qt * 16 + li;
            bf16x8 qf[KS], dof[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag_row(Qs, qj, ks, gq, RS); dof[ks] = frag_row(Ds, qj, ks, gq, RS); }
            const float lse = st_lse[qj];
            f32x4 P[2 * NU], dP[2 * NU];
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t) {
                P[t] = (f32x4){0, 0, 0, 0};
                dP[t] = (f32x4){0, 0, 0, 0};
                if (t < MT) {
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, t * 16 + li, ks, gq, RS), qf[ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Vs, t * 16 + li, ks, gq, RS), dof[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = t * 16 + gq * 4 + e;
                        const float pe = __builtin_amdgcn_exp2f(sc[e] * g.scale2 - lse);
                        float pp = key < m ? pe : 0.f;
                        if (t == 0 && e == 0) pp = (gq == 0 && qj == 0 && !first) ? 0.f : pp;
                        P[t][e] = pp;
                        dP[t][e] = dp[e];
                        part += pp * dp[e];
                    }
                }
            }
            float dlt = group_sum(part);
            if (qj == 0) dlt = dl_c;
            f32x4 acc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 dsf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 2 * u + (j >> 2), e = j & 3;
                    dsf[j] = (bf16)(P[t][e] * (dP[t][e] - dlt) * g.scale);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ks, u, dt, lane, RA, RS), dsf, acc[dt], 0, 0, 0);
            }
            store_tile_rows(opatch, acc, lane, [&](int rr) -> bf16* {
                const int j = qt * 16 + rr;
                return (j >= 1 && j < m) ? dqkv + (size_t)k_row<MODE_TIME>(g, r, j) * lddq + hcol : nullptr; }, &q8o);
            if (qt == 0 && li == 0) {  // column 0 = the CLS query: this group's share of its gradient
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    f32x4* a = (f32x4*)(csum + 2 * DH + dt * 16 + gq * 4);
                    *a += acc[dt];
                }
            }
            if (gq == 0 && qj >= 1) st_dl[qj] = dlt;
        }

        // ---- phase B: dK / dV
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const int kj = kt * 16 + li;
            bf16x8 kb[KS], vb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kb[ks] = frag_row(Ks, kj, ks, gq, RS); vb[ks] = frag_row(Vs, kj, ks, gq, RS); }
            f32x4 dv[DT], dk[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
            const bool cls_dup = kj == 0 && !first;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 pf, dsf;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tile = 2 * u + t;
                    if (tile < MT) {
                        f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Qs, tile * 16 + li, ks, gq, RS), kb[ks], sc, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ds, tile * 16 + li, ks, gq, RS), vb[ks], dp, 0, 0, 0);
                        }
                        const f32x4 l4 = *(const f32x4*)(st_lse + tile * 16 + gq * 4);
                        const f32x4 d4 = *(const f32x4*)(st_dl + tile * 16 + gq * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int qi = tile * 16 + gq * 4 + e;
                            const bool ok = qi < m && kj < m && !(qi == 0 && cls_dup);
                            const float pe = __builtin_amdgcn_exp2f(sc[e] * g.scale2 - l4[e]);
                            const float pp = ok ? pe : 0.f;
                            pf[t * 4 + e] = (bf16)pp;
                            dsf[t * 4 + e] = (bf16)(pp * (dp[e] - d4[e]) * g.scale);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pf[t * 4 + e] = (bf16)0.f; dsf[t * 4 + e] = (bf16)0.f; }
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ds, u, dt, lane, RA, RS), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Qs, u, dt, lane, RA, RS), dsf, dk[dt], 0, 0, 0);
                }
            }
            if (kt == 0 && li == 0) {  // column 0 = the CLS key / value
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    f32x4* a = (f32x4*)(csum + dt * 16 + gq * 4);
                    *a += dk[dt];
                    f32x4* c2 = (f32x4*)(csum + DH + dt * 16 + gq * 4);
                    *c2 += dv[dt];
                }
            }
            auto krowp = [&](int third) {
                return [&, third](int rr) -> bf16* {
                    const int j = kt * 16 + rr;
                    return (j >= 1 && j < m) ? dqkv + (size_t)k_row<MODE_TIME>(g, r, j) * lddq + third * g.W + hcol : nullptr; };
            };
            store_tile_rows(opatch, dk, lane, krowp(1), &q8o);
            store_tile_rows(opatch, dv, lane, krowp(2), &q8o);
        }
    }
    q8o.finish(g, lane);
    // combine the four waves' CLS sums: one partial per block
    __syncthreads();
    if (threadIdx.x < 3 * DH) {
        const int i = threadIdx.x;
        const float v = *(const float*)(smem + 0 * WB + 4 * TB + 2 * RA * 4 + i * 4) + *(const float*)(smem + 1 * WB + 4 * TB + 2 * RA * 4 + i * 4) +
                        *(const float*)(smem + 2 * WB + 4 * TB + 2 * RA * 4 + i * 4) + *(const float*)(smem + 3 * WB + 4 * TB + 2 * RA * 4 + i * 4);
        if (g.cls_parts) cls_acc[(((size_t)(r.b * g.heads + r.h) * g.cls_parts + c) * 3) * DH + i] = v;  // this chunk's slot
        else atomicAdd(cls_acc + ((size_t)(r.b * g.heads + r.h) * 3) * DH + i, v);
    }
}

// ================================================================================================
// FUSED divided-attention FORWARD.  Same ownership as the fused backward: a block per SPACE group, a wave per TIME
// group.  Every key of a patch query lives in its group, so the softmax is a single pass over register-resident score
// tiles (no online rescaling).  The CLS query attends every token of the clip: each group contributes a partial
// softmax state (max, sum, unnormalised O) over ITS keys -- the CLS key itself only in group 0 -- which a tiny merge
// kernel combines into the CLS output row and its log-sum-exp.  cls_part: [B, heads, G, DH + 2] fp32 with G = T (SPACE)
// or ceil(n / TIME_CHUNK) (TIME: partials are merged per wave in registers and per block through LDS first).
// ================================================================================================
template <int MT, bool TR>
__global__ __launch_bounds__(256, 4) void attn_fwd_space_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                   bf16* __restrict__ out, int ldo, float* __restrict__ lse2,
                                                                   float* __restrict__ cls_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K | V tiles of the group
    constexpr int RA = MT * 16, TB = RA * VSTRIDE, NU = (MT + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Grp r = decode<MODE_SPACE>(g, blockIdx.x);
    const int m = g.n + 1;
    char* Ks = smem;
    char* Vs = Ks + TB;
    char* opatch = Vs + TB + wave * 1024;  // this wave's output-staging patch
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    Q8Out q8o;
    q8o.init(g);
    {
        constexpr int PER = (RA * NCH + 255) / 256;
        bf16x8 stg[2][PER];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
                stg[t][i] = sel8(row < m, ldg8(qkv + (size_t)k_row<MODE_SPACE>(g, r, row < m ? row : 0) * g.ld + (1 + t) * g.W + hcol + ch * 8));
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = tid + 256 * i, row = c / NCH, ch = c % NCH;
                if (row < RA) *(bf16x8*)(smem + t * TB + row * VSTRIDE + ch * 16) = stg[t][i];
            }
    }
    // the first query tile's fragments come straight from global memory while the staging settles
    bf16x8 qf[KS];
    {
        const int qj = wave * 16 + li;
        ld_frags(qkv + (size_t)k_row<MODE_SPACE>(g, r, qj < m ? qj : m - 1) * g.ld + hcol, gq, qf);
    }
    __syncthreads();
    for (int qt = wave; qt < MT; qt += 4) {
        const int qj = qt * 16 + li;
        f32x4 st[2 * NU];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * NU; ++t) {
            st[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (t < MT) {
                f32x4 sc = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, t * 16 + li, ks, gq), qf[ks], sc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = t * 16 + gq * 4 + e;
                    const bool ok = key < m && !(key == 0 && qj == 0 && r.sub != 0);  // CLS x CLS: frame 0 only
                    const float v = ok ? sc[e] * g.scale2 : -INFINITY;
                    st[t][e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = group_max(mx);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NU; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pp = __builtin_amdgcn_exp2f(st[t][e] - mx);
                st[t][e] = pp;
                rs += pp;
            }
        const float l = group_sum(rs);
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[2 * u + (j >> 2)][j & 3];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Vs, u, dt, lane, RA), pf, o[dt], 0, 0, 0);
        }
        {
            const float inv = 1.0f / l;
            f32x4 on[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) on[dt] = o[dt] * inv;
            store_tile_rows(opatch, on, lane, [&](int rr) -> bf16* {
                const int j = qt * 16 + rr;
                return (j >= 1 && j < m) ? out + (size_t)k_row<MODE_SPACE>(g, r, j) * ldo + hcol : nullptr; }, &q8o);
        }
        if (qj >= 1 && qj < m) {
            if (gq == 0) lse2[(size_t)k_row<MODE_SPACE>(g, r, qj) * g.heads + r.h] = mx + log2f(l);
        } else if (qj == 0) {  // this frame's partial softmax state of the CLS query
            float* cp = cls_part + ((size_t)(r.b * g.heads + r.h) * g.T + r.sub) * (DH + 2);
            if (gq == 0) { cp[0] = mx; cp[1] = l; }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cp[2 + dt * 16 + gq * 4 + e] = o[dt][e];
        }
        if (qt + 4 < MT) {
            const int qn = (qt + 4) * 16 + li;
            ld_frags(qkv + (size_t)k_row<MODE_SPACE>(g, r, qn < m ? qn : m - 1) * g.ld + hcol, gq, qf);
        }
    }
    q8o.finish(g, lane);
}

template <int MT, bool TR, int RS_ = 0>
__global__ __launch_bounds__(256) void attn_fwd_time_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                  bf16* __restrict__ out, int ldo, float* __restrict__ lse2,
                                                                  float* __restrict__ cls_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // per wave: V tile | CLS state [DH + 2]
    constexpr int RA = MT * 16, RS = RS_ ? RS_ : RA, TB = RS * VSTRIDE, NU = (MT + 1) / 2;  // RS rows of the RA-row geometry are stored
    constexpr int WB = TB + (DH + 2 + 2) * 4 + 1024;
    constexpr int PT = (RS * NCH + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* Vs = smem + wave * WB;
    char* opatch = Vs + TB + (DH + 4) * 4;  // output-staging patch (behind the V tile and the CLS state)
    const int chunks = (g.n + TIME_CHUNK - 1) / TIME_CHUNK;
    Grp r;
    r.h = blockIdx.x % g.heads;
    const int c = (blockIdx.x / g.heads) % chunks;
    r.b = blockIdx.x / (g.heads * chunks); r.nq = g.T; r.nk = g.T + 1;
    const int p_end = (c + 1) * TIME_CHUNK < g.n ? (c + 1) * TIME_CHUNK : g.n;
    const int gq = lane >> 4, li = lane & 15;
    const int hcol = r.h * DH;
    const int m = g.T + 1;
    float Mr = -1e30f, Lr = 0.f;  // running softmax state of the CLS query over this wave's groups (column 0 lanes)
    f32x4 Or[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) Or[dt] = (f32x4){0, 0, 0, 0};
    Q8Out q8o;
    q8o.init(g);

    bf16x8 qf[MT][KS], kf[MT][KS], vst[PT];
    auto issue = [&](int p) {
        r.sub = p;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int j = t * 16 + li;
            const bf16* rowp = qkv + (size_t)k_row<MODE_TIME>(g, r, j < m ? j : m - 1) * g.ld + hcol;
            ld_frags(rowp, gq, qf[t]);
            ld_frags(rowp + g.W, gq, kf[t]);
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
            vst[i] = row < m ? ldg8(qkv + (size_t)k_row<MODE_TIME>(g, r, row) * g.ld + 2 * g.W + hcol + ch * 8) : zero8();
        }
    };
    int p = c * TIME_CHUNK + wave;
    if (p < p_end) issue(p);
    for (; p < p_end; p += 4) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
            if (row < RS) *(bf16x8*)(Vs + row * VSTRIDE + ch * 16) = vst[i];
        }
        bf16x8 qc[MT][KS], kc[MT][KS];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qc[t][ks] = qf[t][ks]; kc[t][ks] = kf[t][ks]; }
        const bool first = p == 0;
        if (p + 4 < p_end) issue(p + 4);
        r.sub = p;
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) {
            const int qj = qt * 16 + li;
            f32x4 st[2 * NU];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t) {
                st[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (t < MT) {
                    f32x4 sc = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t][ks], qc[qt][ks], sc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = t * 16 + gq * 4 + e;
                        const bool ok = key < m && !(key == 0 && qj == 0 && !first);
                        const float v = ok ? sc[e] * g.scale2 : -INFINITY;
                        st[t][e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            }
            mx = group_max(mx);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pp = __builtin_amdgcn_exp2f(st[t][e] - mx);
                    st[t][e] = pp;
                    rs += pp;
                }
            const float l = group_sum(rs);
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[2 * u + (j >> 2)][j & 3];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Vs, u, dt, lane, RA, RS), pf, o[dt], 0, 0, 0);
            }
            {
                const float inv = 1.0f / l;
                f32x4 on[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) on[dt] = o[dt] * inv;
                store_tile_rows(opatch, on, lane, [&](int rr) -> bf16* {
                    const int j = qt * 16 + rr;
                    return (j >= 1 && j < m) ? out + (size_t)k_row<MODE_TIME>(g, r, j) * ldo + hcol : nullptr; }, &q8o);
            }
            if (qj >= 1 && qj < m && gq == 0) lse2[(size_t)k_row<MODE_TIME>(g, r, qj) * g.heads + r.h] = mx + log2f(l);
            if (qt == 0) {  // column 0 = the CLS query: fold this group's state into the running one
                const float Mn = fmaxf(Mr, mx);
                const float a = __builtin_amdgcn_exp2f(Mr - Mn), b = __builtin_amdgcn_exp2f(mx - Mn);
                Lr = Lr * a + l * b;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) Or[dt] = Or[dt] * a + o[dt] * b;
                Mr = Mn;
            }
        }
    }
    q8o.finish(g, lane);
    // the four waves' CLS states -> one partial per block
    float* cst = (float*)(smem + wave * WB + TB);
    if (li == 0) {
        if (gq == 0) { cst[0] = Mr; cst[1] = Lr; }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) cst[2 + dt * 16 + gq * 4 + e] = Or[dt][e];
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int d = threadIdx.x;
        float M = -1e30f;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, *(const float*)(smem + w * WB + TB));
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* cw = (const float*)(smem + w * WB + TB);
            const float sc = __builtin_amdgcn_exp2f(cw[0] - M);
            L += cw[1] * sc;
            O += cw[2 + d] * sc;
        }
        float* cp = cls_part + ((size_t)(r.b * g.heads + r.h) * chunks + c) * (DH + 2);
        if (d == 0) { cp[0] = M; cp[1] = L; }
        cp[2 + d] = O;
    }
}

// ================================================================================================
// FUSED short-sequence FULL attention (S <= 32: the CLIP text tower's 32-token captions, v2/CLIP/clip/model.py:185-187 with
// the causal mask of :330-336).  The (sequence, head) groups are tiny and many -- 6 144 groups of 32 x 64 at 192 pairs -- and
// the streaming kernels spend a block, two of its four waves idle, and one round trip per 64-key tile on each.  Here a WAVE
// owns a group, like in the TIME geometry: every key of a query lives in the group, so the forward softmax is one pass over
// register-resident score tiles, the backward gets D = sum_k P * dP in registers (no delta pass over dO and O) and does dQ
// (phase A) and dK / dV (phase B) from wave-private LDS tiles that are staged once; the next group's rows are in flight in
// registers while the current one is computed; no block-level synchronisation at all.  Waves walk the groups with a grid
// stride (head index fastest: the waves of a block read neighbouring 128-byte head slices of the same token rows).
// ================================================================================================
template <int MT, bool TR>
__global__ __launch_bounds__(256) void attn_fwd_seq_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                 bf16* __restrict__ out, int ldo, float* __restrict__ lse2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // per wave: V tile | output-staging patch
    constexpr int RA = MT * 16, TB = RA * VSTRIDE, NU = (MT + 1) / 2;
    constexpr int WB = TB + 1024;
    constexpr int PT = (RA * NCH + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* Vs = smem + wave * WB;
    char* opatch = Vs + TB;
    const int gq = lane >> 4, li = lane & 15;
    const int m = g.S, groups = g.B * g.heads, stride = gridDim.x * 4;
    const bool causal = g.causal != 0;

    bf16x8 qf[MT][KS], kf[MT][KS], vst[PT];
    auto issue = [&](int gid) {
        const bf16* base = qkv + (size_t)(gid / g.heads) * g.S * g.ld + (gid % g.heads) * DH;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int j = t * 16 + li;
            const bf16* rowp = base + (size_t)(j < m ? j : m - 1) * g.ld;
            ld_frags(rowp, gq, qf[t]);
            ld_frags(rowp + g.W, gq, kf[t]);
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
            vst[i] = row < m ? ldg8(base + (size_t)row * g.ld + 2 * g.W + ch * 8) : zero8();
        }
    };
    int gid = blockIdx.x * 4 + wave;
    if (gid < groups) issue(gid);
    for (; gid < groups; gid += stride) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
            if (row < RA) *(bf16x8*)(Vs + row * VSTRIDE + ch * 16) = vst[i];
        }
        bf16x8 qc[MT][KS], kc[MT][KS];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qc[t][ks] = qf[t][ks]; kc[t][ks] = kf[t][ks]; }
        if (gid + stride < groups) issue(gid + stride);
        const int h = gid % g.heads;
        const size_t row0 = (size_t)(gid / g.heads) * g.S;
        const int hcol = h * DH;
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) {
            const int qj = qt * 16 + li;
            f32x4 st[2 * NU];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t) {
                st[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (t < MT) {
                    f32x4 sc = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[t][ks], qc[qt][ks], sc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = t * 16 + gq * 4 + e;
                        const bool ok = key < m && !(causal && key > qj);
                        const float v = ok ? sc[e] * g.scale2 : -INFINITY;
                        st[t][e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            }
            mx = group_max(mx);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pp = __builtin_amdgcn_exp2f(st[t][e] - mx);
                    st[t][e] = pp;
                    rs += pp;
                }
            const float l = group_sum(rs);
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[2 * u + (j >> 2)][j & 3];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Vs, u, dt, lane, RA), pf, o[dt], 0, 0, 0);
            }
            {
                const float inv = 1.0f / l;
                f32x4 on[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) on[dt] = o[dt] * inv;
                store_tile_rows(opatch, on, lane, [&](int rr) -> bf16* {
                    const int j = qt * 16 + rr;
                    return j < m ? out + (row0 + j) * ldo + hcol : nullptr; });
            }
            if (qj < m && gq == 0) lse2[(row0 + qj) * g.heads + h] = mx + log2f(l);
        }
    }
}

#define SEQ_BWD_WAVES 2  // 21 KiB of tiles per wave (S = 32): two-wave blocks, three of them per CU
template <int MT, bool TR>
__global__ __launch_bounds__(64 * SEQ_BWD_WAVES) void attn_bwd_seq_fused_kernel(AttnGeom g, const bf16* __restrict__ qkv,
                                                                               const bf16* __restrict__ dO, int lddo,
                                                                               const float* __restrict__ lse2,
                                                                               bf16* __restrict__ dqkv, int lddq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RA = MT * 16, TB = RA * VSTRIDE, NU = (MT + 1) / 2;
    constexpr int WB = 4 * TB + 2 * RA * 4 + 1024;  // one wave's region: Q | K | V | dO tiles, lse and D of the query rows, output patch
    constexpr int PT = (RA * NCH + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = smem + wave * WB;
    char* Qs = base;
    char* Ks = Qs + TB;
    char* Vs = Ks + TB;
    char* Ds = Vs + TB;
    float* st_lse = (float*)(Ds + TB);
    float* st_dl = st_lse + RA;
    char* opatch = (char*)(st_dl + RA);
    const int gq = lane >> 4, li = lane & 15;
    const int m = g.S, groups = g.B * g.heads, stride = gridDim.x * SEQ_BWD_WAVES;
    const bool causal = g.causal != 0;

    bf16x8 stg[4][PT];
    float lse_pf = 0.f;
    auto issue = [&](int gid) {
        const int h = gid % g.heads;
        const size_t row0 = (size_t)(gid / g.heads) * g.S;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16* src = t == 3 ? dO + h * DH : qkv + t * g.W + h * DH;
            const int ld = t == 3 ? lddo : g.ld;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
                stg[t][i] = row < m ? ldg8(src + (row0 + row) * ld + ch * 8) : zero8();
            }
        }
        if (lane < m) lse_pf = lse2[(row0 + lane) * g.heads + h];
    };
    int gid = blockIdx.x * SEQ_BWD_WAVES + wave;
    if (gid < groups) issue(gid);
    for (; gid < groups; gid += stride) {
        // registers -> this wave's LDS tiles (rows m..RA-1 arrive as zeros)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int cc = lane + 64 * i, row = cc / NCH, ch = cc % NCH;
                if (row < RA) *(bf16x8*)(base + t * TB + row * VSTRIDE + ch * 16) = stg[t][i];
            }
        if (lane < RA) st_lse[lane] = lane < m ? lse_pf : 0.f;
        if (gid + stride < groups) issue(gid + stride);
        const int hcol = (gid % g.heads) * DH;
        const size_t row0 = (size_t)(gid / g.heads) * g.S;

        // ---- phase A: dQ (+ D_q = sum_k P * dP, exact: every key of the query is in this group)
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) {
            const int qj = qt * 16 + li;
            bf16x8 qf[KS], dof[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag_row(Qs, qj, ks, gq); dof[ks] = frag_row(Ds, qj, ks, gq); }
            const float lse = st_lse[qj];
            f32x4 P[2 * NU], dP[2 * NU];
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * NU; ++t) {
                P[t] = (f32x4){0, 0, 0, 0};
                dP[t] = (f32x4){0, 0, 0, 0};
                if (t < MT) {
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, t * 16 + li, ks, gq), qf[ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Vs, t * 16 + li, ks, gq), dof[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = t * 16 + gq * 4 + e;
                        const bool ok = key < m && !(causal && key > qj);
                        const float pp = ok ? __builtin_amdgcn_exp2f(sc[e] * g.scale2 - lse) : 0.f;
                        P[t][e] = pp;
                        dP[t][e] = dp[e];
                        part += pp * dp[e];
                    }
                }
            }
            const float dlt = group_sum(part);
            f32x4 acc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 dsf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 2 * u + (j >> 2), e = j & 3;
                    dsf[j] = (bf16)(P[t][e] * (dP[t][e] - dlt) * g.scale);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ks, u, dt, lane, RA), dsf, acc[dt], 0, 0, 0);
            }
            store_tile_rows(opatch, acc, lane, [&](int rr) -> bf16* {
                const int j = qt * 16 + rr;
                return j < m ? dqkv + (row0 + j) * lddq + hcol : nullptr; });
            if (gq == 0) st_dl[qj] = dlt;
        }

        // ---- phase B: dK / dV
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const int kj = kt * 16 + li;
            bf16x8 kb[KS], vb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kb[ks] = frag_row(Ks, kj, ks, gq); vb[ks] = frag_row(Vs, kj, ks, gq); }
            f32x4 dv[DT], dk[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dv[dt] = (f32x4){0, 0, 0, 0}; dk[dt] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bf16x8 pf, dsf;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tile = 2 * u + t;
                    if (tile < MT) {
                        f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Qs, tile * 16 + li, ks, gq), kb[ks], sc, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ds, tile * 16 + li, ks, gq), vb[ks], dp, 0, 0, 0);
                        }
                        const f32x4 l4 = *(const f32x4*)(st_lse + tile * 16 + gq * 4);
                        const f32x4 d4 = *(const f32x4*)(st_dl + tile * 16 + gq * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int qi = tile * 16 + gq * 4 + e;
                            const bool ok = qi < m && kj < m && !(causal && kj > qi);
                            const float pp = ok ? __builtin_amdgcn_exp2f(sc[e] * g.scale2 - l4[e]) : 0.f;
                            pf[t * 4 + e] = (bf16)pp;
                            dsf[t * 4 + e] = (bf16)(pp * (dp[e] - d4[e]) * g.scale);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pf[t * 4 + e] = (bf16)0.f; dsf[t * 4 + e] = (bf16)0.f; }
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Ds, u, dt, lane, RA), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_T_lim<TR>(Qs, u, dt, lane, RA), dsf, dk[dt], 0, 0, 0);
                }
            }
            auto krowp = [&](int third) {
                return [&, third](int rr) -> bf16* {
                    const int j = kt * 16 + rr;
                    return j < m ? dqkv + (row0 + j) * lddq + third * g.W + hcol : nullptr; };
            };
            store_tile_rows(opatch, dk, lane, krowp(1));
            store_tile_rows(opatch, dv, lane, krowp(2));
        }
    }
}

// CLS output row = merge of the G partial softmax states of (b, h)
__global__ void attn_cls_merge_kernel(const float* __restrict__ cls_part, int G, int heads, int S, bf16* __restrict__ out,
                                      int ldo, float* __restrict__ lse2, unsigned char* __restrict__ q8 = nullptr,
                                      const float* __restrict__ q8_scale = nullptr, float* __restrict__ q8_amax = nullptr) {
    const int bh = blockIdx.x, d = threadIdx.x;
    const bool live = d < DH;  // (no early exit: every lane takes part in the amax reduction below)
    float ax = 0.f;
    if (live) {
        const float* cp = cls_part + (size_t)bh * G * (DH + 2);
        float M = -1e30f;
        for (int i = 0; i < G; ++i) M = fmaxf(M, cp[(size_t)i * (DH + 2)]);
        float L = 0.f, O = 0.f;
        for (int i = 0; i < G; ++i) {
            const float sc = __builtin_amdgcn_exp2f(cp[(size_t)i * (DH + 2)] - M);
            L += cp[(size_t)i * (DH + 2) + 1] * sc;
            O += cp[(size_t)i * (DH + 2) + 2 + d] * sc;
        }
        const int b = bh / heads, h = bh % heads;
        const size_t row = (size_t)b * S;
        const bf16 ov = (bf16)(O / L);
        out[row * ldo + h * DH + d] = ov;
        if (q8) {  // the CLS row's bytes of the per-tensor e4m3 copy (one byte per thread: 48 rows of 58 416, plumbing)
            const float sc = q8_scale[0], x = (float)ov;
            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(x * (sc > 0.f ? 1.0f / sc : 1.0f), -448.0f), 448.0f), 0.f, 0, false);
            q8[row * ldo + h * DH + d] = (unsigned char)(pk & 0xff);
            ax = fabsf(x);
        }
        if (d == 0) lse2[row * heads + h] = M + log2f(L);
    }
    if (q8) amax_publish(q8_amax, wave_max(ax), threadIdx.x & 63);
}

// cls_acc [B, heads, parts, 3, DH] = fp32 partial sums of (dK, dV, dQ) of the CLS token (parts = 1: the atomically accumulated
// sums of the streaming path; parts = T / the number of TIME chunks: one partial per block of the fused kernels, added here in
// slot order -> run-to-run reproducible) -> bf16 into the CLS row of dqkv; the dQ slot is only used by the fused kernels (the
// split path writes the CLS dQ from its own CLS-query pass).
__global__ void attn_cls_finalize_kernel(const float* __restrict__ cls_acc, int B, int heads, int S, int W, int with_q,
                                         bf16* __restrict__ dqkv, int lddq, int parts, unsigned char* __restrict__ q8 = nullptr,
                                         const float* __restrict__ q8_scale = nullptr, float* __restrict__ q8_amax = nullptr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (b, h, slot, d)
    const int d = idx % DH, slot = (idx / DH) % 3, h = (idx / (3 * DH)) % heads, b = idx / (3 * DH * heads);
    const bool live = idx < B * heads * 3 * DH && !(slot == 2 && !with_q);
    float ax = 0.f;
    if (live) {
        const int third = slot == 2 ? 0 : 1 + slot;
        const float* src = cls_acc + ((size_t)(b * heads + h) * parts * 3 + slot) * DH + d;
        float v = 0.f;
        for (int pidx = 0; pidx < parts; ++pidx) v += src[(size_t)pidx * 3 * DH];
        const bf16 ov = (bf16)v;
        dqkv[(size_t)(b * S) * lddq + third * W + h * DH + d] = ov;
        if (q8) {
            const float sc = q8_scale[0], x = (float)ov;
            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(x * (sc > 0.f ? 1.0f / sc : 1.0f), -448.0f), 448.0f), 0.f, 0, false);
            q8[(size_t)(b * S) * lddq + third * W + h * DH + d] = (unsigned char)(pk & 0xff);
            ax = fabsf(x);
        }
    }
    if (q8) amax_publish(q8_amax, wave_max(ax), threadIdx.x & 63);
}

}  // namespace NS_DH
using namespace NS_DH;

// ------------------------------------------------------------------------------------------------ C ABI
// Per-call dispatch options (include/tvts_hip.h: TVTS_ATTN_NO_TR / _NO_SHARED / _NO_FUSED; 0 = what the step uses).  The library
// keeps no mutable process state: the alternative kernel paths (scalar LDS reads instead of ds_read_tr, per-wave instead of
// block-shared staging, the split passes instead of the fused single-launch kernels) are reachable per call, for parity tests and
// benches, and cannot leak from one caller or thread into another.
#define ATTN_OPTS(opts)                                                                      \
    const bool use_tr = !((opts) & 1), shared = !((opts) & 2), fused = !((opts) & 4);        \
    (void)use_tr; (void)shared; (void)fused

static int make_geom(AttnGeom& g, int mode, int B, int heads, int S, int T, int n, int causal, int ld) {
    if (B <= 0 || heads <= 0 || S <= 0 || ld % 8) return TVTS_EINVAL;
    if ((mode == MODE_SPACE || mode == MODE_TIME) && (T <= 0 || n <= 0 || S != 1 + T * n)) return TVTS_EINVAL;
    if (mode < MODE_FULL || mode > MODE_CLS) return TVTS_EINVAL;
    g.B = B; g.heads = heads; g.S = S; g.T = T; g.n = n; g.causal = causal; g.ld = ld; g.W = heads * DH; g.kv_len = nullptr;
    g.cls_nq = 1; g.cls_q0 = 0; g.cls_qpos = nullptr; g.cls_parts = 0;
    g.drop_thr = 0; g.drop_inv = 1.f; g.drop_seed = nullptr; g.drop_site = 0;
    g.q8_ref = nullptr; g.q8 = nullptr; g.q8_scale = nullptr; g.q8_amax = nullptr;
    g.scale = 1.0f / sqrtf((float)DH);
    g.scale2 = g.scale * 1.4426950408889634f;
    g.q8_only = 0;
    g.ablate = 0;  // timing-ablation bits of the fused backward (opts bits 4..6 of tvts_attn_bwd; results are wrong by construction)
    return TVTS_OK;
}
static int items_q(const AttnGeom& g, int mode) {
    const int nq = mode == MODE_SPACE ? g.n : mode == MODE_TIME ? g.T : mode == MODE_CLS ? 1 : g.S;
    const int groups = mode == MODE_SPACE ? g.B * g.heads * g.T : mode == MODE_TIME ? g.B * g.heads * g.n : g.B * g.heads;
    return groups * ceil_div(nq, 16);
}
static int items_k(const AttnGeom& g, int mode) {
    const int nk = mode == MODE_SPACE ? g.n + 1 : mode == MODE_TIME ? g.T + 1 : g.S;
    const int groups = mode == MODE_SPACE ? g.B * g.heads * g.T : mode == MODE_TIME ? g.B * g.heads * g.n : g.B * g.heads;
    return groups * ceil_div(nk, 16);
}

#define DISPATCH_MODE(KERNEL, mode, ...)                                                        \
    do {                                                                                        \
        if (use_tr) {                                                                         \
            switch (mode) {                                                                     \
                case MODE_FULL: hipLaunchKernelGGL((KERNEL<MODE_FULL, true>), __VA_ARGS__); break;   \
                case MODE_SPACE: hipLaunchKernelGGL((KERNEL<MODE_SPACE, true>), __VA_ARGS__); break; \
                case MODE_TIME: hipLaunchKernelGGL((KERNEL<MODE_TIME, true>), __VA_ARGS__); break;   \
                case MODE_CLS: hipLaunchKernelGGL((KERNEL<MODE_CLS, true>), __VA_ARGS__); break;     \
                default: return TVTS_EINVAL;                                                    \
            }                                                                                   \
        } else {                                                                                \
            switch (mode) {                                                                     \
                case MODE_FULL: hipLaunchKernelGGL((KERNEL<MODE_FULL, false>), __VA_ARGS__); break;   \
                case MODE_SPACE: hipLaunchKernelGGL((KERNEL<MODE_SPACE, false>), __VA_ARGS__); break; \
                case MODE_TIME: hipLaunchKernelGGL((KERNEL<MODE_TIME, false>), __VA_ARGS__); break;   \
                case MODE_CLS: hipLaunchKernelGGL((KERNEL<MODE_CLS, false>), __VA_ARGS__); break;     \
                default: return TVTS_EINVAL;                                                    \
            }                                                                                   \
        }                                                                                       \
    } while (0)

// out[rows, ldo] (heads merged, the layout the output projection consumes), lse2[rows, heads]
struct DropArgs { float p; const unsigned long long* seed; unsigned long long site; };
static int set_drop(AttnGeom& g, const DropArgs* d) {
    if (!d || d->p <= 0.f) return TVTS_OK;
    if (d->p >= 1.f || !d->seed) return TVTS_EINVAL;
    g.drop_thr = (unsigned)((double)d->p * 4294967296.0);
    g.drop_inv = 1.0f / (1.0f - d->p);
    g.drop_seed = d->seed;
    g.drop_site = d->site;
    return TVTS_OK;
}

static int fwd_impl(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const int* kv_len,
                    void* out, int ldo, float* lse2, int opts, hipStream_t stream, const DropArgs* drop = nullptr) {
    ATTN_OPTS(opts);
    AttnGeom g;
    int rc = make_geom(g, mode, B, heads, S, T, n, causal, ld);
    if (rc) return rc;
    if (ldo % 4 || (kv_len && mode != MODE_FULL)) return TVTS_EINVAL;
    g.kv_len = kv_len;
    if ((rc = set_drop(g, drop))) return rc;
    if (g.drop_thr) {  // attention-probability dropout: the block-shared FULL kernel with ds_read_tr fragments only
        if (mode != MODE_FULL) return TVTS_EINVAL;
        const int nb = g.B * g.heads * ceil_div(ceil_div(g.S, 16), 4);
        hipLaunchKernelGGL((attn_fwd_shared_kernel<MODE_FULL, true, true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (fused && use_tr && mode == MODE_FULL && !kv_len && S <= 32) {  // short sequences (text tower): a wave per (sequence, head)
        const int MT = ceil_div(S, 16), groups = B * heads;
        const int lds_bytes = 4 * (MT * 16 * VSTRIDE + 1024);
        const int blocks = ceil_div(groups, 4) < 1536 ? ceil_div(groups, 4) : 1536;
        if (MT == 1) hipLaunchKernelGGL((attn_fwd_seq_fused_kernel<1, true>), dim3(blocks), dim3(256), lds_bytes, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
        else hipLaunchKernelGGL((attn_fwd_seq_fused_kernel<2, true>), dim3(blocks), dim3(256), lds_bytes, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (shared && (mode == MODE_FULL || mode == MODE_SPACE)) {
        const int nq = mode == MODE_SPACE ? g.n : g.S;
        const int groups = mode == MODE_SPACE ? g.B * g.heads * g.T : g.B * g.heads;
        const int nb = groups * ceil_div(ceil_div(nq, 16), 4);
        if (mode == MODE_FULL) { if (use_tr) hipLaunchKernelGGL((attn_fwd_shared_kernel<MODE_FULL, true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
                                 else hipLaunchKernelGGL((attn_fwd_shared_kernel<MODE_FULL, false>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2); }
        else { if (use_tr) hipLaunchKernelGGL((attn_fwd_shared_kernel<MODE_SPACE, true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
               else hipLaunchKernelGGL((attn_fwd_shared_kernel<MODE_SPACE, false>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2); }
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const int blocks = mode == MODE_CLS ? items_q(g, mode) : ceil_div(items_q(g, mode), 4);
    DISPATCH_MODE(attn_fwd_kernel, mode, dim3(blocks), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

extern "C" int ABI(fwd)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                             void* out, int ldo, float* lse2, int opts, hipStream_t stream) {
    return fwd_impl(mode, qkv, ld, B, heads, S, T, n, causal, nullptr, out, ldo, lse2, opts, stream);
}
// FULL attention over sequences padded to S: keys at positions >= kv_len[b] are masked out (the attention_mask of the v1
// text tower, transformers DistilBERT MultiHeadSelfAttention); kv_len is a device int32[B]
extern "C" int ABI(fwd_len)(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo,
                                 float* lse2, hipStream_t stream) {
    if (!kv_len) return TVTS_EINVAL;
    return fwd_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, out, ldo, lse2, 0, stream);
}

extern "C" int ABI(delta)(const void* dO, int lddo, const void* O, int ldo, int rows, int heads, float* delta,
                               hipStream_t stream) {
    if (rows <= 0 || heads <= 0 || lddo % 8 || ldo % 8) return TVTS_EINVAL;
    const long total = (long)rows * heads * 8;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)dO, lddo,
                       (const bf16*)O, ldo, rows, 1, heads, delta);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

static int bwd_dq_impl(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const int* kv_len,
                       const void* dO, int lddo, const float* lse2, const float* delta, void* dqkv, int lddq,
                       int opts, hipStream_t stream, const DropArgs* drop = nullptr) {
    ATTN_OPTS(opts);
    AttnGeom g;
    int rc = make_geom(g, mode, B, heads, S, T, n, causal, ld);
    if (rc) return rc;
    if (lddo % 8 || lddq % 4) return TVTS_EINVAL;
    g.kv_len = kv_len;
    if ((rc = set_drop(g, drop))) return rc;
    if (g.drop_thr) {
        if (mode != MODE_FULL) return TVTS_EINVAL;
        const int nb = g.B * g.heads * ceil_div(ceil_div(g.S, 16), 4);
        hipLaunchKernelGGL((attn_bwd_dq_shared_kernel<MODE_FULL, true, true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv,
                           (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (shared && (mode == MODE_FULL || mode == MODE_SPACE)) {
        const int nq = mode == MODE_SPACE ? g.n : g.S;
        const int groups = mode == MODE_SPACE ? g.B * g.heads * g.T : g.B * g.heads;
        const int nb = groups * ceil_div(ceil_div(nq, 16), 4);
#define DQ_ARGS dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq
        if (mode == MODE_FULL) { if (use_tr) hipLaunchKernelGGL((attn_bwd_dq_shared_kernel<MODE_FULL, true>), DQ_ARGS);
                                 else hipLaunchKernelGGL((attn_bwd_dq_shared_kernel<MODE_FULL, false>), DQ_ARGS); }
        else { if (use_tr) hipLaunchKernelGGL((attn_bwd_dq_shared_kernel<MODE_SPACE, true>), DQ_ARGS);
               else hipLaunchKernelGGL((attn_bwd_dq_shared_kernel<MODE_SPACE, false>), DQ_ARGS); }
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const int blocks = mode == MODE_CLS ? items_q(g, mode) : ceil_div(items_q(g, mode), 4);
    DISPATCH_MODE(attn_bwd_dq_kernel, mode, dim3(blocks), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

extern "C" int ABI(bwd_dq)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                                const void* dO, int lddo, const float* lse2, const float* delta, void* dqkv, int lddq,
                                int opts, hipStream_t stream) {
    return bwd_dq_impl(mode, qkv, ld, B, heads, S, T, n, causal, nullptr, dO, lddo, lse2, delta, dqkv, lddq, opts, stream);
}

// cls_acc: fp32 [B, heads, 3, dh] (dK | dV | dQ of the CLS token), zeroed by the caller before the SPACE/TIME pass, consumed by
// tvts_attn_cls_finalize afterwards (unused for FULL).
static int bwd_dkv_impl(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const int* kv_len,
                        const void* dO, int lddo, const float* lse2, const float* delta, void* dqkv, int lddq,
                        float* cls_acc, int opts, hipStream_t stream, const DropArgs* drop = nullptr) {
    ATTN_OPTS(opts);
    AttnGeom g;
    if (mode == MODE_CLS) return TVTS_EINVAL;  // the CLS query is folded into the SPACE/TIME pass
    int rc = make_geom(g, mode, B, heads, S, T, n, causal, ld);
    if (rc) return rc;
    g.kv_len = kv_len;
    if ((rc = set_drop(g, drop))) return rc;
    if (g.drop_thr) {
        if (mode != MODE_FULL || lddo % 8 || lddq % 4) return TVTS_EINVAL;
        const int nb = g.B * g.heads * ceil_div(ceil_div(g.S, 16), 4);
        hipLaunchKernelGGL((attn_bwd_dkv_shared_kernel<MODE_FULL, true, true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv,
                           (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq, cls_acc);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (lddo % 8 || lddq % 4) return TVTS_EINVAL;
    if ((mode == MODE_SPACE || mode == MODE_TIME) && !cls_acc) return TVTS_EINVAL;
    if (shared && (mode == MODE_FULL || mode == MODE_SPACE)) {
        const int nk = mode == MODE_SPACE ? g.n + 1 : g.S;
        const int groups = mode == MODE_SPACE ? g.B * g.heads * g.T : g.B * g.heads;
        const int nb = groups * ceil_div(ceil_div(nk, 16), 4);
#define DKV_ARGS dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq, cls_acc
        if (mode == MODE_FULL) { if (use_tr) hipLaunchKernelGGL((attn_bwd_dkv_shared_kernel<MODE_FULL, true>), DKV_ARGS);
                                 else hipLaunchKernelGGL((attn_bwd_dkv_shared_kernel<MODE_FULL, false>), DKV_ARGS); }
        else { if (use_tr) hipLaunchKernelGGL((attn_bwd_dkv_shared_kernel<MODE_SPACE, true>), DKV_ARGS);
               else hipLaunchKernelGGL((attn_bwd_dkv_shared_kernel<MODE_SPACE, false>), DKV_ARGS); }
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (shared && mode == MODE_TIME) {
        const int nb = g.B * g.heads * ceil_div(g.n, TIME_CHUNK);
        if (use_tr) hipLaunchKernelGGL((attn_bwd_dkv_time_kernel<true>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq, cls_acc);
        else hipLaunchKernelGGL((attn_bwd_dkv_time_kernel<false>), dim3(nb), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, delta, (bf16*)dqkv, lddq, cls_acc);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const int blocks = ceil_div(items_k(g, mode), 4);
    DISPATCH_MODE(attn_bwd_dkv_kernel, mode, dim3(blocks), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq, cls_acc);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

extern "C" int ABI(bwd_dkv)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                                 const void* dO, int lddo, const float* lse2, const float* delta, void* dqkv, int lddq,
                                 float* cls_acc, int opts, hipStream_t stream) {
    return bwd_dkv_impl(mode, qkv, ld, B, heads, S, T, n, causal, nullptr, dO, lddo, lse2, delta, dqkv, lddq, cls_acc, opts, stream);
}
// whole backward of FULL attention with padded keys masked (see fwd_len).  The dK / dV rows of the padded positions are not
// written: the caller zeroes dqkv beforehand (their true gradient is zero).
extern "C" int ABI(bwd_len)(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                                 const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq,
                                 hipStream_t stream) {
    if (!kv_len || lddo % 8 || ldo % 8 || lddq % 4) return TVTS_EINVAL;
    int rc = ABI(delta)(dO, lddo, O, ldo, B * S, heads, delta, stream);
    if (rc) return rc;
    rc = bwd_dq_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, dO, lddo, lse2, delta, dqkv, lddq, 0, stream);
    if (rc) return rc;
    return bwd_dkv_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, dO, lddo, lse2, delta, dqkv, lddq, nullptr, 0, stream);
}

// the same pair with dropout on the attention probabilities (transformers MultiHeadSelfAttention in training mode, reached from
// v1/model/model_dist_TVTS.py:33-34,131-141): weights = dropout_p(softmax(scores)), the mask drawn from the counter-based
// generator of AttnGeom (seed = seed_dev[0] + site: the same call pair regenerates the same mask; p = 0 is the plain pair)
extern "C" int ABI(fwd_len_drop)(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo,
                                 float* lse2, float p, const long* seed_dev, long site, hipStream_t stream) {
    if (!kv_len) return TVTS_EINVAL;
    const DropArgs d = {p, (const unsigned long long*)seed_dev, (unsigned long long)site};
    return fwd_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, out, ldo, lse2, 0, stream, &d);
}
extern "C" int ABI(bwd_len_drop)(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                                 const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float p,
                                 const long* seed_dev, long site, hipStream_t stream) {
    if (!kv_len || lddo % 8 || ldo % 8 || lddq % 4) return TVTS_EINVAL;
    const DropArgs d = {p, (const unsigned long long*)seed_dev, (unsigned long long)site};
    int rc = ABI(delta)(dO, lddo, O, ldo, B * S, heads, delta, stream);  // D = rowsum(dO o O) with the DROPPED output: unchanged
    if (rc) return rc;
    rc = bwd_dq_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, dO, lddo, lse2, delta, dqkv, lddq, 0, stream, &d);
    if (rc) return rc;
    return bwd_dkv_impl(MODE_FULL, qkv, ld, B, heads, S, 0, 0, 0, kv_len, dO, lddo, lse2, delta, dqkv, lddq, nullptr, 0, stream, &d);
}

// ---- TAIL queries: FULL geometry (no mask) in which only the LAST nq (<= 16) tokens of every sequence are queries, all S tokens
// keys -- the last block of the transcript-sorting head, whose output the model reads at the NT transcript positions only
// (v2/model/sort_transformer.py:131-141: x = self.norm(x[:, x_len:])).  out / lse2 / dO / O / delta are indexed by token row like
// everywhere else (only the query rows are touched); dqkv receives dQ in the query rows and dK / dV in every row -- the caller
// zeroes the dQ third of the other rows.  The CLS-query kernels do the work: one block per (sequence, head), key tiles dealt to
// the four waves.
static __global__ __launch_bounds__(256) void attn_delta_tail_kernel(const bf16* __restrict__ dO, int lddo, const bf16* __restrict__ O,
                                                              int ldo, int B, int S, int nq, const int* __restrict__ qpos, int heads,
                                                              float* __restrict__ delta) {
    const long rh = (long)blockIdx.x * 256 + threadIdx.x;  // (sequence, query, head)
    if (rh >= (long)B * nq * heads) return;
    const int h = (int)(rh % heads), qi = (int)((rh / heads) % nq), b = (int)(rh / ((long)heads * nq));
    const size_t row = (size_t)b * S + (qpos ? qpos[b] : S - nq) + qi;
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const bf16x8 a = ldg8(dO + row * lddo + h * DH + ch * 8);
        const bf16x8 c = ldg8(O + row * ldo + h * DH + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
    }
    delta[row * heads + h] = s;
}
extern "C" int ABI(fwd_tail)(const void* qkv, int ld, int B, int heads, int S, int nq, void* out, int ldo, float* lse2,
                             hipStream_t stream) {
    ATTN_OPTS(0);
    AttnGeom g;
    int rc = make_geom(g, MODE_CLS, B, heads, S, 0, 0, 0, ld);
    if (rc) return rc;
    if (nq < 1 || nq > 16 || nq > S || ldo % 4) return TVTS_EINVAL;
    g.cls_nq = nq; g.cls_q0 = S - nq;
    DISPATCH_MODE(attn_fwd_kernel, MODE_CLS, dim3(B * heads), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
extern "C" int ABI(bwd_tail)(const void* qkv, int ld, int B, int heads, int S, int nq, const void* dO, int lddo, const void* O,
                             int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream) {
    ATTN_OPTS(0);
    AttnGeom g;
    int rc = make_geom(g, MODE_CLS, B, heads, S, 0, 0, 0, ld);
    if (rc) return rc;
    if (nq < 1 || nq > 16 || nq > S || lddo % 8 || ldo % 8 || lddq % 4) return TVTS_EINVAL;
    g.cls_nq = nq; g.cls_q0 = S - nq;
    const long total = (long)B * nq * heads;
    hipLaunchKernelGGL(attn_delta_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)dO, lddo,
                       (const bf16*)O, ldo, B, S, nq, (const int*)nullptr, heads, delta);
    DISPATCH_MODE(attn_bwd_dq_kernel, MODE_CLS, dim3(B * heads), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq);
    const int blocks = ceil_div(B * heads * ceil_div(S, 16), 4);
    DISPATCH_MODE(attn_bwd_dkv_kernel, MODE_CLS, dim3(blocks), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq, (float*)nullptr);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---- ONE query per sequence at token qpos[b] (device int32[B]) that sees the keys 0 .. qpos[b]: the last block of the CLIP text
// tower, whose output the model reads at the EOT token only (v2/CLIP/clip/model.py:343-354: x[arange, text.argmax(-1)]).  Same
// conventions as the tail form; dK / dV of the keys behind the query are not written (their gradient is zero): zero dqkv first.
extern "C" int ABI(fwd_rowq)(const void* qkv, int ld, int B, int heads, int S, const int* qpos, void* out, int ldo, float* lse2,
                             hipStream_t stream) {
    ATTN_OPTS(0);
    AttnGeom g;
    int rc = make_geom(g, MODE_CLS, B, heads, S, 0, 0, 0, ld);
    if (rc) return rc;
    if (!qpos || ldo % 4) return TVTS_EINVAL;
    g.cls_qpos = qpos;
    DISPATCH_MODE(attn_fwd_kernel, MODE_CLS, dim3(B * heads), dim3(256), 0, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
extern "C" int ABI(bwd_rowq)(const void* qkv, int ld, int B, int heads, int S, const int* qpos, const void* dO, int lddo,
                             const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream) {
    ATTN_OPTS(0);
    AttnGeom g;
    int rc = make_geom(g, MODE_CLS, B, heads, S, 0, 0, 0, ld);
    if (rc) return rc;
    if (!qpos || lddo % 8 || ldo % 8 || lddq % 4) return TVTS_EINVAL;
    g.cls_qpos = qpos;
    const long total = (long)B * heads;
    hipLaunchKernelGGL(attn_delta_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)dO, lddo,
                       (const bf16*)O, ldo, B, S, 1, qpos, heads, delta);
    DISPATCH_MODE(attn_bwd_dq_kernel, MODE_CLS, dim3(B * heads), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq);
    const int blocks = ceil_div(B * heads * ceil_div(S, 16), 4);
    DISPATCH_MODE(attn_bwd_dkv_kernel, MODE_CLS, dim3(blocks), dim3(256), 0, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                  lse2, delta, (bf16*)dqkv, lddq, (float*)nullptr);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

extern "C" int ABI(cls_finalize)(const float* cls_acc, int B, int heads, int S, void* dqkv, int lddq,
                                      hipStream_t stream) {
    const int total = B * heads * 3 * DH;
    hipLaunchKernelGGL(attn_cls_finalize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, cls_acc, B, heads, S,
                       heads * DH, 0, (bf16*)dqkv, lddq, 1);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

static __global__ void zero_f32_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// Whole backward of one attention site: D = rowsum(dO*O), dQ, dK, dV (and, for the divided space / time geometries,
// the CLS query and the CLS key/value reduction).  SPACE groups that fit 112 rows take the fused single-launch kernel.
static int bwd_impl(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                    const void* dO, int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv,
                    int lddq, float* cls_acc, long cls_acc_elems, void* q8out, int ldq8, const float* q8_scale, float* q8_amax,
                    int opts, hipStream_t stream) {
    if (mode == MODE_CLS) return TVTS_EINVAL;
    ATTN_OPTS(opts);
    AttnGeom g;
    int rc = make_geom(g, mode, B, heads, S, T, n, causal, ld);
    if (rc) return rc;
    if (lddo % 8 || ldo % 8 || lddq % 4) return TVTS_EINVAL;
    if (q8out) {
        if (ldq8 != lddq || lddq % 8 || !q8_scale) return TVTS_EINVAL;
        g.q8_ref = (const bf16*)dqkv; g.q8 = (unsigned char*)q8out; g.q8_scale = q8_scale; g.q8_amax = q8_amax;
    }
    const bool divided = mode == MODE_SPACE || mode == MODE_TIME;
    if (fused && use_tr && mode == MODE_FULL && S <= 32) {  // short sequences (text tower): one launch, D in registers
        const int MT = ceil_div(S, 16), groups = B * heads;
        const int lds_bytes = SEQ_BWD_WAVES * (4 * MT * 16 * VSTRIDE + 2 * MT * 16 * 4 + 1024);
        const int want = ceil_div(groups, SEQ_BWD_WAVES);
        const int blocks = want < 768 ? want : 768;
        if (MT == 1) hipLaunchKernelGGL((attn_bwd_seq_fused_kernel<1, true>), dim3(blocks), dim3(64 * SEQ_BWD_WAVES), lds_bytes, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, (bf16*)dqkv, lddq);
        else hipLaunchKernelGGL((attn_bwd_seq_fused_kernel<2, true>), dim3(blocks), dim3(64 * SEQ_BWD_WAVES), lds_bytes, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo, lse2, (bf16*)dqkv, lddq);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const bool fused_space = fused && use_tr && mode == MODE_SPACE && n + 1 <= FUSED_MAX_TILES * 16;
    const bool fused_time = fused && use_tr && mode == MODE_TIME && T + 1 <= 32;
    // cls_acc: fp32 scratch for the CLS token's dK / dV / dQ sums.  With room for one partial per block of the fused kernels
    // (B * heads * parts * 3 * dh elements, parts = T for SPACE, ceil(n / TIME_CHUNK) for TIME) the blocks store their shares and
    // the finalize kernel adds them in order (no atomics: reproducible); with B * heads * 3 * dh elements only, or on the streaming
    // path, the shares are accumulated with fp32 atomics.
    const int want_parts = fused_space ? T : fused_time ? ceil_div(n, TIME_CHUNK) : 0;
    const bool parts_ok = want_parts > 0 && cls_acc && cls_acc_elems >= (long)B * heads * want_parts * 3 * DH;
    g.cls_parts = parts_ok ? want_parts : 0;
    g.ablate = (opts >> 4) & 7;
    g.q8_only = (q8out && (opts & 128)) ? 1 : 0;
    if (divided && !parts_ok) {
        if (!cls_acc || cls_acc_elems < (long)B * heads * 3 * DH) return TVTS_EINVAL;
        // a kernel, not hipMemsetAsync: captured memset nodes did not reliably zero this buffer on hipGraph replay (ROCm 7.x:
        // the second and later replays of the captured training step produced garbage CLS gradients, tools/dbg/graph_vs_eager.py)
        const int nz = B * heads * 3 * DH;
        hipLaunchKernelGGL(zero_f32_kernel, dim3(ceil_div(nz, 256)), dim3(256), 0, stream, cls_acc, nz);
    }
    typedef void (*FusedKern)(AttnGeom, const bf16*, const bf16*, int, const float*, const float*, bf16*, int, float*);
    if (fused_space || fused_time) {
        // D for the CLS rows only (the patch rows get theirs inside the fused kernels)
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)(((long)B * heads + 255) / 256)), dim3(256), 0, stream,
                           (const bf16*)dO, lddo, (const bf16*)O, ldo, B, S, heads, delta);
        FusedKern kern = nullptr;
        int lds_bytes = 0, blocks = 0, threads = 0, slot = 0;
        if (fused_space) {
            const int MT = (n + 1 + 15) / 16, RA = MT * 16;
            lds_bytes = 4 * RA * VSTRIDE + 2 * RA * (int)sizeof(float) + (FUSED_THREADS / 64) * 1024;
            switch (MT) {
                case 1: kern = attn_bwd_space_fused_kernel<1, true>; break;
                case 2: kern = attn_bwd_space_fused_kernel<2, true>; break;
                case 3: kern = attn_bwd_space_fused_kernel<3, true>; break;
                case 4: kern = attn_bwd_space_fused_kernel<4, true>; break;
                case 5: kern = attn_bwd_space_fused_kernel<5, true>; break;
                case 6: kern = attn_bwd_space_fused_kernel<6, true>; break;
                default: kern = attn_bwd_space_fused_kernel<7, true>; break;
            }
            blocks = B * heads * T; threads = FUSED_THREADS; slot = MT;
        } else {
            const int MT = (T + 1 + 15) / 16, RA = MT * 16;
            // 17 .. 20 rows (ViT-H/14: 16 frames + CLS): 20 of the 32 rows are stored, two blocks fit a CU instead of one
            const int RS = (MT == 2 && T + 1 <= 20) ? 20 : RA;
            lds_bytes = 4 * (4 * RS * VSTRIDE + 2 * RA * 4 + 3 * DH * 4 + 1024);
            kern = MT == 1 ? attn_bwd_time_fused_kernel<1, true> : RS == 20 ? attn_bwd_time_fused_kernel<2, true, 20> : attn_bwd_time_fused_kernel<2, true>;
            blocks = B * heads * ceil_div(n, TIME_CHUNK); threads = 256; slot = FUSED_MAX_TILES + MT;
        }
        (void)slot;  // (no memo of the attribute call: the library keeps no mutable state; the call is a host-side table write)
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return TVTS_EINVAL;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, stream, g, (const bf16*)qkv, (const bf16*)dO, lddo,
                           lse2, delta, (bf16*)dqkv, lddq, cls_acc);
        TVTS_LAUNCH_CHECK();
        const int total = B * heads * 3 * DH;  // CLS row of dqkv: dK, dV and dQ all come from the accumulators
        hipLaunchKernelGGL(attn_cls_finalize_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, cls_acc, B, heads, S,
                           heads * DH, 1, (bf16*)dqkv, lddq, g.cls_parts ? g.cls_parts : 1, g.q8, g.q8_scale, g.q8_amax);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (q8out) return TVTS_EINVAL;  // only the fused kernels write the copy
    rc = ABI(delta)(dO, lddo, O, ldo, B * S, heads, delta, stream);
    if (rc) return rc;
    rc = ABI(bwd_dq)(mode, qkv, ld, B, heads, S, T, n, causal, dO, lddo, lse2, delta, dqkv, lddq, opts, stream);
    if (rc) return rc;
    rc = ABI(bwd_dkv)(mode, qkv, ld, B, heads, S, T, n, causal, dO, lddo, lse2, delta, dqkv, lddq, cls_acc, opts, stream);
    if (rc) return rc;
    if (divided) {
        rc = ABI(bwd_dq)(MODE_CLS, qkv, ld, B, heads, S, T, n, 0, dO, lddo, lse2, delta, dqkv, lddq, opts, stream);
        if (rc) return rc;
        rc = ABI(cls_finalize)(cls_acc, B, heads, S, dqkv, lddq, stream);
    }
    return rc;
}
extern "C" int ABI(bwd)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                        const void* dO, int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv,
                        int lddq, float* cls_acc, long cls_acc_elems, int opts, hipStream_t stream) {
    return bwd_impl(mode, qkv, ld, B, heads, S, T, n, causal, dO, lddo, O, ldo, lse2, delta, dqkv, lddq, cls_acc, cls_acc_elems,
                    nullptr, 0, nullptr, nullptr, opts, stream);
}
// the same, with the per-tensor e4m3 copy of dqkv written by the kernels themselves (the output gradient of the qkv projection's
// input-gradient and weight-gradient GEMMs, BASELINE config 5); ldq8 == lddq; fused divided geometries only
extern "C" int ABI(bwd_q8)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal,
                           const void* dO, int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv,
                           int lddq, float* cls_acc, long cls_acc_elems, void* q8out, int ldq8, const float* q8_scale,
                           float* q8_amax, int opts, hipStream_t stream) {
    if (!q8out) return TVTS_EINVAL;
    return bwd_impl(mode, qkv, ld, B, heads, S, T, n, causal, dO, lddo, O, ldo, lse2, delta, dqkv, lddq, cls_acc, cls_acc_elems,
                    q8out, ldq8, q8_scale, q8_amax, opts, stream);
}

// Forward of one divided-attention site, patch rows AND the CLS row: fused single-pass kernels + the CLS merge where the
// groups fit (SPACE n + 1 <= 112, TIME T + 1 <= 32), the streaming kernels + the CLS-query kernel otherwise.
// cls_ws: fp32 scratch, at least B * heads * max(T, ceil(n / 28)) * (dh + 2) elements.
static int fwd_divided_impl(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                            float* lse2, float* cls_ws, long cls_ws_elems, void* q8out, int ldq8, const float* q8_scale,
                            float* q8_amax, int opts, hipStream_t stream) {
    if (mode != MODE_SPACE && mode != MODE_TIME) return TVTS_EINVAL;
    ATTN_OPTS(opts);
    AttnGeom g;
    int rc = make_geom(g, mode, B, heads, S, T, n, 0, ld);
    if (rc) return rc;
    if (ldo % 4) return TVTS_EINVAL;
    if (q8out) {  // the e4m3 copy is addressed like the bf16 output: same leading dimension (in bytes), 8-byte pieces
        if (ldq8 != ldo || ldo % 8 || !q8_scale) return TVTS_EINVAL;
        g.q8_ref = (const bf16*)out; g.q8 = (unsigned char*)q8out; g.q8_scale = q8_scale; g.q8_amax = q8_amax;
        g.q8_only = (opts & 128) ? 1 : 0;
    }
    const bool fs = fused && use_tr && mode == MODE_SPACE && n + 1 <= FUSED_MAX_TILES * 16;
    const bool ft = fused && use_tr && mode == MODE_TIME && T + 1 <= 32;
    const int G = mode == MODE_SPACE ? T : ceil_div(n, TIME_CHUNK);
    if ((fs || ft) && cls_ws && cls_ws_elems >= (long)B * heads * G * (DH + 2)) {
        typedef void (*Kern)(AttnGeom, const bf16*, bf16*, int, float*, float*);
        Kern kern = nullptr;
        int lds_bytes, blocks;
        if (fs) {
            const int MT = (n + 1 + 15) / 16;
            switch (MT) {
                case 1: kern = attn_fwd_space_fused_kernel<1, true>; break;
                case 2: kern = attn_fwd_space_fused_kernel<2, true>; break;
                case 3: kern = attn_fwd_space_fused_kernel<3, true>; break;
                case 4: kern = attn_fwd_space_fused_kernel<4, true>; break;
                case 5: kern = attn_fwd_space_fused_kernel<5, true>; break;
                case 6: kern = attn_fwd_space_fused_kernel<6, true>; break;
                default: kern = attn_fwd_space_fused_kernel<7, true>; break;
            }
            lds_bytes = 2 * MT * 16 * VSTRIDE + 4 * 1024;
            blocks = B * heads * T;
        } else {
            const int MT = (T + 1 + 15) / 16;
            const int RS = (MT == 2 && T + 1 <= 20) ? 20 : MT * 16;
            kern = MT == 1 ? attn_fwd_time_fused_kernel<1, true> : RS == 20 ? attn_fwd_time_fused_kernel<2, true, 20> : attn_fwd_time_fused_kernel<2, true>;
            lds_bytes = 4 * (RS * VSTRIDE + (DH + 4) * 4 + 1024);
            blocks = B * heads * G;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, stream, g, (const bf16*)qkv, (bf16*)out, ldo, lse2, cls_ws);
        TVTS_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_cls_merge_kernel, dim3(B * heads), dim3(DH <= 64 ? 64 : 128), 0, stream, cls_ws, G, heads, S,
                           (bf16*)out, ldo, lse2, g.q8, g.q8_scale, g.q8_amax);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    if (q8out) return TVTS_EINVAL;  // only the fused kernels write the copy
    rc = ABI(fwd)(mode, qkv, ld, B, heads, S, T, n, 0, out, ldo, lse2, opts, stream);
    if (rc) return rc;
    return ABI(fwd)(MODE_CLS, qkv, ld, B, heads, S, T, n, 0, out, ldo, lse2, opts, stream);
}
extern "C" int ABI(fwd_divided)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                                float* lse2, float* cls_ws, long cls_ws_elems, int opts, hipStream_t stream) {
    return fwd_divided_impl(mode, qkv, ld, B, heads, S, T, n, out, ldo, lse2, cls_ws, cls_ws_elems, nullptr, 0, nullptr, nullptr, opts, stream);
}
// the same, with the per-tensor e4m3 copy of the output written by the kernels themselves (BASELINE config 5: the attention output is
// the operand of the projection's forward and weight-gradient GEMMs): q8out[r, c] = e4m3(out[r, c] / q8_scale[0]), ldq8 == ldo,
// q8_amax = max(q8_amax, max |out|).  Fused geometries only (-22 otherwise).
extern "C" int ABI(fwd_divided_q8)(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                                   float* lse2, float* cls_ws, long cls_ws_elems, void* q8out, int ldq8, const float* q8_scale,
                                   float* q8_amax, int opts, hipStream_t stream) {
    if (!q8out) return TVTS_EINVAL;
    return fwd_divided_impl(mode, qkv, ld, B, heads, S, T, n, out, ldo, lse2, cls_ws, cls_ws_elems, q8out, ldq8, q8_scale, q8_amax, opts, stream);
}
