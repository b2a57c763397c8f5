// LayerNorm forward / backward for the fp32 residual stream (HBM-bound; one wave64 per row, float4 lanes).
//
// forward : y = (x - mean) * rstd * gamma + beta, x fp32 (optionally gathered rows), y bf16 or fp32;
//           mean / rstd are saved for the backward pass.
// backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ res1 + res2],  g = dy * gamma
//           res1 fp32 or bf16 (the residual-stream gradient), res2 bf16 (a side-branch gradient that only exists as a GEMM
//           operand anyway); dx fp32 and / or a bf16 copy that feeds the next MFMA GEMM; dgamma / dbeta accumulated
//           from per-block partial sums: written to a workspace and combined by a second tiny kernel (or, without
//           a workspace, one fp32 atomic per column per block).
// Reference: nn.LayerNorm as used at v2/model/video_encoder_ViT_B_16.py:79-85 (eps 1e-5, fp32) and
// v2/model/sort_transformer.py:99 (eps 1e-6).
#include "common.h"
#include <type_traits>

#define LN_MAX_IT 5  // 5 * 256 = 1280 columns max

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <>
__device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    *(bf16x4*)p = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

// Both kernels are templated on IT = ceil(W / 256) (register arrays sized to the row, no dead iterations) and walk
// their rows in a grid-stride loop with the NEXT row's loads issued before the current row's reductions: a wave always
// has two rows of traffic in flight instead of one load -> reduce -> store round trip at a time.
// Q8: additionally write the row as OCP e4m3 bytes with ONE scale per row (amax of the bf16-rounded outputs / 448): the fp8
// operand of the next GEMM (BASELINE config 4) leaves the LayerNorm that produced it, bit-identical to tvts_quant_fp8_rows
// run on y, without another pass over the activation.
template <typename TO, int IT, typename TX = float, bool Q8 = false, bool CLS = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, int ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int M, int W, TO* __restrict__ y, int ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     unsigned char* __restrict__ q8 = nullptr, int ldq = 0,
                                                     float* __restrict__ row_scale = nullptr,
                                                     const float* __restrict__ tscale = nullptr, float* __restrict__ amax_acc = nullptr,
                                                     const float* __restrict__ cls_x = nullptr, int cls_period = 0,
                                                     TX* x_refresh = nullptr) {
    // cls_x / cls_period (the hybrid residual stream, round 5): row r with r % cls_period == 0 is the CLS token of clip
    // r / cls_period; its value is carried in fp32 in the compact side array cls_x[M / cls_period][W] and read from THERE (the
    // stream's own row r is stale), and x_refresh row r receives its rounding, so that every later reader of the stream (the
    // residual epilogues, the backward) sees the exact value rounded once instead of an accumulated bf16 sum
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * 4;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float run_amax = 0.f;
    f32x4 gm[IT], bt[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = lane * 4 + it * 256;
        gm[it] = c < W ? load4<float>(gamma + c) : (f32x4){0, 0, 0, 0};
        bt[it] = c < W ? load4<float>(beta + c) : (f32x4){0, 0, 0, 0};
    }
    const float invW = 1.0f / (float)W;
    f32x4 v[IT], nx[IT];
    auto load_row = [&](int rr, f32x4 (&d)[IT]) {
        // (unconditional loads, zeroed afterwards: see ln_bwd_kernel)
        if (CLS && rr % cls_period == 0) {  // (wave-uniform: a wave owns the row)
            const float* cp = cls_x + (size_t)(rr / cls_period) * W;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = lane * 4 + it * 256;
                const f32x4 v = load4<float>(cp + (c < W ? c : 0));
                d[it] = c < W ? v : (f32x4){0, 0, 0, 0};
            }
            return;
        }
        const TX* xp = x + (size_t)(rows ? rows[rr] : rr) * ldx;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = lane * 4 + it * 256;
            const f32x4 v = load4<TX>(xp + (c < W ? c : 0));
            d[it] = c < W ? v : (f32x4){0, 0, 0, 0};
        }
    };
    load_row(r, v);
    for (; r < M; r += stride) {
        const bool more = r + stride < M;
        if (more) load_row(r + stride, nx);
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        const float mean = wave_sum(s) * invW;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[it][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invW + eps);
        f32x4 ob[Q8 ? IT : 1];
        float am = 0.f;
        const bool refresh = CLS && x_refresh && r % cls_period == 0;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * gm[it][e] + bt[it][e];
                store4(y + (size_t)r * ldy + c, o);
                if (refresh) store4(x_refresh + (size_t)r * ldx + c, v[it]);
                if (Q8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ob[it][e] = (float)(bf16)o[e];  // the value the bf16 consumer sees
                        am = fmaxf(am, fabsf(ob[it][e]));
                    }
                }
            }
        }
        if (Q8) {
            am = wave_max(am);
            run_amax = fmaxf(run_amax, am);
            const float scale = q8_scale(tscale, am);
            const float inv = 1.0f / scale;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = lane * 4 + it * 256;
                if (c < W) {
                    float f[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(ob[it][e] * inv, -448.0f), 448.0f);
                    int pk = 0;
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], pk, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
                    *(int*)(q8 + (size_t)r * ldq + c) = pk;
                }
            }
            if (lane == 0 && row_scale) row_scale[r] = scale;
        }
        if (lane == 0) {
            if (mean_out) mean_out[r] = mean;
            if (rstd_out) rstd_out[r] = rstd;
        }
        if (more) {
#pragma unroll
            for (int it = 0; it < IT; ++it) v[it] = nx[it];
        }
    }
    if (Q8) amax_publish(amax_acc, run_amax, lane);
}

template <typename TO, typename TX, bool CLS = false>
static void launch_ln_fwd(int it, dim3 grid, hipStream_t stream, const TX* x, int ldx, const int* rows, const float* gamma,
                          const float* beta, float eps, int M, int W, TO* y, int ldy, float* mean, float* rstd,
                          const float* cls_x = nullptr, int cls_period = 0, TX* x_refresh = nullptr) {
#define LN_FWD_CASE(N) case N: hipLaunchKernelGGL((ln_fwd_kernel<TO, N, TX, false, CLS>), grid, dim3(256), 0, stream, x, ldx, rows, gamma, beta, eps, M, W, y, ldy, mean, rstd, (unsigned char*)nullptr, 0, (float*)nullptr, (const float*)nullptr, (float*)nullptr, cls_x, cls_period, x_refresh); break;
    switch (it) { LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) default: LN_FWD_CASE(5) }
#undef LN_FWD_CASE
}

template <typename TX, bool CLS = false>
static void launch_ln_fwd_q8(int it, dim3 grid, hipStream_t stream, const TX* x, int ldx, const int* rows, const float* gamma,
                             const float* beta, float eps, int M, int W, bf16* y, int ldy, float* mean, float* rstd,
                             unsigned char* q8, int ldq, float* row_scale, const float* tscale, float* amax_acc,
                             const float* cls_x = nullptr, int cls_period = 0, TX* x_refresh = nullptr) {
#define LN_FWD_CASE(N) case N: hipLaunchKernelGGL((ln_fwd_kernel<bf16, N, TX, true, CLS>), grid, dim3(256), 0, stream, x, ldx, rows, gamma, beta, eps, M, W, y, ldy, mean, rstd, q8, ldq, row_scale, tscale, amax_acc, cls_x, cls_period, x_refresh); break;
    switch (it) { LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4) default: LN_FWD_CASE(5) }
#undef LN_FWD_CASE
}

// ================================================================================================================================
// bf16-row forms with EIGHT columns per lane (round 5).  The 4-column kernels above move a bf16 row in 8-byte pieces: the same
// number of memory instructions as an fp32 row at half the bytes -- and the LayerNorm forward did not get faster when the hybrid
// stream halved its input bytes (127 us for 0.46 GB at 150 720 x 768 where the fp32-input form took 136 us for 0.69 GB): the
// kernels are bound by REQUESTS, not bytes.  Here a lane owns 8 consecutive columns (16-byte loads and stores, pair p of a row at
// columns lane * 8 + p * 512), everything else as above: a wave per row, the next row's loads in flight, two-stage dgamma / dbeta.
// Used for the bf16 -> bf16 FORWARD of the space-time blocks (W % 8 == 0): 127-134 -> 93-95 us at 150 720 x 768.  The backward forms
// were built the same way and measured SLOWER than the 4-column kernels (ln_2 199 against 180 us, ln_1 151 against 134, ln_3 236
// against 243: four input streams of two rows in flight need 180-200 registers, i.e. two blocks per CU instead of three; with ONE row
// per wave at three blocks per CU 188 / 136 / 228 us) -- the backward streams ~5 TB/s either way: not kept.
// ================================================================================================================================
__device__ __forceinline__ void widen8(const bf16x8& t, f32x4& lo, f32x4& hi) {
    lo = (f32x4){(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    hi = (f32x4){(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
}
__device__ __forceinline__ bf16x8 narrow8(const f32x4& lo, const f32x4& hi) {
    return (bf16x8){(bf16)lo[0], (bf16)lo[1], (bf16)lo[2], (bf16)lo[3], (bf16)hi[0], (bf16)hi[1], (bf16)hi[2], (bf16)hi[3]};
}

// Q8: the e4m3 copy of the output as in ln_fwd_kernel (the bf16-rounded values under the row's or the tensor's scale: the same
// bytes); y may then be NULL -- under per-tensor scales every consumer of a space-time block's LayerNorm output multiplies the
// e4m3 bytes (forward GEMM and weight gradient), the bf16 tensor would be written and never read.
template <int NP, bool CLS, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_fwd8_kernel(const bf16* __restrict__ x, int ldx, const int* __restrict__ rows,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      int M, int W, bf16* __restrict__ y, int ldy, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, const float* __restrict__ cls_x, int cls_period,
                                                      bf16* x_refresh, unsigned char* __restrict__ q8 = nullptr, int ldq = 0,
                                                      float* __restrict__ row_scale = nullptr, const float* __restrict__ tscale = nullptr,
                                                      float* __restrict__ amax_acc = nullptr) {
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * 4;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float run_amax = 0.f;
    f32x4 gm[NP][2], bt[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane * 8 + p * 512 + h * 4;
            gm[p][h] = c < W ? load4<float>(gamma + c) : (f32x4){0, 0, 0, 0};
            bt[p][h] = c < W ? load4<float>(beta + c) : (f32x4){0, 0, 0, 0};
        }
    const float invW = 1.0f / (float)W;
    struct Row { bf16x8 raw[NP]; bool cls; };
    auto load_row = [&](int rr, Row& w) {
        w.cls = CLS && rr % cls_period == 0;  // (wave-uniform) its value comes from the fp32 side array below
        if (w.cls) return;
        const bf16* xp = x + (size_t)(rows ? rows[rr] : rr) * ldx;
#pragma unroll
        for (int p = 0; p < NP; ++p) {  // (unconditional loads: see ln_bwd_kernel)
            const int c = lane * 8 + p * 512;
            w.raw[p] = *(const bf16x8*)(xp + (c < W ? c : 0));
        }
    };
    Row cur, nxt;
    load_row(r, cur);
    for (; r < M; r += stride) {
        const bool more = r + stride < M;
        if (more) load_row(r + stride, nxt);
        f32x4 v[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = lane * 8 + p * 512;
            v[p][0] = (f32x4){0, 0, 0, 0}; v[p][1] = (f32x4){0, 0, 0, 0};
            if (c < W) {
                if (CLS && cur.cls) {
                    const float* cp = cls_x + (size_t)(r / cls_period) * W + c;
                    v[p][0] = load4<float>(cp); v[p][1] = load4<float>(cp + 4);
                } else {
                    widen8(cur.raw[p], v[p][0], v[p][1]);
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) s += v[p][h][0] + v[p][h][1] + v[p][h][2] + v[p][h][3];
        const float mean = wave_sum(s) * invW;
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (lane * 8 + p * 512 < W) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[p][h][e] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) * invW + eps);
        const bool refresh = CLS && cur.cls && x_refresh;
        bf16x8 ob[Q8 ? NP : 1];
        float am = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = lane * 8 + p * 512;
            if (c < W) {
                f32x4 o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[h][e] = (v[p][h][e] - mean) * rstd * gm[p][h][e] + bt[p][h][e];
                const bf16x8 ob8 = narrow8(o[0], o[1]);
                if (!Q8 || y) *(bf16x8*)(y + (size_t)r * ldy + c) = ob8;
                if (refresh) *(bf16x8*)(x_refresh + (size_t)r * ldx + c) = narrow8(v[p][0], v[p][1]);
                if (Q8) {
                    ob[p] = ob8;  // the values a bf16 consumer would see
#pragma unroll
                    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf((float)ob8[e]));
                }
            }
        }
        if (Q8) {
            am = wave_max(am);
            run_amax = fmaxf(run_amax, am);
            const float scale = q8_scale(tscale, am);
            const float inv = 1.0f / scale;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int c = lane * 8 + p * 512;
                if (c < W) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)ob[p][e] * inv, -448.0f), 448.0f);
                    int p0 = 0, p1 = 0;
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], p0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p0, true);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], p1, false);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], p1, true);
                    typedef __attribute__((ext_vector_type(2))) int i32x2_;
                    *(i32x2_*)(q8 + (size_t)r * ldq + c) = (i32x2_){p0, p1};
                }
            }
            if (lane == 0 && row_scale) row_scale[r] = scale;
        }
        if (lane == 0) {
            if (mean_out) mean_out[r] = mean;
            if (rstd_out) rstd_out[r] = rstd;
        }
        if (more) cur = nxt;
    }
    if (Q8) amax_publish(amax_acc, run_amax, lane);
}

template <bool CLS>
static void launch_ln_fwd8(hipStream_t stream, const bf16* x, int ldx, const int* rows, const float* gamma, const float* beta,
                           float eps, int M, int W, bf16* y, int ldy, float* mean, float* rstd, const float* cls_x = nullptr,
                           int cls_period = 0, bf16* x_refresh = nullptr) {
    int blocks = ceil_div(M, 4);
    if (blocks > 2048) blocks = 2048;
    const int np = ceil_div(W, 512);
#define LN8_CASE(N) case N: hipLaunchKernelGGL((ln_fwd8_kernel<N, CLS>), dim3(blocks), dim3(256), 0, stream, x, ldx, rows, gamma, beta, eps, M, W, y, ldy, mean, rstd, cls_x, cls_period, x_refresh); break;
    switch (np) { LN8_CASE(1) LN8_CASE(2) default: LN8_CASE(3) }
#undef LN8_CASE
}
template <bool CLS>
static void launch_ln_fwd8_q8(hipStream_t stream, const bf16* x, int ldx, const int* rows, const float* gamma, const float* beta,
                              float eps, int M, int W, bf16* y, int ldy, float* mean, float* rstd, unsigned char* q8, int ldq,
                              float* row_scale, const float* tscale, float* amax_acc, const float* cls_x = nullptr,
                              int cls_period = 0, bf16* x_refresh = nullptr) {
    int blocks = ceil_div(M, 4);
    if (blocks > 2048) blocks = 2048;
    const int np = ceil_div(W, 512);
#define LN8_CASE(N) case N: hipLaunchKernelGGL((ln_fwd8_kernel<N, CLS, true>), dim3(blocks), dim3(256), 0, stream, x, ldx, rows, gamma, beta, eps, M, W, y, ldy, mean, rstd, cls_x, cls_period, x_refresh, q8, ldq, row_scale, tscale, amax_acc); break;
    switch (np) { LN8_CASE(1) LN8_CASE(2) default: LN8_CASE(3) }
#undef LN8_CASE
}
static inline bool ln8_ok(int W, int ld_a, int ld_b) { return W % 8 == 0 && W <= 1536 && ld_a % 8 == 0 && ld_b % 8 == 0; }

// LayerNorm forward that also emits the e4m3 copy of its bf16 output with per-row scales (see Q8 above)
extern "C" int tvts_layernorm_fwd_fp8(const void* x, int ldx, int x_bf16, const int* rows, const float* gamma, const float* beta,
                                      float eps, int M, int W, void* y, int ldy, void* q8, int ldq, float* row_scale,
                                      const float* tscale, float* amax_acc, float* mean, float* rstd, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || ldy % 4 || ldq % 4 || !q8 || (!row_scale && !tscale)) return TVTS_EINVAL;
    if (!y && !(x_bf16 && ln8_ok(W, ldx, 8) && ldq % 8 == 0)) return TVTS_EINVAL;  // e4m3-only: the 8-column bf16-row form
    int blocks = ceil_div(M, 4);
    if (blocks > 2048) blocks = 2048;
    const int it = ceil_div(W, 256);
    if (x_bf16 && ln8_ok(W, ldx, y ? ldy : 8) && ldq % 8 == 0)
        launch_ln_fwd8_q8<false>(stream, (const bf16*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd, (unsigned char*)q8, ldq, row_scale, tscale, amax_acc);
    else if (x_bf16) launch_ln_fwd_q8<bf16>(it, dim3(blocks), stream, (const bf16*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd, (unsigned char*)q8, ldq, row_scale, tscale, amax_acc);
    else launch_ln_fwd_q8<float>(it, dim3(blocks), stream, (const float*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd, (unsigned char*)q8, ldq, row_scale, tscale, amax_acc);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// x: the fp32 residual stream, or (x_bf16) a side-branch value that only this LayerNorm consumes and that is therefore
// kept in bf16 (the time residual of the space-time block: the space branch restarts from the block input).
extern "C" int tvts_layernorm_fwd(const void* x, int ldx, int x_bf16, const int* rows, const float* gamma, const float* beta,
                                  float eps, int M, int W, void* y, int ldy, int y_f32, float* mean, float* rstd,
                                  hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || ldy % 4 || (x_bf16 && y_f32)) return TVTS_EINVAL;
    int blocks = ceil_div(M, 4);
    if (blocks > 2048) blocks = 2048;  // 8 blocks x 4 waves per CU, two rows in flight per wave
    const int it = ceil_div(W, 256);
    if (x_bf16 && ln8_ok(W, ldx, ldy)) launch_ln_fwd8<false>(stream, (const bf16*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd);
    else if (x_bf16) launch_ln_fwd<bf16, bf16>(it, dim3(blocks), stream, (const bf16*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd);
    else if (y_f32) launch_ln_fwd<float, float>(it, dim3(blocks), stream, (const float*)x, ldx, rows, gamma, beta, eps, M, W, (float*)y, ldy, mean, rstd);
    else launch_ln_fwd<bf16, float>(it, dim3(blocks), stream, (const float*)x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// The hybrid residual stream (round 5): the stream is bf16, the CLS token's row of every clip -- the row the video embedding is read
// from, and the one row whose rounding error reaches every other token through the attention -- is carried in fp32 in a compact side
// array.  x bf16 [M, W] (rows r % cls_period == 0 stale), cls_x fp32 [M / cls_period, W]; those rows are normalised from cls_x and
// x_refresh (normally x itself; optional) receives their bf16 rounding.  q8 (optional) as in tvts_layernorm_fwd_fp8.
extern "C" int tvts_layernorm_fwd_cls(const void* x, int ldx, const float* cls_x, int cls_period, void* x_refresh, const float* gamma,
                                      const float* beta, float eps, int M, int W, void* y, int ldy, void* q8, int ldq, float* row_scale,
                                      const float* tscale, float* amax_acc, float* mean, float* rstd, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || ldy % 4 || !cls_x || cls_period <= 0 || M % cls_period) return TVTS_EINVAL;
    if (q8 && (ldq % 4 || (!row_scale && !tscale))) return TVTS_EINVAL;
    if (!y && !(q8 && ln8_ok(W, ldx, 8) && ldq % 8 == 0)) return TVTS_EINVAL;  // e4m3-only: the 8-column form
    int blocks = ceil_div(M, 4);
    if (blocks > 2048) blocks = 2048;
    const int it = ceil_div(W, 256);
    if (q8 && ln8_ok(W, ldx, y ? ldy : 8) && ldq % 8 == 0)
        launch_ln_fwd8_q8<true>(stream, (const bf16*)x, ldx, nullptr, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd, (unsigned char*)q8, ldq,
                                row_scale, tscale, amax_acc, cls_x, cls_period, (bf16*)x_refresh);
    else if (q8) launch_ln_fwd_q8<bf16, true>(it, dim3(blocks), stream, (const bf16*)x, ldx, nullptr, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd,
                                   (unsigned char*)q8, ldq, row_scale, tscale, amax_acc, cls_x, cls_period, (bf16*)x_refresh);
    else if (ln8_ok(W, ldx, ldy)) launch_ln_fwd8<true>(stream, (const bf16*)x, ldx, nullptr, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd,
                                                        cls_x, cls_period, (bf16*)x_refresh);
    else launch_ln_fwd<bf16, bf16, true>(it, dim3(blocks), stream, (const bf16*)x, ldx, nullptr, gamma, beta, eps, M, W, (bf16*)y, ldy, mean, rstd,
                                         cls_x, cls_period, (bf16*)x_refresh);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

template <typename TDY> struct RawDy;
template <> struct RawDy<float> { typedef f32x4 T; };
template <> struct RawDy<bf16> { typedef bf16x4 T; };
__device__ __forceinline__ f32x4 widen(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 widen(bf16x4 v) { return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// Q8: additionally write the bf16 copy of dx as OCP e4m3 bytes with one scale per row (amax of the bf16-rounded values / 448) -- the
// output gradient of the e4m3 input-gradient GEMM that consumes dx_bf16 (same bytes as tvts_quant_fp8_rows of dx_bf16)
// blocks per CU of a variant (launch bound and persistent grid): rows of <= 768 columns 3 with a residual input, else 4; wider rows 2
// with a residual input or with a second row in flight (all-bf16 streams), else 3
template <typename TDY, int IT, bool R1, bool R2, typename TX, typename TR1>
constexpr int ln_bwd_per_cu() {
    return IT <= 3 ? ((R1 || R2) ? 3 : 4)
                   : ((R1 || R2 || (sizeof(TDY) == 2 && sizeof(TX) == 2 && (!R1 || sizeof(TR1) == 2))) ? 2 : 3);
}
template <typename TDY, int IT, bool R1, bool R2, typename TX = float, bool Q8 = false, typename TR1 = float, bool CLS = false>
__global__ __launch_bounds__(256, (ln_bwd_per_cu<TDY, IT, R1, R2, TX, TR1>())) void ln_bwd_kernel(const TDY* __restrict__ dy, int lddy, const TX* __restrict__ x,
                                                        int ldx, const int* __restrict__ rows,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const TR1* __restrict__ res1,
                                                        const bf16* __restrict__ res2, int ldr2, int ldr, int M, int W,
                                                        float* __restrict__ dx, int lddx, bf16* __restrict__ dx_bf16,
                                                        int lddxb, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        float* __restrict__ partial,
                                                        unsigned char* __restrict__ q8 = nullptr, int ldq = 0,
                                                        float* __restrict__ row_scale = nullptr,
                                                        const float* __restrict__ tscale = nullptr, float* __restrict__ amax_acc = nullptr,
                                                        const float* __restrict__ cls_x = nullptr, const float* __restrict__ cls_res1 = nullptr,
                                                        float* __restrict__ cls_dx = nullptr, int cls_period = 0) {
    // CLS (the hybrid residual stream): this instantiation walks ONLY the rows r % cls_period == 0, behind a plain launch that has
    // done every row (and dgamma / dbeta): for them the LayerNorm input comes from cls_x (fp32, what the forward normalised), the
    // residual-stream gradient res1 from cls_res1 (fp32) instead of the bf16 stream row, and the result -- which replaces the
    // plain launch's in dx_bf16 (and q8) -- is ALSO stored in fp32 to cls_dx: the CLS token's gradient chain never passes through
    // a bf16 rounding.  (One kernel for both kinds of rows ran the 784 other rows of a clip at half speed: 14 spilled registers.)
    typedef typename RawDy<TDY>::T DyV;
    // a row in flight keeps its inputs in their STORED type (bf16 rows: 2 registers per 4 columns instead of 4) and is widened where it
    // is used; the CLS instantiation reads fp32 side rows and keeps f32x4
    typedef typename std::conditional<CLS, f32x4, typename RawDy<TX>::T>::type XV;
    typedef typename std::conditional<CLS, f32x4, typename RawDy<TR1>::T>::type R1V;
    constexpr bool ALL16 = sizeof(TDY) == 2 && sizeof(TX) == 2 && (!R1 || sizeof(TR1) == 2);
    float run_amax = 0.f;
    __shared__ float red[2][4][IT * 256];  // [gamma|beta][wave][column slot]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[IT], ab[IT], gm[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        ag[it] = (f32x4){0, 0, 0, 0};
        ab[it] = (f32x4){0, 0, 0, 0};
        const int c = lane * 4 + it * 256;
        gm[it] = c < W ? load4<float>(gamma + c) : (f32x4){0, 0, 0, 0};
    }
    const float invW = 1.0f / (float)W;
    const int stride = gridDim.x * 4 * (CLS ? cls_period : 1);
    struct Row { XV x[IT]; DyV d[IT]; R1V r1[R1 ? IT : 1]; bf16x4 r2[R2 ? IT : 1]; float mu, rs; int xr; };
    auto load_row = [&](int rr, Row& w) {
        w.xr = rows ? rows[rr] : rr;
        w.mu = mean[rr];
        w.rs = rstd[rr];
        const bool cls = CLS;
        const size_t ci = cls ? (size_t)(rr / cls_period) * W : 0;
        // UNCONDITIONAL loads (lanes past the row's end re-read its first columns -- always inside the row; their values are never used): inside `if (c < W)`
        // every column group is a basic block of its own, and hipcc's wait-count pass, which must assume that a group may have been
        // skipped, makes each group wait for the loads of the one before it -- IT dependent round trips per row instead of one
        // (the 1280-wide H/14 forms streamed 3.0 - 3.6 TB/s where the 768-wide ones stream ~5)
        // The all-bf16 forms only: with fp32 streams every load in flight at once needs more registers than the launch bounds leave
        // (24 - 40 spilled, and a spill reload in this loop waits for the prefetched row).
        constexpr bool UNCOND = ALL16 && !CLS;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c0 = lane * 4 + it * 256, c = c0 < W ? c0 : 0;
            if (UNCOND || c0 < W) {
                if constexpr (CLS) {
                    w.x[it] = cls_x ? load4<float>(cls_x + ci + c) : load4<TX>(x + (size_t)w.xr * ldx + c);
                    if (R1) w.r1[it] = cls_res1 ? load4<float>(cls_res1 + ci + c) : load4<TR1>(res1 + (size_t)w.xr * ldr + c);
                } else {
                    w.x[it] = *(const XV*)(x + (size_t)w.xr * ldx + c);
                    if (R1) w.r1[it] = *(const R1V*)(res1 + (size_t)w.xr * ldr + c);
                }
                w.d[it] = *(const DyV*)(dy + (size_t)rr * lddy + c);
                if (R2) w.r2[it] = *(const bf16x4*)(res2 + (size_t)w.xr * ldr2 + c);
            }
        }
    };
    // a second row in flight: rows of <= 768 columns always; the wider rows (1024, 1280 columns) when every input stream is bf16 (the
    // hybrid stream's forms: 40 instead of 60 - 80 registers per row in flight).  Without it a wave runs load -> reduce -> store round
    // trips one at a time at 8 - 12 waves per CU: the H/14 backward forms streamed 3.0 - 3.6 TB/s where the B/16 forms stream ~5
    constexpr bool PF = !CLS && (IT <= 3 || ALL16);
    int r = (blockIdx.x * 4 + wave) * (CLS ? cls_period : 1);
    Row cur, nxt;
    if (PF && r < M) load_row(r, cur);
    for (; r < M; r += stride) {
        const bool more = PF && r + stride < M;
        if (PF) { if (more) load_row(r + stride, nxt); }
        else load_row(r, cur);
        f32x4 xh[IT], g[IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
                const f32x4 d = widen(cur.d[it]), xv = widen(cur.x[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[it][e] = (xv[e] - cur.mu) * cur.rs;
                    g[it][e] = d[e] * gm[it][e];
                    s1 += g[it][e];
                    s2 += g[it][e] * xh[it][e];
                    ag[it][e] += d[e] * xh[it][e];
                    ab[it][e] += d[e];
                }
            }
        }
        const float c1 = wave_sum(s1) * invW, c2 = wave_sum(s2) * invW;
        // Q8: the row's outputs are needed twice (amax, then conversion).  The residual forms keep a copy (their launch bound leaves the
        // registers; recomputing keeps the residual registers live through the second sweep: 246 -> 316 us at W = 1280), the
        // residual-free forms recompute (a copy spills at their tighter bound: 130 -> 200 us)
        constexpr bool KEEP = Q8 && (R1 || R2);
        f32x4 ob[KEEP ? IT : 1];
        float am = 0.f;
        auto out_of = [&](int it) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = cur.rs * (g[it][e] - c1 - xh[it][e] * c2);
            if (R1) o += widen(cur.r1[it]);
            if (R2) o += widen(cur.r2[it]);
            return o;
        };
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
                const f32x4 o = out_of(it);
                if (dx) store4(dx + (size_t)cur.xr * lddx + c, o);
                if (dx_bf16) store4(dx_bf16 + (size_t)cur.xr * lddxb + c, o);
                if (CLS && cls_dx) store4(cls_dx + (size_t)(r / cls_period) * W + c, o);
                if (Q8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = (float)(bf16)o[e];  // the value the bf16 consumers see
                        if (KEEP) ob[it][e] = v;
                        am = fmaxf(am, fabsf(v));
                    }
                }
            }
        }
        if (Q8) {  // second sweep over the kept copy or the recomputed outputs
            am = wave_max(am);
            run_amax = fmaxf(run_amax, am);
            const float scale = q8_scale(tscale, am);
            const float inv = 1.0f / scale;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = lane * 4 + it * 256;
                if (c < W) {
                    f32x4 o;
                    if (KEEP) o = ob[it];
                    else o = out_of(it);
                    float f[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf((float)(bf16)o[e] * inv, -448.0f), 448.0f);
                    int pk = 0;
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], pk, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
                    *(int*)(q8 + (size_t)cur.xr * ldq + c) = pk;
                }
            }
            if (lane == 0 && row_scale) row_scale[cur.xr] = scale;
        }
        if (more) cur = nxt;
    }
    if (Q8) amax_publish(amax_acc, run_amax, lane);
    if (!dgamma) return;
    // block reduce the per-wave partials, then one atomic per column per block
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(it * 4 + e) * 64 + lane] = ag[it][e];
            red[1][wave][(it * 4 + e) * 64 + lane] = ab[it][e];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < IT * 4 * 64; idx += 256) {
        const int l = idx & 63, ie = idx >> 6;
        const int c = l * 4 + (ie >> 2) * 256 + (ie & 3);
        if (c < W) {
            const float sg = red[0][0][idx] + red[0][1][idx] + red[0][2][idx] + red[0][3][idx];
            const float sb = red[1][0][idx] + red[1][1][idx] + red[1][2][idx] + red[1][3][idx];
            if (partial) {  // per-block partial sums, combined by ln_dgamma_reduce_kernel (plain stores, deterministic)
                partial[((size_t)blockIdx.x * 2 + 0) * W + c] = sg;
                partial[((size_t)blockIdx.x * 2 + 1) * W + c] = sb;
            } else {        // no workspace: one atomic per column per block
                atomicAdd(dgamma + c, sg);
                atomicAdd(dbeta + c, sb);
            }
        }
    }
}

// dgamma[c] += sum_b partial[b][0][c], dbeta[c] += sum_b partial[b][1][c]
__global__ __launch_bounds__(1024) void ln_dgamma_reduce_kernel(const float* __restrict__ partial, int nblocks, int W,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // 64 columns x 16 row-groups per block: each thread sums nblocks / 16 partials with 8 independent loads in flight
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int k = blockIdx.y;
    __shared__ float acc[16][64];
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < W) {
        int b = part;
        for (; b + 7 * 16 < nblocks; b += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += partial[((size_t)(b + u * 16) * 2 + k) * W + c];
        }
        for (; b < nblocks; b += 16) s[0] += partial[((size_t)b * 2 + k) * W + c];
    }
    acc[part][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (part == 0 && c < W) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += acc[p][threadIdx.x];
        float* dst = k == 0 ? dgamma : dbeta;
        dst[c] += t;
    }
}

template <typename TDY, bool R1, bool R2, typename TX = float, bool Q8 = false, typename TR1 = float, bool CLS = false>
static void launch_ln_bwd(int it, int M, hipStream_t stream, const TDY* dy, int lddy, const TX* x, int ldx, const int* rows,
                          const float* mean, const float* rstd, const float* gamma, const TR1* res1, const bf16* res2,
                          int ldr2, int ldr, int W, float* dx, int lddx, bf16* dxb, int lddxb, float* dgamma, float* dbeta,
                          float* ws, long ws_elems, unsigned char* q8 = nullptr, int ldq = 0, float* row_scale = nullptr,
                          const float* tscale = nullptr, float* amax_acc = nullptr, const float* cls_x = nullptr,
                          const float* cls_res1 = nullptr, float* cls_dx = nullptr, int cls_period = 0) {
    // persistent grid: as many blocks per CU as the variant's registers allow (see __launch_bounds__ above)
    const int per_cu = it <= 3 ? ln_bwd_per_cu<TDY, 3, R1, R2, TX, TR1>() : ln_bwd_per_cu<TDY, 5, R1, R2, TX, TR1>();
    int blocks = ceil_div(CLS ? M / cls_period : M, 4);
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    const dim3 grid(blocks);
    if (CLS) dgamma = dbeta = nullptr;  // the plain launch in front of this one has accumulated them over every row
    float* partial = (dgamma && ws && ws_elems >= (long)blocks * 2 * W) ? ws : nullptr;
#define LN_BWD_CASE(N) case N: hipLaunchKernelGGL((ln_bwd_kernel<TDY, N, R1, R2, TX, Q8, TR1, CLS>), grid, dim3(256), 0, stream, dy, lddy, x, ldx, rows, mean, rstd, gamma, res1, res2, ldr2, ldr, M, W, dx, lddx, dxb, lddxb, dgamma, dbeta, partial, q8, ldq, row_scale, tscale, amax_acc, cls_x, cls_res1, cls_dx, cls_period); break;
    switch (it) { LN_BWD_CASE(1) LN_BWD_CASE(2) LN_BWD_CASE(3) LN_BWD_CASE(4) default: LN_BWD_CASE(5) }
#undef LN_BWD_CASE
    if (partial)
        hipLaunchKernelGGL(ln_dgamma_reduce_kernel, dim3(ceil_div(W, 64), 2), dim3(1024), 0, stream, partial, blocks, W, dgamma, dbeta);
}
template <typename TDY, typename TR1>
static void launch_ln_bwd_res(int it, int M, hipStream_t stream, const TDY* dy, int lddy, const float* x, int ldx,
                              const int* rows, const float* mean, const float* rstd, const float* gamma, const TR1* res1,
                              const bf16* res2, int ldr2, int ldr, int W, float* dx, int lddx, bf16* dxb, int lddxb,
                              float* dgamma, float* dbeta, float* ws, long ws_elems) {
#define LN_ARGS it, M, stream, dy, lddy, x, ldx, rows, mean, rstd, gamma, res1, res2, ldr2, ldr, W, dx, lddx, dxb, lddxb, dgamma, dbeta, ws, ws_elems
    if (res1 && res2) launch_ln_bwd<TDY, true, true, float, false, TR1>(LN_ARGS);
    else if (res1) launch_ln_bwd<TDY, true, false, float, false, TR1>(LN_ARGS);
    else if (res2) launch_ln_bwd<TDY, false, true, float, false, TR1>(LN_ARGS);
    else launch_ln_bwd<TDY, false, false, float, false, TR1>(LN_ARGS);
#undef LN_ARGS
}

extern "C" int tvts_layernorm_bwd(const void* dy, int lddy, int dy_f32, const void* x_, int ldx, int x_bf16, const int* rows,
                                  const float* mean, const float* rstd, const float* gamma, const void* res1_, int res1_bf16,
                                  int ldr, const void* res2_bf16, int ldr2, int M, int W, float* dx, int lddx,
                                  void* dx_bf16, int lddxb, float* dgamma, float* dbeta, float* workspace,
                                  long workspace_elems, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || lddy % 4) return TVTS_EINVAL;
    if ((!dx && !dx_bf16) || (dx && lddx % 4)) return TVTS_EINVAL;
    if ((res1_ && ldr % 4) || (res2_bf16 && ldr2 % 4)) return TVTS_EINVAL;
    const bf16* res2 = (const bf16*)res2_bf16;
    if (dx_bf16 && lddxb % 4) return TVTS_EINVAL;
    if (x_bf16) {  // bf16 input: a side-branch value (ln_1's time residual) or the bf16 residual stream of the space-time blocks:
        // bf16 dy; no residual, or a bf16 res1 (+ the bf16 side branch res2)
        if (dy_f32 || (res1_ && !res1_bf16)) return TVTS_EINVAL;
        const bf16* xb = (const bf16*)x_;
        const bf16* r1 = (const bf16*)res1_;
        const int it = ceil_div(W, 256);
#define LNB_ARGS it, M, stream, (const bf16*)dy, lddy, xb, ldx, rows, mean, rstd, gamma, r1, res2, ldr2, ldr, W, dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems
        if (r1 && res2) launch_ln_bwd<bf16, true, true, bf16, false, bf16>(LNB_ARGS);
        else if (r1) launch_ln_bwd<bf16, true, false, bf16, false, bf16>(LNB_ARGS);
        else if (res2) launch_ln_bwd<bf16, false, true, bf16, false, bf16>(LNB_ARGS);
        else launch_ln_bwd<bf16, false, false, bf16, false, bf16>(LNB_ARGS);
#undef LNB_ARGS
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const float* x = (const float*)x_;
    const int it = ceil_div(W, 256);
    if (res1_ && res1_bf16) {  // the residual-stream gradient carried in bf16 (the space-time block's backward): bf16 dy only
        if (dy_f32) return TVTS_EINVAL;
        launch_ln_bwd_res<bf16, bf16>(it, M, stream, (const bf16*)dy, lddy, x, ldx, rows, mean, rstd, gamma, (const bf16*)res1_, res2, ldr2,
                                      ldr, W, dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems);
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    const float* res1 = (const float*)res1_;
    if (dy_f32)
        launch_ln_bwd_res<float, float>(it, M, stream, (const float*)dy, lddy, x, ldx, rows, mean, rstd, gamma, res1, res2, ldr2, ldr, W,
                                        dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems);
    else
        launch_ln_bwd_res<bf16, float>(it, M, stream, (const bf16*)dy, lddy, x, ldx, rows, mean, rstd, gamma, res1, res2, ldr2, ldr, W,
                                       dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// LayerNorm backward that also emits the e4m3 copy of its bf16 output with per-row scales (Q8 above): the forms of the space-time
// block's backward -- bf16 dy with (ln_2) fp32 residual, (ln_3) fp32 + bf16 residuals, (ln_1) bf16 x and no residual, (ln_post) fp32 x
// and no residual.  Every row (no row list), dx_bf16 required.
extern "C" int tvts_layernorm_bwd_fp8(const void* dy, int lddy, const void* x_, int ldx, int x_bf16, const float* mean,
                                      const float* rstd, const float* gamma, const void* res1_, int res1_bf16, int ldr, const void* res2_bf16,
                                      int ldr2, int M, int W, float* dx, int lddx, void* dx_bf16, int lddxb, void* q8, int ldq,
                                      float* row_scale, const float* tscale, float* amax_acc, float* dgamma, float* dbeta,
                                      float* workspace, long workspace_elems, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || lddy % 4 || !dx_bf16 || lddxb % 4 || !q8 || ldq % 4 || (!row_scale && !tscale))
        return TVTS_EINVAL;
    if ((dx && lddx % 4) || (res1_ && ldr % 4) || (res2_bf16 && ldr2 % 4)) return TVTS_EINVAL;
    const bf16* res2 = (const bf16*)res2_bf16;
    const int it = ceil_div(W, 256);
#define Q8_ARGS it, M, stream, (const bf16*)dy, lddy, x, ldx, (const int*)nullptr, mean, rstd, gamma, res1, res2, ldr2, ldr, W, dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems, (unsigned char*)q8, ldq, row_scale, tscale, amax_acc
    if (x_bf16) {
        const bf16* x = (const bf16*)x_;
        if (res1_ && !res1_bf16) return TVTS_EINVAL;
        const bf16* res1 = (const bf16*)res1_;
        if (res1 && res2) launch_ln_bwd<bf16, true, true, bf16, true, bf16>(Q8_ARGS);
        else if (res1) launch_ln_bwd<bf16, true, false, bf16, true, bf16>(Q8_ARGS);
        else if (res2) return TVTS_EINVAL;
        else launch_ln_bwd<bf16, false, false, bf16, true, bf16>(Q8_ARGS);
    } else if (res1_ && res1_bf16) {
        const float* x = (const float*)x_;
        const bf16* res1 = (const bf16*)res1_;
        if (res2) launch_ln_bwd<bf16, true, true, float, true, bf16>(Q8_ARGS);
        else launch_ln_bwd<bf16, true, false, float, true, bf16>(Q8_ARGS);
    } else {
        const float* x = (const float*)x_;
        const float* res1 = (const float*)res1_;
        if (res1 && res2) launch_ln_bwd<bf16, true, true, float, true>(Q8_ARGS);
        else if (res1) launch_ln_bwd<bf16, true, false, float, true>(Q8_ARGS);
        else if (res2) return TVTS_EINVAL;
        else launch_ln_bwd<bf16, false, false, float, true>(Q8_ARGS);
    }
#undef Q8_ARGS
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// LayerNorm backward on the hybrid residual stream (tvts_layernorm_fwd_cls): bf16 dy, bf16 x [M, W], every row; res1 (optional) the
// bf16 stream gradient, res2 (optional, only with res1) a bf16 side branch; dx_bf16 required.  For the rows r % cls_period == 0:
// input from cls_x (optional), residual-stream gradient from cls_res1 (optional; fp32 [M / cls_period, W]) instead of res1's row,
// result also to cls_dx (optional, fp32).  q8 (optional) as in tvts_layernorm_bwd_fp8.
extern "C" int tvts_layernorm_bwd_cls(const void* dy, int lddy, const void* x_, int ldx, const float* cls_x, const float* cls_res1,
                                      float* cls_dx, int cls_period, const float* mean, const float* rstd, const float* gamma,
                                      const void* res1_bf16, int ldr, const void* res2_bf16, int ldr2, int M, int W, void* dx_bf16,
                                      int lddxb, void* q8, int ldq, float* row_scale, const float* tscale, float* amax_acc,
                                      float* dgamma, float* dbeta, float* workspace, long workspace_elems, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || lddy % 4 || !dx_bf16 || lddxb % 4) return TVTS_EINVAL;
    if (cls_period <= 0 || M % cls_period || (cls_res1 && !res1_bf16) || (res2_bf16 && !res1_bf16)) return TVTS_EINVAL;
    if ((res1_bf16 && ldr % 4) || (res2_bf16 && ldr2 % 4) || (q8 && (ldq % 4 || (!row_scale && !tscale)))) return TVTS_EINVAL;
    const bf16* x = (const bf16*)x_;
    const bf16* res1 = (const bf16*)res1_bf16;
    const bf16* res2 = (const bf16*)res2_bf16;
    const int it = ceil_div(W, 256);
#define CLS_ARGS(c) it, M, stream, (const bf16*)dy, lddy, x, ldx, (const int*)nullptr, mean, rstd, gamma, res1, res2, ldr2, ldr, W, (float*)nullptr, 0, (bf16*)dx_bf16, lddxb, dgamma, dbeta, workspace, workspace_elems, (unsigned char*)q8, ldq, row_scale, tscale, amax_acc, c ? cls_x : nullptr, c ? cls_res1 : nullptr, c ? cls_dx : nullptr, c ? cls_period : 0
    // every row through the plain bf16-stream kernel (the CLS rows' results there come from the stream's bf16 rows: close, and
    // replaced below), then the CLS rows alone from the fp32 side arrays
    if (q8) {
        if (res1 && res2) { launch_ln_bwd<bf16, true, true, bf16, true, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, true, true, bf16, true, bf16, true>(CLS_ARGS(1)); }
        else if (res1) { launch_ln_bwd<bf16, true, false, bf16, true, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, true, false, bf16, true, bf16, true>(CLS_ARGS(1)); }
        else { launch_ln_bwd<bf16, false, false, bf16, true, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, false, false, bf16, true, bf16, true>(CLS_ARGS(1)); }
    } else {
        if (res1 && res2) { launch_ln_bwd<bf16, true, true, bf16, false, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, true, true, bf16, false, bf16, true>(CLS_ARGS(1)); }
        else if (res1) { launch_ln_bwd<bf16, true, false, bf16, false, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, true, false, bf16, false, bf16, true>(CLS_ARGS(1)); }
        else { launch_ln_bwd<bf16, false, false, bf16, false, bf16>(CLS_ARGS(0)); launch_ln_bwd<bf16, false, false, bf16, false, bf16, true>(CLS_ARGS(1)); }
    }
#undef CLS_ARGS
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
