// LayerNorm forward / backward for the fp32 residual stream (HBM-bound; one wave64 per row, float4 lanes).
//
// forward : y = (x - mean) * rstd * gamma + beta, x fp32 (optionally gathered rows), y bf16 or fp32;
//           mean / rstd are saved for the backward pass.
// backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ res1 + res2],  g = dy * gamma
//           res1 fp32 (the residual-stream gradient), res2 bf16 (a side-branch gradient that only exists as a GEMM
//           operand anyway); dx fp32 and / or a bf16 copy that feeds the next MFMA GEMM; dgamma / dbeta accumulated
//           with per-block partial sums and one fp32 atomic per column per block.
// Reference: nn.LayerNorm as used at v2/model/video_encoder_ViT_B_16.py:79-85 (eps 1e-5, fp32) and
// v2/model/sort_transformer.py:99 (eps 1e-6).
#include "common.h"

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

template <typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int M, int W, TO* __restrict__ y, int ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int xr = rows ? rows[r] : r;
    const float* xp = x + (size_t)xr * ldx;
    f32x4 v[LN_MAX_IT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            v[it] = load4<float>(xp + c);
            s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[it][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)W + eps);
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            const f32x4 g = load4<float>(gamma + c), b = load4<float>(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * g[e] + b[e];
            store4(y + (size_t)r * ldy + c, o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[r] = mean;
        if (rstd_out) rstd_out[r] = rstd;
    }
}

extern "C" int tvts_layernorm_fwd(const float* x, int ldx, const int* rows, const float* gamma, const float* beta,
                                  float eps, int M, int W, void* y, int ldy, int y_f32, float* mean, float* rstd,
                                  hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || ldy % 4) return TVTS_EINVAL;
    const dim3 grid(ceil_div(M, 4)), block(256);
    if (y_f32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, stream, x, ldx, rows, gamma, beta, eps, M, W, (float*)y,
                           ldy, mean, rstd);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, stream, x, ldx, rows, gamma, beta, eps, M, W, (bf16*)y,
                           ldy, mean, rstd);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

template <typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                     int ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ res1,
                                                     const bf16* __restrict__ res2, int ldr2, int ldr, int M, int W,
                                                     float* __restrict__ dx, int lddx, bf16* __restrict__ dx_bf16,
                                                     int lddxb, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][4][256 * LN_MAX_IT / 64 * 64];  // [gamma|beta][wave][column slot]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[LN_MAX_IT], ab[LN_MAX_IT], gm[LN_MAX_IT];
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        ag[it] = (f32x4){0, 0, 0, 0};
        ab[it] = (f32x4){0, 0, 0, 0};
        const int c = lane * 4 + it * 256;
        gm[it] = c < W ? load4<float>(gamma + c) : (f32x4){0, 0, 0, 0};
    }
    const float invW = 1.0f / (float)W;
    for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) {
        const int xr = rows ? rows[r] : r;
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[LN_MAX_IT], g[LN_MAX_IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAX_IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
                const f32x4 xv = load4<float>(x + (size_t)xr * ldx + c);
                const f32x4 d = load4<TDY>(dy + (size_t)r * lddy + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[it][e] = (xv[e] - mu) * rs;
                    g[it][e] = d[e] * gm[it][e];
                    s1 += g[it][e];
                    s2 += g[it][e] * xh[it][e];
                    ag[it][e] += d[e] * xh[it][e];
                    ab[it][e] += d[e];
                }
            }
        }
        const float c1 = wave_sum(s1) * invW, c2 = wave_sum(s2) * invW;
#pragma unroll
        for (int it = 0; it < LN_MAX_IT; ++it) {
            const int c = lane * 4 + it * 256;
            if (c < W) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - c1 - xh[it][e] * c2);
                if (res1) o += load4<float>(res1 + (size_t)xr * ldr + c);
                if (res2) o += load4<bf16>(res2 + (size_t)xr * ldr2 + c);
                if (dx) store4(dx + (size_t)xr * lddx + c, o);
                if (dx_bf16) store4(dx_bf16 + (size_t)xr * lddxb + c, o);
            }
        }
    }
    if (!dgamma) return;
    // block reduce the per-wave partials, then one atomic per column per block
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(it * 4 + e) * 64 + lane] = ag[it][e];
            red[1][wave][(it * 4 + e) * 64 + lane] = ab[it][e];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < LN_MAX_IT * 4 * 64; idx += 256) {
        const int l = idx & 63, ie = idx >> 6;
        const int c = l * 4 + (ie >> 2) * 256 + (ie & 3);
        if (c < W) {
            const float sg = red[0][0][idx] + red[0][1][idx] + red[0][2][idx] + red[0][3][idx];
            const float sb = red[1][0][idx] + red[1][1][idx] + red[1][2][idx] + red[1][3][idx];
            atomicAdd(dgamma + c, sg);
            atomicAdd(dbeta + c, sb);
        }
    }
}

extern "C" int tvts_layernorm_bwd(const void* dy, int lddy, int dy_f32, const float* x, int ldx, const int* rows,
                                  const float* mean, const float* rstd, const float* gamma, const float* res1,
                                  int ldr, const void* res2_bf16, int ldr2, int M, int W, float* dx, int lddx,
                                  void* dx_bf16, int lddxb, float* dgamma, float* dbeta, hipStream_t stream) {
    if (M <= 0 || W <= 0 || W % 4 || W > 256 * LN_MAX_IT || ldx % 4 || lddy % 4) return TVTS_EINVAL;
    if ((!dx && !dx_bf16) || (dx && lddx % 4)) return TVTS_EINVAL;
    if ((res1 && ldr % 4) || (res2_bf16 && ldr2 % 4)) return TVTS_EINVAL;
    const bf16* res2 = (const bf16*)res2_bf16;
    if (dx_bf16 && lddxb % 4) return TVTS_EINVAL;
    int blocks = ceil_div(M, 4);
    if (blocks > 512) blocks = 512;
    if (dy_f32)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, lddy, x, ldx, rows,
                           mean, rstd, gamma, res1, res2, ldr2, ldr, M, W, dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)dy, lddy, x, ldx, rows,
                           mean, rstd, gamma, res1, res2, ldr2, ldr, M, W, dx, lddx, (bf16*)dx_bf16, lddxb, dgamma, dbeta);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
