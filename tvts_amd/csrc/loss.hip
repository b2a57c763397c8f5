// Loss heads in fp32: cosine-similarity InfoNCE over the gathered embeddings and the transcript-sorting CE.
//
//   sim_matrix          v2/model/model_dist_TVTSv2_ViT_B_16.py:119-127  (a / max(|a|, eps))
//   NormSoftmaxLoss     v2/model/loss.py:13-25   (both directions, temperature 0.05, no 1/2)
//   2 * CrossEntropy    v2/trainer/trainer.py:487-492
// The G x G products themselves go through tvts_gemm_small_f32; this file holds the row kernels.
#include "common.h"

// xn = x / max(|x|, eps); inv[r] = 1 / max(|x|, eps); flag[r] = |x| > eps
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int R, int E, float eps,
                                                     float* __restrict__ xn, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) { const float v = x[(size_t)r * E + c]; s += v * v; }
    const float nrm = sqrtf(wave_sum(s));
    const float iv = 1.0f / fmaxf(nrm, eps);
    for (int c = lane; c < E; c += 64) xn[(size_t)r * E + c] = x[(size_t)r * E + c] * iv;
    if (lane == 0) inv[r] = nrm > eps ? iv : -iv;  // sign carries the clamp flag
}
extern "C" int tvts_l2norm_rows(const float* x, int R, int E, float eps, float* xn, float* inv, hipStream_t stream) {
    hipLaunchKernelGGL(l2norm_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, stream, x, R, E, eps, xn, inv);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// dx = inv * (dxn - xn * <xn, dxn>)   (clamped rows: dx = inv * dxn)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ xn,
                                                         const float* __restrict__ inv, int R, int E, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += xn[(size_t)r * E + c] * dxn[(size_t)r * E + c];
    s = wave_sum(s);
    const float iv = inv[r];
    const bool clamped = iv < 0.f;
    for (int c = lane; c < E; c += 64) {
        const float d = dxn[(size_t)r * E + c];
        dx[(size_t)r * E + c] = clamped ? -iv * d : iv * (d - xn[(size_t)r * E + c] * s);
    }
}
extern "C" int tvts_l2norm_rows_bwd(const float* dxn, const float* xn, const float* inv, int R, int E, float* dx,
                                    hipStream_t stream) {
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, stream, dxn, xn, inv, R, E, dx);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// x[G,G] = sim / temperature.  lse[0..G) = row log-sum-exp, lse[G..2G) = column log-sum-exp.  One wave each.
__global__ __launch_bounds__(256) void infonce_lse_kernel(const float* __restrict__ x, int G, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= 2 * G) return;
    const bool col = w >= G;
    const int idx = col ? w - G : w;
    const size_t stride = col ? (size_t)G : 1, base = col ? (size_t)idx : (size_t)idx * G;
    float m = -INFINITY;
    for (int c = lane; c < G; c += 64) m = fmaxf(m, x[base + c * stride]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < G; c += 64) s += __expf(x[base + c * stride] - m);
    s = wave_sum(s);
    if (lane == 0) lse[w] = m + __logf(s);
}
// loss += -(1/G) sum_i [(x_ii - rowlse_i) + (x_ii - collse_i)];  dx_ij = (e^{x_ij-rowlse_i} + e^{x_ij-collse_j} - 2 d_ij)/G
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* __restrict__ x, const float* __restrict__ lse, int G,
                                                           float* __restrict__ dx, float* __restrict__ loss) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)G * G) return;
    const int i = (int)(idx / G), j = (int)(idx % G);
    const float v = x[idx];
    float d = __expf(v - lse[i]) + __expf(v - lse[G + j]);
    if (i == j) {
        d -= 2.f;
        atomicAdd(loss, -(2.f * v - lse[i] - lse[G + j]) / (float)G);
    }
    if (dx) dx[idx] = d / (float)G;
}
extern "C" int tvts_infonce(const float* x, int G, float* lse, float* dx, float* loss, hipStream_t stream) {
    if (G <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(infonce_lse_kernel, dim3(ceil_div(2 * G, 4)), dim3(256), 0, stream, x, G, lse);
    hipLaunchKernelGGL(infonce_grad_kernel, dim3((unsigned)(((long)G * G + 255) / 256)), dim3(256), 0, stream, x, lse, G, dx,
                       loss);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// loss += scale * mean_r (lse_r - x[r,label_r]);  dlogits = scale * (softmax - onehot) / R
__global__ void ce_kernel(const float* __restrict__ logits, const int* __restrict__ labels, int R, int C, float scale,
                          float* __restrict__ dlogits, float* __restrict__ loss) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* p = logits + (size_t)r * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += __expf(p[c] - m);
    const float lse = m + __logf(s);
    const int lb = labels[r];
    atomicAdd(loss, scale * (lse - p[lb]) / (float)R);
    if (dlogits)
        for (int c = 0; c < C; ++c) dlogits[(size_t)r * C + c] = scale * (__expf(p[c] - lse) - (c == lb ? 1.f : 0.f)) / (float)R;
}
extern "C" int tvts_cross_entropy(const float* logits, const int* labels, int R, int C, float scale, float* dlogits,
                                  float* loss, hipStream_t stream) {
    if (R <= 0 || C <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(ce_kernel, dim3(ceil_div(R, 64)), dim3(64), 0, stream, logits, labels, R, C, scale, dlogits, loss);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
