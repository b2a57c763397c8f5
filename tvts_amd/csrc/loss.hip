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

// x[G,G] = sim / temperature.  lse[0..G) = row log-sum-exp (one wave per row), lse[G..2G) = column log-sum-exp.
__global__ __launch_bounds__(256) void infonce_row_lse_kernel(const float* __restrict__ x, int G, float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= G) return;
    const float* p = x + (size_t)w * G;
    float m = -INFINITY;
    for (int c = lane; c < G; c += 64) m = fmaxf(m, p[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < G; c += 64) s += __expf(p[c] - m);
    s = wave_sum(s);
    if (lane == 0) lse[w] = m + __logf(s);
}
// columns: a block owns 64 adjacent columns (lane = column, so every load is a coalesced 256-byte row segment instead of
// a stride-G gather), its 4 waves split the rows and keep an online (max, sum) per column, merged through LDS.
__global__ __launch_bounds__(256) void infonce_col_lse_kernel(const float* __restrict__ x, int G, float* __restrict__ lse) {
    __shared__ float sm[4][64], ss[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float m = -INFINITY, s = 0.f;
    if (col < G) {
        for (int r = wave; r < G; r += 4) {
            const float v = x[(size_t)r * G + col];
            if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
            else s += __expf(v - m);
        }
    }
    sm[wave][lane] = m; ss[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < G) {
        float mm = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += ss[w][lane] > 0.f ? ss[w][lane] * __expf(sm[w][lane] - mm) : 0.f;
        lse[G + col] = mm + __logf(tot);
    }
}
// dx_ij = (e^{x_ij-rowlse_i} + e^{x_ij-collse_j} - 2 d_ij)/G
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* __restrict__ x, const float* __restrict__ lse, int G,
                                                           float* __restrict__ dx) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)G * G) return;
    const int i = (int)(idx / G), j = (int)(idx % G);
    const float v = x[idx];
    float d = __expf(v - lse[i]) + __expf(v - lse[G + j]);
    if (i == j) d -= 2.f;
    dx[idx] = d / (float)G;
}
// A block's 256 partial sums -> one value, always in the same order (lane butterfly, then the four waves left to right): the
// loss scalars are run-to-run reproducible (they were fp32 atomics up to round 2).
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
// loss += -(1/G) sum_i [(x_ii - rowlse_i) + (x_ii - collse_i)]: ONE block, thread t takes i = t, t + 256, ... in order
__global__ __launch_bounds__(256) void infonce_loss_kernel(const float* __restrict__ x, const float* __restrict__ lse, int G,
                                                           float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += 256) s += 2.f * x[(size_t)i * G + i] - lse[i] - lse[G + i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] += -s / (float)G;
}
extern "C" int tvts_infonce(const float* x, int G, float* lse, float* dx, float* loss, hipStream_t stream) {
    if (G <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(infonce_row_lse_kernel, dim3(ceil_div(G, 4)), dim3(256), 0, stream, x, G, lse);
    hipLaunchKernelGGL(infonce_col_lse_kernel, dim3(ceil_div(G, 64)), dim3(256), 0, stream, x, G, lse);
    if (dx)
        hipLaunchKernelGGL(infonce_grad_kernel, dim3((unsigned)(((long)G * G + 255) / 256)), dim3(256), 0, stream, x, lse, G, dx);
    hipLaunchKernelGGL(infonce_loss_kernel, dim3(1), dim3(256), 0, stream, x, lse, G, loss);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// loss += scale * mean_r (lse_r - x[r,label_r]);  dlogits = scale * (softmax - onehot) / R.  ONE block walks the rows (R = B * NT,
// a few thousand at most, C = 4) so that the loss sum has a fixed order.
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int* __restrict__ labels, int R, int C,
                                                 float scale, float* __restrict__ dlogits, float* __restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float* p = logits + (size_t)r * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, p[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(p[c] - m);
        const float lse = m + __logf(s);
        const int lb = labels[r];
        acc += lse - p[lb];
        if (dlogits)
            for (int c = 0; c < C; ++c)
                dlogits[(size_t)r * C + c] = scale * (__expf(p[c] - lse) - (c == lb ? 1.f : 0.f)) / (float)R;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) loss[0] += scale * acc / (float)R;
}
extern "C" int tvts_cross_entropy(const float* logits, const int* labels, int R, int C, float scale, float* dlogits,
                                  float* loss, hipStream_t stream) {
    if (R <= 0 || C <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, stream, logits, labels, R, C, scale, dlogits, loss);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- validation metrics
// Rank of the ground truth in a text x video similarity matrix (x[i, j] = <text_i, video_j>, n_text = q * n_vid, the
// captions of video j are rows j*q .. j*q + q - 1), exactly as v2/model/metric.py ranks with its "sort, then find the
// positions equal to the ground-truth distance" formulation:
//   mode 0 (t2v_metrics :16-126, ties broken optimistically): ranks[i] = #{ j : x[i, j] > x[i, i / q] }
//   mode 1 (v2t_metrics :129-187, ties averaged, closest own caption): for video j,
//          ranks[j] = min over its captions c of  #{ k : x[k, j] > x[c, j] } + (#{ k : x[k, j] == x[c, j] } - 1) / 2
// One block per query; comparisons are exact fp32 like the reference's numpy.
// valid (optional, n_text bytes): captions that exist (MSRVTT videos with fewer captions, metric.py:104-111,160-176):
// a missing caption is no candidate and no ground truth in mode 1; mode 0 ranks every row, the caller drops the masked ones.
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ x, long ld, int n_text, int n_vid, int q,
                                                             int mode, const unsigned char* __restrict__ valid,
                                                             float* __restrict__ ranks) {
    __shared__ int red[2][4];
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = mode == 0 ? n_vid : n_text;               // candidates of this query
    const long cs = mode == 0 ? 1 : ld;                      // stride between candidates
    const float* base = mode == 0 ? x + (long)i * ld : x + i;
    const int ngt = mode == 0 ? 1 : q;
    float best = 3.0e38f;
    for (int c = 0; c < ngt; ++c) {
        const int gt_idx = mode == 0 ? i / q : i * q + c;
        if (mode == 1 && valid && !valid[gt_idx]) continue;  // block-uniform
        const float gt = base[(long)gt_idx * cs];
        int greater = 0, equal = 0;
        for (int k = threadIdx.x; k < n; k += 256) {
            const float v = base[(long)k * cs];
            const bool ok = !(mode == 1 && valid && !valid[k]);
            greater += ok && v > gt;
            equal += ok && v == gt;
        }
        greater = (int)wave_sum((float)greater);  // counts < 2^24: exact in fp32
        equal = (int)wave_sum((float)equal);
        __syncthreads();
        if (lane == 0) { red[0][wave] = greater; red[1][wave] = equal; }
        __syncthreads();
        const int g_all = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const int e_all = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float rank = mode == 0 ? (float)g_all : (float)g_all + 0.5f * (float)(e_all - 1);
        best = rank < best ? rank : best;
    }
    if (threadIdx.x == 0) ranks[i] = best;
}
extern "C" int tvts_retrieval_ranks(const float* sims, long ld, int n_text, int n_vid, int mode,
                                    const unsigned char* valid, float* ranks, hipStream_t stream) {
    if (n_text <= 0 || n_vid <= 0 || n_text % n_vid || (mode != 0 && mode != 1) || n_text >= (1 << 24)) return TVTS_EINVAL;
    const int q = n_text / n_vid;
    hipLaunchKernelGGL(retrieval_rank_kernel, dim3(mode == 0 ? n_text : n_vid), dim3(256), 0, stream, sims, ld, n_text, n_vid, q,
                       mode, valid, ranks);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
