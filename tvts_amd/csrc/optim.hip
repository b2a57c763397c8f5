// Optimizer-side kernels over the FLAT fp32 parameter / gradient / moment buffers.
//
//   tvts_adamw_hf        multi-tensor AdamW with Hugging Face semantics (transformers==4.10.2 `AdamW`, the optimizer
//                        the reference builds at v2/train_dist_TVTSv2_ViT_B_16.py:118-125): bias-corrected step,
//                        eps added OUTSIDE the correction, decoupled decay `p -= lr*wd*p` AFTER the Adam update.
//                        One launch for all ~400 tensors: the flat buffers are cut into 1024-element chunks and a
//                        byte table gives each chunk its parameter group (255 = frozen / padding).  The same pass
//                        applies the data-parallel 1/world gradient scale and refreshes the bf16 weight shadow.
//   tvts_cast_f32_bf16   shadow refresh when an external optimizer owned the update (drop-in path)
//   tvts_transpose_bf16_batched   [N,K] -> [K,N] copies of every GEMM weight (dgrad operand), one launch
//   tvts_pad_rows_bf16 / tvts_add_rows_f32   K-padding of the 14x14 patch-embedding weight and its wgrad (H/14)
#include "common.h"

struct AdamGroups { float lr[4]; float wd[4]; float step_size[4]; };

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow,
                                                    const unsigned char* __restrict__ chunk_group, AdamGroups hp, float beta1,
                                                    float beta2, float omb1, float omb2, float eps, float grad_scale,
                                                    const int* __restrict__ step_dev, const float* __restrict__ hyper_dev,
                                                    double beta1d, double beta2d) {
    const int grp = chunk_group[blockIdx.x];
    if (grp > 3) return;
    if (hyper_dev) {  // lr[4] | wd[4] in device memory: a captured launch follows the LR schedule without re-capture
        hp.lr[grp] = hyper_dev[grp];
        hp.wd[grp] = hyper_dev[4 + grp];
    }
    if (step_dev) {  // step counter lives in device memory (hipGraph replay): bias correction computed here
        const double st = (double)step_dev[0];
        // the betas in double, as the host path has them: both paths then produce the same step size bit for bit (with the
        // float betas the step differed by 8e-6 relative, enough to flip a few dozen bf16 shadow roundings per step)
        const double bc1 = 1.0 - pow(beta1d, st), bc2 = 1.0 - pow(beta2d, st);
        hp.step_size[grp] = (float)((double)hp.lr[grp] * sqrt(bc2) / bc1);
    }
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    f32x4 pv = *(f32x4*)(p + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
    const f32x4 gv = *(const f32x4*)(g + i);
    const float lr = hp.lr[grp], wd = hp.wd[grp], ss = hp.step_size[grp];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] * grad_scale;
        mv[e] = beta1 * mv[e] + omb1 * gg;
        vv[e] = beta2 * vv[e] + omb2 * gg * gg;
        pv[e] -= ss * mv[e] / (sqrtf(vv[e]) + eps);
        if (wd > 0.f) pv[e] -= lr * wd * pv[e];
    }
    *(f32x4*)(p + i) = pv;
    *(f32x4*)(m + i) = mv;
    *(f32x4*)(v + i) = vv;
    if (shadow) *(bf16x4*)(shadow + i) = (bf16x4){(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
}

extern "C" int tvts_adamw_hf(float* p, const float* g, float* m, float* v, void* shadow_bf16,
                             const unsigned char* chunk_group, int nchunks, const float* lr4, const float* wd4, int step,
                             const int* step_dev, const float* hyper_dev, double beta1, double beta2, double eps,
                             float grad_scale, hipStream_t stream) {
    if (nchunks <= 0 || (step <= 0 && !step_dev)) return TVTS_EINVAL;
    if (hyper_dev && !step_dev) return TVTS_EINVAL;  // the device table is only consulted together with the device step counter
    if (step <= 0) step = 1;
    AdamGroups hp;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    for (int i = 0; i < 4; ++i) {
        hp.lr[i] = lr4[i];
        hp.wd[i] = wd4[i];
        hp.step_size[i] = (float)((double)lr4[i] * sqrt(bc2) / bc1);
    }
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, stream, p, g, m, v, (bf16*)shadow_bf16, chunk_group, hp,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, grad_scale, step_dev, hyper_dev, beta1, beta2);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(src + i * 4);
        *(bf16x4*)(dst + i * 4) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
}
__global__ __launch_bounds__(256) void widen_kernel(const bf16* __restrict__ src, float* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const bf16x4 v = *(const bf16x4*)(src + i * 4);
        *(f32x4*)(dst + i * 4) = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}
// bf16 -> fp32 (the bf16 gradient payload read back into the fp32 gradient buffer after its all-reduce)
extern "C" int tvts_cast_bf16_f32(const void* src, float* dst, long n, hipStream_t stream) {
    if (n <= 0 || n % 4) return TVTS_EINVAL;
    const size_t n4 = (size_t)n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)src, dst, n4);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
extern "C" int tvts_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream) {
    if (n <= 0 || n % 4) return TVTS_EINVAL;
    const size_t n4 = (size_t)n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (bf16*)dst, n4);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// tile table entry: {src_off, dst_off (elements, as two int32 halves each), R, C, tile_r, tile_c}
struct TrTile { long long src_off, dst_off; int R, C, tr, tc; };

__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                const TrTile* __restrict__ tiles) {
    __shared__ __attribute__((aligned(16))) bf16 t[64][72];  // 144-byte rows: 16-byte stores stay aligned, column reads spread over the banks
    const TrTile e = tiles[blockIdx.x];
    const bf16* s = src + e.src_off;
    bf16* d = dst + e.dst_off;
    const int r0 = e.tr * 64, c0 = e.tc * 64;
    if (e.R % 8 == 0 && e.C % 8 == 0 && e.src_off % 8 == 0 && e.dst_off % 8 == 0) {
        // 16 bytes per lane on both sides (2-byte accesses ran this kernel at a third of the HBM rate: 1.8 ms for the 0.63 G bf16
        // weights of ViT-H/14): a lane loads 8 consecutive columns of a row, and stores 8 consecutive rows of a column
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x + 256 * k, r = i >> 3, c8 = (i & 7) * 8;
            if (r0 + r < e.R && c0 + c8 < e.C) *(bf16x8*)&t[r][c8] = *(const bf16x8*)(s + (size_t)(r0 + r) * e.C + c0 + c8);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x + 256 * k, c = i >> 3, r8 = (i & 7) * 8;
            if (c0 + c < e.C && r0 + r8 < e.R) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = t[r8 + j][c];
                *(bf16x8*)(d + (size_t)(c0 + c) * e.R + r0 + r8) = v;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (r0 + r < e.R && c0 + c < e.C) t[r][c] = s[(size_t)(r0 + r) * e.C + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < e.R && c0 + c < e.C) d[(size_t)(c0 + c) * e.R + r0 + r] = t[r][c];
    }
}
extern "C" int tvts_transpose_bf16_batched(const void* src, void* dst, const void* tiles, int ntiles, hipStream_t stream) {
    if (ntiles <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(ntiles), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst,
                       (const TrTile*)tiles);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// Zero-padded bf16 row copy: dst[r, 0:cols_pad] = src[r, 0:cols] | 0.  The H/14 patch-embedding weight ([W, 588]) gets
// its K padded to 640 this way so the MFMA GEMM's 64-wide K loop and 16-byte row alignment hold.
__global__ __launch_bounds__(256) void pad_rows_kernel(const bf16* __restrict__ src, int lds_, bf16* __restrict__ dst, int ldd,
                                                       int cols, int cols_pad) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols_pad; c += 256)
        dst[(size_t)r * ldd + c] = c < cols ? src[(size_t)r * lds_ + c] : (bf16)0.f;
}
extern "C" int tvts_pad_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, int cols_pad,
                                  hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || cols_pad < cols || ld_dst < cols_pad || ld_src < cols) return TVTS_EINVAL;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(rows), dim3(256), 0, stream, (const bf16*)src, ld_src, (bf16*)dst, ld_dst, cols,
                       cols_pad);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// dst[r, c] += src[r, c] for c < cols (fp32, independent row strides): folds the padded wgrad scratch back into the
// [W, 588] gradient of the patch-embedding weight.
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds_,
                                                       int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) dst[(size_t)r * ldd + c] += src[(size_t)r * lds_ + c];
}
extern "C" int tvts_add_rows_f32(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols,
                                 hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || ld_dst < cols || ld_src < cols) return TVTS_EINVAL;
    hipLaunchKernelGGL(add_rows_kernel, dim3(rows), dim3(256), 0, stream, dst, ld_dst, src, ld_src, cols);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---- fp8 (OCP e4m3) quantisation with one scale per tensor: scale = amax / 448, q = rne(x / scale), |q| <= 448
//      (the weight / activation format of tvts_gemm_nt_fp8; torch.float8_e4m3fn bit patterns)
// 16 bytes per lane and access: 8 bf16 or 4 fp32 (cols % 8 == 0 resp. % 4, rows 16-byte aligned)
template <typename T> struct Vec16;
template <> struct Vec16<bf16> { typedef bf16x8 V; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 V; static constexpr int N = 4; };

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, long ld, int rows, int cols, float* __restrict__ amax) {
    typedef typename Vec16<T>::V V;
    constexpr int N = Vec16<T>::N;
    float m = 0.f;
    for (long r = blockIdx.x; r < rows; r += gridDim.x)
        for (int c = threadIdx.x * N; c < cols; c += 256 * N) {
            const V v = *(const V*)(x + r * ld + c);
#pragma unroll
            for (int e = 0; e < N; ++e) m = fmaxf(m, fabsf((float)v[e]));
        }
    m = wave_max(m);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)  // one atomic per block (every wave hammering the single address serialises the whole kernel)
        atomicMax((unsigned*)amax, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));  // non-negative floats order like their bits
}
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const T* __restrict__ x, long ld, int rows, int cols,
                                                        const float* __restrict__ amax, unsigned char* __restrict__ out, long ldo,
                                                        float* __restrict__ scale_out) {
    typedef typename Vec16<T>::V V;
    constexpr int N = Vec16<T>::N;
    const float am = amax[0];
    const float scale = am > 0.f ? am / 448.0f : 1.0f;
    const float inv = 1.0f / scale;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) scale_out[0] = scale;
    for (long r = blockIdx.x; r < rows; r += gridDim.x)
        for (int c = threadIdx.x * N; c < cols; c += 256 * N) {
            const V v = *(const V*)(x + r * ld + c);
            int pk[N / 4];
#pragma unroll
            for (int h = 0; h < N / 4; ++h) {
                float f[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf((float)v[h * 4 + e] * inv, -448.0f), 448.0f);
                int p = 0;
                p = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], p, false);
                p = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], p, true);
                pk[h] = p;
            }
            if (N == 8) *(long*)(out + r * ldo + c) = ((long)(unsigned)pk[N / 4 - 1] << 32) | (unsigned)pk[0];
            else *(int*)(out + r * ldo + c) = pk[0];
        }
}
static __global__ void zero1_kernel(float* p) { *p = 0.f; }
extern "C" int tvts_amax(const void* x, int is_f32, long ld, int rows, int cols, float* amax, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || cols % (is_f32 ? 4 : 8) || ld % (is_f32 ? 4 : 8)) return TVTS_EINVAL;
    hipLaunchKernelGGL(zero1_kernel, dim3(1), dim3(1), 0, stream, amax);  // a kernel, not a memset node (see attention.hip)
    const int blocks = rows < 1024 ? rows : 1024;
    if (is_f32) hipLaunchKernelGGL(amax_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, ld, rows, cols, amax);
    else hipLaunchKernelGGL(amax_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, ld, rows, cols, amax);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
extern "C" int tvts_quant_fp8(const void* x, int is_f32, long ld, int rows, int cols, const float* amax, void* out, long ldo,
                              float* scale_out, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || cols % (is_f32 ? 4 : 8) || ldo % 8 || ld % (is_f32 ? 4 : 8)) return TVTS_EINVAL;
    const int blocks = rows < 2048 ? rows : 2048;
    if (is_f32) hipLaunchKernelGGL(quant_fp8_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, ld, rows, cols, amax,
                                   (unsigned char*)out, ldo, scale_out);
    else hipLaunchKernelGGL(quant_fp8_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, ld, rows, cols, amax,
                            (unsigned char*)out, ldo, scale_out);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---- all weights of the fp8 path in three launches (was 3-4 launches per weight, 224 weights on ViT-H/14): one table entry per
//      weight -- fp32 master [rows, cols] (contiguous), its e4m3 copy, optionally the bf16 TRANSPOSED shadow [cols, rows] and its
//      e4m3 copy (the input-gradient operand), amax / scale scalars.  Pass 1 zeroes the amax scalars, pass 2 takes max |w| of the
//      master (blocks x tensors, one atomicMax per block), pass 3 converts both copies under the one scale (a bf16 rounding past
//      the master's amax saturates at +-448).  Same bit patterns as tvts_amax + tvts_quant_fp8 per tensor.
struct Q8Desc {
    const float* w; unsigned char* q; const bf16* wt; unsigned char* qt; float* amax; float* scale; float* scale_t;
    int rows, cols;
};
__global__ void q8_multi_zero_kernel(const Q8Desc* __restrict__ tab, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) *tab[i].amax = 0.f;
}
__global__ __launch_bounds__(256) void q8_multi_amax_kernel(const Q8Desc* __restrict__ tab) {
    const Q8Desc d = tab[blockIdx.y];
    const long n4 = (long)d.rows * d.cols / 4;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(d.w + i * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax((unsigned*)d.amax, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float e, float inv) {
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a * inv, -448.0f), 448.0f), fminf(fmaxf(b * inv, -448.0f), 448.0f), p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c * inv, -448.0f), 448.0f), fminf(fmaxf(e * inv, -448.0f), 448.0f), p, true);
    return p;
}
__global__ __launch_bounds__(256) void q8_multi_quant_kernel(const Q8Desc* __restrict__ tab) {
    const Q8Desc d = tab[blockIdx.y];
    const float am = *d.amax;
    const float scale = am > 0.f ? am / 448.0f : 1.0f;
    const float inv = 1.0f / scale;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *d.scale = scale; if (d.scale_t) *d.scale_t = scale; }
    const long n4 = (long)d.rows * d.cols / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(d.w + i * 4);
        *(int*)(d.q + i * 4) = pack4_e4m3(v[0], v[1], v[2], v[3], inv);
        if (d.wt) {
            const bf16x4 t = *(const bf16x4*)(d.wt + i * 4);
            *(int*)(d.qt + i * 4) = pack4_e4m3((float)t[0], (float)t[1], (float)t[2], (float)t[3], inv);
        }
    }
}
extern "C" int tvts_quant_fp8_multi(const void* table, int n, hipStream_t stream) {
    if (!table || n <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(q8_multi_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const Q8Desc*)table, n);
    hipLaunchKernelGGL(q8_multi_amax_kernel, dim3(32, n), dim3(256), 0, stream, (const Q8Desc*)table);
    hipLaunchKernelGGL(q8_multi_quant_kernel, dim3(32, n), dim3(256), 0, stream, (const Q8Desc*)table);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---- activations: ONE scale per row (token), amax and conversion in a single pass.  A wave owns a row: it is read once into
//      registers (rows up to 5120 columns; wider rows are read a second time, from cache), reduced to its amax with wave
//      shuffles, scaled by 448 / amax and written as e4m3 bytes + the row's scale.  3 bytes of traffic per element instead of
//      the 5 of tvts_amax + tvts_quant_fp8, and a tighter scale than one amax for the whole tensor.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16* __restrict__ x, long ld, int rows, int cols,
                                                             unsigned char* __restrict__ out, long ldo, float* __restrict__ row_scale,
                                                             const float* __restrict__ tscale, float* __restrict__ amax_acc) {
    constexpr int MAXV = 10;  // 10 x 64 lanes x 8 bf16 = 5120 columns held in registers
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const bool in_regs = cols <= MAXV * 512;
    float run_amax = 0.f;
    for (long r = wave0; r < rows; r += nwaves) {
        const bf16* xr = x + r * ld;
        bf16x8 v[MAXV];
        float m = 0.f;
        if (in_regs) {
#pragma unroll
            for (int t = 0; t < MAXV; ++t) {
                const int c = (t * 64 + lane) * 8;
                if (c < cols) {
                    v[t] = *(const bf16x8*)(xr + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[t][e]));
                }
            }
        } else {
            for (int c = lane * 8; c < cols; c += 512) {
                const bf16x8 u = *(const bf16x8*)(xr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)u[e]));
            }
        }
        m = wave_max(m);
        run_amax = fmaxf(run_amax, m);
        const float scale = q8_scale(tscale, m);
        const float inv = 1.0f / scale;
        if (lane == 0 && row_scale) row_scale[r] = scale;
        unsigned char* orow = out + r * ldo;
        auto emit = [&](const bf16x8& u, int c) {
            int pk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float f[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf((float)u[h * 4 + e] * inv, -448.0f), 448.0f);
                int q = 0;
                q = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], q, false);
                q = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], q, true);
                pk[h] = q;
            }
            *(long*)(orow + c) = ((long)(unsigned)pk[1] << 32) | (unsigned)pk[0];
        };
        if (in_regs) {
#pragma unroll
            for (int t = 0; t < MAXV; ++t) {
                const int c = (t * 64 + lane) * 8;
                if (c < cols) emit(v[t], c);
            }
        } else {
            for (int c = lane * 8; c < cols; c += 512) emit(*(const bf16x8*)(xr + c), c);
        }
    }
    amax_publish(amax_acc, run_amax, lane);
}
// next step's scales from this step's amax values, for n tensors at once: scale[i] = amax[i] / 448 (unchanged while amax[i] == 0:
// a tensor that was not produced this step keeps its scale), amax[i] = 0.  A scale slot never stays 0: a tensor whose maximum was 0
// in the calibration step (the time branch's attention output under the reference's zero-initialised timeattn.qkv) gets the scale
// 1.0 -- the value q8_scale() quantises under when it meets a zero scale -- so that producer and consumers agree (the GEMMs
// dequantise with the slot's value: a raw 0 there would force the tensor's products to zero for one step)
__global__ void fp8_update_scales_kernel(float* __restrict__ amax, float* __restrict__ scale, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i];
    if (a > 0.f) scale[i] = a / 448.0f;
    else if (!(scale[i] > 0.f)) scale[i] = 1.0f;
    amax[i] = 0.f;
}
extern "C" int tvts_fp8_update_scales(float* amax, float* scale, int n, hipStream_t stream) {
    if (n <= 0 || !amax || !scale) return TVTS_EINVAL;
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, amax, scale, n);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
extern "C" int tvts_quant_fp8_rows(const void* x, long ld, int rows, int cols, void* out, long ldo, float* row_scale,
                                   const float* tscale, float* amax_acc, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || cols % 8 || ld % 8 || ldo % 8 || !x || !out || (!row_scale && !tscale)) return TVTS_EINVAL;
    const long want = ((long)rows + 3) / 4;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, ld, rows, cols,
                       (unsigned char*)out, ldo, row_scale, tscale, amax_acc);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// probe used by tests: what does ds_read_b64_tr_b16 return?  in: 16 x 64 bf16 row-major tile (row stride 160 B in LDS)
__global__ void probe_tr16_kernel(const bf16* __restrict__ in, bf16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char tile[16 * 160];
    const int lane = threadIdx.x;
    for (int c = lane; c < 16 * 8; c += 64) *(bf16x8*)(tile + (c >> 3) * 160 + (c & 7) * 16) = *(const bf16x8*)(in + (c >> 3) * 64 + (c & 7) * 8);
    __syncthreads();
    const int gq = lane >> 4, i = lane & 15;
    const char* p = tile + (gq * 4 + (i >> 2)) * 160 + (i & 3) * 8;
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))p);
    const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = tb[e];
}
extern "C" int tvts_probe_tr16(const void* in, void* out, hipStream_t stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, (const bf16*)in, (bf16*)out);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
