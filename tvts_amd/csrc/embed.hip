// Token-assembly kernels around the transformer stacks (all HBM-bound gathers / scatters, fp32 unless noted).
//
//  tvts_patch_gather        kept patches of every frame -> bf16 im2col rows [B*T*n, 3*p*p] in conv-weight order
//                           (v2/model/video_encoder_ViT_B_16.py:180-184 + tube-mask gather :201-215; only kept
//                           patches are embedded -- output-equivalent)
//  tvts_vit_assemble(+_bwd) CLS + patch embedding + spatial/temporal position (:185-198) -> tokens [B*S, W]
//  tvts_text_embed(+_bwd)   token embedding gather + positional add (model_dist_TVTSv2_ViT_B_16.py:98-100)
//  tvts_text_mean(+_bwd)    clip-major captions [NT*B,E] -> mean over NT and the [B,NT,E] copy for the sort head (:69-76)
//  tvts_sort_assemble(+_bwd) video tokens + type_embed[0] | caption embeddings + type_embed[1]
//                           (v2/model/sort_transformer.py:124-128)
#include "common.h"

// ---------------------------------------------------------------------------------------------- patch gather
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ video, const int* __restrict__ keep,
                                                           int B, int T, int n, int img, int p, bf16* __restrict__ out,
                                                           int ldo) {
    const int K = 3 * p * p;
    const int row = blockIdx.x;  // (b, f, i)
    const int i = row % n, f = (row / n) % T, b = row / (n * T);
    const int g = img / p;
    const int pi = keep[b * n + i];
    const int gy = pi / g, gx = pi % g;
    const float* fr = video + ((size_t)(b * T + f) * 3) * img * img;
    for (int c8 = threadIdx.x * 8; c8 < K; c8 += 256 * 8) {
        const int ch = c8 / (p * p), rem = c8 % (p * p), py = rem / p, px = rem % p;  // px multiple of 8
        const float* src = fr + ((size_t)ch * img + gy * p + py) * img + gx * p + px;
        const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
        bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)c[0], (bf16)c[1], (bf16)c[2], (bf16)c[3]};
        *(bf16x8*)(out + (size_t)row * ldo + c8) = o;
    }
}

// any patch size (H/14: p = 14, K = 588): one thread per output column, columns K..ldo-1 are written as zeros so the
// patch-embedding GEMM can run with its K padded to a multiple of 64.
__global__ __launch_bounds__(256) void patch_gather_any_kernel(const float* __restrict__ video, const int* __restrict__ keep,
                                                               int B, int T, int n, int img, int p, bf16* __restrict__ out,
                                                               int ldo) {
    const int K = 3 * p * p;
    const int row = blockIdx.x;
    const int i = row % n, f = (row / n) % T, b = row / (n * T);
    const int g = img / p;
    const int pi = keep[b * n + i];
    const int gy = pi / g, gx = pi % g;
    const float* fr = video + ((size_t)(b * T + f) * 3) * img * img;
    for (int c = threadIdx.x; c < ldo; c += 256) {
        float v = 0.f;
        if (c < K) {
            const int ch = c / (p * p), rem = c % (p * p), py = rem / p, px = rem % p;
            v = fr[((size_t)ch * img + gy * p + py) * img + gx * p + px];
        }
        out[(size_t)row * ldo + c] = (bf16)v;
    }
}

extern "C" int tvts_patch_gather(const float* video, const int* keep, int B, int T, int n, int img, int patch, void* out,
                                 int ldo, hipStream_t stream) {
    if (B <= 0 || T <= 0 || n <= 0 || patch <= 0 || img % patch || ldo % 8 || ldo < 3 * patch * patch) return TVTS_EINVAL;
    if (patch % 8 || ldo != 3 * patch * patch)
        hipLaunchKernelGGL(patch_gather_any_kernel, dim3(B * T * n), dim3(256), 0, stream, video, keep, B, T, n, img, patch,
                           (bf16*)out, ldo);
    else
        hipLaunchKernelGGL(patch_gather_kernel, dim3(B * T * n), dim3(256), 0, stream, video, keep, B, T, n, img, patch,
                           (bf16*)out, ldo);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// v1 (TVTS) tubelet embedding: Conv3d(3, W, kernel = stride = (tb, p, p)) over [B, C, T, H, W] (v1/model/video_encoder.py:78-99)
// as an im2col product over the KEPT patches of every tube only.  Row (b, tube, i) = patch keep[b, tube, i] of tube `tube`,
// columns in the Conv3d weight's (c, t, py, px) order: K = 3 * tb * p * p.  keep differs from tube to tube here
// (v1/data_loader/YTTemporal_dataset.py:211-215); video arrives as [B, T, 3, H, W] (the model permutes it itself, :179).
__global__ __launch_bounds__(256) void patch_gather_tube_kernel(const float* __restrict__ video, const int* __restrict__ keep,
                                                                int B, int tubes, int tb, int n, int img, int p,
                                                                bf16* __restrict__ out, int ldo) {
    const int pp = p * p, K = 3 * tb * pp;
    const int row = blockIdx.x;  // (b, tube, i)
    const int i = row % n, tu = (row / n) % tubes, b = row / (n * tubes);
    const int g = img / p;
    const int pi = keep[(size_t)(b * tubes + tu) * n + i];
    const int gy = pi / g, gx = pi % g;
    const int T = tubes * tb;
    for (int c8 = threadIdx.x * 8; c8 < K; c8 += 256 * 8) {
        const int ch = c8 / (tb * pp), r1 = c8 % (tb * pp), t = r1 / pp, rem = r1 % pp, py = rem / p, px = rem % p;  // px % 8 == 0
        const float* src = video + ((((size_t)(b * T + tu * tb + t) * 3 + ch) * img + gy * p + py) * img + gx * p + px);
        const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
        bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)c[0], (bf16)c[1], (bf16)c[2], (bf16)c[3]};
        *(bf16x8*)(out + (size_t)row * ldo + c8) = o;
    }
}
extern "C" int tvts_patch_gather_tube(const float* video, const int* keep, int B, int tubes, int tubelet, int n, int img,
                                      int patch, void* out, int ldo, hipStream_t stream) {
    if (B <= 0 || tubes <= 0 || tubelet <= 0 || n <= 0 || patch <= 0 || img % patch || patch % 8 || ldo != 3 * tubelet * patch * patch)
        return TVTS_EINVAL;
    hipLaunchKernelGGL(patch_gather_tube_kernel, dim3(B * tubes * n), dim3(256), 0, stream, video, keep, B, tubes, tubelet, n, img,
                       patch, (bf16*)out, ldo);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// uint8 wire format: the frames arrive as the decoder leaves them -- uint8, H x W x 3 interleaved, already
// resized on the host -- and the rest of the reference's transform chain runs here, fused into the tube-mask gather:
// crop (video_transform.CenterCrop / RandomCrop = an offset per sample), ClipToTensor (float32 / 255) and Normalize
// ((v - mean) / std, fp32, same operation order -> bit-identical to v2/video_transforms/video_transform.py:24-75,
// functional.py:81-97), then the bf16 rounding of the im2col row.  4x less PCIe / HBM traffic than fp32 frames.
// ytab / xtab (optional): the RESIZE in front of the crop (video_transform.Resize -> PIL nearest-neighbour, videoaug.py:12,21):
// frames are the decoder's Hs x Ws pictures, ytab[H0] / xtab[W0] give the source row / column of every row / column of the
// (virtual) resized H0 x W0 picture -- Pillow's own index table, built by tvts_amd/data_loader/transforms.py.
__global__ __launch_bounds__(256) void patch_gather_u8_kernel(const unsigned char* __restrict__ frames, int H0, int W0,
                                                              const int* __restrict__ crop, const int* __restrict__ keep,
                                                              int B, int T, int n, int img, int p, float m0, float m1,
                                                              float m2, float s0, float s1, float s2,
                                                              bf16* __restrict__ out, int ldo, const int* __restrict__ ytab,
                                                              const int* __restrict__ xtab, int Hs, int Ws) {
    const int K = 3 * p * p;
    const int row = blockIdx.x;  // (b, f, i)
    const int i = row % n, f = (row / n) % T, b = row / (n * T);
    const int g = img / p;
    const int pi = keep[b * n + i];
    const int gy = pi / g, gx = pi % g;
    // centre crop: int(round((H0 - img) / 2.)) with Python's round-half-to-even (video_transform.py:454-455)
    const int dy = H0 - img, dx = W0 - img;
    const int y0 = crop ? crop[2 * b] : (dy >> 1) + ((dy & 1) & (dy >> 1)), x0 = crop ? crop[2 * b + 1] : (dx >> 1) + ((dx & 1) & (dx >> 1));
    const unsigned char* fr = frames + (size_t)(b * T + f) * Hs * Ws * 3;
    for (int c = threadIdx.x; c < ldo; c += 256) {
        float v = 0.f;
        if (c < K) {
            const int ch = c / (p * p), rem = c % (p * p), py = rem / p, px = rem % p;
            int sy = y0 + gy * p + py, sx = x0 + gx * p + px;  // pixel of the (resized) H0 x W0 picture
            if (ytab) { sy = ytab[sy]; sx = xtab[sx]; }
            const float u = (float)fr[((size_t)sy * Ws + sx) * 3 + ch];
            const float mean = ch == 0 ? m0 : ch == 1 ? m1 : m2, sd = ch == 0 ? s0 : ch == 1 ? s1 : s2;
            v = (u / 255.0f - mean) / sd;
        }
        out[(size_t)row * ldo + c] = (bf16)v;
    }
}
extern "C" int tvts_patch_gather_u8(const unsigned char* frames, int H0, int W0, const int* crop, const int* keep, int B,
                                    int T, int n, int img, int patch, const float* mean3, const float* std3, void* out,
                                    int ldo, hipStream_t stream) {
    if (B <= 0 || T <= 0 || n <= 0 || patch <= 0 || img % patch || H0 < img || W0 < img || ldo % 8 || ldo < 3 * patch * patch)
        return TVTS_EINVAL;
    hipLaunchKernelGGL(patch_gather_u8_kernel, dim3(B * T * n), dim3(256), 0, stream, frames, H0, W0, crop, keep, B, T, n, img,
                       patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16*)out, ldo, nullptr, nullptr, H0, W0);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// the same with the nearest-neighbour resize Hs x Ws -> H0 x W0 in front (ytab / xtab: device int32 [H0] / [W0])
extern "C" int tvts_patch_gather_u8_resized(const unsigned char* frames, int Hs, int Ws, const int* ytab, const int* xtab, int H0,
                                            int W0, const int* crop, const int* keep, int B, int T, int n, int img, int patch,
                                            const float* mean3, const float* std3, void* out, int ldo, hipStream_t stream) {
    if (B <= 0 || T <= 0 || n <= 0 || patch <= 0 || img % patch || H0 < img || W0 < img || ldo % 8 || ldo < 3 * patch * patch ||
        !ytab || !xtab || Hs <= 0 || Ws <= 0)
        return TVTS_EINVAL;
    hipLaunchKernelGGL(patch_gather_u8_kernel, dim3(B * T * n), dim3(256), 0, stream, frames, H0, W0, crop, keep, B, T, n, img,
                       patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16*)out, ldo, ytab, xtab, Hs, Ws);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- ViT token assemble
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch, int ldp,
                                                           const float* __restrict__ cls, const float* __restrict__ pos,
                                                           const float* __restrict__ temporal, const int* __restrict__ keep,
                                                           int keep_per_frame, int B, int T, int n, int W,
                                                           float* __restrict__ tok, int ldt) {
    const int S = 1 + T * n;
    const int row = blockIdx.x;  // b*S + s
    const int b = row / S, s = row % S;
    for (int c = threadIdx.x * 4; c < W; c += 1024) {
        f32x4 v;
        if (s == 0) {
            v = *(const f32x4*)(cls + c) + *(const f32x4*)(pos + c);
        } else {
            const int f = (s - 1) / n, i = (s - 1) % n;
            v = *(const f32x4*)(patch + (size_t)((b * T + f) * n + i) * ldp + c) +
                *(const f32x4*)(pos + (size_t)(1 + keep[keep_per_frame ? (b * T + f) * n + i : b * n + i]) * W + c) +
                *(const f32x4*)(temporal + (size_t)f * W + c);
        }
        *(f32x4*)(tok + (size_t)row * ldt + c) = v;
    }
}

extern "C" int tvts_vit_assemble(const float* patch, int ldp, const float* cls, const float* pos, const float* temporal,
                                 const int* keep, int keep_per_frame, int B, int T, int n, int W, float* tok, int ldt,
                                 hipStream_t stream) {
    if (W % 4 || ldp % 4 || ldt % 4) return TVTS_EINVAL;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(B * (1 + T * n)), dim3(256), 0, stream, patch, ldp, cls, pos, temporal,
                       keep, keep_per_frame, B, T, n, W, tok, ldt);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// one block per (b, f): d_patch rows (bf16, compact im2col row order), dpos scatter, dtemporal / dcls sums
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const float* __restrict__ dtok, int ldt,
                                                               const int* __restrict__ keep, int keep_per_frame, int B, int T,
                                                               int n, int W, bf16* __restrict__ dpatch, int ldp,
                                                               float* __restrict__ dcls,
                                                               float* __restrict__ dpos, float* __restrict__ dtemporal,
                                                               float* __restrict__ part_t, float* __restrict__ part_c,
                                                               float* __restrict__ part_p) {
    // part_t [B][T][W] / part_c [B][W] / part_p [B * T][n][W] (optional, all or none): the block's sums and its slot values go to
    // partials that colsum_partials_kernel / pos_rows_from_partials_kernel add in a fixed order, instead of fp32 atomics
    const int S = 1 + T * n;
    const int b = blockIdx.x / T, f = blockIdx.x % T;
    for (int c = threadIdx.x; c < W; c += 256) {
        float ts = 0.f;
        for (int i = 0; i < n; ++i) {
            const float v = dtok[(size_t)(b * S + 1 + f * n + i) * ldt + c];
            dpatch[(size_t)((b * T + f) * n + i) * ldp + c] = (bf16)v;
            if (part_p) part_p[((size_t)(b * T + f) * n + i) * W + c] = v;
            else atomicAdd(dpos + (size_t)(1 + keep[keep_per_frame ? (b * T + f) * n + i : b * n + i]) * W + c, v);
            ts += v;
        }
        if (part_t) part_t[((size_t)b * T + f) * W + c] = ts;
        else atomicAdd(dtemporal + (size_t)f * W + c, ts);
        if (f == 0) {
            const float v0 = dtok[(size_t)(b * S) * ldt + c];
            if (part_c) part_c[(size_t)b * W + c] = v0;
            else { atomicAdd(dcls + c, v0); atomicAdd(dpos + c, v0); }
        }
    }
}

// tube masks (one keep list per clip, shared by its frames -- the TVTSv2 batch): a block per (clip, group of ASM_SLOTS patch
// slots) sweeps the T frames of a slot, so that the positional-embedding gradient gets ONE atomic per (slot, column) instead of
// one per frame, and the temporal-embedding sums stay in registers over the block's slots (116 M -> 23 M fp32 atomics at 192
// clips x 8 frames x 98 patches x 768 columns: the atomics were the kernel's run time, 465 us); 16-byte loads, 8-byte stores.
#define ASM_SLOTS 14
#define ASM_MAXT 16
// part_t / part_c (optional): per-block partials [B * groups][T][W] of the temporal-embedding sums and [B][W] of the CLS rows,
// added in block order by colsum_partials_kernel (no atomics: the sums are run-to-run reproducible).  part_p (optional): the
// frame sums of every kept patch slot [B][n][W]; pos_rows_from_partials_kernel then adds, for every row of the positional
// embedding, the slots that point at it in clip order.  Without part_p the rows are a scatter of fp32 atomics.
__global__ __launch_bounds__(256) void vit_assemble_bwd_tube_kernel(const float* __restrict__ dtok, int ldt,
                                                                    const int* __restrict__ keep, int B, int T, int n, int W,
                                                                    bf16* __restrict__ dpatch, int ldp, float* __restrict__ dcls,
                                                                    float* __restrict__ dpos, float* __restrict__ dtemporal,
                                                                    float* __restrict__ part_t, float* __restrict__ part_c,
                                                                    float* __restrict__ part_p) {
    const int S = 1 + T * n;
    const int groups = (n + ASM_SLOTS - 1) / ASM_SLOTS;
    const int b = blockIdx.x / groups, grp = blockIdx.x % groups;
    const int i0 = grp * ASM_SLOTS, i1 = (i0 + ASM_SLOTS < n) ? i0 + ASM_SLOTS : n;
    for (int c = threadIdx.x * 4; c < W; c += 1024) {
        f32x4 ts[ASM_MAXT];
#pragma unroll
        for (int f = 0; f < ASM_MAXT; ++f) ts[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = i0; i < i1; ++i) {
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < ASM_MAXT; ++f) {
                if (f < T) {
                    const f32x4 v = *(const f32x4*)(dtok + (size_t)(b * S + 1 + f * n + i) * ldt + c);
                    *(bf16x4*)(dpatch + (size_t)((b * T + f) * n + i) * ldp + c) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    ps += v;
                    ts[f] += v;
                }
            }
            if (part_p) {
                *(f32x4*)(part_p + ((size_t)b * n + i) * W + c) = ps;
            } else {
                float* dp = dpos + (size_t)(1 + keep[b * n + i]) * W + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dp + e, ps[e]);
            }
        }
#pragma unroll
        for (int f = 0; f < ASM_MAXT; ++f) {
            if (f < T) {
                if (part_t) *(f32x4*)(part_t + ((size_t)blockIdx.x * T + f) * W + c) = ts[f];
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(dtemporal + (size_t)f * W + c + e, ts[f][e]);
                }
            }
        }
        if (grp == 0) {
            const f32x4 v0 = *(const f32x4*)(dtok + (size_t)(b * S) * ldt + c);
            if (part_c) *(f32x4*)(part_c + (size_t)b * W + c) = v0;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { atomicAdd(dcls + c + e, v0[e]); atomicAdd(dpos + c + e, v0[e]); }
            }
        }
    }
}

// out[c] (and out2[c]) += sum_p part[p][c] for c < W: 64 columns x 16 partial-groups per block, every thread adds its partials
// in order, the 16 groups are combined in order
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ part, int nparts, int W,
                                                               float* __restrict__ out, float* __restrict__ out2) {
    __shared__ float acc[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (c < W)
        for (int p = grp; p < nparts; p += 16) s += part[(size_t)p * W + c];
    acc[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && c < W) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += acc[g][threadIdx.x];
        out[c] += t;
        if (out2) out2[c] += t;
    }
}

// dpos[1 + r] += sum over the clips b of the slot sums part_p[b][i] with keep[b][i] == r, in a fixed order: a block per (row r of the
// positional embedding, 256 columns).  Its threads first look up "their" clip's slot (256 clips at a time; a tube mask holds a
// position at most once, further matches of a malformed one are added too); then 4 thread groups take every 4th clip, a thread
// four columns, and the groups are combined in group order.
__global__ __launch_bounds__(256) void pos_rows_from_partials_kernel(const float* __restrict__ part_p, const int* __restrict__ keep,
                                                                     int B, int n, int W, float* __restrict__ dpos) {
    __shared__ int first[256], more[256];
    __shared__ f32x4 comb[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int grp = tid >> 6, c = blockIdx.y * 256 + (tid & 63) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 256) {
        int f = -1, cnt = 0;
        if (b0 + tid < B) {
            const int* kp = keep + (size_t)(b0 + tid) * n;
            for (int i = 0; i < n; ++i)
                if (kp[i] == r) { if (f < 0) f = i; ++cnt; }
        }
        first[tid] = f; more[tid] = cnt > 1;
        __syncthreads();
        const int nb = B - b0 < 256 ? B - b0 : 256;
        if (c < W) {
            for (int k = grp; k < nb; k += 4) {
                const int fi = first[k];
                if (fi < 0) continue;
                acc += *(const f32x4*)(part_p + ((size_t)(b0 + k) * n + fi) * W + c);
                if (more[k]) {
                    const int* kp = keep + (size_t)(b0 + k) * n;
                    for (int i = fi + 1; i < n; ++i)
                        if (kp[i] == r) acc += *(const f32x4*)(part_p + ((size_t)(b0 + k) * n + i) * W + c);
                }
            }
        }
        __syncthreads();
    }
    comb[tid] = acc;
    __syncthreads();
    if (grp == 0 && c < W) {
        const f32x4 t = ((comb[tid] + comb[64 + tid]) + comb[128 + tid]) + comb[192 + tid];
        *(f32x4*)(dpos + (size_t)(1 + r) * W + c) += t;
    }
}

extern "C" int tvts_vit_assemble_bwd(const float* dtok, int ldt, const int* keep, int keep_per_frame, int B, int T, int n, int W,
                                     void* dpatch, int ldp, float* dcls, float* dpos, int n_pos, float* dtemporal, float* workspace,
                                     long workspace_elems, hipStream_t stream) {
    if (!keep_per_frame && T <= ASM_MAXT && W % 4 == 0 && ldt % 4 == 0 && ldp % 4 == 0) {
        const int groups = (n + ASM_SLOTS - 1) / ASM_SLOTS;
        const long need = (long)B * groups * T * W + (long)B * W;
        float* part_t = (workspace && workspace_elems >= need) ? workspace : nullptr;
        float* part_c = part_t ? workspace + (size_t)B * groups * T * W : nullptr;
        float* part_p = (part_t && n_pos > 0 && workspace_elems >= need + (long)B * n * W) ? workspace + need : nullptr;
        hipLaunchKernelGGL(vit_assemble_bwd_tube_kernel, dim3(B * groups), dim3(256), 0, stream, dtok, ldt, keep, B, T, n, W,
                           (bf16*)dpatch, ldp, dcls, dpos, dtemporal, part_t, part_c, part_p);
        if (part_p)
            hipLaunchKernelGGL(pos_rows_from_partials_kernel, dim3(n_pos, ceil_div(W, 256)), dim3(256), 0, stream, part_p, keep, B, n, W, dpos);
        if (part_t) {
            hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(T * W, 64)), dim3(1024), 0, stream, part_t, B * groups, T * W,
                               dtemporal, (float*)nullptr);
            hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(W, 64)), dim3(1024), 0, stream, part_c, B, W, dcls, dpos);
        }
        TVTS_LAUNCH_CHECK();
        return TVTS_OK;
    }
    {   // one keep list per frame / tubelet (v1), or shapes the tube kernel does not take
        const long need = (long)B * T * W + (long)B * W + (long)B * T * n * W;
        const bool ordered = workspace && workspace_elems >= need && n_pos > 0 && keep_per_frame && W % 4 == 0;
        float* part_t = ordered ? workspace : nullptr;
        float* part_c = ordered ? part_t + (size_t)B * T * W : nullptr;
        float* part_p = ordered ? part_c + (size_t)B * W : nullptr;
        hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(B * T), dim3(256), 0, stream, dtok, ldt, keep, keep_per_frame, B, T, n, W,
                           (bf16*)dpatch, ldp, dcls, dpos, dtemporal, part_t, part_c, part_p);
        if (ordered) {
            // the (clip, frame) pairs are the gather's "clips": keep[B * T][n], slot values [B * T][n][W]
            hipLaunchKernelGGL(pos_rows_from_partials_kernel, dim3(n_pos, ceil_div(W, 256)), dim3(256), 0, stream, part_p, keep, B * T, n, W, dpos);
            hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(T * W, 64)), dim3(1024), 0, stream, part_t, B, T * W,
                               dtemporal, (float*)nullptr);
            hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(W, 64)), dim3(1024), 0, stream, part_c, B, W, dcls, dpos);
        }
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- text embedding
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ ids, int ld_ids, int N, int L,
                                                         const float* __restrict__ emb, const float* __restrict__ pos,
                                                         int Wt, float* __restrict__ x, int ldx) {
    const int row = blockIdx.x;  // n*L + l
    const int nn = row / L, l = row % L;
    const int id = ids[(size_t)nn * ld_ids + l];
    for (int c = threadIdx.x * 4; c < Wt; c += 1024)
        *(f32x4*)(x + (size_t)row * ldx + c) = *(const f32x4*)(emb + (size_t)id * Wt + c) + *(const f32x4*)(pos + (size_t)l * Wt + c);
}
extern "C" int tvts_text_embed(const int* ids, int ld_ids, int N, int L, const float* emb, const float* pos, int Wt,
                               float* x, int ldx, hipStream_t stream) {
    if (Wt % 4 || ldx % 4) return TVTS_EINVAL;
    hipLaunchKernelGGL(text_embed_kernel, dim3(N * L), dim3(256), 0, stream, ids, ld_ids, N, L, emb, pos, Wt, x, ldx);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const float* __restrict__ dx, int ldx, const int* __restrict__ ids,
                                                             int ld_ids, int N, int L, int Wt, float* __restrict__ demb,
                                                             float* __restrict__ dpos) {
    const int row = blockIdx.x;
    const int nn = row / L, l = row % L;
    const int id = ids[(size_t)nn * ld_ids + l];
    for (int c = threadIdx.x; c < Wt; c += 256) {
        const float v = dx[(size_t)row * ldx + c];
        if (v != 0.f) {
            atomicAdd(demb + (size_t)id * Wt + c, v);
            atomicAdd(dpos + (size_t)l * Wt + c, v);
        }
    }
}
// the same sums without atomics (run-to-run reproducible).  Positions: dpos[l] += sum_n dx[n * L + l], a block per (64 columns,
// position), 16 thread groups over the captions, each in caption order, combined in group order.
__global__ __launch_bounds__(1024) void text_pos_bwd_kernel(const float* __restrict__ dx, int ldx, int N, int L, int Wt,
                                                            float* __restrict__ dpos) {
    __shared__ float acc[16][64];
    const int l = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Wt)
        for (int nn = grp; nn < N; nn += 16) s += dx[((size_t)nn * L + l) * ldx + c];
    acc[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && c < Wt) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += acc[g][threadIdx.x];
        dpos[(size_t)l * Wt + c] += t;
    }
}
// Tokens: the caller hands over the rows sorted by token id (`order`, ties in row order) and the starts of the runs of equal ids
// (`seg`, N * L + 1 entries, padded with N * L: a fixed grid, empty runs exit).  A block per run adds its rows in order --
// G = 256 / (Wt / 4) thread groups take every G-th row, each thread four columns, the groups are combined in group order.
#define TOK_LONG_RUN 64     // a run of more rows than this, listed among the first TOK_LONG_SLOTS runs, is summed by a block per 64 columns
#define TOK_LONG_SLOTS 64
__global__ __launch_bounds__(256) void text_tok_bwd_kernel(const float* __restrict__ dx, int ldx, const int* __restrict__ ids,
                                                           int ld_ids, int L, int Wt, const int* __restrict__ order,
                                                           const int* __restrict__ seg, float* __restrict__ demb) {
    __shared__ f32x4 part[256];
    const int lo = seg[blockIdx.x], hi = seg[blockIdx.x + 1];
    if (lo >= hi) return;
    if (hi - lo > TOK_LONG_RUN && blockIdx.x < TOK_LONG_SLOTS) return;  // text_tok_long_bwd_kernel's
    const int row0 = order[lo];
    const int id = ids[(size_t)(row0 / L) * ld_ids + row0 % L];
    const int cpt = Wt / 4 < 256 ? Wt / 4 : 256;  // column-threads per group
    const int G = 256 / cpt, grp = threadIdx.x / cpt, ct = threadIdx.x % cpt;
    for (int c0 = 0; c0 < Wt; c0 += cpt * 4) {
        const int c = c0 + ct * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (grp < G && c < Wt) {
            // sixteen rows in flight (the two id runs every caption shares -- start and end token -- are N rows long): the row
            // numbers first, then the rows (clamped, not predicated: a branch per load would serialise them), added in row order
            for (int k = lo + grp; k < hi; k += 16 * G) {
                int idx[16];
                f32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int kk = k + u * G;
                    idx[u] = order[kk < hi ? kk : hi - 1];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(dx + (size_t)idx[u] * ldx + c);
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (k + u * G < hi) a += v[u];
            }
        }
        part[threadIdx.x] = a;
        __syncthreads();
        if (grp == 0 && c < Wt) {
            f32x4 t = part[ct];
            for (int g = 1; g < G; ++g) t += part[g * cpt + ct];
            *(f32x4*)(demb + (size_t)id * Wt + c) += t;
        }
        __syncthreads();
    }
}
// the long runs (every caption's start / end token: N rows each): a block per (run, 64 columns), 16 thread groups over the rows,
// a thread four columns, sixteen rows in flight per thread, the groups combined in group order
__global__ __launch_bounds__(256) void text_tok_long_bwd_kernel(const float* __restrict__ dx, int ldx, const int* __restrict__ ids,
                                                                int ld_ids, int L, int Wt, const int* __restrict__ order,
                                                                const int* __restrict__ seg, float* __restrict__ demb) {
    __shared__ f32x4 part[256];
    const int lo = seg[blockIdx.x], hi = seg[blockIdx.x + 1];
    if (hi - lo <= TOK_LONG_RUN) return;
    const int row0 = order[lo];
    const int id = ids[(size_t)(row0 / L) * ld_ids + row0 % L];
    const int grp = threadIdx.x >> 4, c = blockIdx.y * 64 + (threadIdx.x & 15) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < Wt) {
        for (int k = lo + grp; k < hi; k += 16 * 16) {
            int idx[16];
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int kk = k + u * 16;
                idx[u] = order[kk < hi ? kk : hi - 1];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(dx + (size_t)idx[u] * ldx + c);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k + u * 16 < hi) a += v[u];
        }
    }
    part[threadIdx.x] = a;
    __syncthreads();
    if (grp == 0 && c < Wt) {
        f32x4 t = part[threadIdx.x];
        for (int g = 1; g < 16; ++g) t += part[g * 16 + threadIdx.x];
        *(f32x4*)(demb + (size_t)id * Wt + c) += t;
    }
}
extern "C" int tvts_text_embed_bwd(const float* dx, int ldx, const int* ids, int ld_ids, int N, int L, int Wt, float* demb,
                                   float* dpos, const int* order, const int* seg, hipStream_t stream) {
    if (order != nullptr && seg != nullptr && Wt % 4 == 0 && ldx % 4 == 0) {
        hipLaunchKernelGGL(text_pos_bwd_kernel, dim3(ceil_div(Wt, 64), L), dim3(1024), 0, stream, dx, ldx, N, L, Wt, dpos);
        hipLaunchKernelGGL(text_tok_bwd_kernel, dim3(N * L), dim3(256), 0, stream, dx, ldx, ids, ld_ids, L, Wt, order, seg, demb);
        const int slots = N * L < TOK_LONG_SLOTS ? N * L : TOK_LONG_SLOTS;
        hipLaunchKernelGGL(text_tok_long_bwd_kernel, dim3(slots, ceil_div(Wt, 64)), dim3(256), 0, stream, dx, ldx, ids, ld_ids, L, Wt,
                           order, seg, demb);
    } else {
        hipLaunchKernelGGL(text_embed_bwd_kernel, dim3(N * L), dim3(256), 0, stream, dx, ldx, ids, ld_ids, N, L, Wt, demb, dpos);
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- caption mean
__global__ void text_mean_kernel(const float* __restrict__ t, int NT, int B, int E, float* __restrict__ mean,
                                 float* __restrict__ before) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (b, e)
    if (idx >= B * E) return;
    const int b = idx / E, e = idx % E;
    float s = 0.f;
    for (int i = 0; i < NT; ++i) {
        const float v = t[(size_t)(i * B + b) * E + e];
        s += v;
        if (before) before[((size_t)b * NT + i) * E + e] = v;
    }
    mean[idx] = s / (float)NT;
}
extern "C" int tvts_text_mean(const float* t, int NT, int B, int E, float* mean, float* before, hipStream_t stream) {
    hipLaunchKernelGGL(text_mean_kernel, dim3(ceil_div(B * E, 256)), dim3(256), 0, stream, t, NT, B, E, mean, before);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
__global__ void text_mean_bwd_kernel(const float* __restrict__ dmean, int NT, int B, int E, float* __restrict__ dt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (i, b, e)
    if (idx >= NT * B * E) return;
    const int e = idx % E, b = (idx / E) % B;
    dt[idx] = dmean[(size_t)b * E + e] / (float)NT;
}
extern "C" int tvts_text_mean_bwd(const float* dmean, int NT, int B, int E, float* dt, hipStream_t stream) {
    hipLaunchKernelGGL(text_mean_bwd_kernel, dim3(ceil_div(NT * B * E, 256)), dim3(256), 0, stream, dmean, NT, B, E, dt);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- sort head input
// xs[b, s] = tok[b*S + off + s] + type[0]  (s < Sv);  xs[b, Sv + i] = text[b, i] + type[1]
__global__ __launch_bounds__(256) void sort_assemble_kernel(const float* __restrict__ tok, int ldt, int S, int off, int Sv,
                                                            const float* __restrict__ text, int NT,
                                                            const float* __restrict__ type, int E, float* __restrict__ xs,
                                                            int ldx) {
    const int So = Sv + NT;
    const int row = blockIdx.x;
    const int b = row / So, s = row % So;
    const float* src = s < Sv ? tok + (size_t)(b * S + off + s) * ldt : text + ((size_t)b * NT + (s - Sv)) * E;
    const float* ty = type + (s < Sv ? 0 : E);
    for (int c = threadIdx.x * 4; c < E; c += 1024)
        *(f32x4*)(xs + (size_t)row * ldx + c) = *(const f32x4*)(src + c) + *(const f32x4*)(ty + c);
}
extern "C" int tvts_sort_assemble(const float* tok, int ldt, int B, int S, int off, int Sv, const float* text, int NT,
                                  const float* type, int E, float* xs, int ldx, hipStream_t stream) {
    if (E % 4 || ldt % 4 || ldx % 4) return TVTS_EINVAL;
    hipLaunchKernelGGL(sort_assemble_kernel, dim3(B * (Sv + NT)), dim3(256), 0, stream, tok, ldt, S, off, Sv, text, NT, type,
                       E, xs, ldx);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// d_out[b*S + r] (bf16, feeds the output-projection dgrad/wgrad) = [r >= off ? dxs[b, r-off] : 0] + [r == 0 ? dvid[b] : 0]
// dtype[0] += sum over video rows, dtype[1] += sum over caption rows.
// grid (B, row chunks of 32): each block converts its rows and adds its partial column sums with atomics.
__global__ __launch_bounds__(256) void sort_assemble_bwd_kernel(const float* __restrict__ dxs, int ldx, int S, int off, int Sv,
                                                                int NT, const float* __restrict__ dvid, int E,
                                                                bf16* __restrict__ dout, int ldo, float* __restrict__ dtype,
                                                                float* __restrict__ part0, float* __restrict__ part1) {
    const int So = Sv + NT;
    const int b = blockIdx.x;
    const int r0 = blockIdx.y * 32;
    int r1 = r0 + 32;
    r1 = r1 < S ? r1 : S;
    for (int c = threadIdx.x; c < E; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int r = r0; r < r1; ++r) {
            float v = 0.f;
            if (dxs && r >= off) {
                v = dxs[(size_t)(b * So + r - off) * ldx + c];
                s0 += v;
            }
            if (r == 0 && dvid) v += dvid[(size_t)b * E + c];
            dout[(size_t)(b * S + r) * ldo + c] = (bf16)v;
        }
        if (dxs) {
            if (blockIdx.y == 0)
                for (int i = 0; i < NT; ++i) s1 += dxs[(size_t)(b * So + Sv + i) * ldx + c];
            if (part0) {  // per-block partials, added in block order by colsum_partials_kernel
                part0[((size_t)b * gridDim.y + blockIdx.y) * E + c] = s0;
                if (blockIdx.y == 0) part1[(size_t)b * E + c] = s1;
            } else {
                atomicAdd(dtype + c, s0);
                if (blockIdx.y == 0) atomicAdd(dtype + E + c, s1);
            }
        }
    }
}
// workspace (optional, >= B * (ceil(S / 32) + 1) * E floats): the type-embedding gradient (a plain sum over token rows, one of the
// lr 1e-4 parameters: its noise is what Adam amplifies) as ordered per-block partials instead of fp32 atomics
extern "C" int tvts_sort_assemble_bwd(const float* dxs, int ldx, int B, int S, int off, int Sv, int NT, const float* dvid,
                                      int E, void* dout, int ldo, float* dtype, float* workspace, long workspace_elems,
                                      hipStream_t stream) {
    const int chunks = ceil_div(S, 32);
    const bool parts = dxs && dtype && workspace && workspace_elems >= (long)B * (chunks + 1) * E;
    float* part0 = parts ? workspace : nullptr;
    float* part1 = parts ? workspace + (size_t)B * chunks * E : nullptr;
    hipLaunchKernelGGL(sort_assemble_bwd_kernel, dim3(B, chunks), dim3(256), 0, stream, dxs, ldx, S, off, Sv, NT, dvid,
                       E, (bf16*)dout, ldo, dtype, part0, part1);
    if (parts) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(E, 64)), dim3(1024), 0, stream, part0, B * chunks, E, dtype,
                           (float*)nullptr);
        hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(E, 64)), dim3(1024), 0, stream, part1, B, E, dtype + E,
                           (float*)nullptr);
    }
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- row gather/scatter (fp32)
// dst[r] = src[rows[r]]   (gather)   or   dst[rows[r]] += src[r]   (scatter_add)
__global__ void rows_gather_kernel(const float* __restrict__ src, int lds_, const int* __restrict__ rows, int R, int W,
                                   float* __restrict__ dst, int ldd, int scatter) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        if (scatter) dst[(size_t)rows[r] * ldd + c] += src[(size_t)r * lds_ + c];
        else dst[(size_t)r * ldd + c] = src[(size_t)rows[r] * lds_ + c];
    }
}
extern "C" int tvts_rows_gather(const float* src, int ld_src, const int* rows, int R, int W, float* dst, int ld_dst,
                                int scatter_add, hipStream_t stream) {
    hipLaunchKernelGGL(rows_gather_kernel, dim3(R), dim3(256), 0, stream, src, ld_src, rows, R, W, dst, ld_dst, scatter_add);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- rows between a token-row matrix and a packed [R, W] one
// (the "used rows" blocks: the last block of the sort head / the text tower runs on the R rows the model reads, engine.py::_used_rows_*)
//   mode 0 gather        packed[r] = full[rows[r]]: source full_f32 if given else full_bf16; packed_f32 and / or packed_bf16 written
//   mode 1 scatter       full[rows[r]] = packed[r], per element type given on BOTH sides
//   mode 2 scatter-add   full_f32[rows[r]] += packed_f32[r]; full_bf16[rows[r]] = bf16(that sum) when given (the bf16 copy of the row)
// rows are distinct (token rows of distinct captions / clips): no two blocks touch one row.
__global__ __launch_bounds__(256) void rows_move_kernel(int mode, const int* __restrict__ rows, int W, float* full_f32, int ldf,
                                                        bf16* full_bf16, int ldfb, float* packed_f32, int ldp, bf16* packed_bf16, int ldpb) {
    const int r = blockIdx.x;
    const size_t fr = (size_t)rows[r];
    for (int c = threadIdx.x; c < W; c += 256) {
        if (mode == 0) {
            const float v = full_f32 ? full_f32[fr * ldf + c] : (float)full_bf16[fr * ldfb + c];
            if (packed_f32) packed_f32[(size_t)r * ldp + c] = v;
            if (packed_bf16) packed_bf16[(size_t)r * ldpb + c] = full_f32 ? (bf16)v : full_bf16[fr * ldfb + c];
        } else if (mode == 1) {
            if (full_f32 && packed_f32) full_f32[fr * ldf + c] = packed_f32[(size_t)r * ldp + c];
            if (full_bf16 && packed_bf16) full_bf16[fr * ldfb + c] = packed_bf16[(size_t)r * ldpb + c];
        } else {
            const float v = full_f32[fr * ldf + c] + packed_f32[(size_t)r * ldp + c];
            full_f32[fr * ldf + c] = v;
            if (full_bf16) full_bf16[fr * ldfb + c] = (bf16)v;
        }
    }
}
extern "C" int tvts_rows_move(int mode, const int* rows, int R, int W, float* full_f32, int ld_full_f32, void* full_bf16, int ld_full_bf16,
                              float* packed_f32, int ld_packed_f32, void* packed_bf16, int ld_packed_bf16, hipStream_t stream) {
    if (!rows || R < 0 || W <= 0 || mode < 0 || mode > 2 || (!full_f32 && !full_bf16) || (!packed_f32 && !packed_bf16)) return TVTS_EINVAL;
    if (mode == 2 && (!full_f32 || !packed_f32)) return TVTS_EINVAL;
    if (mode == 1 && !((full_f32 && packed_f32) || (full_bf16 && packed_bf16))) return TVTS_EINVAL;
    if (R == 0) return TVTS_OK;
    hipLaunchKernelGGL(rows_move_kernel, dim3(R), dim3(256), 0, stream, mode, rows, W, full_f32, ld_full_f32, (bf16*)full_bf16, ld_full_bf16,
                       packed_f32, ld_packed_f32, (bf16*)packed_bf16, ld_packed_bf16);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// a column range of a bf16 row-major matrix set to zero (dQ of the rows that are no queries in the used-rows attention backward)
__global__ __launch_bounds__(256) void zero_cols_bf16_kernel(bf16* x, int ld, long rows, int cols8) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
    const long n = rows * cols8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols8;
        const int c = (int)(i - r * cols8);
        *(u32x4_*)(x + (size_t)r * ld + c * 8) = (u32x4_){0u, 0u, 0u, 0u};
    }
}
extern "C" int tvts_zero_cols_bf16(void* x, int ld, long rows, int cols, hipStream_t stream) {
    if (!x || rows < 0 || cols <= 0 || (cols % 8) || (ld % 8) || ((size_t)x % 16)) return TVTS_EINVAL;
    if (rows == 0) return TVTS_OK;
    const long n = rows * (cols / 8);
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_cols_bf16_kernel, dim3((int)blocks), dim3(256), 0, stream, (bf16*)x, ld, rows, cols / 8);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- hidden-state dropout (v1 text tower)
// out = x * m / (1 - p) (+ residual); m: element (r, c) is kept iff the upper 32 bits of splitmix64(seed + (r * cols + c) * phi)
// are >= p * 2^32 -- the generator of the attention-probability dropout (attention.hip::drop_keep)
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ x, int ldx, int rows, int cols, unsigned thr,
                                                           float inv, const unsigned long long* __restrict__ seed_dev,
                                                           unsigned long long site, const float* __restrict__ residual, int ldr,
                                                           float* __restrict__ out, int ldo, bf16* __restrict__ outb, int ldob) {
    const unsigned long long seed = seed_dev[0] + site;
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        unsigned long long z = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        float v = ((unsigned)(z >> 32) >= thr) ? x[(size_t)r * ldx + c] * inv : 0.f;
        if (residual) v += residual[(size_t)r * ldr + c];
        if (out) out[(size_t)r * ldo + c] = v;
        if (outb) outb[(size_t)r * ldob + c] = (bf16)v;
    }
}
extern "C" int tvts_dropout_rows(const float* x, int ldx, int rows, int cols, float p, const long* seed_dev, long site,
                                 const float* residual, int ldr, float* out, int ldo, void* out_bf16, int ldob, hipStream_t stream) {
    if (!x || rows <= 0 || cols <= 0 || p < 0.f || p >= 1.f || !seed_dev || (!out && !out_bf16)) return TVTS_EINVAL;
    const long n = (long)rows * cols;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, rows, cols,
                       (unsigned)((double)p * 4294967296.0), 1.0f / (1.0f - p), (const unsigned long long*)seed_dev,
                       (unsigned long long)site, residual, ldr, out, ldo, (bf16*)out_bf16, ldob);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- ReLU (v1 txt_proj)
// y = max(x, 0) / dx = dy * (x > 0): the nn.ReLU in front of the v1 text projection (v1/model/model_dist_TVTS.py:65-68)
__global__ void relu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    out[i] = dy ? (v > 0.f ? dy[i] : 0.f) : (v > 0.f ? v : 0.f);
}
extern "C" int tvts_relu(const float* x, const float* dy, float* out, long n, hipStream_t stream) {
    if (n <= 0) return TVTS_EINVAL;
    hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, dy, out, n);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}

// ---------------------------------------------------------------------------------------------- tube mask on the device
// The reference draws the tube mask per sample in the dataset worker: a shuffled arange(ppf) cut to its first n_keep
// entries, shared by every frame of the clip (v2/data_loader/YTTemporal_dataset.py:207-213).  Here one block per sample
// gives every patch index a counter-based 54-bit random key (splitmix64 of seed, global sample number, patch index), sorts
// the (key, index) words in LDS and keeps the indices of the n_keep smallest: a uniformly random, unsorted permutation
// prefix, reproducible for (seed, sample number) whatever the batch split or the rank layout.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void tube_mask_kernel(unsigned long long seed, unsigned long long first_sample, int ppf,
                                                        int n_keep, int p2, int* __restrict__ keep) {
    __shared__ unsigned long long w[1024];
    const unsigned long long base = splitmix64(seed + splitmix64(first_sample + blockIdx.x));
    for (int i = threadIdx.x; i < p2; i += 256)
        w[i] = i < ppf ? ((splitmix64(base + (unsigned long long)i) & ~0x3FFull) | (unsigned long long)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < p2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = w[i], b = w[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { w[i] = b; w[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n_keep; i += 256) keep[(size_t)blockIdx.x * n_keep + i] = (int)(w[i] & 0x3FFull);
}

extern "C" int tvts_tube_mask(long seed_, long first_sample_, int B, int ppf, int n_keep, int* keep, hipStream_t stream) {
    const unsigned long long seed = (unsigned long long)seed_, first_sample = (unsigned long long)first_sample_;
    if (B < 0 || ppf <= 0 || ppf > 1024 || n_keep < 0 || n_keep > ppf || (B > 0 && n_keep > 0 && keep == nullptr)) return TVTS_EINVAL;
    if (B == 0 || n_keep == 0) return TVTS_OK;
    int p2 = 2;
    while (p2 < ppf) p2 <<= 1;
    hipLaunchKernelGGL(tube_mask_kernel, dim3(B), dim3(256), 0, stream, seed, first_sample, ppf, n_keep, p2, keep);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
