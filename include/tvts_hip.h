/* C ABI of libtvts_hip.so -- the MI355X (gfx950) kernels under the TVTSv2 pretrain step.
 *
 * The reference (TencentARC/TVTS v2) has no FFI: its hot path is stock PyTorch ops.
 * Each entry point below replaces the ATen op(s) at the cited reference site; INTEGRATION.md shows the
 * ctypes binding a maintainer adds.  Conventions: raw device pointers, explicit sizes / leading
 * dimensions in ELEMENTS, asynchronous on `stream`, no allocation inside, re-entrant, return 0 or a
 * hipError_t (> 0) / -22 for an invalid argument.  bf16 = 16-bit brain float, row-major everywhere.
 *
 * The library keeps NO mutable process state: entry points are called from the main thread, from PyTorch's autograd worker
 * thread and from communication hooks.  Where a kernel family has alternative dispatch paths (tile sizes, split
 * counts, fused / split passes) the entry point takes a per-call `opts` word -- 0 is the automatic choice the training step
 * uses, the TVTS_*  bits below select the alternatives for parity tests and benches (tests/test_abi.py::test_two_threads).
 */
#ifndef TVTS_HIP_H
#define TVTS_HIP_H
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TVTS_ACT_NONE = 0, TVTS_ACT_QUICK_GELU = 1, TVTS_ACT_GELU_ERF = 2,
       TVTS_GATE_ADD_BF16 = 3 /* gate_act only: gate_h is a bf16 matrix ADDED to the result -- the bf16 residual stream */ };
enum { TVTS_ATTN_FULL = 0, TVTS_ATTN_SPACE = 1, TVTS_ATTN_TIME = 2, TVTS_ATTN_CLS = 3 };
/* `opts` of the GEMM entry points (OR them; 0 = automatic) */
enum {
    TVTS_GEMM_TILE_128 = 1,     /* force the persistent 128x128 kernel */
    TVTS_GEMM_TILE_256 = 2,     /* force the pipelined 256x256 kernel (-22 if the shape cannot take it) */
    TVTS_GEMM_FP8_K32 = 4,      /* tvts_gemm_nt_fp8*: the 16x16x32 fp8 MFMA main loop (bf16 issue rate) instead of the K = 128 scaled
                                   MFMA; both accumulate the same products in fp32 */
    TVTS_TN_NO_EARLY_DMA = 4,   /* tvts_gemm_tn_bf16: LDS-DMA through the builtin path (the one operands past 4 GiB take) */
    TVTS_TN_AFAST_0 = 8,        /* tvts_gemm_tn_bf16: tile walk of an m-range, b-dimension fastest ... */
    TVTS_TN_AFAST_1 = 16,       /* ... or a-dimension fastest (default: the shorter one) */
    TVTS_GEMM_STREAMK = 32,     /* tvts_gemm_nt_bf16 / tvts_gemm_tn_bf16: force the stream-K walk (work split by K stage, ordered sum of the
                                   partial tiles in the block that arrives last); -22 without a workspace or on a shape it cannot take */
    TVTS_GEMM_NO_STREAMK = 64,  /* ... never take it */
    TVTS_GEMM_RING = 128,       /* tvts_gemm_nt_bf16: force the ring form of the 128-column kernel (one block per CU, three 64-deep stages
                                   in flight: the small-batch kernel; same bits as TVTS_GEMM_TILE_128); -22 if an operand is too large
                                   for its 32-bit offsets.  Automatic where its cost model wins (csrc/gemm.hip, nt_use_ring) */
    TVTS_GEMM_NO_RING = 16384,  /* ... never take it */
    TVTS_GEMM_F32_PATCH = 4194304,   /* tvts_gemm_nt_bf16, 256 x 256 kernel, plain bf16 results (round 6): the fp32 LDS patch of the epilogue instead of the
                                   bf16-first patch (the tile rounded in the accumulator layout, two 16-row slabs per 4 KiB patch) -- the same
                                   bits either way (tests/test_bench_path_gpu.py), the old kernel for A/Bs */
    TVTS_GEMM_CLOCK_SAMPLE = 2097152, /* tvts_gemm_nt_bf16, plain bf16 result on the 256 x 256 kernel (round 6, bench.py's instrumented step): block 0
                                   writes {s_memtime, s_memrealtime} at its first instruction and behind its last tile into the LAST 32 bytes
                                   of `workspace` (4 x u64: cycles0, ticks0, cycles1, ticks1) -- shader cycles over constant-rate ticks =
                                   the clock the launch ran at.  A separate instantiation: the kernels of every other call are unchanged;
                                   ignored where the call takes another kernel or the stream-K walk */
    TVTS_GEMM_SIDE_DERIV = 1048576 /* tvts_gemm_nt_bf16 / _fp8 / _fp8_gate (round 5): the activation forms store act'(x) in `preact` instead of the
                                   pre-activation x, the gate forms take `gate_h` as that derivative and multiply by it as is -- the forward
                                   epilogue evaluates the sigmoid / erf parts anyway, the input-gradient epilogue then needs no
                                   transcendental at all.  Both GEMMs of an MLP must agree on it (the stored tensor changes meaning). */
};
/* measurement hook: tile rows of the forced ring kernel, 128 / 192 / 256 (0 = its own choice between 128 and 192) */
#define TVTS_GEMM_RING_ROWS(r) ((r) == 128 ? (1 << 16) : (r) == 192 ? (2 << 16) : (r) == 256 ? (3 << 16) : 0)
/* persistent grid of the 256x256 NT kernels (bf16 and fp8): at most n blocks, one per CU (8 .. 256, multiple of 8; 0 = the whole
 * chip) -- leaves CUs to kernels of other streams (the RCCL kernels of the side stream when world > 1), and a measurement
 * hook (tools/gemm_cus.py) */
#define TVTS_GEMM_CUS(n) ((((n) / 8) & 63) << 8)
/* tvts_gemm_tn_bf16: number of contraction ranges (0 = automatic: the smallest count that fills >= 93 % of a round) */
#define TVTS_TN_SPLITS(s) ((s) << 8)
/* `opts` of the attention entry points */
enum {
    TVTS_ATTN_NO_TR = 1,        /* scalar LDS reads instead of ds_read_b64_tr_b16 fragments */
    TVTS_ATTN_NO_SHARED = 2,    /* per-wave instead of block-shared K / V (Q / dO) staging */
    TVTS_ATTN_NO_FUSED = 4      /* the split passes instead of the fused single-launch kernels */
    /* bits 4..6 (tvts_attn_bwd only): timing ablations of the fused backward, results wrong by construction */
};

/* ---- GEMM (gemm.hip).  nn.Linear forward / dgrad: v2/model/video_encoder_ViT_B_16.py:26-27,41,74,105-109;
 *      v2/CLIP/clip/model.py:175-181; v2/model/sort_transformer.py:21-23,41-42.  K % 64 == 0, N % 4 == 0.
 *      out = [gate'(gate_h) *] act(A.B^T + bias) [+ residual]; preact (bf16) receives A.B^T + bias when act != 0.
 *      gate_act = TVTS_GATE_ADD_BF16: out = A.B^T + bias + gate_h (bf16 residual: x + proj(.) / x + mlp(.) of
 *      video_encoder_ViT_B_16.py:117-123 with the residual stream kept in bf16; `residual` is the fp32 form of the same sum)
 *      workspace (optional, workspace_bytes >= tvts_gemm_nt_workspace_bytes()): scratch of the stream-K walk the entry point
 *      takes when an output of few tiles would leave the persistent grid a fraction of a round (the reference's own per-GPU
 *      batches of 12 / 24 pairs): fp32 partial tiles + one arrival counter per output tile.  Its first 64 KiB must be ZERO
 *      on the first call and are zero again after every call; one workspace serves one stream (calls in flight at the same
 *      time need workspaces of their own).  Without it the tile-granular walk is the only one. */
int tvts_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias,
                      const float* residual, int ldr, int act, void* preact, int ldp, const void* gate_h, int ldh,
                      int gate_act, void* out, int ldc, int out_f32, void* workspace, long workspace_bytes, int opts,
                      hipStream_t stream);
long tvts_gemm_nt_workspace_bytes(void);
/* the kernel tvts_gemm_nt_bf16 picks for an [M, N] result under `opts` -- 256 / 128 = the output tile of the pipelined / the
 * double-buffered kernel, 1128 / 1192 = the ring kernel (TVTS_GEMM_RING) with 128 / 192 tile rows: lets a parity test assert that
 * the kernel it means to exercise is the one that ran */
int tvts_gemm_nt_select(int M, int N, int opts);
/* weight gradient: out[Na,Nb] (+)= P[M,Na]^T . Q[M,Nb], bf16 in, fp32 out (autograd of the Linear sites above) */
/* colsum (optional): colsum[a] += sum_m P[m,a] -- the bias gradient, fused into the same pass.
 * workspace (optional, workspace_elems floats): scratch for the split-M partials; with it the kernel stores
 * partials that are combined in range order (deterministic), without it the partials meet through fp32 atomics.
 * counters (optional, n_counters ints, ZERO on the first call and zero again after every call; one array per stream): with them
 * and TVTS_GEMM_STREAMK in `opts` the block that arrives last at an output tile adds that tile's partials itself (same order, same
 * bits as the separate reduce pass, one launch less -- and measured slower: the partials then travel at agent scope and one block
 * sums them); otherwise a reduce pass follows the kernel. */
int tvts_gemm_tn_bf16(const void* P, int ldp, const void* Q, int ldq, int M, int Na, int Nb, float* out, int ldo,
                      int accumulate, float* colsum, float* workspace, long workspace_elems, int* counters, int n_counters,
                      int opts, hipStream_t stream);
/* Several weight gradients in ONE launch + ONE ordered reduce launch (the six of a ViT block at the reference's per-GPU batches, where
 * each alone fills a fraction of a round): problems = HOST array of n records
 *   struct { const void* P; int ldp; const void* Q; int ldq; int M, Na, Nb; float* out; int ldo; int accumulate; float* colsum; }
 * (the arguments of tvts_gemm_tn_bf16); table_dev = device memory for the plan (tvts_gemm_tn_grouped_table_bytes(n)), table_host = HOST
 * staging memory of the same size, owned by the caller (page-locked, so that the copy is truly asynchronous) and left untouched until
 * the stream has passed this call.  upload != 0 builds the plan in table_host and copies it with ONE hipMemcpyAsync on `stream`, in
 * front of the kernels that read it -- no host wait, ordered on whatever stream the caller launches on; upload == 0 re-uses the plan a
 * previous call with the SAME problems and workspace left in table_dev (table_host may be NULL).  Every problem's result has the bits
 * of tvts_gemm_tn_bf16 called with the same range count. */
int tvts_gemm_tn_bf16_grouped(const void* problems, int n, void* table_dev, void* table_host, long table_bytes, int upload,
                              float* workspace, long workspace_elems, int opts, hipStream_t stream);
long tvts_gemm_tn_grouped_table_bytes(int n);
/* the tile (128: 128x128 kernel, two blocks per CU; 256: pipelined 256x256 kernel) tvts_gemm_tn_bf16 picks for M rows into an
 * [Na, Nb] output under `opts` */
int tvts_gemm_tn_select(int M, int Na, int Nb, int opts);
/* fp8 (OCP e4m3) operands with scales in device memory, fp32 accumulate: the GEMM of BASELINE config 4's weight / activation
 * path (nn.Linear sites of video_encoder_ViT_H_14.py); K % 128 == 0, lda / ldb % 16 == 0 (bytes).  scale_b: one scale for the
 * weight; scale_a: one scale for the tensor, or (scale_a_rows != 0) M per-row scales as written by tvts_quant_fp8_rows */
int tvts_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* scale_a,
                     int scale_a_rows, const float* scale_b, const float* bias, const float* residual, int ldr, int act, void* preact, int ldp,
                     void* out, int ldc, int out_f32, void* q8out, int ldq8, const float* q8_scale, float* q8_amax, int opts,
                     hipStream_t stream);
/* q8out (optional; activation forms with a bf16 result, here and in tvts_gemm_nt_fp8_gate): the epilogue also writes
 * q8out[m, n] = e4m3(out[m, n] / q8_scale[0]) and folds max |out| into q8_amax -- the per-tensor copy the next GEMMs of
 * BASELINE config 5 read (the following layer's forward and weight gradient), without a quantiser pass over the result.  With q8out,
 * `out` may be NULL: the e4m3 copy is then the only result (round 5: in the per-tensor regime nothing reads the bf16 tensor) */
/* input gradient of such a layer (autograd of nn.Linear + the GELU of video_encoder_ViT_H_14.py's Mlp): out[M,N] (bf16) =
 * gate_act'(gate_h[M,N]) * (scale_a[m] * scale_b * (A[M,K] B[N,K]^T)), A = e4m3 copy of the output gradient (per-token scales),
 * B = e4m3 copy of the transposed weight; the un-gated input gradients take tvts_gemm_nt_fp8 itself */
int tvts_gemm_nt_fp8_gate(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const float* scale_a,
                          int scale_a_rows, const float* scale_b, const float* bias, const void* gate_h, int ldh, int gate_act,
                          void* out, int ldc, void* q8out, int ldq8, const float* q8_scale, float* q8_amax, int opts,
                          hipStream_t stream);
/* weight gradient of such a layer on e4m3 operands: out[Na,Nb] (+)= scale_p * scale_q * sum_m P8[m,Na] * Q8[m,Nb].  The contraction runs
 * over the tokens, so the operands carry ONE scale per tensor (device scalars; tvts_quant_fp8 / tvts_quant_fp8_rows2 write such
 * copies) -- not the per-token scales of the forward / input-gradient operands.  Na, Nb, ldp, ldq (bytes) multiples of 16.
 * workspace: split-M partials, reduced in range order (deterministic).  colsum (optional): colsum[a] += scale_p * sum_m P8[m,a] --
 * the bias gradient from the SAME e4m3 bytes the weight gradient is made of (a ones operand on the matrix pipe). */
int tvts_gemm_tn_fp8(const void* P8, int ldp, const void* Q8, int ldq, int M, int Na, int Nb, const float* scale_p,
                     const float* scale_q, float* out, int ldo, int accumulate, float* colsum, float* workspace,
                     long workspace_elems, int opts, hipStream_t stream);
/* (main loop of tvts_gemm_nt_fp8*: v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales, the fp8 issue rate of gfx950;
 * TVTS_GEMM_FP8_K32 selects the 16x16x32 fp8 form) */
/* per-tensor fp8 quantisation: amax[0] = max |x| ; q = rne(x * 448 / amax) as e4m3, scale_out[0] = amax / 448 */
int tvts_amax(const void* x, int is_f32, long ld, int rows, int cols, float* amax, hipStream_t stream);
int tvts_quant_fp8(const void* x, int is_f32, long ld, int rows, int cols, const float* amax, void* out, long ldo,
                   float* scale_out, hipStream_t stream);
/* the same for MANY weights in three launches (every fp8 weight of the model, once per optimizer step).  table: n device records
 *   struct { const float* w; uint8_t* q; const bf16* wt; uint8_t* qt; float* amax; float* scale; float* scale_t; int rows, cols; }
 * w = fp32 master [rows, cols] contiguous, q its e4m3 copy; wt (optional) the bf16 transposed copy [cols, rows] and qt its e4m3
 * copy under the SAME scale (amax of the master); rows * cols % 4 == 0 */
int tvts_quant_fp8_multi(const void* table, int n, hipStream_t stream);
/* per-row (per-token) quantisation of a bf16 activation in one pass: row_scale[r] = amax(x[r,:]) / 448 (1 for an all-zero
 * row), out[r,:] = e4m3(x[r,:] / row_scale[r]); cols % 8 == 0 */
int tvts_quant_fp8_rows(const void* x, long ld, int rows, int cols, void* out, long ldo, float* row_scale,
                        const float* tscale, float* amax_acc, hipStream_t stream);
/* Per-tensor (delayed) scaling -- the form the e4m3 WEIGHT GRADIENT needs, whose contraction over the tokens cannot carry per-token
 * scales: with tscale (device scalar) every row is quantised under that one scale instead of its own amax / 448 (row_scale may then be
 * null), so the same bytes serve tvts_gemm_nt_fp8* (scale_a = tscale, scale_a_rows = 0) and tvts_gemm_tn_fp8; amax_acc (device scalar,
 * optional, in either mode) receives max(amax_acc, max |x|) of the call.  tvts_fp8_update_scales turns a step's amax values into the
 * next step's scales: scale[i] = amax[i] / 448 where amax[i] > 0, amax[i] = 0.  The same two arguments exist on the LayerNorm forms
 * below. */
int tvts_fp8_update_scales(float* amax, float* scale, int n, hipStream_t stream);
/* strided fp32 matmul for the tiny products (text_projection model_dist..B_16.py:108, head sort_transformer.py:113,
 * sim_matrix model_dist..B_16.py:126): C[i,j] (+)= alpha * sum_k A[i*sai+k*sak] * B[k*sbk+j*sbj] + bias[j] */
int tvts_gemm_small_f32(const float* A, long sai, long sak, const float* B, long sbk, long sbj, int M, int N, int K,
                        float alpha, const float* bias, float* C, long ldc, int accumulate, hipStream_t stream);
/* a FEW rows through a linear layer with fp32 result and fp32 residual: out[r, n] = residual[r, n] + bias[n] + sum_k A[r * lda + k] W[n * ldw + k]
 * (A, W bf16; N % 16 == 0, K % 32 == 0; bias / residual optional).  The CLS rows of the hybrid residual stream through the blocks'
 * residual-adding projections (`x + attn(...)`, `x + mlp(...)`, video_encoder_ViT_B_16.py:121-124): one row per clip, lda = S * K. */
int tvts_rows_linear_bf16(const void* A, long lda, const void* W, int ldw, int R, int N, int K, const float* bias, const float* residual,
                          int ldr, float* out, int ldo, hipStream_t stream);
/* bias gradient: out[n] += sum_m X[m,n].  workspace (optional): partial sums of row ranges, added in range order (deterministic, and
 * a grid over the rows as well as the columns); without it one block per 64 columns walks every row */
int tvts_colsum_bf16(const void* X, int ld, int M, int N, float* out, float* workspace, long workspace_elems, hipStream_t stream);

/* ---- LayerNorm (norm.hip): video_encoder_ViT_B_16.py:79-85 (eps 1e-5), sort_transformer.py:99 (eps 1e-6) */
int tvts_layernorm_fwd(const void* x, int ldx, int x_bf16, const int* rows, const float* gamma, const float* beta, float eps, int M,
                       int W, void* y, int ldy, int y_f32, float* mean, float* rstd, hipStream_t stream);
/* the same, additionally writing the bf16 output as OCP e4m3 bytes q8[M, W] (ldq bytes per row) with one scale per row
 * (row_scale[M] = amax(row) / 448): the fp8 operand of the GEMM that follows (BASELINE config 4), identical to
 * tvts_quant_fp8_rows run on y.  y may be NULL (bf16 x, W % 8 == 0, W <= 1536, ldx % 8 == ldq % 8 == 0): the e4m3 bytes are
 * then the only output -- under per-tensor scales (tscale) the GEMMs that follow never read the bf16 tensor.  The same holds
 * for tvts_layernorm_fwd_cls with q8. */
int tvts_layernorm_fwd_fp8(const void* x, int ldx, int x_bf16, const int* rows, const float* gamma, const float* beta, float eps,
                           int M, int W, void* y, int ldy, void* q8, int ldq, float* row_scale, const float* tscale, float* amax_acc,
                           float* mean, float* rstd, hipStream_t stream);
/* dx = LayerNorm backward (+ res1 + res2): res1 is the residual-stream gradient, fp32 or (res1_bf16 != 0, bf16 dy only) bf16 --
 * the space-time block's backward carries it in bf16, the precision its GEMM consumers read it in anyway; res2 a bf16 side branch */
int tvts_layernorm_bwd(const void* dy, int lddy, int dy_f32, const void* x, int ldx, int x_bf16, const int* rows, const float* mean,
                       const float* rstd, const float* gamma, const void* res1, int res1_bf16, int ldr, const void* res2_bf16, int ldr2,
                       int M, int W, float* dx, int lddx, void* dx_bf16, int lddxb, float* dgamma, float* dbeta,
                       float* workspace, long workspace_elems, hipStream_t stream);
/* the same with the e4m3 copy of dx_bf16 (one scale per row, the bytes tvts_quant_fp8_rows would write): the output gradient of the
 * e4m3 input-gradient GEMM that consumes dx_bf16.  bf16 dy, every row, dx_bf16 required; x fp32 with res1 / res1 + res2 / no
 * residual, or x bf16 without residuals */
int tvts_layernorm_bwd_fp8(const void* dy, int lddy, const void* x, int ldx, int x_bf16, const float* mean, const float* rstd,
                           const float* gamma, const void* res1, int res1_bf16, int ldr, const void* res2_bf16, int ldr2, int M, int W, float* dx,
                           int lddx, void* dx_bf16, int lddxb, void* q8, int ldq, float* row_scale, const float* tscale, float* amax_acc,
                           float* dgamma, float* dbeta, float* workspace, long workspace_elems, hipStream_t stream);
/* The HYBRID residual stream of the space-time blocks (round 5; replaces the fp32 activations `x + ...` of
 * video_encoder_ViT_B_16.py:113-124 byte for byte except one row per clip): the stream x [M, W] is bf16, but the row of each clip's CLS
 * token (rows r % cls_period == 0) -- the row the video embedding is read from, and the one row whose rounding error reaches every
 * other token through the attention -- is carried in fp32 in a compact side array cls_x [M / cls_period, W].  Forward: those rows are
 * normalised from cls_x (their stream rows are stale) and x_refresh (normally x itself, optional) receives their bf16 rounding, so that
 * later readers of the stream see the exact value rounded once.  q8 .. amax_acc optional, as in tvts_layernorm_fwd_fp8. */
int tvts_layernorm_fwd_cls(const void* x, int ldx, const float* cls_x, int cls_period, void* x_refresh, const float* gamma,
                           const float* beta, float eps, int M, int W, void* y, int ldy, void* q8, int ldq, float* row_scale,
                           const float* tscale, float* amax_acc, float* mean, float* rstd, hipStream_t stream);
/* Backward on the hybrid stream: bf16 dy, bf16 x, every row; res1_bf16 (optional) the bf16 stream gradient, res2_bf16 (optional, with
 * res1) a bf16 side branch; dx_bf16 required.  Rows r % cls_period == 0: input from cls_x (optional), stream gradient from cls_res1
 * (optional, fp32 [M / cls_period, W]) instead of res1's row, result ALSO to cls_dx (optional, fp32) -- the CLS token's gradient chain
 * never passes through a bf16 rounding.  q8 .. amax_acc optional, as in tvts_layernorm_bwd_fp8. */
int tvts_layernorm_bwd_cls(const void* dy, int lddy, const void* x, int ldx, const float* cls_x, const float* cls_res1, float* cls_dx,
                           int cls_period, const float* mean, const float* rstd, const float* gamma, const void* res1_bf16, int ldr,
                           const void* res2_bf16, int ldr2, int M, int W, void* dx_bf16, int lddxb, void* q8, int ldq, float* row_scale,
                           const float* tscale, float* amax_acc, float* dgamma, float* dbeta, float* workspace, long workspace_elems,
                           hipStream_t stream);

/* ---- attention (attention.hip), head dim 64, packed qkv [rows, 3*heads*64]:
 *      divided space-time attention video_encoder_ViT_B_16.py:11-15,38-76; causal text attention
 *      CLIP/clip/model.py:185-187,330-336; sort-head attention sort_transformer.py:45-53 */
int tvts_attn_fwd(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, void* out, int ldo,
                  float* lse2, int opts, hipStream_t stream);
int tvts_attn_delta(const void* dO, int lddo, const void* O, int ldo, int rows, int heads, float* delta,
                    hipStream_t stream);
int tvts_attn_bwd_dq(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
                     int lddo, const float* lse2, const float* delta, void* dqkv, int lddq, int opts, hipStream_t stream);
int tvts_attn_bwd_dkv(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
                      int lddo, const float* lse2, const float* delta, void* dqkv, int lddq, float* cls_acc, int opts,
                      hipStream_t stream);
int tvts_attn_cls_finalize(const float* cls_acc, int B, int heads, int S, void* dqkv, int lddq, hipStream_t stream);
/* whole backward of one attention site (D = rowsum(dO*O), dQ, dK, dV, CLS query + CLS key/value reduction of the divided
 * geometries) = the autograd of VarAttention.forward video_encoder_ViT_B_16.py:38-76; delta [rows, heads] and
 * cls_acc (cls_acc_elems fp32 elements) are scratch.  SPACE groups of <= 112 tokens run as ONE fused launch.  cls_acc holds the
 * CLS token's dK / dV / dQ shares: with >= B * heads * max(T, ceil(n / 28)) * 3 * dh elements every block of the fused kernels
 * stores its share and they are added in a fixed order (run-to-run reproducible); with B * heads * 3 * dh elements (the
 * minimum) the shares are accumulated with fp32 atomics. */
int tvts_attn_bwd(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
              int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float* cls_acc,
              long cls_acc_elems, int opts, hipStream_t stream);
/* forward of one divided-attention site including the CLS row (VarAttention.forward video_encoder_ViT_B_16.py:38-76);
 * cls_ws: fp32 scratch of >= B * heads * max(T, ceil(n / 28)) * (dh + 2) elements (partial softmax states of the CLS query) */
int tvts_attn_fwd_divided(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                      float* lse2, float* cls_ws, long cls_ws_elems, int opts, hipStream_t stream);
/* the two site entry points with the per-tensor e4m3 copy of their result written by the kernels themselves (BASELINE config 5: the
 * attention output feeds the projection's forward and weight-gradient GEMMs, dqkv the qkv projection's input- and weight-gradient
 * GEMMs): q8out[r, c] = e4m3(result[r, c] / q8_scale[0]) with the result's leading dimension (ldq8 == ldo resp. lddq, in bytes),
 * q8_amax = max(q8_amax, max |result|).  Fused geometries only (SPACE n + 1 <= 112, TIME T + 1 <= 32), -22 otherwise. */
int tvts_attn_fwd_divided_q8(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                      float* lse2, float* cls_ws, long cls_ws_elems, void* q8out, int ldq8, const float* q8_scale, float* q8_amax,
                      int opts, hipStream_t stream);
int tvts_attn_bwd_q8(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
              int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float* cls_acc,
              long cls_acc_elems, void* q8out, int ldq8, const float* q8_scale, float* q8_amax, int opts, hipStream_t stream);

/* the same entry points for head dim 80 (ViT-H/14, 1280 / 16 heads); qkv is [rows, 3*heads*80] */
/* FULL attention over sequences padded to S with the padded keys masked: keys at positions >= kv_len[b] (device int32[B]) get
 * probability 0 -- the attention_mask of the v1 text tower (transformers DistilBERT MultiHeadSelfAttention, reached from
 * v1/model/model_dist_TVTS.py:131-141).  bwd_len = delta + dQ + dK/dV; the dK / dV rows of the padded positions are NOT written
 * (their gradient is zero): zero dqkv first. */
int tvts_attn_fwd_len(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo, float* lse2,
                      hipStream_t stream);
int tvts_attn_bwd_len(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                      const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
/* the same pair with DROPOUT on the attention probabilities -- transformers' DistilBERT MultiHeadSelfAttention in training mode
 * (weights = dropout(softmax(scores)), attention_dropout 0.1), which the v1 model runs in every pretraining step
 * (v1/model/model_dist_TVTS.py:33-34 text_model.train(), :131-141).  Counter-based mask: probability (b, h, q, k) is kept iff the
 * upper 32 bits of splitmix64(seed_dev[0] + site + index * 0x9E3779B97F4A7C15) are >= p * 2^32, kept values are scaled by
 * 1 / (1 - p); seed_dev is DEVICE memory (a replayed hipGraph draws new masks when the caller advances it), the backward
 * regenerates the forward's mask from the same (seed, site).  p = 0: the plain pair. */
int tvts_attn_fwd_len_drop(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo, float* lse2,
                    float p, const long* seed_dev, long site, hipStream_t stream);
int tvts_attn_bwd_len_drop(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                    const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float p,
                    const long* seed_dev, long site, hipStream_t stream);
/* FULL attention (no mask) whose only QUERIES are the last nq (<= 16) tokens of every sequence, all S tokens keys: the last block
 * of the transcript-sorting head -- SortTransformer.forward_features reads its output at the transcript positions only
 * (v2/model/sort_transformer.py:131-141: x = self.norm(x[:, x_len:])), so the other rows of that block's attention output, MLP and
 * residual are never used by the loss.  out / lse2 / dO / O / delta are indexed by token row (only the query rows are touched);
 * bwd_tail = delta + dQ (query rows) + dK / dV (every row): the caller zeroes the dQ third of the non-query rows of dqkv. */
int tvts_attn_fwd_tail(const void* qkv, int ld, int B, int heads, int S, int nq, void* out, int ldo, float* lse2,
                       hipStream_t stream);
int tvts_attn_bwd_tail(const void* qkv, int ld, int B, int heads, int S, int nq, const void* dO, int lddo, const void* O, int ldo,
                       const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
int tvts_attn80_fwd_tail(const void* qkv, int ld, int B, int heads, int S, int nq, void* out, int ldo, float* lse2,
                         hipStream_t stream);
int tvts_attn80_bwd_tail(const void* qkv, int ld, int B, int heads, int S, int nq, const void* dO, int lddo, const void* O, int ldo,
                         const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
/* ONE query per sequence, at token qpos[b] (device int32[B]), that sees the keys 0 .. qpos[b]: the last block of the CLIP text tower,
 * whose output the model reads at the EOT token only (v2/CLIP/clip/model.py:343-354).  Conventions of the tail form; the dK / dV
 * rows behind the query are not written (zero gradient): zero dqkv first. */
int tvts_attn_fwd_rowq(const void* qkv, int ld, int B, int heads, int S, const int* qpos, void* out, int ldo, float* lse2,
                       hipStream_t stream);
int tvts_attn_bwd_rowq(const void* qkv, int ld, int B, int heads, int S, const int* qpos, const void* dO, int lddo, const void* O,
                       int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
int tvts_attn80_fwd_rowq(const void* qkv, int ld, int B, int heads, int S, const int* qpos, void* out, int ldo, float* lse2,
                         hipStream_t stream);
int tvts_attn80_bwd_rowq(const void* qkv, int ld, int B, int heads, int S, const int* qpos, const void* dO, int lddo, const void* O,
                         int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
int tvts_attn80_fwd_len(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo, float* lse2,
                        hipStream_t stream);
int tvts_attn80_bwd_len(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                        const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, hipStream_t stream);
/* the same pair with DROPOUT on the attention probabilities -- transformers' DistilBERT MultiHeadSelfAttention in training mode
 * (weights = dropout(softmax(scores)), attention_dropout 0.1), which the v1 model runs in every pretraining step
 * (v1/model/model_dist_TVTS.py:33-34 text_model.train(), :131-141).  Counter-based mask: probability (b, h, q, k) is kept iff the
 * upper 32 bits of splitmix64(seed_dev[0] + site + index * 0x9E3779B97F4A7C15) are >= p * 2^32, kept values are scaled by
 * 1 / (1 - p); seed_dev is DEVICE memory (a replayed hipGraph draws new masks when the caller advances it), the backward
 * regenerates the forward's mask from the same (seed, site).  p = 0: the plain pair. */
int tvts_attn80_fwd_len_drop(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, void* out, int ldo, float* lse2,
                    float p, const long* seed_dev, long site, hipStream_t stream);
int tvts_attn80_bwd_len_drop(const void* qkv, int ld, int B, int heads, int S, const int* kv_len, const void* dO, int lddo,
                    const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float p,
                    const long* seed_dev, long site, hipStream_t stream);
int tvts_attn80_fwd(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, void* out, int ldo,
                  float* lse2, int opts, hipStream_t stream);
int tvts_attn80_delta(const void* dO, int lddo, const void* O, int ldo, int rows, int heads, float* delta,
                    hipStream_t stream);
int tvts_attn80_bwd_dq(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
                     int lddo, const float* lse2, const float* delta, void* dqkv, int lddq, int opts, hipStream_t stream);
int tvts_attn80_bwd_dkv(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
                      int lddo, const float* lse2, const float* delta, void* dqkv, int lddq, float* cls_acc, int opts,
                      hipStream_t stream);
int tvts_attn80_cls_finalize(const float* cls_acc, int B, int heads, int S, void* dqkv, int lddq, hipStream_t stream);
/* whole backward of one attention site (D = rowsum(dO*O), dQ, dK, dV, CLS query + CLS key/value reduction of the divided
 * geometries) = the autograd of VarAttention.forward video_encoder_ViT_B_16.py:38-76; delta [rows, heads] and
 * cls_acc (cls_acc_elems fp32 elements) are scratch.  SPACE groups of <= 112 tokens run as ONE fused launch.  cls_acc holds the
 * CLS token's dK / dV / dQ shares: with >= B * heads * max(T, ceil(n / 28)) * 3 * dh elements every block of the fused kernels
 * stores its share and they are added in a fixed order (run-to-run reproducible); with B * heads * 3 * dh elements (the
 * minimum) the shares are accumulated with fp32 atomics. */
int tvts_attn80_bwd(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
              int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float* cls_acc,
              long cls_acc_elems, int opts, hipStream_t stream);
/* forward of one divided-attention site including the CLS row (VarAttention.forward video_encoder_ViT_B_16.py:38-76);
 * cls_ws: fp32 scratch of >= B * heads * max(T, ceil(n / 28)) * (dh + 2) elements (partial softmax states of the CLS query) */
int tvts_attn80_fwd_divided(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                      float* lse2, float* cls_ws, long cls_ws_elems, int opts, hipStream_t stream);
/* the two site entry points with the per-tensor e4m3 copy of their result written by the kernels themselves (BASELINE config 5: the
 * attention output feeds the projection's forward and weight-gradient GEMMs, dqkv the qkv projection's input- and weight-gradient
 * GEMMs): q8out[r, c] = e4m3(result[r, c] / q8_scale[0]) with the result's leading dimension (ldq8 == ldo resp. lddq, in bytes),
 * q8_amax = max(q8_amax, max |result|).  Fused geometries only (SPACE n + 1 <= 112, TIME T + 1 <= 32), -22 otherwise. */
int tvts_attn80_fwd_divided_q8(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, void* out, int ldo,
                      float* lse2, float* cls_ws, long cls_ws_elems, void* q8out, int ldq8, const float* q8_scale, float* q8_amax,
                      int opts, hipStream_t stream);
int tvts_attn80_bwd_q8(int mode, const void* qkv, int ld, int B, int heads, int S, int T, int n, int causal, const void* dO,
              int lddo, const void* O, int ldo, const float* lse2, float* delta, void* dqkv, int lddq, float* cls_acc,
              long cls_acc_elems, void* q8out, int ldq8, const float* q8_scale, float* q8_amax, int opts, hipStream_t stream);

/* ---- token assembly (embed.hip): video_encoder_ViT_B_16.py:176-216; model_dist..B_16.py:69-76,98-100;
 *      sort_transformer.py:124-128 */
int tvts_patch_gather(const float* video, const int* keep, int B, int T, int n, int img, int patch, void* out, int ldo,
                      hipStream_t stream);
/* uint8 H x W x 3 frames (resized on the host): crop + ClipToTensor + Normalize (video_transforms/video_transform.py:24-75,
 * functional.py:81-97; base_dataset.py:125-127) fused into the tube-mask gather; mean3 / std3 are HOST arrays, crop is a
 * device [B,2] (top, left) or NULL for the centre crop */
int tvts_patch_gather_u8(const unsigned char* frames, int H0, int W0, const int* crop, const int* keep, int B, int T, int n,
                         int img, int patch, const float* mean3, const float* std3, void* out, int ldo, hipStream_t stream);
/* the same with the transform chain's Resize in front (video_transforms/videoaug.py:12,21 -> video_transform.py:171-188 ->
 * functional.py:47-64: PIL nearest-neighbour to (H0, W0)): frames are the decoder's Hs x Ws pictures, ytab[H0] / xtab[W0]
 * (device int32) hold Pillow's source index of every resized row / column (tvts_amd/data_loader/transforms.py) */
int tvts_patch_gather_u8_resized(const unsigned char* frames, int Hs, int Ws, const int* ytab, const int* xtab, int H0, int W0,
                                 const int* crop, const int* keep, int B, int T, int n, int img, int patch, const float* mean3,
                                 const float* std3, void* out, int ldo, hipStream_t stream);
/* tube mask drawn on the device (replaces the per-sample np.random.shuffle(arange(ppf))[:n_keep] of the dataset worker,
 * v2/data_loader/YTTemporal_dataset.py:207-213): keep[b, :] = the n_keep patch indices with the smallest counter-based
 * random keys of sample number first_sample + b -- an unsorted prefix of a uniformly random permutation, shared by all
 * frames of the clip; ppf <= 1024; seed / first_sample are taken as unsigned 64-bit values */
int tvts_tube_mask(long seed, long first_sample, int B, int ppf, int n_keep, int* keep, hipStream_t stream);
/* keep_per_frame 0: keep[B, n], one tube mask for all frames of a clip (v2); 1: keep[B, T, n], one per frame / tubelet (v1) */
int tvts_vit_assemble(const float* patch, int ldp, const float* cls, const float* pos, const float* temporal,
                      const int* keep, int keep_per_frame, int B, int T, int n, int W, float* tok, int ldt, hipStream_t stream);
/* workspace (optional, fp32 scratch): with >= B * ceil(n / 14) * T * W + B * W elements (tube masks) the temporal-embedding and
 * class-embedding sums are ordered per-block partials; with B * n * W more and n_pos > 0 (the rows of dpos behind the class row:
 * the patches per frame) the positional-embedding rows are gathered in clip order too -- every sum run-to-run reproducible.
 * One keep list per frame / tubelet (v1): B * T * W + B * W + B * T * n * W elements for the same.  Without the room (or
 * n_pos = 0) they are a scatter of fp32 atomics. */
int tvts_vit_assemble_bwd(const float* dtok, int ldt, const int* keep, int keep_per_frame, int B, int T, int n, int W,
                          void* dpatch, int ldp, float* dcls, float* dpos, int n_pos, float* dtemporal, float* workspace,
                          long workspace_elems, hipStream_t stream);
/* v1 (TVTS) Conv3d tubelet embedding as im2col over the kept patches of every tube (v1/model/video_encoder.py:78-99,199-206):
 * video fp32 [B, tubes * tubelet, 3, img, img], keep int32 [B, tubes, n] -> bf16 rows [B * tubes * n, 3 * tubelet * patch^2]
 * in the Conv3d weight's (c, t, py, px) column order; patch % 8 == 0 */
int tvts_patch_gather_tube(const float* video, const int* keep, int B, int tubes, int tubelet, int n, int img, int patch,
                           void* out, int ldo, hipStream_t stream);
int tvts_text_embed(const int* ids, int ld_ids, int N, int L, const float* emb, const float* pos, int Wt, float* x, int ldx,
                    hipStream_t stream);
/* order / seg (optional, device int32): the rows 0 .. N * L - 1 sorted by token id (ties in row order) and the starts of the runs
 * of equal ids in that list (N * L + 1 entries, non-decreasing, padded with N * L) -- with them both embedding gradients are
 * ordered sums (run-to-run reproducible), without them (NULL) a scatter of fp32 atomics like nn.Embedding's backward.  The runs
 * may be listed in any order; runs of more than 64 rows among the FIRST 64 are summed by a block per 64 columns instead of one
 * block (list the long ones first: every caption's start / end token makes a run of N rows) */
int tvts_text_embed_bwd(const float* dx, int ldx, const int* ids, int ld_ids, int N, int L, int Wt, float* demb, float* dpos,
                        const int* order, const int* seg, hipStream_t stream);
int tvts_text_mean(const float* t, int NT, int B, int E, float* mean, float* before, hipStream_t stream);
int tvts_text_mean_bwd(const float* dmean, int NT, int B, int E, float* dt, hipStream_t stream);
int tvts_sort_assemble(const float* tok, int ldt, int B, int S, int off, int Sv, const float* text, int NT,
                       const float* type, int E, float* xs, int ldx, hipStream_t stream);
/* workspace (optional, >= B * (ceil(S / 32) + 1) * E fp32 elements): the type-embedding gradient as ordered partials (no atomics) */
int tvts_sort_assemble_bwd(const float* dxs, int ldx, int B, int S, int off, int Sv, int NT, const float* dvid, int E,
                           void* dout, int ldo, float* dtype, float* workspace, long workspace_elems, hipStream_t stream);
/* hidden-state dropout of the v1 text tower in training mode (transformers DistilBERT: Embeddings.dropout after the embedding
 * LayerNorm, FFN.dropout after lin2; p = 0.1; v1/model/model_dist_TVTS.py:33-34): out[r, c] = x[r, c] * m / (1 - p) (+ residual[r, c]),
 * m from the same counter-based generator as the attention dropout with index r * cols + c; out (fp32) and / or out_bf16 may be
 * given.  The backward is the same call on the gradient (same seed / site => same mask). */
int tvts_dropout_rows(const float* x, int ldx, int rows, int cols, float p, const long* seed_dev, long site,
                      const float* residual, int ldr, float* out, int ldo, void* out_bf16, int ldob, hipStream_t stream);
/* out = relu(x) (dy NULL) or out = dy * (x > 0) (its backward): the nn.ReLU of v1's txt_proj (v1/model/model_dist_TVTS.py:65-68) */
int tvts_relu(const float* x, const float* dy, float* out, long n, hipStream_t stream);
int tvts_rows_gather(const float* src, int ld_src, const int* rows, int R, int W, float* dst, int ld_dst, int scatter_add,
                     hipStream_t stream);
/* rows between a token-row matrix ("full") and a packed [R, W] matrix: the last block of the sort head (sort_transformer.py:131-141:
   the model reads the NT transcript rows) and of the text tower (CLIP/clip/model.py:343-354: the EOT row) run on those R rows only.
   mode 0 gather: packed[r] = full[rows[r]] (source full_f32 if given, else full_bf16; packed_f32 and / or packed_bf16 written);
   mode 1 scatter: full[rows[r]] = packed[r] for every element type given on both sides; mode 2: full_f32[rows[r]] += packed_f32[r] and
   full_bf16[rows[r]] = bf16(sum) when given.  rows must be distinct. */
int tvts_rows_move(int mode, const int* rows, int R, int W, float* full_f32, int ld_full_f32, void* full_bf16, int ld_full_bf16,
                   float* packed_f32, int ld_packed_f32, void* packed_bf16, int ld_packed_bf16, hipStream_t stream);
/* x[r, 0:cols] = 0 of a bf16 matrix (cols, ld multiples of 8): dQ of the rows that are no queries in the used-rows attention backward */
int tvts_zero_cols_bf16(void* x, int ld, long rows, int cols, hipStream_t stream);

/* ---- losses (loss.hip): model_dist..B_16.py:119-127, loss.py:13-25, trainer.py:487-492 */
int tvts_l2norm_rows(const float* x, int R, int E, float eps, float* xn, float* inv, hipStream_t stream);
int tvts_l2norm_rows_bwd(const float* dxn, const float* xn, const float* inv, int R, int E, float* dx, hipStream_t stream);
int tvts_infonce(const float* x, int G, float* lse, float* dx, float* loss, hipStream_t stream);
int tvts_cross_entropy(const float* logits, const int* labels, int R, int C, float scale, float* dlogits, float* loss,
                       hipStream_t stream);
/* validation: rank of the ground truth per query in sims[n_text, n_vid] (text x video);
 * mode 0 = model/metric.py:16-126 t2v_metrics (ranks[n_text], optimistic ties), mode 1 = :129-187 v2t_metrics
 * (ranks[n_vid], averaged ties, closest own caption); valid: optional n_text bytes, 0 = caption missing (query_masks) */
int tvts_retrieval_ranks(const float* sims, long ld, int n_text, int n_vid, int mode, const unsigned char* valid,
                         float* ranks, hipStream_t stream);

/* ---- optimizer (optim.hip): transformers.AdamW as built at train_dist_TVTSv2_ViT_B_16.py:118-125 */
/* step_dev (optional): the step counter in device memory (hipGraph replay); hyper_dev (optional, needs step_dev): lr[4] | wd[4]
 * in device memory, read instead of the host lr4 / wd4 so that a captured launch follows the LR schedule */
int tvts_adamw_hf(float* p, const float* g, float* m, float* v, void* shadow_bf16, const unsigned char* chunk_group,
                  int nchunks, const float* lr4, const float* wd4, int step, const int* step_dev, const float* hyper_dev,
                  double beta1, double beta2, double eps, float grad_scale, hipStream_t stream);
int tvts_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream);
int tvts_cast_bf16_f32(const void* src, float* dst, long n, hipStream_t stream);
int tvts_transpose_bf16_batched(const void* src, void* dst, const void* tiles, int ntiles, hipStream_t stream);
/* H/14 (patch 14, K = 588): video_encoder_ViT_H_14.py:336-337 conv1 weight padded to K = 640 for the MFMA GEMM */
int tvts_pad_rows_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, int cols_pad,
                       hipStream_t stream);
int tvts_add_rows_f32(float* dst, int ld_dst, const float* src, int ld_src, int rows, int cols, hipStream_t stream);
int tvts_probe_tr16(const void* in, void* out, hipStream_t stream);

/* ---- timing helper for bench.py: HIP events on the stream the kernels run on */
int tvts_event_create(void** ev);
int tvts_event_record(void* ev, hipStream_t stream);
int tvts_event_elapsed_ms(void* start, void* stop, float* ms);
int tvts_event_destroy(void* ev);
/* clock under load (SURVEY.md 8d "confirm the peak on the box: clocks x CUs x MFMA rate"): the rate of the constant counter
   (s_memrealtime) in kHz, the CU count and the sheet's maximum shader clock of the device; the shader cycles themselves are sampled
   INSIDE a GEMM launch (TVTS_GEMM_CLOCK_SAMPLE).  No reference site: the reference has no roofline accounting. */
/* p[0:nbytes] = 0 on the stream (hipMemsetAsync): optimizer.zero_grad() of the flat gradient buffer (v2/trainer/trainer.py:476) and
   the step's small accumulators */
int tvts_zero_bytes(void* p, long nbytes, hipStream_t stream);
int tvts_device_clock_info(int device, int* wall_clock_khz, int* cu_count, int* max_shader_khz);

#ifdef __cplusplus
}
#endif
#endif
