// Shared device helpers for the TVTSv2 gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

#define TVTS_OK 0
#define TVTS_EINVAL (-22)

#define TVTS_LAUNCH_CHECK()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// ACT_ADD_BF16 only exists in the GATE slot of the GEMM epilogues: the bf16 side input is ADDED (the bf16 residual stream of the
// space-time blocks) instead of multiplying the result by an activation derivative
enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2, ACT_ADD_BF16 = 3 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// QuickGELU x*sigmoid(1.702x) and its derivative on the fast transcendental path (v_exp_f32 + v_rcp_f32, ~1 ulp):
// the epilogue evaluates 64 K of these per output tile while the matrix pipe waits.
__device__ __forceinline__ float fast_sigmoid(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
// erf-GELU on the same fast path (round 3; libm's erff is a branchy ~50-instruction routine that spilled the fp8 GELU / gate
// epilogues of the H/14 GEMMs and cost them half their rate): Phi(x) = 0.5 (1 + erf(x / sqrt 2)) by Abramowitz & Stegun 7.1.26,
//   erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) e^{-z^2},  t = 1 / (1 + p z),  z >= 0,  |error| <= 1.5e-7,
// i.e. <= 7.5e-8 absolute in Phi -- five orders below the bf16 rounding of the outputs these epilogues write.  The tail
// q = 1 - Phi(|x|) is formed without cancellation; e^{-z^2} = e^{-x^2 / 2} is also the Gaussian of the derivative.
// (fp contraction is pinned by hand in the erf forms: left to hipcc, the derivative's polynomial was contracted differently in
// different kernels -- the ring and the double-buffered NT kernel disagreed by one bf16 ulp in a few elements per million)
__device__ __forceinline__ void gelu_erf_parts(float x, float& cdf, float& gauss) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    gauss = __builtin_amdgcn_exp2f((-0.72134752044448170f * x) * x);
    float poly = 1.061405429f;
    poly = __builtin_fmaf(poly, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float q = ((0.5f * poly) * t) * gauss;  // 1 - Phi(|x|)
    cdf = x >= 0.f ? 1.0f - q : q;
}
__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == ACT_QUICK_GELU) return x * fast_sigmoid(1.702f * x);
    if (act == ACT_GELU_ERF) {
#pragma clang fp contract(off)
        float cdf, gs;
        gelu_erf_parts(x, cdf, gs);
        return x * cdf;
    }
    return x;
}
// d act(x) / dx
__device__ __forceinline__ float act_bwd(float x, int act) {
    if (act == ACT_QUICK_GELU) {
        const float s = fast_sigmoid(1.702f * x);
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    if (act == ACT_GELU_ERF) {
#pragma clang fp contract(off)
        float cdf, gs;
        gelu_erf_parts(x, cdf, gs);
        return __builtin_fmaf(x * 0.3989422804014327f, gs, cdf);
    }
    return 1.0f;
}

// the GATE slot of a GEMM epilogue: v * act'(h) (input gradient through an activation) or v + h (bf16 residual).
// deriv (TVTS_GEMM_SIDE_DERIV, round 5): h already IS act'(pre-activation) -- the forward's epilogue, which evaluates the sigmoid /
// erf parts for the activation anyway, stored the derivative in place of the pre-activation -- so the gate is one multiply instead of
// a second evaluation of the transcendentals (64 K of them per output tile while the matrix pipe waits)
__device__ __forceinline__ float gate_apply(float v, float h, int gate, int deriv = 0) {
    return gate == ACT_ADD_BF16 ? v + h : deriv ? v * h : v * act_bwd(h, gate);
}
// act(x) and, in `side`, what the forward epilogue's side output stores for the pre-activation x: x itself, or (deriv) act'(x) --
// from ONE evaluation of the sigmoid / erf parts (left to the compiler's CSE the erf form was evaluated twice: +64 us on the H/14 fc1)
__device__ __forceinline__ float act_fwd_side(float x, int act, int deriv, float& side) {
    if (act == ACT_QUICK_GELU) {
        const float s = fast_sigmoid(1.702f * x);
        side = deriv ? s * (1.0f + 1.702f * x * (1.0f - s)) : x;
        return x * s;
    }
    if (act == ACT_GELU_ERF) {
#pragma clang fp contract(off)
        float cdf, gs;
        gelu_erf_parts(x, cdf, gs);
        side = deriv ? __builtin_fmaf(x * 0.3989422804014327f, gs, cdf) : x;
        return x * cdf;
    }
    side = x;
    return x;
}

// The same two results for a PAIR of elements on the packed fp32 pipe (v_pk_mul / v_pk_add / v_pk_fma_f32 take two fp32 values per
// lane in one full-rate issue; only the transcendentals and the sign select stay per element): the GEMM epilogues are bound by VALU
// issue while the matrix pipe waits -- round 5 counted ~14 instructions per output element in the QuickGELU + derivative form.
// Every operation and its order are those of the scalar form above, so the two produce the same bits.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ pk_fma(f32x2_ a, f32x2_ b, f32x2_ c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_ act_fwd_side_pk(f32x2_ x, int act, int deriv, f32x2_& side) {
    if (act == ACT_QUICK_GELU) {
        const f32x2_ t = 1.702f * x;
        const f32x2_ z = -1.4426950408889634f * t;
        const f32x2_ d = 1.0f + (f32x2_){__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
        const f32x2_ s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        side = deriv ? s * (1.0f + t * (1.0f - s)) : x;
        return x * s;
    }
    if (act == ACT_GELU_ERF) {
#pragma clang fp contract(off)
        const f32x2_ z = __builtin_elementwise_abs(x) * 0.70710678118654752f;
        const f32x2_ r = pk_fma((f32x2_){0.3275911f, 0.3275911f}, z, (f32x2_){1.0f, 1.0f});
        const f32x2_ t = {__builtin_amdgcn_rcpf(r[0]), __builtin_amdgcn_rcpf(r[1])};
        const f32x2_ a = (-0.72134752044448170f * x) * x;
        const f32x2_ gauss = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        f32x2_ poly = {1.061405429f, 1.061405429f};
        poly = pk_fma(poly, t, (f32x2_){-1.453152027f, -1.453152027f});
        poly = pk_fma(poly, t, (f32x2_){1.421413741f, 1.421413741f});
        poly = pk_fma(poly, t, (f32x2_){-0.284496736f, -0.284496736f});
        poly = pk_fma(poly, t, (f32x2_){0.254829592f, 0.254829592f});
        const f32x2_ q = ((0.5f * poly) * t) * gauss;
        const f32x2_ p = 1.0f - q;
        const f32x2_ cdf = {x[0] >= 0.f ? p[0] : q[0], x[1] >= 0.f ? p[1] : q[1]};
        side = deriv ? pk_fma(x * 0.3989422804014327f, gauss, cdf) : x;
        return x * cdf;
    }
    side = x;
    return x;
}

// XCD-aware bijective remap of a 1-D block id: blocks that land on one XCD (bid % 8) get a
// contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Per-tensor (delayed) scaling of the e4m3 copies (BASELINE config 5, fp8 weight gradients): the quantising kernels take
//   tscale   the tensor's scale of THIS step (device scalar, = last step's amax / 448): every row is quantised under it instead of
//            under its own amax / 448 -- one copy then serves the forward / input-gradient GEMMs (a constant "per-row" scale)
//            AND the weight-gradient GEMM, whose contraction over the tokens cannot carry per-token scales;
//   amax_acc the running max |x| of this step (device scalar, >= 0): becomes the next step's scale (tvts_fp8_update_scales).
// A wave folds the amax of all the rows it walked into one atomic, and only when it would raise the value.
__device__ __forceinline__ void amax_publish(float* amax_acc, float wave_amax, int lane) {
    if (amax_acc && lane == 0 && wave_amax > *(volatile float*)amax_acc) atomicMax((int*)amax_acc, __float_as_int(wave_amax));
}
__device__ __forceinline__ float q8_scale(const float* tscale, float row_amax) {
    if (tscale) {
        const float s = tscale[0];
        return s > 0.f ? s : 1.0f;
    }
    return row_amax > 0.f ? row_amax / 448.0f : 1.0f;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
