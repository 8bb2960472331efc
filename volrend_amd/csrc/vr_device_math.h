// vr_device_math.h -- gfx950 device math for the ray-march kernel.
//
// The kernel is compiled with -ffp-contract=off: every fused multiply-add in
// here is spelled __builtin_fmaf explicitly, so the rounding of each operation
// is a property of this source and not of the optimiser.  Policy<0> ("strict")
// rounds after every operator exactly as the reference source reads
// (include/volrend/cuda/rt_core.cuh, internal/lumisphere.hpp, cuda/common.cuh);
// Policy<1> fuses a*b+c where an nvcc -fmad=true build plausibly does
// (DESIGN.md "FP contract" lists the rules and every site).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

template <int FMA>
struct Policy;

template <>
struct Policy<0> {
    static __device__ __forceinline__ float madd(float a, float b, float c) { return a * b + c; }
    static __device__ __forceinline__ float msub(float a, float b, float c) { return a * b - c; }
    static __device__ __forceinline__ float nmadd(float a, float b, float c) { return c - a * b; }
    static __device__ __forceinline__ double dmadd(double a, double b, double c) { return a * b + c; }
    static __device__ __forceinline__ double dmsub(double a, double b, double c) { return a * b - c; }
};
template <>
struct Policy<1> {
    static __device__ __forceinline__ float madd(float a, float b, float c) {
        return __builtin_fmaf(a, b, c);
    }
    static __device__ __forceinline__ float msub(float a, float b, float c) {
        return __builtin_fmaf(a, b, -c);
    }
    static __device__ __forceinline__ float nmadd(float a, float b, float c) {
        return __builtin_fmaf(-a, b, c);
    }
    static __device__ __forceinline__ double dmadd(double a, double b, double c) {
        return __builtin_fma(a, b, c);
    }
    static __device__ __forceinline__ double dmsub(double a, double b, double c) {
        return __builtin_fma(a, b, -c);
    }
};

// PTX min.f32 / max.f32 semantics == v_min_f32 / v_max_f32: NaN loses, -0 < +0.
static __device__ __forceinline__ float vmin(float a, float b) { return __builtin_fminf(a, b); }
static __device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }

static __device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
static __device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// binary16 bits -> binary32, exact (v_cvt_f32_f16)
static __device__ __forceinline__ float h2f(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
}

// b * (float)h and fma(b, (float)h, c) with the binary16 operand taken straight from one
// half of a packed word: v_fma_mix_f32 converts it on the fly (exact) and rounds once, so
//   mul_half(b, w)    == b * h2f(half(w))           (the product plus -0.0 is the product),
//   fma_half(b, w, c) == fmaf(b, h2f(half(w)), c)
// bit for bit -- one VALU instruction instead of v_cvt_f32_f16 + v_mul_f32 / v_fma_f32.
template <int HI>
static __device__ __forceinline__ float mul_half(float b, uint32_t w) {
    float d;
    if (HI)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
            : "=v"(d) : "v"(b), "v"(w), "s"(0x80000000u));
    else
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]"
            : "=v"(d) : "v"(b), "v"(w), "s"(0x80000000u));
    return d;
}
template <int HI>
static __device__ __forceinline__ float fma_half(float b, uint32_t w, float c) {
    float d;
    if (HI)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
            : "=v"(d) : "v"(b), "v"(w), "v"(c));
    else
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(b), "v"(w), "v"(c));
    return d;
}

// b * (float)h + c with TWO roundings (product, then sum) as one two-instruction unit:
// bit-identical to mul_half(b, w) + c.  Written as one asm block so that the product never
// lives longer than one instruction -- left to itself the scheduler computes every product of
// a dot product up front (they are independent) and pays ~30 VGPRs for it.
template <int HI>
static __device__ __forceinline__ float mul_add_half(float b, uint32_t w, float c) {
    float d, t;
    if (HI)
        asm("v_fma_mix_f32 %1, %2, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\t"
            "v_add_f32 %0, %1, %5"
            : "=v"(d), "=&v"(t) : "v"(b), "v"(w), "s"(0x80000000u), "v"(c));
    else
        asm("v_fma_mix_f32 %1, %2, %3, %4 op_sel_hi:[0,1,0]\n\t"
            "v_add_f32 %0, %1, %5"
            : "=v"(d), "=&v"(t) : "v"(b), "v"(w), "s"(0x80000000u), "v"(c));
    return d;
}

// vr_expf: the deterministic expf of DESIGN.md.  The reference calls CUDA's
// expf (rt_core.cuh:119,160), whose bits depend on NVIDIA's ex2.approx; this is
// a pure IEEE-op algorithm so host oracle and device agree bit for bit:
//   clamp to [-104, 89]; k = rint(x*log2e); r = x - k*ln2 (two-step Cody-Waite);
//   degree-5 Horner (Cephes coefficients); result = (y*2^(k>>1)) * 2^(k-(k>>1)).
static __device__ __forceinline__ float vr_expf(float x) {
    // Branch-free form of the spec: v_med3_f32 is the clamp (a NaN stays a NaN
    // through the fma chain), v_ldexp_f32 is the single-rounding power-of-two
    // scaling (identical to the spec's two exact-then-rounded multiplies).
    x = __builtin_amdgcn_fmed3f(x, -104.0f, 89.0f);
    const float kf = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = __builtin_fmaf(kf, -0.693145751953125f, x);
    r = __builtin_fmaf(kf, -1.428606765330187045037746429443359375e-06f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    return __builtin_amdgcn_ldexpf(y, (int)kf);
}

// Two independent vr_expf in one instruction stream: the multiplies / fmas / adds are packed
// (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: two binary32 operations per lane and issue slot,
// each rounded exactly like its scalar form), clamp / rint / ldexp stay per component.  Every
// component goes through the operation sequence of vr_expf above -- same bits.
typedef float float2v __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ float2v splat2(float v) { return (float2v){v, v}; }
static __device__ __forceinline__ float2v vr_expf2(float2v x) {
    x.x = __builtin_amdgcn_fmed3f(x.x, -104.0f, 89.0f);
    x.y = __builtin_amdgcn_fmed3f(x.y, -104.0f, 89.0f);
    const float2v t = x * splat2(1.44269502162933349609375f);
    const float2v kf = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
    float2v r = __builtin_elementwise_fma(kf, splat2(-0.693145751953125f), x);
    r = __builtin_elementwise_fma(kf, splat2(-1.428606765330187045037746429443359375e-06f), r);
    float2v p = splat2(1.9875691500e-4f);
    p = __builtin_elementwise_fma(p, r, splat2(1.3981999507e-3f));
    p = __builtin_elementwise_fma(p, r, splat2(8.3334519073e-3f));
    p = __builtin_elementwise_fma(p, r, splat2(4.1665795894e-2f));
    p = __builtin_elementwise_fma(p, r, splat2(1.6666665459e-1f));
    p = __builtin_elementwise_fma(p, r, splat2(5.0000001201e-1f));
    const float2v r2 = r * r;
    float2v y = __builtin_elementwise_fma(p, r2, r);
    y = y + splat2(1.0f);
    return (float2v){__builtin_amdgcn_ldexpf(y.x, (int)kf.x), __builtin_amdgcn_ldexpf(y.y, (int)kf.y)};
}

template <int FMA>
static __device__ __forceinline__ float norm3(const float* d) {
    using P = Policy<FMA>;
    float s = P::madd(d[0], d[0], d[1] * d[1]);
    s = P::madd(d[2], d[2], s);
    return __builtin_sqrtf(s);
}

template <int FMA>
static __device__ __forceinline__ void normalize3(float* d) {
    const float inv = 1.f / norm3<FMA>(d);
    d[0] *= inv;
    d[1] *= inv;
    d[2] *= inv;
}

// column-major 3x3 (first 9 floats of the 4x3 c2w) times vector
template <int FMA>
static __device__ __forceinline__ void mv3(const float* m, const float* v, float* out) {
    using P = Policy<FMA>;
    out[0] = P::madd(m[6], v[2], P::madd(m[0], v[0], m[3] * v[1]));
    out[1] = P::madd(m[7], v[2], P::madd(m[1], v[0], m[4] * v[1]));
    out[2] = P::madd(m[8], v[2], P::madd(m[2], v[0], m[5] * v[1]));
}

template <int FMA>
static __device__ __forceinline__ float dot3(const float* u, const float* v) {
    using P = Policy<FMA>;
    return P::madd(u[2], v[2], P::madd(u[0], v[0], u[1] * v[1]));
}

template <int FMA>
static __device__ __forceinline__ void cross3(const float* a, const float* b, float* out) {
    using P = Policy<FMA>;
    out[0] = P::msub(a[1], b[2], a[2] * b[1]);
    out[1] = P::msub(a[2], b[0], a[0] * b[2]);
    out[2] = P::msub(a[0], b[1], a[1] * b[0]);
}

}  // namespace vr
