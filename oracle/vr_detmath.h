/*
 * vr_detmath.h -- ORACLE-side deterministic scalar math (test infrastructure).
 *
 * The reference calls CUDA libdevice expf (<= 2 ulp, built on the ex2.approx
 * hardware instruction -- not reproducible off an NVIDIA GPU).  Parity between
 * the CPU oracle and the gfx950 kernel therefore uses ONE published algorithm,
 * written independently on both sides from the spec in DESIGN.md ("vr_expf"):
 *
 *   NaN -> NaN;  x clamped to [-104, 89]
 *   k  = rint(x * 0x1.715476p+0f)                      (round-to-nearest-even)
 *   r  = fma(k, -0x1.62e400p-1f, x);  r = fma(k, -0x1.7f7d1cp-20f, r)
 *   p  = Horner degree-5 in r (Cephes expf coefficients), y = fma(p, r*r, r) + 1
 *   result = (y * 2^(k>>1)) * 2^(k-(k>>1))             (two exact power-of-two scalings,
 *                                                       one rounding when subnormal)
 * Only IEEE fp32 mul/add/fma/rint and integer ops: bit-reproducible anywhere.
 * Max error measured against double exp: < 1 ulp (tests/test_oracle_math.py).
 */
#ifndef VR_DETMATH_H_
#define VR_DETMATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float vr_bits2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t vr_f2bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static inline float vr_det_expf(float x) {
    if (x != x) return x;
    x = x < -104.0f ? -104.0f : x;
    x = x > 89.0f ? 89.0f : x;
    const float kf = rintf(x * 1.44269502162933349609375f);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.428606765330187045037746429443359375e-06f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    const int k = (int)kf;
    const int k1 = k >> 1; /* arithmetic shift: floor(k/2) */
    const int k2 = k - k1;
    const float s1 = vr_bits2f((uint32_t)(k1 + 127) << 23);
    const float s2 = vr_bits2f((uint32_t)(k2 + 127) << 23);
    return (y * s1) * s2;
}

/* IEEE binary16 bits -> binary32, exact (== CUDA __half2float) */
static inline float vr_half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else {
            /* subnormal: normalise */
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while ((man & 0x400u) == 0);
            out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        out = sign | 0x7F800000u | (man << 13);
    } else {
        out = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    return vr_bits2f(out);
}

/* PTX / GCN min,max semantics: NaN loses, -0 < +0 */
static inline float vr_minf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline float vr_maxf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? b : a;
    return a > b ? a : b;
}

#endif
