/* Host shim for <cuda_fp16.h> (ORACLE / test infrastructure only). */
#ifndef VR_REF_SHIM_CUDA_FP16_H_
#define VR_REF_SHIM_CUDA_FP16_H_
#include <cstdint>
#include <cstring>
struct __half {
    uint16_t x;
};
typedef __half half;
/* exact binary16 -> binary32 */
static inline float __half2float(__half hv) {
    const uint16_t h = hv.x;
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else {
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
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}
#endif
