// valu_rate4 -- 64-bit address arithmetic and integer multiplies on gfx950 (same harness as valu_rate2.hip)
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define OP4(T1) \
    asm volatile(T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) "\n" T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) \
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(s0), "v"(s1), "s"(sb) : "vcc", "s20", "s21");
#define DEFK(NAME, T1)                                                                        \
    __global__ void NAME(unsigned* out, int iters, unsigned long long sb) {                   \
        unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;           \
        unsigned s0 = 0x00010003u + threadIdx.x, s1 = 96;                                     \
        for (int i = 0; i < iters; ++i) { REP32(OP4(T1)) }                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(q0 + q1 + q2 + q3);           \
    }
#define X(i) #i
// %0..%3: 64-bit register pairs; %4, %5: 32-bit vector sources; %6: scalar pair
#define T_MAD64(i) "v_mad_u64_u32 %" X(i) ", s[20:21], %4, %5, %" X(i)
#define T_LSHL_ADD64(i) "v_lshl_add_u64 %" X(i) ", %" X(i) ", 3, %6"
#define T_MUL_LO(i) "v_mul_lo_u32 %" X(i) ", %4, %5"   /* writes the low half of the pair */
#define T_MUL_U24(i) "v_mul_u32_u24 %" X(i) ", %4, %5"
#define T_MAD_U24(i) "v_mad_u32_u24 %" X(i) ", %4, %5, %4"
#define T_ADD_CO_PAIR(i) "v_add_co_u32 %" X(i) ", vcc, %4, %" X(i) "\n v_addc_co_u32 %5, vcc, 0, %5, vcc"
#define T_MUL_HI(i) "v_mul_hi_u32 %" X(i) ", %4, %5"
#define OP4W(T1) \
    asm volatile(T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) "\n" T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) \
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(s0), "v"(s1), "s"(sb) : "vcc");
#define DEFK32(NAME, T1)                                                                      \
    __global__ void NAME(unsigned* out, int iters, unsigned long long sb) {                   \
        unsigned w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;                     \
        unsigned s0 = 0x00010003u + threadIdx.x, s1 = 96;                                     \
        for (int i = 0; i < iters; ++i) { REP32(OP4W(T1)) }                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = w0 + w1 + w2 + w3;                       \
    }
DEFK(k_mad_u64_u32, T_MAD64) DEFK(k_lshl_add_u64, T_LSHL_ADD64) DEFK32(k_mul_lo_u32, T_MUL_LO)
DEFK32(k_mul_u32_u24, T_MUL_U24) DEFK32(k_mad_u32_u24, T_MAD_U24) DEFK32(k_mul_hi_u32, T_MUL_HI)

template <typename K>
void run(const char* name, K kern) {
    const int iters = 64, waves = 5, blocks = 256 * 4 * waves;
    unsigned* out;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0x100000000ull);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0x100000000ull);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_inst = (double)iters * 32 * 8 * waves;
    printf("{\"inst\": \"%s\", \"ns_per_inst_per_simd\": %.3f, \"cycles_at_2p2GHz\": %.2f}\n", name,
           best * 1e6 / n_inst, best * 1e6 / n_inst * 2.2);
    (void)hipFree(out);
}

int main() {
#define R(n) run(#n, n)
    R(k_mad_u64_u32); R(k_lshl_add_u64); R(k_mul_lo_u32); R(k_mul_u32_u24); R(k_mad_u32_u24); R(k_mul_hi_u32);
    return 0;
}
