// valu_rate2 -- issue cost of the integer / conversion / compare instructions of the march round on
// gfx950 (companion of valu_rate.hip): 8 independent destination registers x 32 x 64 iterations,
// 5 waves per SIMD, wall clock.  Prints SIMD cycles per wave64 instruction at 2.2 GHz.
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
// T = instruction text with %D (destination / accumulator register) and %S (a second source)
#define OP8(T1) \
    asm volatile(T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) "\n" T1(4) "\n" T1(5) "\n" T1(6) "\n" T1(7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s0), "v"(s1) : "vcc");

#define DEFK(NAME, T1)                                                                        \
    __global__ void NAME(unsigned* out, int iters) {                                          \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,      \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                       \
        unsigned s0 = 0x3f800001u + threadIdx.x, s1 = 3;                                      \
        for (int i = 0; i < iters; ++i) { REP32(OP8(T1)) }                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;   \
    }
#define X(i) #i
#define T_ADD_U32(i) "v_add_u32 %" X(i) ", %" X(i) ", %8"
#define T_AND(i) "v_and_b32 %" X(i) ", %" X(i) ", %8"
#define T_LSHL(i) "v_lshlrev_b32 %" X(i) ", 1, %" X(i)
#define T_LSHR(i) "v_lshrrev_b32 %" X(i) ", %9, %" X(i)
#define T_BFE(i) "v_bfe_u32 %" X(i) ", %" X(i) ", 3, 5"
#define T_LSHL_OR(i) "v_lshl_or_b32 %" X(i) ", %" X(i) ", 1, %8"
#define T_ADD_LSHL(i) "v_add_lshl_u32 %" X(i) ", %" X(i) ", %8, 2"
#define T_OR3(i) "v_or3_b32 %" X(i) ", %" X(i) ", %8, %9"
#define T_MOV(i) "v_mov_b32 %" X(i) ", %8"
#define T_CNDMASK(i) "v_cndmask_b32 %" X(i) ", %" X(i) ", %8, vcc"
#define T_CMP(i) "v_cmp_lt_f32 vcc, %" X(i) ", %8"
#define T_CMP64(i) "v_cmp_lt_f32 s[20:21], %" X(i) ", %8"
#define T_MED3(i) "v_med3_f32 %" X(i) ", %" X(i) ", 0, %8"
#define T_MAX(i) "v_max_f32 %" X(i) ", %" X(i) ", %8"
#define T_MIN3(i) "v_min3_f32 %" X(i) ", %" X(i) ", %8, %9"
#define T_MUL(i) "v_mul_f32 %" X(i) ", %" X(i) ", %8"
#define T_MULNEG(i) "v_mul_f32 %" X(i) ", %" X(i) ", -%8"
#define T_FMAC(i) "v_fmac_f32 %" X(i) ", %8, %9"
#define T_FRACT(i) "v_fract_f32 %" X(i) ", %" X(i)
#define T_CVTU(i) "v_cvt_u32_f32 %" X(i) ", %" X(i)
#define T_CVTF(i) "v_cvt_f32_u32 %" X(i) ", %" X(i)
#define T_LDEXP(i) "v_ldexp_f32 %" X(i) ", %" X(i) ", %9"
#define T_RNDNE(i) "v_rndne_f32 %" X(i) ", %" X(i)
#define T_MBCNT(i) "v_mbcnt_lo_u32_b32 %" X(i) ", %8, %" X(i)
#define T_ALIGNBIT(i) "v_alignbit_b32 %" X(i) ", %8, %" X(i) ", 8"
#define T_RCP(i) "v_rcp_f32 %" X(i) ", %" X(i)
#define T_DIVFIXUP(i) "v_div_fixup_f32 %" X(i) ", %" X(i) ", %8, %9"
#define T_SUB(i) "v_sub_f32 %" X(i) ", %8, %" X(i)
#define T_PKMUL(i) "v_pk_mul_f32 v[" X(i) "0:" X(i) "1], v[" X(i) "0:" X(i) "1], v[" X(i) "0:" X(i) "1]"
DEFK(k_add_u32, T_ADD_U32) DEFK(k_and, T_AND) DEFK(k_lshl, T_LSHL) DEFK(k_lshr, T_LSHR) DEFK(k_bfe, T_BFE)
DEFK(k_lshl_or, T_LSHL_OR) DEFK(k_add_lshl, T_ADD_LSHL) DEFK(k_or3, T_OR3) DEFK(k_mov, T_MOV)
DEFK(k_cndmask, T_CNDMASK) DEFK(k_cmp, T_CMP) DEFK(k_cmp64, T_CMP64) DEFK(k_med3, T_MED3) DEFK(k_max, T_MAX)
DEFK(k_min3, T_MIN3) DEFK(k_mul, T_MUL) DEFK(k_mulneg, T_MULNEG) DEFK(k_fmac, T_FMAC) DEFK(k_fract, T_FRACT)
DEFK(k_cvtu, T_CVTU) DEFK(k_cvtf, T_CVTF) DEFK(k_ldexp, T_LDEXP) DEFK(k_rndne, T_RNDNE) DEFK(k_mbcnt, T_MBCNT)
DEFK(k_alignbit, T_ALIGNBIT) DEFK(k_rcp, T_RCP) DEFK(k_divfixup, T_DIVFIXUP) DEFK(k_sub, T_SUB)

template <typename K>
void run(const char* name, K kern) {
    const int iters = 64, waves = 5, blocks = 256 * 4 * waves;
    unsigned* out;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_inst = (double)iters * 32 * 8 * waves;
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ns_per_inst_per_simd\": %.3f, \"cycles_at_2p2GHz\": %.2f}\n",
           name, waves, best * 1e6 / n_inst, best * 1e6 / n_inst * 2.2);
    (void)hipFree(out);
}

int main() {
#define R(n) run(#n, n)
    R(k_add_u32); R(k_and); R(k_lshl); R(k_lshr); R(k_bfe); R(k_lshl_or); R(k_add_lshl); R(k_or3); R(k_mov);
    R(k_cndmask); R(k_cmp); R(k_cmp64); R(k_med3); R(k_max); R(k_min3); R(k_mul); R(k_mulneg); R(k_fmac);
    R(k_fract); R(k_cvtu); R(k_cvtf); R(k_ldexp); R(k_rndne); R(k_mbcnt); R(k_alignbit); R(k_rcp);
    R(k_divfixup); R(k_sub);
    return 0;
}
