// valu_rate3 -- vector instructions with SCALAR operands (uniform parameters in SGPRs, lane masks,
// literal constants) on gfx950; same harness as valu_rate2.hip.
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define OP8(T1) \
    asm volatile(T1(0) "\n" T1(1) "\n" T1(2) "\n" T1(3) "\n" T1(4) "\n" T1(5) "\n" T1(6) "\n" T1(7) \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s0), "s"(u0), "s"(m0) : "vcc", "s20", "s21");
#define DEFK(NAME, T1)                                                                        \
    __global__ void NAME(unsigned* out, int iters, unsigned u0, unsigned long long m0) {      \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,      \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                       \
        unsigned s0 = 0x3f800001u + threadIdx.x;                                              \
        asm volatile("s_mov_b64 vcc, %0" :: "s"(m0) : "vcc");                                 \
        for (int i = 0; i < iters; ++i) { REP32(OP8(T1)) }                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;   \
    }
#define X(i) #i
#define T_ADD_VV(i) "v_add_f32 %" X(i) ", %8, %" X(i)
#define T_ADD_SV(i) "v_add_f32 %" X(i) ", %9, %" X(i)
#define T_ADD_LIT(i) "v_add_f32 %" X(i) ", 0x4b800000, %" X(i)
#define T_ADD_INL(i) "v_add_f32 %" X(i) ", 1.0, %" X(i)
#define T_FMA_S(i) "v_fma_f32 %" X(i) ", %" X(i) ", %9, %" X(i)
#define T_FMAAK(i) "v_fmaak_f32 %" X(i) ", %" X(i) ", %8, 0x3c088908"
#define T_CND_VCC(i) "v_cndmask_b32 %" X(i) ", %" X(i) ", %8, vcc"
#define T_CND_S(i) "v_cndmask_b32 %" X(i) ", %" X(i) ", %8, %10"
#define T_CND_CONST(i) "v_cndmask_b32 %" X(i) ", 0, 1, %10"
#define T_LSHR_S(i) "v_lshrrev_b32 %" X(i) ", %9, %" X(i)
#define T_MOV_S(i) "v_mov_b32 %" X(i) ", %9"
#define T_RFL(i) "v_readfirstlane_b32 s20, %" X(i)
#define T_CMP_S(i) "v_cmp_lt_f32 vcc, %9, %" X(i)
#define T_XOR(i) "v_xor_b32 %" X(i) ", %8, %" X(i)
#define T_LSHL_V(i) "v_lshlrev_b32 %" X(i) ", %8, %" X(i)
#define T_MAX_VV(i) "v_max_f32 %" X(i) ", %8, %" X(i)
#define T_MIN_U(i) "v_min_u32 %" X(i) ", %8, %" X(i)
#define T_SUB_U(i) "v_sub_u32 %" X(i) ", %8, %" X(i)
#define T_DIV_FMAS(i) "v_div_fmas_f32 %" X(i) ", %" X(i) ", %8, %" X(i)
#define T_DIV_SCALE(i) "v_div_scale_f32 %" X(i) ", vcc, %" X(i) ", %8, %" X(i)
#define T_DIV_SCALE_S(i) "v_div_scale_f32 %" X(i) ", s[20:21], %" X(i) ", %8, %" X(i)
#define T_CND64_VCC(i) "v_cndmask_b32_e64 %" X(i) ", %" X(i) ", %8, vcc"
#define T_ADDC(i) "v_addc_co_u32 %" X(i) ", vcc, 0, %" X(i) ", vcc"
#define T_ADD_CO(i) "v_add_co_u32 %" X(i) ", vcc, %8, %" X(i)
#define T_CMP_CND(i) "v_cmp_lt_f32 vcc, %8, %" X(i) "\n v_cndmask_b32 %" X(i) ", %" X(i) ", %8, vcc"
#define T_CMP64_CND(i) "v_cmp_lt_f32 s[20:21], %8, %" X(i) "\n s_nop 1\n v_cndmask_b32 %" X(i) ", %" X(i) ", %8, s[20:21]"
DEFK(k_add_vv, T_ADD_VV) DEFK(k_add_sv, T_ADD_SV) DEFK(k_add_literal, T_ADD_LIT) DEFK(k_add_inline, T_ADD_INL)
DEFK(k_fma_sgpr, T_FMA_S) DEFK(k_fmaak_literal, T_FMAAK) DEFK(k_cndmask_vcc, T_CND_VCC) DEFK(k_cndmask_sgpr, T_CND_S)
DEFK(k_cndmask_0_1_sgpr, T_CND_CONST) DEFK(k_lshr_sgpr, T_LSHR_S) DEFK(k_mov_sgpr, T_MOV_S)
DEFK(k_readfirstlane, T_RFL) DEFK(k_cmp_sgpr, T_CMP_S) DEFK(k_xor, T_XOR) DEFK(k_lshl_vv, T_LSHL_V)
DEFK(k_max_vv, T_MAX_VV) DEFK(k_min_u32, T_MIN_U) DEFK(k_sub_u32, T_SUB_U)
DEFK(k_div_fmas, T_DIV_FMAS) DEFK(k_div_scale_vcc, T_DIV_SCALE) DEFK(k_div_scale_sgpr, T_DIV_SCALE_S)
DEFK(k_cndmask_e64_vcc, T_CND64_VCC) DEFK(k_addc_vcc, T_ADDC) DEFK(k_add_co, T_ADD_CO)
DEFK(k_cmp_then_cndmask_vcc_PAIR, T_CMP_CND) DEFK(k_cmp_then_cndmask_sgpr_PAIR, T_CMP64_CND)

template <typename K>
void run(const char* name, K kern) {
    const int iters = 64, waves = 5, blocks = 256 * 4 * waves;
    unsigned* out;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0x3f800000u, 0x5555aaaa5555aaaaull);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, 0x3f800000u, 0x5555aaaa5555aaaaull);
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
    R(k_add_vv); R(k_add_sv); R(k_add_literal); R(k_add_inline); R(k_fma_sgpr); R(k_fmaak_literal);
    R(k_cndmask_vcc); R(k_cndmask_sgpr); R(k_cndmask_0_1_sgpr); R(k_lshr_sgpr); R(k_mov_sgpr);
    R(k_readfirstlane); R(k_cmp_sgpr); R(k_xor); R(k_lshl_vv); R(k_max_vv); R(k_min_u32); R(k_sub_u32);
    R(k_div_fmas); R(k_div_scale_vcc); R(k_div_scale_sgpr); R(k_cndmask_e64_vcc); R(k_addc_vcc); R(k_add_co);
    R(k_cmp_then_cndmask_vcc_PAIR); R(k_cmp_then_cndmask_sgpr_PAIR);
    return 0;
}
