// valu_rate -- what does one wave64 vector instruction cost on gfx950?  Issue rate of the
// instructions the shade / march rounds are made of, one kind at a time: a loop of 8 independent
// chains x 32 instructions, timed with the shader clock, at 1 and at 5 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = 1.0000001f;
    unsigned hw = (unsigned)(threadIdx.x * 2654435761u) | 0x3c003c00u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_fma_f32
            REP32(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 1) {  // v_add_f32
            REP32(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 2) {  // v_fma_mix_f32 (fp16 operand converted on the fly)
            REP32(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[0,1,0]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(hw));)
        } else if (KIND == 3) {  // v_pk_fma_f32
            REP32(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                               "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 4) {  // v_nop
            REP32(asm volatile("v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop");)
        } else if (KIND == 5) {  // v_cvt_f32_f16
            REP32(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
                               "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(hw));)
        } else if (KIND == 6) {  // v_mul_f64
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
            REP32(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                               "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));)
            a0 += (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 7) {  // v_lshl_or_b32 (integer, 3 operands)
            unsigned u0 = hw, u1 = hw + 1, u2 = hw + 2, u3 = hw + 3;
            REP32(asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4\n"
                               "v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4"
                               : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(hw));)
            a0 += (float)(u0 + u1 + u2 + u3);
        } else if (KIND == 8) {  // v_exp_f32 (transcendental)
            REP32(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                               "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 9) {  // dependent chain of v_fma_f32 (latency)
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                               "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0"
                               : "+v"(a0) : "v"(b));)
        } else if (KIND == 10) {  // s_cmp (scalar)
            REP32(asm volatile("s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0\n"
                               "s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0\n s_cmp_eq_u32 0, 0" ::: "scc");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    const int iters = 64, cus = 256;
    const int blocks = cus * 4 * waves_per_simd;  // 64-thread blocks: one wave each
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    hipMalloc(&cyc, (size_t)blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double n_inst = (double)iters * 32 * 8;
    // shader clock ticks per instruction as one wave sees it, and SIMD-cycles per instruction from the wall clock
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"clk_per_inst_per_wave\": %.2f, \"kernel_us\": %.1f, "
           "\"ns_per_inst_per_simd\": %.3f}\n",
           name, waves_per_simd, mean / n_inst, ms * 1e3, ms * 1e6 / (n_inst * waves_per_simd));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {1, 5}) {
        run<0>("v_fma_f32", w);
        run<1>("v_add_f32", w);
        run<2>("v_fma_mix_f32", w);
        run<3>("v_pk_fma_f32", w);
        run<4>("v_nop", w);
        run<5>("v_cvt_f32_f16", w);
        run<6>("v_mul_f64", w);
        run<7>("v_lshl_or_b32", w);
        run<8>("v_exp_f32", w);
        run<9>("v_fma_f32 dependent chain", w);
        run<10>("s_cmp_eq_u32", w);
    }
    return 0;
}
