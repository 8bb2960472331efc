// exec_mask_rate -- does a wave64 vector instruction get cheaper on gfx950 when whole 16- or 32-lane groups of
// its EXEC mask are empty?  (If it did, compacting the live rays of a thinning wave into its low lanes would
// make the drain phase cheaper without any exchange between waves.)  A loop of 8 independent v_fma_f32 / v_med3_f32
// / v_fma_mix chains under different EXEC masks, W waves per SIMD, timed with the wall clock; the first wave of
// every SIMD also reports shader clocks per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/exec_mask_rate.hip -o /tmp/exec_mask_rate && /tmp/exec_mask_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

__global__ __launch_bounds__(64) void k(float* out, unsigned long long mask, int iters, int kind) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
          a7 = a0 + 7;
    const float b = 1.0000001f;
    if (!((mask >> (threadIdx.x & 63)) & 1ull)) {  // lanes outside the mask leave: EXEC = mask for the loop
        out[blockIdx.x * 64 + threadIdx.x] = 0.f;
        return;
    }
    for (int i = 0; i < iters; ++i) {
        if (kind == 0) {
            REP32(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else {
            REP32(asm volatile("v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n"
                               "v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
    struct { const char* name; unsigned long long m; } masks[] = {
        {"all 64 lanes", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"low 16", 0xFFFFull}, {"low 15", 0x7FFFull}, {"low 14", 0x3FFFull},
        {"low 12", 0xFFFull}, {"low 10", 0x3FFull}, {"low 9", 0x1FFull}, {"low 8", 0xFFull}, {"low 4", 0xFull}, {"low 2", 3ull},
        {"lane 0", 1ull}, {"every 4th lane (16 lanes)", 0x1111111111111111ull}, {"every 8th lane (8 lanes)", 0x0101010101010101ull},
        {"every 16th lane (4 lanes)", 0x0001000100010001ull}, {"lanes 0-7 and 32-39 (16)", 0x000000FF000000FFull},
        {"lanes 0-3 of every 16 (16)", 0x000F000F000F000Full}, {"lanes 0-1 of every 16 (8)", 0x0003000300030003ull},
        {"every 2nd lane (32)", 0x5555555555555555ull}, {"high 16", 0xFFFF000000000000ull}, {"high 8", 0xFF00000000000000ull}};
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int w : {1, 4}) {
            for (auto& mk : masks) {
                const int grid = cus * 4 * w;  // w waves per SIMD
                hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, mk.m, 10, kind);
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, mk.m, iters, kind);
                hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                const double insts = (double)iters * 256.0;  // per wave
                printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"exec\": \"%s\", \"kernel_us\": %.1f, \"ns_per_inst_per_simd\": %.3f}\n",
                       kind == 0 ? "v_fma_f32" : "v_med3_f32", w, mk.name, us, us * 1e3 / (insts * w));
            }
        }
    return 0;
}
