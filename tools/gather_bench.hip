// Random 128-byte gather ceiling of the chip: what rate of L2<->fabric traffic can the CUs
// sustain when every request is a random, 128-byte aligned record (the access pattern of
// the SH-record fetch in render_kernel)?  Groups of 8 lanes read one record as 8 x 16 B,
// `U` independent records in flight per group and iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int U, int LANES>  // LANES lanes of 16 B per record: 8 -> 128 B, 4 -> 64 B
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ table, uint32_t n_records,
                                             int iters, uint32_t* out) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    const uint32_t grp = gid / LANES, vec = gid % LANES;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = mix(grp * 2654435761u + (uint32_t)(it * U + u) * 40503u) % n_records;
            q[u] = table[(uint64_t)r * 8u + vec];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// Divergent 4-byte gathers (the child-word loads of the march phase) from a table that stays
// in the vector L1 (16 KB) or in L2 (2 MB): lane-loads per second the TA/TCP path sustains.
template <int U>
__global__ __launch_bounds__(64) void gather4(const uint32_t* __restrict__ table, uint32_t mask,
                                              int iters, uint32_t* out) {
    const uint32_t gid = blockIdx.x * 64u + threadIdx.x;
    uint32_t acc = gid * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = table[mix(acc + (uint32_t)u * 40503u) & mask];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u] + (uint32_t)u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
int run4(const uint32_t* table, uint32_t words, int waves_per_cu, int cus, uint32_t* out) {
    const int blocks = waves_per_cu * cus;
    const int iters = 4096 / U;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather4<U>), dim3(blocks), dim3(64), 0, 0, table, words - 1, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather4<U>), dim3(blocks), dim3(64), 0, 0, table, words - 1, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)blocks * 64 * iters * U;
    printf("{\"dword_gather_table_bytes\": %u, \"in_flight_per_lane\": %d, \"waves_per_cu\": %d, "
           "\"ms\": %.3f, \"Glane_loads_per_s\": %.1f, \"wave_instr_per_us_per_cu\": %.2f}\n",
           words * 4, U, waves_per_cu, ms, loads / ms / 1e6, loads / 64 / ms / 1e3 / cus);
    return 0;
}

template <int U, int LANES>
int run(const uint4* table, uint32_t n_records, int waves_per_cu, int cus, uint32_t* out) {
    const int blocks = waves_per_cu * cus;
    const int iters = 2048 / U;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<U, LANES>), dim3(blocks), dim3(64), 0, 0, table, n_records, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<U, LANES>), dim3(blocks), dim3(64), 0, 0, table, n_records, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double records = (double)blocks * (64 / LANES) * iters * U;
    printf("{\"lanes_per_record\": %d, \"record_bytes\": %d, \"in_flight_per_group\": %d, "
           "\"waves_per_cu\": %d, \"ms\": %.3f, \"Grecords_per_s\": %.3f, \"line_GB_per_s\": %.1f}\n",
           LANES, LANES * 16, U, waves_per_cu, ms, records / ms / 1e6, records * 128.0 / ms / 1e6);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t n_records = 1u << 24;  // 2 GiB of 128-byte records
    uint4* table;
    uint32_t* out;
    CHECK(hipMalloc((void**)&table, (size_t)n_records * 128));
    CHECK(hipMalloc((void**)&out, 64));
    CHECK(hipMemset(table, 1, (size_t)n_records * 128));
    printf("# %s, %d CUs, table 2 GiB; line_GB_per_s counts one 128-byte line per record\n",
           prop.gcnArchName, cus);
    for (int w : {8, 16, 20, 32}) {
        if (run<1, 8>(table, n_records, w, cus, out)) return 1;
        if (run<2, 8>(table, n_records, w, cus, out)) return 1;
        if (run<4, 8>(table, n_records, w, cus, out)) return 1;
        if (run<8, 8>(table, n_records, w, cus, out)) return 1;
    }
    for (int w : {20, 32}) {
        if (run<2, 4>(table, n_records, w, cus, out)) return 1;
        if (run<8, 4>(table, n_records, w, cus, out)) return 1;
    }
    // the hash makes neighbouring lanes hit unrelated words: 64 distinct lines per instruction
    // for the 2 MB table, a 16 KB table stays in the vector L1
    for (uint32_t words : {4096u, 524288u}) {
        for (int w : {20, 32}) {
            if (run4<1>((const uint32_t*)table, words, w, cus, out)) return 1;
            if (run4<4>((const uint32_t*)table, words, w, cus, out)) return 1;
        }
    }
    return 0;
}
