// vr_experiment_hooks.h -- timing experiments and profiling builds of vr_kernels.hip; nothing in
// here is part of the product.  Included twice by vr_kernels.hip:
//   VR_HOOKS_PART 1 (inside namespace vr::{anonymous}, before the kernels): the VR_EXP_* / TL_* /
//                   TL3_* macros the kernels use.  In the product build (VR_ABLATE == 0,
//                   VR_TIMELINE == 0) every one of them expands to the plain expression / nothing.
//   VR_HOOKS_PART 2 (global namespace, end of the file): the host-side reader of the
//                   -DVR_TIMELINE=3 tallies.
// volrend_amd/build.py refuses to build the product library with -DVR_ABLATE / -DVR_TIMELINE.
#if VR_HOOKS_PART == 1
// -DVR_ABLATE=n removes work ON PURPOSE (wrong pictures, never shipped), -DVR_TIMELINE=n adds
// shader-clock reads around the phases of the render kernel.  The kernels only use the
// VR_EXP_* / TL_* / TL3_* names below.
#ifndef VR_ABLATE
#define VR_ABLATE 0
#endif
#ifndef VR_TIMELINE
#define VR_TIMELINE 0  // 1: per-phase cycle sums into sched_stats; 3: time-resolved tallies of a launch instead
#endif
#if VR_ABLATE == 0
#define VR_EXP_RECORD_CHUNK(v, j, leaf) ((v)[j])   // 16-byte chunk j of a record (register path)
#define VR_EXP_RECORD_LEAF(leaf) (leaf)            // the record an item names (DMA path)
#define VR_EXP_RECORD_DMA 1                        // record DMAs are issued
#define VR_EXP_FUSED_COLOUR 1                      // hit samples queue colour work
#define VR_EXP_STEAL 1                             // waves steal from the ray queues of other XCDs
#define VR_EXP_TOP_ENTRY(load) (load)              // the top-grid entry of a cell
#define VR_EXP_BRICK_WORD(load) (load)             // the brick entry of a sample
#else
#define VR_EXP_RECORD_CHUNK(v, j, leaf) \
    (VR_ABLATE == 1 ? (v)[0] : VR_ABLATE == 2 ? make_uint4((leaf) + (j), (leaf), (leaf), (leaf)) : (v)[j])
#define VR_EXP_RECORD_LEAF(leaf) (VR_ABLATE == 5 ? ((leaf) & 0x3FFu) : (leaf))  // 5: a 128 KB window
#define VR_EXP_RECORD_DMA (VR_ABLATE != 4)         // 4: no record fetch at all
#define VR_EXP_FUSED_COLOUR (VR_ABLATE != 6)       // 6: the kernel marches without colour work
#define VR_EXP_STEAL (VR_ABLATE != 8)              // 8: every wave stays with the ray queue of its XCD (same pictures)
// 9 / 10: the march without its memory latency -- every sample "finds" an empty leaf of depth G0 + 3
// without a brick load (9) and without the top load either (10): rays cross the whole volume in
// finest-level steps, nothing is shaded; what a round costs then is its instruction chain alone
#define VR_EXP_BRICK_WORD(load) (VR_ABLATE == 9 || VR_ABLATE == 10 ? (kLeafBit | (2u << 29)) : (load))
#define VR_EXP_TOP_ENTRY(load) (VR_ABLATE == 10 ? make_uint2(0u, 0u) : (load))
#endif
#if VR_TIMELINE == 1
#define TL_MARK() (tl_mark = __builtin_readcyclecounter())
#define TL_ADD(v) do { const unsigned long long n_ = __builtin_readcyclecounter(); \
                       (v) += n_ - tl_mark; tl_mark = n_; } while (0)
#elif VR_TIMELINE == 3
#define TL_MARK() (tl3_mark = (uint32_t)__builtin_readcyclecounter())
#define TL_ADD(v) TL3_TIME(TL3_IDX_##v)
#else
#define TL_MARK() ((void)0)
#define TL_ADD(v) ((void)0)
#endif
#if VR_TIMELINE == 3
// time-resolved tallies: per bucket of 2^15 shader clocks (14.9 us at 2.2 GHz; 64 buckets, counted
// from the start of the wave -- the waves of a persistent launch all start within ~20 us) the
// cycles spent in the retire / refill block, marching and shading, the march rounds, the marching
// lanes, and the waves that ended.  A wave counts in 768 bytes of its LDS (two scalar registers of
// state: the kernel has none to spare) and adds its table to one of 64 global copies when it
// ends; read (and cleared) by vr_exp_tl3_read() -- tools/tail_profile.py.
constexpr int kTl3Buckets = 64, kTl3Copies = 64, kTl3Rows = 6;
__device__ unsigned long long vr_tl3[kTl3Copies][kTl3Rows][kTl3Buckets];
#define TL3_IDX_tl_refill 0
#define TL3_IDX_tl_march 1
#define TL3_IDX_tl_shade_load 2
#define TL3_IDX_tl_shade_math 2
#define TL3_IDX_tl_shade_acc 2
#define TL3_BUCKET(now_) ((((now_) - tl3_clk0) >> 15) & (uint32_t)(kTl3Buckets - 1))
#define TL3_DECL() __shared__ uint32_t tl3_h[3 * kTl3Buckets];                                      \
                   for (int k_ = threadIdx.x & 63; k_ < 3 * kTl3Buckets; k_ += 64) tl3_h[k_] = 0;   \
                   __syncthreads();                                                                 \
                   const uint32_t tl3_clk0 = (uint32_t)__builtin_readcyclecounter();                \
                   uint32_t tl3_mark = tl3_clk0
// words of a bucket: [0] refill | march << 16 (units of 16 clocks), [1] shade | rounds << 16, [2] lanes
#define TL3_TIME(idx_) do { const uint32_t n_ = (uint32_t)__builtin_readcyclecounter();              \
        const uint32_t d_ = (n_ - tl3_mark) >> 4; tl3_mark = n_;                                     \
        if ((idx_) >= 0 && lane == 0)                                                               \
            atomicAdd(&tl3_h[3 * TL3_BUCKET(n_) + ((idx_) >> 1)], d_ << (16 * ((idx_) & 1))); } while (0)
#define TL3_ROUND(go_) do { const uint32_t l_ = (uint32_t)__builtin_popcountll(                       \
                                __builtin_amdgcn_ballot_w64(go_));                                  \
        if (lane == 0) { const uint32_t b_ = TL3_BUCKET((uint32_t)__builtin_readcyclecounter());    \
                         atomicAdd(&tl3_h[3 * b_ + 1], 1u << 16); atomicAdd(&tl3_h[3 * b_ + 2], l_); } } while (0)
#define TL3_SHADE(n_) ((void)0)
#define TL3_END() do { __syncthreads();                                                             \
        unsigned long long(*h_)[kTl3Buckets] = vr_tl3[blockIdx.x % kTl3Copies];                     \
        const uint32_t a_ = tl3_h[3 * lane], b_ = tl3_h[3 * lane + 1], c_ = tl3_h[3 * lane + 2];    \
        if (a_ | b_ | c_) {                                                                         \
            atomicAdd(&h_[0][lane], (unsigned long long)(a_ & 0xFFFFu));                            \
            atomicAdd(&h_[1][lane], (unsigned long long)(a_ >> 16));                                \
            atomicAdd(&h_[2][lane], (unsigned long long)(b_ & 0xFFFFu));                            \
            atomicAdd(&h_[3][lane], (unsigned long long)(b_ >> 16));                                \
            atomicAdd(&h_[4][lane], (unsigned long long)c_);                                        \
        }                                                                                           \
        if (lane == 0) atomicAdd(&h_[5][TL3_BUCKET((uint32_t)__builtin_readcyclecounter())], 1ull); } while (0)
#else
#define TL3_DECL() ((void)0)
#define TL3_ROUND(go_) ((void)0)
#define TL3_SHADE(n_) ((void)0)
#define TL3_END() ((void)0)
#endif
// declarations / dumps of the cycle tallies (sched_stats words: see tools/quick_ab.py, bench.py)
#if VR_TIMELINE == 1
#define TL_DECL_FUSED()                                                                          \
    unsigned long long tl_refill = 0, tl_march = 0, tl_shade_load = 0, tl_shade_math = 0,        \
                       tl_shade_acc = 0, tl_total0 = __builtin_readcyclecounter(), tl_mark = 0,  \
                       tl_drained = 0 /* when this wave found the ray queue empty */
#define TL_QUEUE_DRY() (tl_drained = __builtin_readcyclecounter())
#define TL_DUMP_FUSED()                                                                           \
    do {                                                                                          \
        if (!COUNT && p.sched_stats && lane == 0) {                                               \
            atomicAdd(&p.sched_stats[0], tl_refill);                                              \
            atomicAdd(&p.sched_stats[1], tl_march);                                               \
            atomicAdd(&p.sched_stats[2], tl_shade_load);                                          \
            atomicAdd(&p.sched_stats[3], tl_shade_math);                                          \
            atomicAdd(&p.sched_stats[4], tl_shade_acc);                                           \
            atomicAdd(&p.sched_stats[5], (unsigned long long)__builtin_readcyclecounter() - tl_total0); \
            atomicAdd(&p.sched_stats[6], 1ull);                                                   \
            /* the wave's tail: from the moment the queue was empty to its last retired ray */    \
            atomicAdd(&p.sched_stats[7], (unsigned long long)__builtin_readcyclecounter() - tl_drained); \
        }                                                                                         \
    } while (0)
#else
#define TL_DECL_FUSED() ((void)0)
#define TL_QUEUE_DRY() ((void)0)
#define TL_DUMP_FUSED() ((void)0)
#endif
#elif VR_HOOKS_PART == 2
#if VR_TIMELINE == 3
// experiment builds only (see the hooks at the top of this file): out = [kTl3Rows][kTl3Buckets] sums
extern "C" int vr_exp_tl3_read(unsigned long long* out, int reset) {
    using namespace vr;
    static unsigned long long host[kTl3Copies][kTl3Rows][kTl3Buckets];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(vr_tl3), sizeof(host)) != hipSuccess) return -1;
    for (int r = 0; r < kTl3Rows; ++r)
        for (int b = 0; b < kTl3Buckets; ++b) {
            unsigned long long v = 0;
            for (int c = 0; c < kTl3Copies; ++c) v += host[c][r][b];
            out[r * kTl3Buckets + b] = v;
        }
    if (reset) {
        for (auto& c : host) for (auto& r : c) for (auto& v : r) v = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(vr_tl3), host, sizeof(host)) != hipSuccess) return -1;
    }
    return kTl3Buckets;
}
#endif
#endif  // VR_HOOKS_PART
