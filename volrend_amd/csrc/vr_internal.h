// vr_internal.h -- shared between the C-ABI host layer (vr_api.cpp) and the
// gfx950 kernels (vr_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "volrend_hip.h"

namespace vr {

constexpr int kMaxBatch = 512;   // frames per launch (VR_MAX_BATCH)
constexpr int kTableChunk = 48;  // frames per prepare_launch_kernel call (4 KB kernarg limit)
constexpr int kTouchLeafShift = 7;  // log2 of the bytes one bit of the records' distinct-line bitmap stands for
                                    // (layout studies patch it: 6 = one bit per 64-byte SH9 record slot)

// Per-frame part of a launch: pose and buffers.  Lives in device memory (one
// small table per launch slot) because lanes of one wave may hold rays of
// different frames.
struct FrameDesc {
    float xf[12];  // column-major 4x3 c2w (CameraSpec.transform)
    uint8_t* rgba;
    float* accum;
    const float* depth;
    unsigned long long* counters;  // optional VrCounters (7 x u64)
};

struct FrameTable {
    int32_t first;  // index of f[0] in the device table
    int32_t n;
    FrameDesc f[kTableChunk];
};

// Everything else the render kernel needs, passed BY VALUE as the kernel argument
// (the reference passes CameraSpec + TreeSpec + RenderOptions by value and the
// 12-float pose through a 48-byte H2D copy per frame, src/camera.cpp:67-75).
struct KParams {
    // ---- tree (TreeSpec, data_spec.hpp:23-50), device layout ----
    const uint32_t* nodes;    // one word per child slot (vr_kernels.hip)
    const uint16_t* leaves;   // padded coefficient records
    const uint2* top;         // N == 2: 8^top_levels cells, see "lookup structure" in vr_kernels.hip
    const uint32_t* bricks;   // N == 2: n_bricks * 8^brick_levels entries
    const float* extra;
    float offset[3];
    float scale[3];
    int32_t N, N3;
    int64_t capacity;         // nodes
    int32_t data_dim;
    int32_t format;
    int32_t basis_dim;
    int32_t leaf_stride_h;    // fp16 elements between padded records
    int32_t max_depth;        // deepest leaf level (child words read - 1)
    int32_t top_levels;       // G0: the top grid has 2^G0 cells per axis (0 = no lookup structure)
    int32_t brick_levels;     // BL: a brick has 2^BL entries per axis (0 = no bricks)
    int32_t brick_blocked;    // 8^3 bricks stored in 4 x 4 x 2 line blocks (per tree, fixed at upload)
    float ndc_width, ndc_height, ndc_focal;
    // ---- camera intrinsics (CameraSpec, data_spec.hpp:11-22); poses are per frame ----
    int32_t width, height;
    float fx, fy;
    // ---- options (RenderOptions, render_options.hpp:11-53) ----
    float step_size, sigma_thresh, stop_thresh, background_brightness;
    float bbox[6];
    int32_t basis_min, basis_max;
    int32_t render_depth;
    int32_t enable_probe;
    int32_t probe_disp_size;
    const float* probe_coeffs;
    // rodrigues(opt.rot_dirs): the per-frame uniform part (angle, axis, cos, sin)
    // is evaluated once on the host (volrend.cu:59-64), the per-ray part on device
    int32_t rot_enabled;
    float rot_k[3];
    float rot_cos, rot_sin;
    // ---- frames / sharding ----
    const FrameDesc* frames;     // device table, n_frames entries
    int32_t n_frames;
    int64_t pitch;
    int32_t offscreen;
    int32_t layout;
    int32_t tile_w, tile_h;      // multiples of 8
    int32_t tiles_x, tiles_y;
    int32_t rank, world;
    int32_t n_local_tiles;
    int32_t wblocks_per_tile_x;  // tile_w / 8
    int32_t wblocks_per_tile;    // (tile_w/8)*(tile_h/8)
    int64_t n_wave_blocks;       // per frame: n_local_tiles * wblocks_per_tile
    uint32_t total_rays;         // n_frames * n_wave_blocks * 64
    // ---- persistent scheduling ----
    uint32_t* queue_head;        // 8 x (head, count) word pairs, 16 words apart, reset by prepare_launch_kernel
    int32_t n_queues;            // 1 or 8 (one ray-id range per XCD)
    int32_t chunk_max;           // largest ray-id chunk a wave takes at once (multiple of 64)
    const uint32_t* ray_buf;     // rays (written by raygen_kernel), blocked SoA; queue x's rays compacted to the
    uint32_t* ray_buf_rw;        // front of its region (vr_kernels.hip "Ray queues")
    int32_t basis_words;         // basis_fn values a ray carries in registers (0 for RGBA)
    int32_t ray_tail_words;      // words of a ray record behind its 16-word head: the 3 words of the
                                 // view direction (ray_vdir) or the basis_words basis values
    int32_t ray_vdir;            // SH trees: the record carries the view direction, the basis is
                                 // evaluated when a lane takes the ray
    int32_t refill_min;          // refill once this many lanes are idle
    int32_t march_max;           // march steps per lane between two shade checks
    int32_t drain_flush;         // drain phase: a ray blocked by its full colour queue gets a partial shade round
                                 // at once when at most this many lanes of the wave still march (0 = never)
    int32_t max_iter;            // guard: march rounds of a wave without a retired ray before it cuts its rays
    int32_t instrumented;        // any frame carries counters -> FULL flavour
    int32_t any_accum;           // some frame of the launch asks for its fp32 accumulators
    int32_t records_nt;          // record DMA loads carry the non-temporal hint (large lookup structures)
    int32_t frame_group;         // ray-id order: frames per group (block major, frame minor inside)
    int32_t super_block;         // ray-id order: blocks of a tile visited in SxS super-blocks
    uint32_t* status;            // device word: bit0 = iteration cap hit
    unsigned long long* sched_stats;  // 8 x u64 scheduling tallies (instrumented flavours)
    // distinct-line meter (instrumented flavours, vr_touch_enable): one bit per 128-byte line of
    // leaves / nodes / top / bricks, set by every access; NULL when off
    uint32_t* touch[4];
};

// vr_kernels.hip
hipError_t launch_prepare(const KParams& p, const FrameTable& tbl, hipStream_t stream);
hipError_t launch_render(const KParams& p, int fp_mode, int n_cus, int waves_override, int gen_waves,
                         hipStream_t stream);
hipError_t launch_assemble(uint8_t* frame, int64_t pitch, const uint8_t* gathered, int width,
                           int height, int tile_w, int tile_h, int world, int n_frames,
                           int64_t out_stride, int64_t rank_stride, int64_t in_stride,
                           hipStream_t stream);
hipError_t launch_probe(const KParams& p, const float probe[3], float* out_dev,
                        hipStream_t stream);
// upload-time re-layout of the reference arrays into the device layout
int leaf_stride_halfs(int data_dim);
hipError_t launch_relayout(const int32_t* child, const uint16_t* data, const int32_t* perm,
                           uint32_t* nodes, uint16_t* leaves, int64_t n_slots, int N3,
                           int data_dim, int stride_h, hipStream_t stream);
// codebook decode of a quantised tree.npz into the reference's flat data layout (device)
hipError_t launch_decode_quant(const uint16_t* colors, const uint16_t* map, const uint16_t* sigma,
                               const uint16_t* retained, uint16_t* data, int64_t n_slots,
                               int n_quant, int n_ret, int data_dim, hipStream_t stream);
// number of set bits of a bitmap of n_words 32-bit words, added to *out
hipError_t launch_popcount(const uint32_t* words, uint64_t n_words, unsigned long long* out,
                           hipStream_t stream);
// N == 2 lookup structure (top grid + bricks), built from the re-laid-out node words
hipError_t launch_build_lookup(const uint32_t* nodes, const int32_t* brick_root, int n_bricks,
                               uint2* top, uint32_t* bricks, int top_levels, int brick_levels,
                               int brick_blocked, uint32_t* error_flag, hipStream_t stream);

}  // namespace vr
