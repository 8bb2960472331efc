// vr_internal.h -- shared between the C-ABI host layer (vr_api.cpp) and the
// gfx950 kernels (vr_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "volrend_hip.h"

namespace vr {

// Everything the render kernel needs, passed BY VALUE as the kernel argument
// (the reference passes CameraSpec + TreeSpec + RenderOptions by value and the
// 12-float pose through a 48-byte H2D copy per frame, src/camera.cpp:67-75;
// here the pose rides in the argument block, so a frame is exactly one launch).
struct KParams {
    // ---- tree (TreeSpec, data_spec.hpp:23-50) ----
    const uint32_t* nodes;    // device layout: one word per child slot (vr_kernels.hip)
    const uint16_t* leaves;   // device layout: padded coefficient records
    const uint32_t* grid;     // device layout: top-level restart grid (N == 2) or NULL
    const float* extra;
    float offset[3];
    float scale[3];
    int32_t N, N3;
    int32_t data_dim;
    int32_t format;
    int32_t basis_dim;
    int32_t leaf_stride_h;    // fp16 elements between padded records
    int32_t max_depth;        // deepest leaf level (child words read - 1)
    int32_t grid_levels;      // G: grid has 2^G cells per axis (0 = no grid)
    int32_t xcd_remap;        // 1: contiguous wave-block range per XCD
    int32_t march_max;        // empty-space steps per lane between two shade phases
    float ndc_width, ndc_height, ndc_focal;
    // ---- camera (CameraSpec, data_spec.hpp:11-22) ----
    float xf[12];
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
    // ---- frame / sharding ----
    uint8_t* rgba;
    int64_t pitch;
    const float* depth;
    float* accum;
    int32_t offscreen;
    int32_t layout;
    int32_t tile_w, tile_h;      // multiples of 8
    int32_t tiles_x, tiles_y;
    int32_t rank, world;
    int32_t n_local_tiles;
    int32_t wblocks_per_tile_x;  // tile_w / 8
    int32_t wblocks_per_tile;    // (tile_w/8)*(tile_h/8)
    int64_t n_wave_blocks;       // n_local_tiles * wblocks_per_tile
    uint32_t* status;            // device word: bit0 = iteration cap hit
    unsigned long long* counters;  // optional VrCounters (7 x u64), instrumentation
};

// vr_kernels.hip
hipError_t launch_render(const KParams& p, int fp_mode, hipStream_t stream);
hipError_t launch_assemble(uint8_t* frame, int64_t pitch, const uint8_t* gathered, int width,
                           int height, int tile_w, int tile_h, int world, hipStream_t stream);
hipError_t launch_probe(const KParams& p, const float probe[3], float* out_dev,
                        hipStream_t stream);
// upload-time re-layout of the reference arrays into the device layout
int leaf_stride_halfs(int data_dim);
hipError_t launch_relayout(const int32_t* child, const uint16_t* data, uint32_t* nodes,
                           uint16_t* leaves, int64_t n_slots, int N3, int data_dim, int stride_h,
                           hipStream_t stream);
hipError_t launch_build_grid(const uint32_t* nodes, uint32_t* grid, int G, hipStream_t stream);

}  // namespace vr
