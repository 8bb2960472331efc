// volrend::RenderOptions -- field for field the reference's
// include/volrend/render_options.hpp:11-53 (render_depth is always present: the device
// backend always exists here).
#pragma once
#include "volrend/common.hpp"

// Max global basis
#define VOLREND_GLOBAL_BASIS_MAX 25

namespace volrend {

struct RenderOptions {
    // * BASIC RENDERING
    float step_size = 1e-4f;      // epsilon added to steps to avoid re-hitting the current box
    float sigma_thresh = 1e-2f;   // sigma below this counts as 0
    float stop_thresh = 1e-2f;    // stop marching when the remaining light is below this
    float background_brightness = 1.f;

    // * VISUALIZATION
    // [minx, miny, minz, maxx, maxy, maxz] relative to the tree bounding box [0, 1]
    float render_bbox[6] = {0.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    // Range of basis functions to use (no effect for RGBA)
    int basis_minmax[2] = {0, VOLREND_GLOBAL_BASIS_MAX - 1};
    // Rotation applied to viewdirs for all rays
    float rot_dirs[3] = {0.f, 0.f, 0.f};

    // * ADVANCED VISUALIZATION
    bool show_grid = false;
    int grid_max_depth = 4;
    bool render_depth = false;

    // * Probe for inspecting lumispheres
    bool enable_probe = false;
    float probe[3] = {0.f, 0.f, 1.f};
    int probe_disp_size = 100;
};

}  // namespace volrend
