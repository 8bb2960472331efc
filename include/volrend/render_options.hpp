// volrend::RenderOptions: the per-frame knobs of the ray march.  Names, types and defaults are
// those of the reference (include/volrend/render_options.hpp:11-53) so that callers compile
// unchanged; VrRenderOptions (volrend_hip.h) is the C-ABI image of this struct.  `render_depth`
// exists unconditionally: the device backend is always present here.
#pragma once
#include "volrend/common.hpp"

#define VOLREND_GLOBAL_BASIS_MAX 25  // most basis functions per channel (SH degree 4)

namespace volrend {

struct RenderOptions {
    // -- marching (rt_core.cuh:108-188)
    float step_size = 1e-4f, sigma_thresh = 1e-2f, stop_thresh = 1e-2f;
    //    step_size: added to every leaf-exit distance; sigma_thresh: densities up to this are
    //    skipped; stop_thresh: a ray ends once its remaining light falls below this
    float background_brightness = 1.f;

    // -- what is shown
    float render_bbox[6] = {0.f, 0.f, 0.f, 1.f, 1.f, 1.f};    // min xyz, max xyz in tree space [0, 1]
    int basis_minmax[2] = {0, VOLREND_GLOBAL_BASIS_MAX - 1};  // basis functions outside are zeroed
    float rot_dirs[3] = {0.f, 0.f, 0.f};                      // axis-angle turn of every view direction

    // -- carried for source compatibility: the device kernel ignores them, as upstream's does
    bool show_grid = false;
    int grid_max_depth = 4;

    bool render_depth = false;                                // grey depth image instead of colour

    // -- lumisphere probe (volrend.cu:100-134, 175-191)
    bool enable_probe = false;
    float probe[3] = {0.f, 0.f, 1.f};
    int probe_disp_size = 100;
};

}  // namespace volrend

// The member ORDER is the reference's too (a VOLREND_CUDA build: render_options.hpp:40-52), so
// that aggregate initialisation and offset-based bindings (embind value_object fields) carry over.
#include <cstddef>
static_assert(offsetof(volrend::RenderOptions, step_size) == 0 &&
                  offsetof(volrend::RenderOptions, background_brightness) == 12 &&
                  offsetof(volrend::RenderOptions, render_bbox) == 16 &&
                  offsetof(volrend::RenderOptions, basis_minmax) == 40 &&
                  offsetof(volrend::RenderOptions, rot_dirs) == 48 &&
                  offsetof(volrend::RenderOptions, show_grid) == 60 &&
                  offsetof(volrend::RenderOptions, grid_max_depth) == 64 &&
                  offsetof(volrend::RenderOptions, render_depth) == 68 &&
                  offsetof(volrend::RenderOptions, enable_probe) == 69 &&
                  offsetof(volrend::RenderOptions, probe) == 72 &&
                  offsetof(volrend::RenderOptions, probe_disp_size) == 84 &&
                  sizeof(volrend::RenderOptions) == 88,
              "RenderOptions layout differs from the reference's");
