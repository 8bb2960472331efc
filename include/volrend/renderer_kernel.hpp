// volrend::launch_renderer -- the reference's device entry point
// (include/volrend/cuda/renderer_kernel.hpp:9-12) over the HIP C ABI.
//   cudaArray_t image_arr  -> linear device RGBA8 buffer (pitch = width*4 unless given)
//   cudaArray_t depth_arr  -> linear device R32F buffer or nullptr
//   cudaStream_t           -> hipStream_t passed as void*
// Asynchronous like the reference: returns after enqueueing on `stream`.
#pragma once
#include <vector>

#include "volrend/camera.hpp"
#include "volrend/n3tree.hpp"
#include "volrend/render_options.hpp"

namespace volrend {

void launch_renderer(const N3Tree& tree, const Camera& cam, const RenderOptions& options,
                     void* image_rgba8_dev, const float* depth_dev, void* stream,
                     bool offscreen = false);

// The same for a list of poses known up front (the volrend_headless loop,
// main_headless.cpp:207-225): one launch per <= VR_MAX_BATCH poses.  transforms[i] is
// the 12-float column-major 4x3 c2w of images[i].
void launch_renderer_batch(const N3Tree& tree, const Camera& cam,
                           const std::vector<const float*>& transforms,
                           const RenderOptions& options, const std::vector<void*>& images,
                           void* stream, bool offscreen = true);

}  // namespace volrend
