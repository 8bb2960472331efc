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

// launch_renderer only enqueues, so it cannot report what a launch found out on the device: the
// sticky status word of the tree (vr_tree_status; bit 0 = rays hit the sample guard, their
// pixels are wrong).  Call this once the stream is idle -- the end of a render loop, or after the
// copy of a frame -- to get the reference's loud failure (src/cuda/common.cu:8-21 prints and
// exits; this throws std::runtime_error and clears the word).  Synchronous.
void check_render_status(const N3Tree& tree);
// The same on a stream the caller is about to wait for anyway (vr_tree_status_on): enqueues the
// read of the word (and its clear) on `stream`, synchronises THAT stream only and throws as above.  Other streams
// that render the same tree are not waited for (check_render_status copies on the legacy stream,
// which waits for every blocking stream of the device); the word is the tree's, so bits set by
// their launches show up in whichever check comes first.
void check_render_status(const N3Tree& tree, void* stream);

}  // namespace volrend
