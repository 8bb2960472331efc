// volrend::VolumeRenderer -- the reference's renderer facade (include/volrend/renderer.hpp:11-42,
// src/cuda_renderer.cpp:83-195) without OpenGL: the frame is a linear RGBA8 buffer in device
// memory instead of a GL renderbuffer shared through CUDA-GL interop.
//
//   VolumeRenderer r;            // the reference needs a current GL context here; this does not
//   r.resize(800, 800);
//   r.set(tree);                 // narrows options.basis_minmax to the tree's basis, as upstream
//   r.camera.center = ...;  r.options.step_size = ...;
//   r.render();                  // camera._update() + clear + launch_renderer (interactive path:
//                                // offscreen = false, i.e. composited over what is in the frame)
//   r.read_frame(host_rgba8);    // or frame(): the device pointer, valid until the next-but-one render()
//
// What upstream draws with GL BEFORE the ray march -- meshes, the probe cube, the octree grid --
// reaches the kernel as two images: the RGBA8 colour it composites over and an R32F depth that
// ends each ray (cuda_renderer.cpp:83-126; volrend.cu:142-147,152-165).  Here a caller that has
// such images hands them in as device buffers (set_underlay); without them the frame is cleared to
// background_brightness and the depth to 1e9 exactly as upstream's glClear calls do
// (cuda_renderer.cpp:85-92).  `meshes` is not offered: rasterising them is GL work (north_star:
// no GL fallback).
#pragma once
#include <cstdint>
#include <memory>

#include "volrend/camera.hpp"
#include "volrend/n3tree.hpp"
#include "volrend/render_options.hpp"

namespace volrend {

// The stream and the frames live on the device of the tree that is set (the calling thread's
// current device while there is none): set() of a tree on another device moves them.
struct VolumeRenderer {
    explicit VolumeRenderer();
    ~VolumeRenderer();
    VolumeRenderer(const VolumeRenderer&) = delete;
    VolumeRenderer& operator=(const VolumeRenderer&) = delete;

    // Render the currently set tree (asynchronous, like upstream: the frame is complete once
    // stream() is idle; read_frame() waits)
    void render();

    // Set volumetric data to render (must be uploaded: N3Tree::open does that)
    void set(N3Tree& tree);

    // Clear the volumetric data
    void clear();

    // Resize the buffer
    void resize(int width, int height);

    // Name identifying the renderer backend
    const char* get_backend();

    // Camera instance
    Camera camera;

    // Rendering options
    RenderOptions options;

    // ---- in place of the GL framebuffer ----
    // Device images every render() starts from (both optional, camera.width x camera.height,
    // dense rows): what upstream's meshes leave in the colour and the R32F depth attachment.
    // The buffers stay the caller's; every render() copies them on the stream of the frame it
    // writes (the two frames alternate between two streams: next_stream()).  Ordering against the
    // work that PRODUCES them: pass the stream (hipStream_t) it runs on as `producer_stream` and
    // render() (a) makes its copies wait for everything enqueued there so far and (b) makes that
    // stream wait for the copies, so the producer may overwrite the images for the next frame
    // right after render() returns.  Without a producer stream the images must be complete (and
    // stay untouched) on the host's clock: synchronise before render(), and after it before
    // writing them again.  They belong to the frame size of the moment: after resize() hand
    // them in again (render() refuses a stale pair).
    void set_underlay(const void* rgba8_dev, const float* depth_dev, void* producer_stream = nullptr);
    // The frame the last render() wrote (device memory, RGBA8, width * 4 bytes per row)
    const uint8_t* frame() const;
    // Waits for the last render() and copies its frame to host memory (width * height * 4 bytes).
    // Throws if a launch reported rays cut by the sample guard (check_render_status,
    // volrend/renderer_kernel.hpp): a wrong frame is not handed out as a good one.
    void read_frame(void* host_rgba8);
    // The stream (hipStream_t) the last render() was enqueued on: frame() is complete once it is
    // idle.  The two frames alternate between two streams of their own, so that a render() issued
    // before the previous frame has been consumed starts under that launch's tail.
    void* stream() const;
    // The stream the NEXT render() will use (created on first use): for callers that order their
    // own device work against it.
    void* next_stream();

   private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace volrend
