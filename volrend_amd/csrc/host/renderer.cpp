// renderer.cpp -- volrend::launch_renderer over vr_render / vr_render_batch.
#include <stdexcept>
#include <string>

#include "volrend/renderer_kernel.hpp"

namespace volrend {
namespace {

VrRenderOptions to_c(const RenderOptions& o) {
    VrRenderOptions c;
    vr_default_options(&c);
    c.step_size = o.step_size;
    c.sigma_thresh = o.sigma_thresh;
    c.stop_thresh = o.stop_thresh;
    c.background_brightness = o.background_brightness;
    for (int i = 0; i < 6; ++i) c.render_bbox[i] = o.render_bbox[i];
    c.basis_minmax[0] = o.basis_minmax[0];
    c.basis_minmax[1] = o.basis_minmax[1];
    for (int i = 0; i < 3; ++i) {
        c.rot_dirs[i] = o.rot_dirs[i];
        c.probe[i] = o.probe[i];
    }
    c.show_grid = o.show_grid;
    c.grid_max_depth = o.grid_max_depth;
    c.render_depth = o.render_depth;
    c.enable_probe = o.enable_probe;
    c.probe_disp_size = o.probe_disp_size;
    return c;
}

VrCamera to_c(const Camera& cam, const float* transform12) {
    VrCamera c;
    for (int i = 0; i < 12; ++i) c.transform[i] = transform12[i];
    c.width = cam.width;
    c.height = cam.height;
    c.fx = cam.fx;
    c.fy = cam.fy;
    return c;
}

void check(int rc, const char* what) {
    if (rc != VR_OK) throw std::runtime_error(std::string(what) + ": " + vr_last_error());
}

}  // namespace

void launch_renderer(const N3Tree& tree, const Camera& cam, const RenderOptions& options,
                     void* image_rgba8_dev, const float* depth_dev, void* stream, bool offscreen) {
    const VrCamera c = to_c(cam, glm::value_ptr(cam.transform));
    const VrRenderOptions o = to_c(options);
    VrFrame f;
    vr_default_frame(&f);
    f.rgba = image_rgba8_dev;
    f.depth = depth_dev;
    f.offscreen = offscreen ? 1 : 0;
    check(vr_render(tree.device, &c, &o, &f, stream), "vr_render");
}

void launch_renderer_batch(const N3Tree& tree, const Camera& cam,
                           const std::vector<const float*>& transforms,
                           const RenderOptions& options, const std::vector<void*>& images,
                           void* stream, bool offscreen) {
    if (transforms.size() != images.size())
        throw std::invalid_argument("launch_renderer_batch: one image per pose");
    const VrRenderOptions o = to_c(options);
    for (size_t first = 0; first < transforms.size(); first += VR_MAX_BATCH) {
        const int n = (int)std::min<size_t>(VR_MAX_BATCH, transforms.size() - first);
        VrCamera cams[VR_MAX_BATCH];
        VrFrame frames[VR_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            cams[i] = to_c(cam, transforms[first + i]);
            vr_default_frame(&frames[i]);
            frames[i].rgba = images[first + i];
            frames[i].offscreen = offscreen ? 1 : 0;
        }
        check(vr_render_batch(tree.device, n, cams, &o, frames, stream), "vr_render_batch");
    }
}

namespace {
void throw_on_status(uint32_t status) {
    if (status != 0)
        throw std::runtime_error(
            "render status 0x" + std::to_string(status) +
            ": rays hit the sample guard (step_size too small for this scene?): the frames of "
            "these launches are wrong");
}
}  // namespace

void check_render_status(const N3Tree& tree) {
    if (!tree.device) return;
    uint32_t status = 0;
    check(vr_tree_status(tree.device, &status, 1), "vr_tree_status");
    throw_on_status(status);
}

void check_render_status(const N3Tree& tree, void* stream) {
    if (!tree.device) return;
    uint32_t status = 0;
    check(vr_tree_status_on(tree.device, &status, 1, stream), "vr_tree_status_on");
    throw_on_status(status);
}

}  // namespace volrend
