/*
 * ref_driver.cpp -- appended AFTER the reference's src/cuda/volrend.cu in one
 * translation unit (see Makefile), so the reference's own static device
 * functions (device::render_kernel, device::trace_ray, screen2worlddir, ...)
 * are callable.  This file contains no restated algorithm: it only builds the
 * reference's argument structs and runs its kernel body once per pixel.
 * ORACLE / test infrastructure only.
 */
#include <atomic>
#include <thread>
#include <vector>

namespace volrend {
/* Stubs for members whose definitions live in reference TUs we do not build
 * (src/n3tree.cpp, src/camera.cpp, src/cuda/common.cu). */
N3Tree::N3Tree() {}
N3Tree::~N3Tree() {}
struct Camera::DragState {};
Camera::Camera(int width, int height, float fx, float fy)
    : width(width), height(height), fx(fx), fy(fy) {}
Camera::~Camera() {}
cudaError_t cuda_assert(const cudaError_t code, const char* const file, const int line,
                        const bool abort) {
    (void)file; (void)line; (void)abort;
    return code;
}
}  // namespace volrend

extern "C" {

struct RefTree {
    const int32_t* child;
    const uint16_t* data;
    const float* extra;
    float offset[3];
    float scale[3];
    int32_t N;
    int32_t data_dim;
    int32_t format;
    int32_t basis_dim;
    float ndc_width, ndc_height, ndc_focal;
};
struct RefCamera {
    float transform[12];
    int32_t width, height;
    float fx, fy;
};
struct RefOptions {
    float step_size, sigma_thresh, stop_thresh, background_brightness;
    float render_bbox[6];
    int32_t basis_minmax[2];
    float rot_dirs[3];
    int32_t show_grid, grid_max_depth, render_depth, enable_probe;
    float probe[3];
    int32_t probe_disp_size;
};

}  // extern "C"

namespace {
using namespace volrend;

void fill_tree(N3Tree& t, const RefTree* d) {
    t.N = d->N;
    t.data_dim = d->data_dim;
    t.data_format.format = static_cast<decltype(t.data_format.format)>(d->format);
    t.data_format.basis_dim = d->basis_dim;
    for (int i = 0; i < 3; ++i) {
        t.offset[i] = d->offset[i];
        t.scale[i] = d->scale[i];
    }
    t.use_ndc = d->ndc_width > 0;
    t.ndc_width = d->ndc_width;
    t.ndc_height = d->ndc_height;
    t.ndc_focal = d->ndc_focal;
    /* host memory plays the role of device memory */
    t.device.data = reinterpret_cast<__half*>(const_cast<uint16_t*>(d->data));
    t.device.child = const_cast<int32_t*>(d->child);
    t.device.offset = t.offset.data();
    t.device.scale = t.scale.data();
    t.device.extra = const_cast<float*>(d->extra);
}

RenderOptions make_opts(const RefOptions* o) {
    RenderOptions r;
    r.step_size = o->step_size;
    r.sigma_thresh = o->sigma_thresh;
    r.stop_thresh = o->stop_thresh;
    r.background_brightness = o->background_brightness;
    for (int i = 0; i < 6; ++i) r.render_bbox[i] = o->render_bbox[i];
    r.basis_minmax[0] = o->basis_minmax[0];
    r.basis_minmax[1] = o->basis_minmax[1];
    for (int i = 0; i < 3; ++i) r.rot_dirs[i] = o->rot_dirs[i];
    r.show_grid = o->show_grid != 0;
    r.grid_max_depth = o->grid_max_depth;
    r.render_depth = o->render_depth != 0;
    r.enable_probe = o->enable_probe != 0;
    for (int i = 0; i < 3; ++i) r.probe[i] = o->probe[i];
    r.probe_disp_size = o->probe_disp_size;
    return r;
}
}  // namespace

extern "C" {

/* Runs the reference's render_kernel (volrend.cu:78-173) for every pixel of the
 * rectangle.  rgba is the full frame (pitch width*4) and, for offscreen=0, must
 * already hold the colour to composite over; depth is the R32F mesh depth. */
int ref_render(const RefTree* td, const RefCamera* cd, const RefOptions* od, int offscreen,
               int x0, int y0, int w, int h, uint8_t* rgba, float* depth,
               const float* probe_coeffs, int nthreads) {
    N3Tree tree;
    fill_tree(tree, td);
    Camera cam(cd->width, cd->height, cd->fx, cd->fy);
    float xf[12];
    for (int i = 0; i < 12; ++i) xf[i] = cd->transform[i];
    cam.device.transform = xf;
    const RenderOptions opt = make_opts(od);

    VrShimSurface img{rgba, (size_t)cd->width * 4, cd->width * 4, cd->height};
    VrShimSurface dep{reinterpret_cast<uint8_t*>(depth), (size_t)cd->width * 4, cd->width * 4,
                      cd->height};
    // work items = spans of 64 pixels of the rectangle (rows differ a lot in cost: a row per
    // item leaves most of a many-core host idle at the end of the frame)
    const long long n_px = (long long)w * h;
    std::atomic<long long> next{0};
    auto work = [&]() {
        blockDim.x = 1;
        threadIdx.x = 0;
        for (;;) {
            const long long first = next.fetch_add(64);
            if (first >= n_px) break;
            const long long last = first + 64 < n_px ? first + 64 : n_px;
            for (long long i = first; i < last; ++i) {
                const int row = (int)(i / w), x = x0 + (int)(i % w);
                blockIdx.x = (unsigned)((y0 + row) * cd->width + x);
                device::render_kernel(&img, depth ? &dep : nullptr, internal::CameraSpec(cam),
                                      internal::TreeSpec(tree), opt,
                                      const_cast<float*>(probe_coeffs), offscreen != 0);
            }
        }
    };
    if (nthreads <= 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    cam.device.transform = nullptr;
    return 0;
}

/* The fp32 accumulators: the reference's own call sequence
 * volrend.cu:136-150 (screen2worlddir .. trace_ray) with out[] exposed. */
int ref_trace(const RefTree* td, const RefCamera* cd, const RefOptions* od, int x0, int y0,
              int w, int h, float* accum /* full frame, 4 floats per pixel */) {
    N3Tree tree;
    fill_tree(tree, td);
    Camera cam(cd->width, cd->height, cd->fx, cd->fy);
    float xf[12];
    for (int i = 0; i < 12; ++i) xf[i] = cd->transform[i];
    cam.device.transform = xf;
    const RenderOptions opt = make_opts(od);
    const internal::CameraSpec cs(cam);
    const internal::TreeSpec ts(tree);
    for (int y = y0; y < y0 + h; ++y)
        for (int x = x0; x < x0 + w; ++x) {
            float dir[3], cen[3], out[4] = {0.f, 0.f, 0.f, 0.f};
            screen2worlddir(x, y, cs, dir, cen);
            float vdir[3] = {dir[0], dir[1], dir[2]};
            maybe_world2ndc(ts, dir, cen);
            for (int i = 0; i < 3; ++i) cen[i] = ts.offset[i] + ts.scale[i] * cen[i];
            rodrigues(opt.rot_dirs, vdir);
            device::trace_ray(ts, dir, vdir, cen, opt, 1e9f, out);
            float* o = accum + 4 * ((size_t)y * cd->width + x);
            o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; o[3] = out[3];
        }
    cam.device.transform = nullptr;
    return 0;
}

/* retrieve_cursor_lumisphere_kernel, volrend.cu:175-191 */
int ref_probe_coeffs(const RefTree* td, const RefOptions* od, float* out) {
    N3Tree tree;
    fill_tree(tree, td);
    device::retrieve_cursor_lumisphere_kernel(internal::TreeSpec(tree), make_opts(od), out);
    return 0;
}

/* query_single_from_root, n3tree_query.hpp:13-48 */
int64_t ref_query(const RefTree* td, float xyz[3], float* cube_sz) {
    N3Tree tree;
    fill_tree(tree, td);
    const internal::TreeSpec ts(tree);
    const half* val;
    internal::query_single_from_root(ts, xyz, &val, cube_sz);
    return (int64_t)((val - ts.data) / ts.data_dim);
}

/* maybe_precalc_basis, lumisphere.hpp:9-87 */
int ref_basis(const RefTree* td, const float dir[3], float out[25]) {
    N3Tree tree;
    fill_tree(tree, td);
    internal::maybe_precalc_basis(internal::TreeSpec(tree), dir, out);
    return 0;
}

}  // extern "C"
