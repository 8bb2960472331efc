// volume_renderer.cpp -- volrend::VolumeRenderer over a linear device frame (no OpenGL);
// see include/volrend/renderer.hpp.  Follows src/cuda_renderer.cpp:83-195 step for step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "volrend/renderer.hpp"
#include "volrend/renderer_kernel.hpp"

namespace volrend {
namespace {
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("VolumeRenderer: ") + what + ": " + hipGetErrorString(e));
}
// The renderer's stream and frames live on the device of the tree it renders (launch_renderer
// runs there whatever the calling thread's current device is); the thread's device is restored.
class OnDevice {
   public:
    explicit OnDevice(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev_) == hipSuccess && prev_ != device)
            switched_ = hipSetDevice(device) == hipSuccess;
    }
    ~OnDevice() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    OnDevice(const OnDevice&) = delete;
    OnDevice& operator=(const OnDevice&) = delete;

   private:
    int prev_ = 0;
    bool switched_ = false;
};
}  // namespace

struct VolumeRenderer::Impl {
    const N3Tree* tree = nullptr;
    // Two frames, as upstream's two framebuffers (cuda_renderer.cpp:210-214): render() writes
    // one while the consumer may still read the other -- and each frame has its OWN stream, so
    // that a loop that calls render() again before it consumes the previous frame gets the
    // next launch's start under the previous launch's tail (a lone frame drains for ~0.3 ms
    // while its longest rays finish: one frame per launch runs at 0.40 instead of 0.58 ms on
    // two alternating streams, profiles/r05_stream_overlap.jsonl).
    hipStream_t streams[2] = {nullptr, nullptr};  // (both or none)
    uint8_t* rgba[2] = {nullptr, nullptr};
    float* depth[2] = {nullptr, nullptr};
    int buf_index = 0, last = -1;
    int width = 0, height = 0;  // size of the allocations
    const void* under_rgba = nullptr;
    const float* under_depth = nullptr;
    int under_w = 0, under_h = 0;  // the size the underlay buffers were handed in for
    hipStream_t under_stream = nullptr;  // where the underlay is produced (nullptr: ordered by the host)
    bool under_ordered = false;          // (set_underlay was given a producer stream)
    hipEvent_t under_ready = nullptr, under_taken = nullptr;
    int device = -1;               // where stream and frames live (-1: nothing created yet)

    // The tree's device (the current one while no tree is set).  Moving to another device drops
    // the stream and the frames: they are re-created there.
    int target_device() const {
        int d = 0;
        if (tree && tree->device) {
            VrTreeInfo info;
            if (vr_tree_info(tree->device, &info) == VR_OK) return info.device;
        }
        (void)hipGetDevice(&d);
        return d;
    }
    void start() {  // cuda_renderer.cpp:59-81 without the GL objects
        const int want = target_device();
        if (streams[0] && device != want) {
            OnDevice on(device);
            sync_all();
            release();
            for (auto& st : streams) {
                (void)hipStreamDestroy(st);
                st = nullptr;
            }
        }
        if (!streams[0]) {
            OnDevice on(want);
            for (auto& st : streams) hip_check(hipStreamCreate(&st), "hipStreamCreate");
            device = want;
        }
    }
    void sync_all() {
        for (auto st : streams)
            if (st) (void)hipStreamSynchronize(st);
    }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (rgba[i]) (void)hipFree(rgba[i]);
            if (depth[i]) (void)hipFree(depth[i]);
            rgba[i] = nullptr;
            depth[i] = nullptr;
        }
        width = height = 0;
        last = -1;
    }
    void allocate(int w, int h) {
        if (w == width && h == height) return;
        sync_all();
        release();
        if (w <= 0 || h <= 0) return;
        for (int i = 0; i < 2; ++i) {
            hip_check(hipMalloc((void**)&rgba[i], (size_t)w * h * 4), "hipMalloc(frame)");
            hip_check(hipMalloc((void**)&depth[i], (size_t)w * h * 4), "hipMalloc(depth)");
        }
        width = w;
        height = h;
    }
    ~Impl() {
        OnDevice on(device);
        sync_all();
        release();
        for (auto st : streams)
            if (st) (void)hipStreamDestroy(st);
        if (under_ready) (void)hipEventDestroy(under_ready);
        if (under_taken) (void)hipEventDestroy(under_taken);
    }
};

VolumeRenderer::VolumeRenderer() : impl_(std::make_unique<Impl>()) {}
VolumeRenderer::~VolumeRenderer() = default;

void VolumeRenderer::render() {
    Impl& m = *impl_;
    m.start();
    OnDevice on(m.device);
    m.allocate(camera.width, camera.height);
    if (m.width <= 0) return;
    if ((m.under_rgba || m.under_depth) && (m.under_w != m.width || m.under_h != m.height))
        throw std::runtime_error("VolumeRenderer::render: the underlay was set for " +
                                 std::to_string(m.under_w) + "x" + std::to_string(m.under_h) +
                                 ", the frame is " + std::to_string(m.width) + "x" +
                                 std::to_string(m.height) + " (set_underlay again after resize)");
    const size_t px = (size_t)m.width * m.height;
    uint8_t* frame = m.rgba[m.buf_index];
    float* depth = m.depth[m.buf_index];
    hipStream_t stream = m.streams[m.buf_index];  // the frame's own stream (see Impl)
    const bool ordered = m.under_ordered && (m.under_rgba || m.under_depth);
    if (ordered) {  // the copies below wait for what the producer has enqueued so far
        if (!m.under_ready) {
            hip_check(hipEventCreateWithFlags(&m.under_ready, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&m.under_taken, hipEventDisableTiming), "hipEventCreate");
        }
        hip_check(hipEventRecord(m.under_ready, m.under_stream), "hipEventRecord(underlay ready)");
        hip_check(hipStreamWaitEvent(stream, m.under_ready, 0), "hipStreamWaitEvent(underlay ready)");
    }
    // glClearNamedFramebufferfv: colour = (b, b, b, 1) converted to RGBA8 the GL way
    // (round(clamp(b, 0, 1) * 255)), depth attachment = 1e9 (cuda_renderer.cpp:85-92)
    if (m.under_rgba) {
        hip_check(hipMemcpyAsync(frame, m.under_rgba, px * 4, hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync(underlay colour)");
    } else {
        const float b = std::min(std::max(options.background_brightness, 0.f), 1.f);
        const uint32_t c = (uint32_t)std::lround(b * 255.f);
        hip_check(hipMemsetD32Async((hipDeviceptr_t)frame, (int)(c | c << 8 | c << 16 | 0xFF000000u),
                                    px, stream), "hipMemsetD32Async(frame)");
    }
    if (m.under_depth) {
        hip_check(hipMemcpyAsync(depth, m.under_depth, px * 4, hipMemcpyDeviceToDevice, stream),
                  "hipMemcpyAsync(underlay depth)");
    } else {
        const float inf = 1e9f;
        uint32_t bits;
        static_assert(sizeof(bits) == sizeof(inf), "");
        __builtin_memcpy(&bits, &inf, 4);
        hip_check(hipMemsetD32Async((hipDeviceptr_t)depth, (int)bits, px, stream),
                  "hipMemsetD32Async(depth)");
    }
    if (ordered) {  // ... and the producer's next writes wait for the copies
        hip_check(hipEventRecord(m.under_taken, stream), "hipEventRecord(underlay taken)");
        hip_check(hipStreamWaitEvent(m.under_stream, m.under_taken, 0), "hipStreamWaitEvent(underlay taken)");
    }
    camera._update();  // cuda_renderer.cpp:97
    if (m.tree != nullptr)  // cuda_renderer.cpp:114-120: the interactive path composites (offscreen = false)
        launch_renderer(*m.tree, camera, options, frame, depth, stream, false);
    m.last = m.buf_index;
    m.buf_index ^= 1;
}

void VolumeRenderer::set(N3Tree& tree) {  // cuda_renderer.cpp:171-180
    if (!tree.is_cuda_loaded())
        throw std::runtime_error("VolumeRenderer::set: the tree is not on the device (N3Tree::open uploads it)");
    impl_->tree = &tree;
    impl_->start();  // (moves the stream and the frames to the tree's device if need be)
    options.basis_minmax[0] = 0;
    options.basis_minmax[1] = std::max(tree.data_format.basis_dim - 1, 0);
}

void VolumeRenderer::clear() { impl_->tree = nullptr; }  // cuda_renderer.cpp:220

void VolumeRenderer::resize(int width, int height) {  // cuda_renderer.cpp:128-169
    if (camera.width == width && camera.height == height && impl_->width == width &&
        impl_->height == height)
        return;
    impl_->start();
    OnDevice on(impl_->device);
    camera.width = width;
    camera.height = height;
    impl_->allocate(width, height);
}

const char* VolumeRenderer::get_backend() { return "HIP"; }  // upstream: "CUDA" (cuda_renderer.cpp:225)

void VolumeRenderer::set_underlay(const void* rgba8_dev, const float* depth_dev, void* producer_stream) {
    impl_->under_rgba = rgba8_dev;
    impl_->under_depth = depth_dev;
    impl_->under_stream = static_cast<hipStream_t>(producer_stream);
    impl_->under_ordered = producer_stream != nullptr;
    impl_->under_w = camera.width;
    impl_->under_h = camera.height;
}

const uint8_t* VolumeRenderer::frame() const {
    return impl_->last < 0 ? nullptr : impl_->rgba[impl_->last];
}

void VolumeRenderer::read_frame(void* host_rgba8) {
    Impl& m = *impl_;
    if (m.last < 0) throw std::runtime_error("VolumeRenderer::read_frame: nothing rendered yet");
    OnDevice on(m.device);
    hip_check(hipMemcpyAsync(host_rgba8, m.rgba[m.last], (size_t)m.width * m.height * 4,
                             hipMemcpyDeviceToHost, m.streams[m.last]), "hipMemcpyAsync(read_frame)");
    // what the launches found out on the device (the sample guard) surfaces here, loudly, instead of
    // a wrong frame handed out as a good one -- read on THIS frame's stream, behind its copy: the
    // other frame's launch, if one is in flight, is not waited for
    if (m.tree)
        check_render_status(*m.tree, m.streams[m.last]);
    else
        hip_check(hipStreamSynchronize(m.streams[m.last]), "hipStreamSynchronize");
}

void* VolumeRenderer::next_stream() {
    impl_->start();
    return impl_->streams[impl_->buf_index];
}

void* VolumeRenderer::stream() const {  // the stream of the frame frame() names
    return impl_->last < 0 ? impl_->streams[0] : impl_->streams[impl_->last];
}

}  // namespace volrend
