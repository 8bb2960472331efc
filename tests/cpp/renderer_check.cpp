// renderer_check -- volrend::VolumeRenderer (reference include/volrend/renderer.hpp:11-42,
// src/cuda_renderer.cpp:83-195) on a real GPU, without OpenGL.  Driven by
// tests/test_gpu_renderer.py, which renders the same thing with the CPU oracle's
// compositing path (offscreen = 0).
//
//   renderer_check <tree.npz> <spec.txt> <out.raw>
// spec: "size W H FX FY", "background_brightness b", "step_size s", "burst n" (render() n times per camera),
//       "loop n" (afterwards: n render() calls back to back, timed -> "loop_ms_per_frame x"),
//       "underlay <rgba.raw> <depth.raw>" (optional), "underlay_stream 1" (the underlay is PRODUCED on a stream
//       of the caller's right before every render() and overwritten with junk right after it, no host
//       synchronisation in between: set_underlay's producer_stream must order both),
//       one "cam cx cy cz bx by bz" per frame (camera centre and v_back; v_world_up stays +z).
// stdout: one "transform f0 .. f11" line per frame (what Camera::_update made of the vectors)
//         and "basis_minmax a b backend NAME".
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "volrend/renderer.hpp"

static std::vector<char> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char* argv[]) {
    using namespace volrend;
    if (argc < 4) return 2;
    try {
        N3Tree tree(argv[1]);
        VolumeRenderer r;
        if (r.frame() != nullptr) return 6;  // nothing rendered yet
        std::ifstream spec(argv[2]);
        int w = 0, h = 0, burst = 1, loop = 0, under_stream = 0;
        std::string under_rgba, under_depth;
        std::vector<std::vector<float>> cams;
        for (std::string line; std::getline(spec, line);) {
            std::istringstream is(line);
            std::string key;
            if (!(is >> key)) continue;
            if (key == "size") is >> w >> h >> r.camera.fx >> r.camera.fy;
            else if (key == "background_brightness") is >> r.options.background_brightness;
            else if (key == "step_size") is >> r.options.step_size;
            else if (key == "burst") is >> burst;
            else if (key == "loop") is >> loop;
            else if (key == "underlay") is >> under_rgba >> under_depth;
            else if (key == "underlay_stream") is >> under_stream;
            else if (key == "cam") {
                std::vector<float> c(6);
                for (float& v : c) is >> v;
                cams.push_back(c);
            } else return 2;
        }
        r.resize(w, h);
        if (r.camera.width != w || r.camera.height != h) return 7;
        r.options.basis_minmax[1] = 24;
        r.set(tree);  // must narrow basis_minmax to the tree's basis (cuda_renderer.cpp:176-177)
        void *d_rgba = nullptr, *d_depth = nullptr, *d_true_rgba = nullptr, *d_true_depth = nullptr;
        hipStream_t producer = nullptr;
        const size_t img_bytes = (size_t)w * h * 4;
        if (!under_rgba.empty()) {
            const std::vector<char> a = slurp(under_rgba), b = slurp(under_depth);
            if (a.size() != img_bytes || b.size() != img_bytes) return 8;
            if (hipMalloc(&d_rgba, a.size()) != hipSuccess || hipMalloc(&d_depth, b.size()) != hipSuccess) return 4;
            if (under_stream) {
                // the images a mesh pass would render: kept aside, copied into the underlay buffers
                // ON THE PRODUCER STREAM before every render() and trashed there right after it
                if (hipStreamCreate(&producer) != hipSuccess) return 4;
                if (hipMalloc(&d_true_rgba, a.size()) != hipSuccess || hipMalloc(&d_true_depth, b.size()) != hipSuccess) return 4;
                if (hipMemcpy(d_true_rgba, a.data(), a.size(), hipMemcpyHostToDevice) != hipSuccess) return 4;
                if (hipMemcpy(d_true_depth, b.data(), b.size(), hipMemcpyHostToDevice) != hipSuccess) return 4;
                if (hipMemset(d_rgba, 0x5A, img_bytes) != hipSuccess || hipMemset(d_depth, 0, img_bytes) != hipSuccess) return 4;
                r.set_underlay(d_rgba, (const float*)d_depth, producer);
                if (r.next_stream() == nullptr) return 9;
            } else {
                if (hipMemcpy(d_rgba, a.data(), a.size(), hipMemcpyHostToDevice) != hipSuccess) return 4;
                if (hipMemcpy(d_depth, b.data(), b.size(), hipMemcpyHostToDevice) != hipSuccess) return 4;
                r.set_underlay(d_rgba, (const float*)d_depth);
            }
        }
        auto render_once = [&]() {
            if (producer) {
                if (r.next_stream() == r.stream() && r.frame() != nullptr) throw std::runtime_error("frames do not alternate streams");
                (void)hipMemcpyAsync(d_rgba, d_true_rgba, img_bytes, hipMemcpyDeviceToDevice, producer);
                (void)hipMemcpyAsync(d_depth, d_true_depth, img_bytes, hipMemcpyDeviceToDevice, producer);
            }
            r.render();
            if (producer) {  // junk: a depth of 0 would end every ray at once
                (void)hipMemsetAsync(d_rgba, 0x5A, img_bytes, producer);
                (void)hipMemsetAsync(d_depth, 0, img_bytes, producer);
            }
        };
        std::vector<uint8_t> host((size_t)w * h * 4 * (cams.size() + 1));
        for (size_t i = 0; i < cams.size(); ++i) {
            r.camera.center = glm::vec3(cams[i][0], cams[i][1], cams[i][2]);
            r.camera.v_back = glm::vec3(cams[i][3], cams[i][4], cams[i][5]);
            // burst > 1: render() again before the previous frame is consumed -- the two frames
            // alternate between two streams, consecutive launches overlap; the LAST frame is read
            for (int b = 0; b < burst; ++b) render_once();
            r.read_frame(host.data() + (size_t)w * h * 4 * i);
            const float* t = glm::value_ptr(r.camera.transform);
            printf("transform");
            for (int k = 0; k < 12; ++k) printf(" %.9g", t[k]);
            printf("\n");
        }
        if (loop > 0 && !cams.empty()) {
            // a render() loop as an interactive caller runs it: `loop` frames, the camera stepping
            // through the given ones, the frame consumed only at the end (tools/cli_bench.py)
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < loop; ++k) {
                const std::vector<float>& c = cams[(size_t)k % cams.size()];
                r.camera.center = glm::vec3(c[0], c[1], c[2]);
                r.camera.v_back = glm::vec3(c[3], c[4], c[5]);
                r.render();
            }
            (void)hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("loop_ms_per_frame %.5f\n", ms / loop);
        }
        // without a tree render() leaves the cleared frame / the underlay (cuda_renderer.cpp:114)
        r.clear();
        render_once();
        r.read_frame(host.data() + (size_t)w * h * 4 * cams.size());
        FILE* fp = fopen(argv[3], "wb");
        if (!fp || fwrite(host.data(), 1, host.size(), fp) != host.size()) return 5;
        fclose(fp);
        printf("basis_minmax %d %d backend %s\n", r.options.basis_minmax[0], r.options.basis_minmax[1],
               r.get_backend());
        (void)hipDeviceSynchronize();
        for (void* q : {d_rgba, d_depth, d_true_rgba, d_true_depth})
            if (q) (void)hipFree(q);
        if (producer) (void)hipStreamDestroy(producer);
    } catch (const std::exception& e) {
        printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    return 0;
}
