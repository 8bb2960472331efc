// gpu_check -- the kept C++ API on a real GPU: N3Tree::open -> volrend::launch_renderer /
// launch_renderer_batch (reference include/volrend/cuda/renderer_kernel.hpp:9-12,
// include/volrend/render_options.hpp:11-53) -> raw RGBA8 frames on stdout's file.
// Driven by tests/test_gpu_cpp_api.py, which renders the same spec with the CPU oracle.
//
//   gpu_check <tree.npz> <spec.txt> <out.raw>
// spec: "size W H FX FY", any number of "<option> v...", "mode single|batch",
//       one "pose f0 .. f11" per frame (column-major 4x3 c2w).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "volrend/n3tree.hpp"
#include "volrend/renderer_kernel.hpp"

#define HIP_OK(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            return 4;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char* argv[]) {
    using namespace volrend;
    if (argc < 4) return 2;
    try {
        N3Tree tree(argv[1]);  // open() + upload
        if (!tree.is_cuda_loaded()) return 3;
        std::ifstream spec(argv[2]);
        Camera cam;
        RenderOptions opt;
        std::string mode = "single";
        std::vector<std::vector<float>> poses;
        for (std::string line; std::getline(spec, line);) {
            std::istringstream is(line);
            std::string key;
            if (!(is >> key)) continue;
            if (key == "size") is >> cam.width >> cam.height >> cam.fx >> cam.fy;
            else if (key == "mode") is >> mode;
            else if (key == "step_size") is >> opt.step_size;
            else if (key == "sigma_thresh") is >> opt.sigma_thresh;
            else if (key == "stop_thresh") is >> opt.stop_thresh;
            else if (key == "background_brightness") is >> opt.background_brightness;
            else if (key == "render_bbox") for (float& v : opt.render_bbox) is >> v;
            else if (key == "basis_minmax") is >> opt.basis_minmax[0] >> opt.basis_minmax[1];
            else if (key == "rot_dirs") for (float& v : opt.rot_dirs) is >> v;
            else if (key == "render_depth") is >> opt.render_depth;
            else if (key == "show_grid") is >> opt.show_grid;
            else if (key == "grid_max_depth") is >> opt.grid_max_depth;
            else if (key == "enable_probe") is >> opt.enable_probe;
            else if (key == "probe") for (float& v : opt.probe) is >> v;
            else if (key == "probe_disp_size") is >> opt.probe_disp_size;
            else if (key == "pose") {
                std::vector<float> p(12);
                for (float& v : p) is >> v;
                poses.push_back(p);
            } else {
                fprintf(stderr, "unknown spec key %s\n", key.c_str());
                return 2;
            }
        }
        const size_t frame_bytes = (size_t)cam.width * cam.height * 4;
        uint8_t* dev = nullptr;
        HIP_OK(hipMalloc((void**)&dev, frame_bytes * poses.size()));
        HIP_OK(hipMemset(dev, 0, frame_bytes * poses.size()));
        hipStream_t stream;
        HIP_OK(hipStreamCreate(&stream));
        if (mode == "batch") {
            std::vector<const float*> tr;
            std::vector<void*> imgs;
            for (size_t i = 0; i < poses.size(); ++i) {
                tr.push_back(poses[i].data());
                imgs.push_back(dev + frame_bytes * i);
            }
            launch_renderer_batch(tree, cam, tr, opt, imgs, stream, true);
        } else {
            for (size_t i = 0; i < poses.size(); ++i) {
                for (int k = 0; k < 12; ++k) cam.transform[k / 3][k % 3] = poses[i][k];
                launch_renderer(tree, cam, opt, dev + frame_bytes * i, nullptr, stream, true);
            }
        }
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<uint8_t> host(frame_bytes * poses.size());
        HIP_OK(hipMemcpy(host.data(), dev, host.size(), hipMemcpyDeviceToHost));
        FILE* fp = fopen(argv[3], "wb");
        if (!fp || fwrite(host.data(), 1, host.size(), fp) != host.size()) return 5;
        fclose(fp);
        HIP_OK(hipFree(dev));
        HIP_OK(hipStreamDestroy(stream));
        printf("frames=%zu w=%d h=%d\n", poses.size(), cam.width, cam.height);
    } catch (const std::exception& e) {
        printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    return 0;
}
