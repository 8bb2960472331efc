// volrend::N3Tree -- read-only N^3 tree loader, same public surface as the reference's
// include/volrend/n3tree.hpp:24-105.  Differences forced by the backend swap:
//   * `device` (five raw CUDA pointers) becomes an opaque vr_tree_t handle owned by the
//     HIP library (include/volrend_hip.h); is_cuda_loaded() keeps its name.
//   * the host arrays are plain NpyArray objects of the bundled npz reader
//     (volrend/internal/npz.hpp) instead of cnpy::NpyArray.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <tuple>
#include <vector>

#include "volrend/common.hpp"
#include "volrend/data_format.hpp"
#include "volrend/internal/npz.hpp"
#include "volrend/vecmath.hpp"
#include "volrend_hip.h"

namespace volrend {

struct N3Tree {
    N3Tree();
    explicit N3Tree(const std::string& path);
    ~N3Tree();
    N3Tree(const N3Tree&) = delete;
    N3Tree& operator=(const N3Tree&) = delete;

    // Open npz
    void open(const std::string& path);
    // Open memory data stream
    void open_mem(const char* data, uint64_t size);

    // Spatial branching factor. Only 2 really supported upstream; any N renders here.
    int N = 0;
    // Size of data stored on each leaf
    int data_dim = 0;
    // Data format (SH, SG etc)
    DataFormat data_format;
    // Capacity
    int capacity = 0;

    // Scaling for coordinates
    std::array<float, 3> scale{};
    // Translation
    std::array<float, 3> offset{};

    bool is_data_loaded();
    bool is_cuda_loaded();  // device copy present (name kept from the reference)

    // Clear the CPU memory (keeps child_ like the reference, src/n3tree.cpp:441-447)
    void clear_cpu_memory();

    // Index pack/unpack
    int pack_index(int nd, int i, int j, int k);
    std::tuple<int, int, int, int> unpack_index(int packed);

    // NDC config
    bool use_ndc = false;
    float ndc_width = 0, ndc_height = 0, ndc_focal = 0;
    glm::vec3 ndc_avg_up, ndc_avg_back, ndc_avg_cen;

    // Device copy (replaces `mutable struct { __half* data; ... } device`)
    vr_tree_t device = nullptr;
    // open()/open_mem() upload to the current device unless this is cleared first
    // (host-only tools and tests; the reference always uploads when built with CUDA)
    static bool upload_on_open;
    // Quantised files (scripts/compress_octree.py): decode the codebooks on the device
    // during the upload instead of the host loop of src/n3tree.cpp:310-339.  data_ then
    // stays empty until decode_quantized_host() is called.  Ignored when !upload_on_open.
    static bool device_decode;
    // Materialise data_ of a quantised tree on the host (no-op otherwise)
    void decode_quantized_host();

    // Main data holder
    internal::NpyArray data_;
    // Child link data holder
    internal::NpyArray child_;
    // Optional extra data, only used for SG/ASG
    internal::NpyArray extra_;
    // Codebook members of a quantised file awaiting the decode (empty otherwise)
    internal::NpyArray quant_colors_, quant_map_, sigma_, data_retained_;

   private:
    void load_npz(internal::NpzFile& npz);
    void load_device();  // load_cuda
    void free_device();  // free_cuda

    std::string npz_path_, poses_bounds_path_;
    bool data_loaded_ = false;
    int N2_ = 0, N3_ = 0;
    bool device_loaded_ = false;
};

}  // namespace volrend
