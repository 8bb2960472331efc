// n3tree.cpp -- volrend::N3Tree over the HIP C ABI.  Loader semantics follow the
// reference's src/n3tree.cpp:111-362 (keys, legacy inference, quantised codebooks, LLFF
// NDC sidecar); the device side is vr_tree_upload / vr_tree_free instead of
// N3Tree::load_cuda / free_cuda (src/cuda/n3tree.cu:9-49).
#include "volrend/n3tree.hpp"

#include <cstdio>
#include <cstring>
#include <chrono>
#include <fstream>
#include <stdexcept>

namespace volrend {
namespace {

// DataFormat.  parse: leading letters name the basis, the rest is its dimension.
}  // namespace

void DataFormat::parse(const std::string& str) {
    size_t k = 0;
    while (k < str.size() && std::isalpha((unsigned char)str[k])) ++k;
    if (k == str.size()) {
        basis_dim = -1;
        format = RGBA;
        return;
    }
    basis_dim = std::atoi(str.c_str() + k);
    const std::string head = str.substr(0, k);
    format = head == "ASG" ? ASG : head == "SG" ? SG : head == "SH" ? SH : RGBA;
}

std::string DataFormat::to_string() const {
    static const char* names[] = {"RGBA", "SH", "SG", "ASG"};
    std::string out = (int)format < 4 ? names[(int)format] : "UNKNOWN";
    if (basis_dim != -1) out += std::to_string(basis_dim);
    return out;
}

namespace {

// LLFF poses_bounds.npy: mean pose & image geometry (reference src/n3tree.cpp:20-52)
void unpack_llff_poses_bounds(const internal::NpyArray& pb, float& width, float& height,
                              float& focal, glm::vec3& up, glm::vec3& backward, glm::vec3& cen) {
    height = (float)pb.as_double(4);
    width = (float)pb.as_double(9);
    focal = (float)pb.as_double(14);
    cen = glm::vec3(0.f);
    backward = glm::vec3(0.f);
    up = glm::vec3(0.f);
    glm::vec3 right(0.f);
    const size_t block = 17;
    float bd_min = 1e9f;
    for (size_t off = 0; off + block <= pb.num_vals; off += block) {
        for (int r = 0; r < 3; ++r) {
            right[r] += (float)pb.as_double(off + 5 * r + 1);
            up[r] -= (float)pb.as_double(off + 5 * r + 0);
            backward[r] += (float)pb.as_double(off + 5 * r + 2);
            cen[r] += (float)pb.as_double(off + 5 * r + 3);
        }
        bd_min = std::min(bd_min, (float)std::min(pb.as_double(off + 15), pb.as_double(off + 16)));
    }
    const size_t total = pb.num_vals / block;
    cen = cen / ((float)total * bd_min * 0.75f);
    backward = glm::normalize(backward);
    right = glm::normalize(glm::cross(up, backward));
    up = glm::normalize(glm::cross(backward, right));
}

}  // namespace

bool N3Tree::upload_on_open = true;
bool N3Tree::device_decode = true;

N3Tree::N3Tree() {}
N3Tree::N3Tree(const std::string& path) { open(path); }
N3Tree::~N3Tree() { free_device(); }

void N3Tree::open(const std::string& path) {
    clear_cpu_memory();
    data_loaded_ = false;
    npz_path_ = path;
    if (path.size() <= 4 || path.substr(path.size() - 4) != ".npz")
        throw std::runtime_error("N3Tree::open: expected a .npz file: " + path);
    poses_bounds_path_ = path.substr(0, path.size() - 4) + "_poses_bounds.npy";
    if (!std::ifstream(path)) {
        printf("Can't load because file does not exist: %s\n", path.c_str());
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    internal::NpzFile npz = internal::npz_load(path);
    load_npz(npz);
    const auto t1 = std::chrono::steady_clock::now();

    use_ndc = bool(std::ifstream(poses_bounds_path_));
    if (use_ndc) {
        fprintf(stderr, "INFO: Found poses_bounds.npy for NDC: %s\n", poses_bounds_path_.c_str());
        const internal::NpyArray pb = internal::npy_load(poses_bounds_path_);
        unpack_llff_poses_bounds(pb, ndc_width, ndc_height, ndc_focal, ndc_avg_up, ndc_avg_back,
                                 ndc_avg_cen);
    }
    if (upload_on_open) {
        const auto t2 = std::chrono::steady_clock::now();
        load_device();
        const auto t3 = std::chrono::steady_clock::now();
        fprintf(stderr, "INFO: tree ready: npz load%s %.1f ms, device upload%s %.1f ms\n",
                !quant_map_.empty() && !data_.empty() ? " + host codebook decode" : "",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                !quant_map_.empty() && data_.empty() ? " + device codebook decode" : "",
                std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
    data_loaded_ = true;
}

void N3Tree::open_mem(const char* data, uint64_t size) {
    data_loaded_ = false;
    clear_cpu_memory();
    npz_path_.clear();
    internal::NpzFile npz = internal::npz_load_mem(reinterpret_cast<const uint8_t*>(data), size);
    load_npz(npz);
    use_ndc = false;
    if (upload_on_open) load_device();
    data_loaded_ = true;
}

void N3Tree::load_npz(internal::NpzFile& npz) {
    auto need = [&](const char* key) -> internal::NpyArray& {
        auto it = npz.find(key);
        if (it == npz.end()) throw std::runtime_error(std::string("tree.npz lacks '") + key + "'");
        return it->second;
    };
    data_dim = (int)need("data_dim").as_double();
    if (npz.count("data_format")) {
        data_format.parse(npz["data_format"].as_string());
    } else if (data_dim == 4) {  // legacy files
        data_format.format = DataFormat::RGBA;
        data_format.basis_dim = -1;
        fprintf(stderr, "INFO: Legacy file with no format specifier; spherical basis disabled\n");
    } else {
        data_format.format = DataFormat::SH;
        data_format.basis_dim = (data_dim - 1) / 3;
        fprintf(stderr,
                "INFO: Legacy file with no format specifier; autodetect spherical harmonics order\n");
    }
    fprintf(stderr, "INFO: Data format %s\n", data_format.to_string().c_str());

    if (npz.count("invradius3")) {
        const internal::NpyArray& a = npz["invradius3"];
        if (a.num_vals < 3) throw std::runtime_error("invradius3 must hold 3 values");
        for (int i = 0; i < 3; ++i) scale[i] = (float)a.as_double(i);
    } else {
        scale[0] = scale[1] = scale[2] = (float)need("invradius").as_double();
    }
    printf("INFO: Scale %f %f %f\n", scale[0], scale[1], scale[2]);
    {
        const internal::NpyArray& a = need("offset");
        if (a.num_vals < 3) throw std::runtime_error("offset must hold 3 values");
        for (int i = 0; i < 3; ++i) offset[i] = (float)a.as_double(i);
    }

    child_ = std::move(need("child"));
    if (child_.kind != 'i' || child_.word_size != 4 || child_.shape.size() != 4)
        throw std::runtime_error("child must be int32 [capacity, N, N, N]");
    if (child_.shape[1] < 2 || child_.shape[1] > 16 || child_.shape[2] != child_.shape[1] ||
        child_.shape[3] != child_.shape[1] || child_.shape[0] == 0 || child_.shape[0] > 0x7FFFFFFFu)
        throw std::runtime_error("child must be int32 [capacity, N, N, N] with 2 <= N <= 16");
    N = (int)child_.shape[1];
    if (N != 2) fprintf(stderr, "WARNING: N != 2 is rendered by the generic (slower) kernel.\n");
    N2_ = N * N;
    N3_ = N * N * N;

    if (npz.count("quant_colors")) {
        // median-cut codebooks (scripts/compress_octree.py:106-119), decode as
        // reference src/n3tree.cpp:279-340
        fprintf(stderr, "INFO: Decoding quantized colors\n");
        const internal::NpyArray& qc = npz["quant_colors"];
        if (qc.word_size != 2) throw std::runtime_error("codebook must be stored in half precision");
        const internal::NpyArray& qm = need("quant_map");
        // ranks first: nothing below may index a shape that is not there
        if (qm.shape.size() != 5 || qc.shape.size() != 3 || qc.shape[1] != 65536 ||
            qc.shape[2] != 3)
            throw std::runtime_error("quant_map / quant_colors have unexpected shapes");
        if (qm.shape[1] == 0 || qm.shape[1] > 0x7FFFFFFFu)
            throw std::runtime_error("quant_map capacity out of range");
        capacity = (int)qm.shape[1];
        const size_t n_q = qm.shape[0];
        if (qc.shape[0] != n_q) throw std::runtime_error("codebook and map basis numbers does not match");
        need("sigma");
        quant_colors_ = std::move(npz["quant_colors"]);
        quant_map_ = std::move(npz["quant_map"]);
        sigma_ = std::move(npz["sigma"]);
        data_retained_ = npz.count("data_retained") ? std::move(npz["data_retained"])
                                                    : internal::NpyArray();
        if (sigma_.word_size != 2 || quant_map_.word_size != 2 ||
            (!data_retained_.empty() && data_retained_.word_size != 2))
            throw std::runtime_error("quantised arrays must be 16-bit");
        if (!data_retained_.empty() && data_retained_.shape.size() != 6)
            throw std::runtime_error("data_retained must be [n_retained, capacity, N, N, N, 3]");
        data_ = internal::NpyArray();
        {  // every codebook array is indexed per slot: check before anything reads them
            const size_t n_slots = (size_t)capacity * N3_;
            const size_t n_ret = data_retained_.empty() ? 0 : data_retained_.shape[0];
            if ((size_t)capacity != child_.shape[0] || data_dim < 1 || data_dim > 4096 ||
                quant_map_.num_vals != n_q * n_slots || sigma_.num_vals != n_slots ||
                quant_colors_.num_vals != n_q * 65536 * 3 ||
                (n_ret && data_retained_.num_vals != n_ret * n_slots * 3) ||
                3 * (n_q + n_ret) + 1 > (size_t)data_dim)
                throw std::runtime_error("quantised arrays do not match capacity / data_dim");
        }
        if (!(device_decode && upload_on_open)) decode_quantized_host();
    } else {
        internal::NpyArray& d = need("data");
        if (d.shape.size() != 5 || d.shape[0] == 0 || d.shape[0] > 0x7FFFFFFFu)
            throw std::runtime_error("data must be float16 [capacity, N, N, N, data_dim]");
        capacity = (int)d.shape[0];
        if (d.word_size != 2) throw std::runtime_error("data must be stored in half precision");
        data_ = std::move(d);
    }
    if ((size_t)capacity != child_.shape[0])
        throw std::runtime_error("child and data disagree on the capacity");
    if (data_dim < 1 || data_dim > 4096) throw std::runtime_error("data_dim out of range");
    const size_t n_slots = (size_t)capacity * N3_;
    if (!data_.empty() && data_.num_vals != n_slots * (size_t)data_dim)
        throw std::runtime_error("data does not have capacity * N^3 * data_dim values");
    if (npz.count("extra_data"))
        extra_ = std::move(npz["extra_data"]);
    else
        extra_ = internal::NpyArray();
}

void N3Tree::decode_quantized_host() {
    // reference src/n3tree.cpp:296-340
    if (quant_map_.empty() || !data_.empty()) return;
    const size_t n_q = quant_map_.shape[0];
    const size_t n_ret = data_retained_.empty() ? 0 : data_retained_.shape[0];
    const size_t n_basis = n_q + n_ret;
    const size_t n_child = (size_t)capacity * N3_;
    if (3 * n_basis + 1 > (size_t)data_dim)
        throw std::runtime_error("quantised basis functions do not fit data_dim");
    data_.shape = {(size_t)capacity, (size_t)N, (size_t)N, (size_t)N, (size_t)data_dim};
    data_.word_size = 2;
    data_.kind = 'f';
    data_.num_vals = n_child * data_dim;
    data_.data_holder.assign(data_.num_vals * 2, 0);
    uint16_t* out = data_.data<uint16_t>();
    const internal::NpyArray &cs = sigma_, &cm = quant_map_, &cc = quant_colors_,
                             &cr = data_retained_;
    const uint16_t* sigma = cs.data<uint16_t>();
    const uint16_t* map = cm.data<uint16_t>();
    const uint16_t* colors = cc.data<uint16_t>();
    for (size_t i = 0; i < n_child; ++i) {
        const size_t off = i * data_dim;
        for (size_t j = 0; j < n_q; ++j) {
            const uint16_t* c = colors + (j * 65536 + map[j * n_child + i]) * 3;
            for (size_t k = 0; k < 3; ++k) out[off + j + n_ret + k * n_basis] = c[k];
        }
        out[off + data_dim - 1] = sigma[i];
    }
    if (n_ret) {
        const uint16_t* ret = cr.data<uint16_t>();
        for (size_t i = 0; i < n_child; ++i)
            for (size_t j = 0; j < n_ret; ++j)
                for (size_t k = 0; k < 3; ++k)
                    out[i * data_dim + j + k * n_basis] = ret[(j * n_child + i) * 3 + k];
    }
}

void N3Tree::load_device() {
    free_device();
    VrTreeDesc d;
    vr_default_tree_desc(&d);
    // const access: stored members stay zero-copy views into the mapped file
    const internal::NpyArray& cchild = child_;
    const internal::NpyArray& cdata = data_;
    const internal::NpyArray& cextra = extra_;
    d.child = cchild.data<int32_t>();
    const bool quantised = data_.empty() && !quant_map_.empty();
    if (!quantised) d.data = cdata.data<uint16_t>();
    if (!extra_.empty()) {
        d.extra = cextra.data<float>();
        d.extra_count = extra_.num_bytes() / sizeof(float);
    }
    for (int i = 0; i < 3; ++i) {
        d.offset[i] = offset[i];
        d.scale[i] = scale[i];
    }
    d.N = N;
    d.capacity = capacity;
    d.data_dim = data_dim;
    d.format = (int)data_format.format;
    d.basis_dim = data_format.basis_dim;
    d.ndc_width = use_ndc ? ndc_width : -1.f;
    d.ndc_height = ndc_height;
    d.ndc_focal = ndc_focal;
    d.memory = 0;
    int rc;
    if (quantised) {
        const internal::NpyArray &cs = sigma_, &cm = quant_map_, &cc = quant_colors_,
                                 &cr = data_retained_;
        VrQuantDesc q;
        q.quant_colors = cc.data<uint16_t>();
        q.quant_map = cm.data<uint16_t>();
        q.sigma = cs.data<uint16_t>();
        q.data_retained = cr.empty() ? nullptr : cr.data<uint16_t>();
        q.n_quant = (int)cm.shape[0];
        q.n_retained = cr.empty() ? 0 : (int)cr.shape[0];
        rc = vr_tree_upload_quantized(&d, &q, &device);
    } else {
        rc = vr_tree_upload(&d, &device);
    }
    if (rc != VR_OK)
        throw std::runtime_error(std::string("vr_tree_upload: ") + vr_last_error());
    device_loaded_ = true;
}

void N3Tree::free_device() {
    if (device) vr_tree_free(device);
    device = nullptr;
    device_loaded_ = false;
}

bool N3Tree::is_data_loaded() { return data_loaded_; }
bool N3Tree::is_cuda_loaded() { return device_loaded_; }

void N3Tree::clear_cpu_memory() {
    // keep child_ (the reference keeps it for wireframes)
    data_.clear();
    quant_colors_.clear();
    quant_map_.clear();
    sigma_.clear();
    data_retained_.clear();
}

int N3Tree::pack_index(int nd, int i, int j, int k) { return nd * N3_ + i * N2_ + j * N + k; }

std::tuple<int, int, int, int> N3Tree::unpack_index(int packed) {
    const int k = packed % N;
    packed /= N;
    const int j = packed % N;
    packed /= N;
    const int i = packed % N;
    packed /= N;
    return std::tuple<int, int, int, int>{packed, i, j, k};
}

}  // namespace volrend
