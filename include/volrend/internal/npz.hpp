// Minimal .npy / .npz reader for the tree files (stored + deflate members, ZIP64, '<U'
// strings, in-memory archives).  Written for this repo; plays the role cnpy plays in the
// reference (3rdparty/cnpy) without sharing code with it.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace volrend {
namespace internal {

struct NpyArray {
    // Either owned bytes (data_holder) or a zero-copy view into a memory-mapped .npz
    // (stored, i.e. uncompressed, members of npz_load: no copy of a GB-sized `data` array).
    std::vector<uint8_t> data_holder;
    const uint8_t* view = nullptr;
    size_t view_bytes = 0;
    std::shared_ptr<void> mapping;  // keeps the mmap alive while views exist
    std::vector<size_t> shape;
    size_t word_size = 0;
    char kind = 0;  // numpy dtype kind: 'f', 'i', 'u', 'U', 'b', ...
    bool fortran_order = false;
    size_t num_vals = 0;

    const uint8_t* bytes() const { return view ? view : data_holder.data(); }
    template <typename T>
    T* data() {  // writable access materialises a view
        if (view) own();
        return reinterpret_cast<T*>(data_holder.data());
    }
    template <typename T>
    const T* data() const { return reinterpret_cast<const T*>(bytes()); }
    size_t num_bytes() const { return view ? view_bytes : data_holder.size(); }
    bool empty() const { return num_bytes() == 0; }
    void own() {
        if (!view) return;
        data_holder.assign(view, view + view_bytes);
        view = nullptr;
        view_bytes = 0;
        mapping.reset();
    }
    void clear() {
        data_holder.clear();
        data_holder.shrink_to_fit();
        view = nullptr;
        view_bytes = 0;
        mapping.reset();
    }
    // scalar / first element as double (any numeric dtype)
    double as_double(size_t i = 0) const;
    // '<U..' or '|S..' array as an ASCII string
    std::string as_string() const;
};

using NpzFile = std::map<std::string, NpyArray>;

NpyArray npy_load(const std::string& path);
NpyArray npy_parse(const uint8_t* bytes, size_t size);
NpzFile npz_load(const std::string& path);
NpzFile npz_load_mem(const uint8_t* bytes, size_t size);

}  // namespace internal
}  // namespace volrend
