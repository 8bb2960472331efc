// RGBA8 PNG writer (reference include/volrend/internal/imwrite.hpp); dependency-free
// apart from zlib's crc32 / adler32 -- libpng headers are not available everywhere.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace volrend {
namespace internal {
// No filtering, pixels STORED (compression level 0), as the reference configures libpng
// (src/imwrite.cpp:29-31): the cost is two checksums and the I/O.
bool write_png_file(const std::string& filename, const uint8_t* rgba, int width, int height);
// The same file into caller memory: png_stored_size(w, h) bytes exactly; returns the bytes written.
size_t png_stored_size(int width, int height);
size_t encode_png_stored(const uint8_t* rgba, int width, int height, uint8_t* out);
}  // namespace internal
}  // namespace volrend
