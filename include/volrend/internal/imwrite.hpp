// RGBA8 PNG writer (reference include/volrend/internal/imwrite.hpp); dependency-free
// apart from zlib's deflate/crc32 -- libpng headers are not available everywhere.
#pragma once
#include <cstdint>
#include <string>

namespace volrend {
namespace internal {
// No filtering, fastest deflate level: like the reference's level-0 / no-filter choice
// (src/imwrite.cpp:29-31) the cost is dominated by I/O, not compression.
bool write_png_file(const std::string& filename, const uint8_t* rgba, int width, int height);
}  // namespace internal
}  // namespace volrend
