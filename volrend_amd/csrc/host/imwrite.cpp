// imwrite.cpp -- RGBA8 PNG writer: IHDR + one zlib-deflated IDAT (filter 0 on every
// row) + IEND.  Plays the role of the reference's libpng writer (src/imwrite.cpp:14-79).
#include "volrend/internal/imwrite.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace volrend {
namespace internal {
namespace {
void put32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}
bool write_chunk(FILE* fp, const char type[4], const uint8_t* data, size_t len) {
    uint8_t hdr[8];
    put32(hdr, (uint32_t)len);
    std::memcpy(hdr + 4, type, 4);
    uint32_t crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, (uInt)len);
    uint8_t tail[4];
    put32(tail, crc);
    return fwrite(hdr, 1, 8, fp) == 8 && (len == 0 || fwrite(data, 1, len, fp) == len) &&
           fwrite(tail, 1, 4, fp) == 4;
}
}  // namespace

bool write_png_file(const std::string& filename, const uint8_t* rgba, int width, int height) {
    if (!rgba || width <= 0 || height <= 0) {
        fprintf(stderr, "PNG write failed\n");
        return false;
    }
    FILE* fp = fopen(filename.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "PNG destination could not be opened\n");
        return false;
    }
    const size_t row = (size_t)width * 4;
    std::vector<uint8_t> raw((row + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0;  // filter type: none
        std::memcpy(&raw[(row + 1) * y + 1], rgba + row * y, row);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    bool ok = compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) == Z_OK;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)width);
    put32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 6;   // colour type RGBA
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    ok = ok && fwrite(sig, 1, 8, fp) == 8 && write_chunk(fp, "IHDR", ihdr, 13) &&
         write_chunk(fp, "IDAT", comp.data(), clen) && write_chunk(fp, "IEND", nullptr, 0);
    fclose(fp);
    if (!ok) fprintf(stderr, "PNG write failed\n");
    return ok;
}

}  // namespace internal
}  // namespace volrend
