// imwrite.cpp -- RGBA8 PNG writer: IHDR + one IDAT holding a zlib stream of STORED blocks
// (filter 0 on every row) + IEND.  The reference asks libpng for compression level 0 and no
// filter (src/imwrite.cpp:29-31): the pixels are stored, not deflated, and so they are here --
// the stream is assembled by hand (RFC 1950 header, RFC 1951 stored blocks of up to 65535
// bytes, Adler-32), zlib only lends its crc32 / adler32.  The file is built in one buffer and
// written with one call: an 800x800 frame costs a pass of checksums, not 88 ms of deflate.
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
// header + payload + CRC of one chunk whose payload already sits at p + 8
uint8_t* seal_chunk(uint8_t* p, const char type[4], size_t len) {
    put32(p, (uint32_t)len);
    std::memcpy(p + 4, type, 4);
    uLong crc = crc32(0L, Z_NULL, 0);
    // (crc32 takes a 32-bit length; a frame's IDAT can exceed it only beyond 2^32 bytes,
    // refused below)
    crc = crc32(crc, p + 4, (uInt)(4 + len));
    put32(p + 8 + len, (uint32_t)crc);
    return p + 12 + len;
}
constexpr size_t kStored = 65535;  // largest stored block
}  // namespace

size_t png_stored_size(int width, int height) {
    const size_t raw = ((size_t)width * 4 + 1) * (size_t)height;
    const size_t blocks = (raw + kStored - 1) / kStored;
    const size_t idat = 2 + blocks * 5 + raw + 4;
    return 8 + (12 + 13) + (12 + idat) + 12;
}

size_t encode_png_stored(const uint8_t* rgba, int width, int height, uint8_t* out) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    const size_t row = (size_t)width * 4;
    const size_t raw = (row + 1) * (size_t)height;
    uint8_t* p = out;
    std::memcpy(p, sig, 8);
    p += 8;
    put32(p + 8, (uint32_t)width);
    put32(p + 12, (uint32_t)height);
    p[16] = 8;  // bit depth
    p[17] = 6;  // colour type RGBA
    p[18] = p[19] = p[20] = 0;
    p = seal_chunk(p, "IHDR", 13);

    uint8_t* const idat = p;
    uint8_t* q = idat + 8;
    *q++ = 0x78;  // CM = 8 (deflate), 32 KB window
    *q++ = 0x01;  // FLEVEL 0, no dictionary; (0x7801 % 31 == 0)
    // the filtered scanlines (one zero byte + the row), cut into stored blocks
    uLong adler = adler32(0L, Z_NULL, 0);
    size_t left_in_block = 0, left_total = raw;
    auto emit = [&](const uint8_t* src, size_t n) {
        while (n) {
            if (left_in_block == 0) {
                const size_t len = left_total < kStored ? left_total : kStored;
                *q++ = left_total <= kStored ? 1 : 0;  // BFINAL, BTYPE = 00
                *q++ = (uint8_t)(len & 0xFF);
                *q++ = (uint8_t)(len >> 8);
                *q++ = (uint8_t)(~len & 0xFF);
                *q++ = (uint8_t)((~len >> 8) & 0xFF);
                left_in_block = len;
            }
            const size_t take = n < left_in_block ? n : left_in_block;
            std::memcpy(q, src, take);
            q += take;
            src += take;
            n -= take;
            left_in_block -= take;
            left_total -= take;
        }
    };
    static const uint8_t filter_none = 0;
    for (int y = 0; y < height; ++y) {
        emit(&filter_none, 1);
        emit(rgba + row * y, row);
        adler = adler32(adler, &filter_none, 1);
        adler = adler32(adler, rgba + row * y, (uInt)row);
    }
    put32(q, (uint32_t)adler);
    q += 4;
    p = seal_chunk(idat, "IDAT", (size_t)(q - (idat + 8)));
    p = seal_chunk(p, "IEND", 0);
    return (size_t)(p - out);
}

bool write_png_file(const std::string& filename, const uint8_t* rgba, int width, int height) {
    if (!rgba || width <= 0 || height <= 0 ||
        ((size_t)width * 4 + 1) * (size_t)height >= 0xFFFF0000ull) {
        fprintf(stderr, "PNG write failed\n");
        return false;
    }
    FILE* fp = fopen(filename.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "PNG destination could not be opened\n");
        return false;
    }
    std::vector<uint8_t> file(png_stored_size(width, height));
    const size_t n = encode_png_stored(rgba, width, height, file.data());
    const bool ok = n == file.size() && fwrite(file.data(), 1, n, fp) == n;
    if (fclose(fp) != 0 || !ok) {
        fprintf(stderr, "PNG write failed\n");
        return false;
    }
    return true;
}

}  // namespace internal
}  // namespace volrend
