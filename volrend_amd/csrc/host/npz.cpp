// npz.cpp -- .npy / .npz reader (see volrend/internal/npz.hpp).
// Walks the ZIP central directory (robust against data descriptors and ZIP64 local
// headers, both of which numpy's savez produces), inflates deflated members with zlib.
#include "volrend/internal/npz.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <future>
#include <fstream>

namespace volrend {
namespace internal {
namespace {

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("npz: " + what); }

// header dict, e.g. {'descr': '<f2', 'fortran_order': False, 'shape': (3, 2, 2, 2), }
void parse_header(const std::string& h, NpyArray& a) {
    auto find_value = [&](const char* key) -> size_t {
        const size_t k = h.find(key);
        if (k == std::string::npos) fail(std::string("header lacks ") + key);
        const size_t c = h.find(':', k);
        if (c == std::string::npos) fail("malformed header");
        return c + 1;
    };
    {  // descr
        size_t p = find_value("'descr'");
        const size_t q1 = h.find('\'', p);
        const size_t q2 = h.find('\'', q1 + 1);
        if (q1 == std::string::npos || q2 == std::string::npos) fail("malformed descr");
        const std::string d = h.substr(q1 + 1, q2 - q1 - 1);  // like "<f2", "|u1", "<U4"
        if (d.size() < 3) fail("unsupported dtype " + d);
        if (d[0] == '>') fail("big-endian arrays are not supported");
        a.kind = d[1];
        if (d.find_first_not_of("0123456789", 2) != std::string::npos) fail("unsupported dtype " + d);
        if (d.size() > 8) fail("unsupported dtype " + d);
        const size_t n = (size_t)std::stoul(d.substr(2));
        a.word_size = a.kind == 'U' ? n * 4 : n;  // numpy stores UCS4
    }
    {  // fortran_order
        size_t p = find_value("'fortran_order'");
        while (p < h.size() && h[p] == ' ') ++p;
        a.fortran_order = h.compare(p, 4, "True") == 0;
    }
    {  // shape
        size_t p = find_value("'shape'");
        const size_t l = h.find('(', p), r = h.find(')', p);
        if (l == std::string::npos || r == std::string::npos) fail("malformed shape");
        a.shape.clear();
        size_t i = l + 1;
        while (i < r) {
            while (i < r && (h[i] == ' ' || h[i] == ',')) ++i;
            if (i >= r) break;
            size_t j = i;
            while (j < r && h[j] >= '0' && h[j] <= '9') ++j;
            if (j == i) fail("malformed shape");
            if (j - i > 18) fail("malformed shape");
            a.shape.push_back((size_t)std::stoull(h.substr(i, j - i)));
            i = j;
        }
    }
    if (a.word_size == 0 || a.word_size > 1024 || a.shape.size() > 16) fail("unsupported array header");
    a.num_vals = 1;
    for (size_t s : a.shape) {
        if (s != 0 && a.num_vals > (SIZE_MAX / a.word_size) / s) fail("array size overflows");
        a.num_vals *= s;
    }
}

// Parses the npy preamble; returns the offset of the raw data.
size_t parse_npy_preamble(const uint8_t* b, size_t size, NpyArray& a) {
    if (size < 10 || std::memcmp(b, "\x93NUMPY", 6) != 0) fail("not an npy stream");
    const int major = b[6];
    size_t hlen, off;
    if (major == 1) {
        hlen = rd16(b + 8);
        off = 10;
    } else {
        if (size < 12) fail("truncated npy header");
        hlen = rd32(b + 8);
        off = 12;
    }
    if (off + hlen > size) fail("truncated npy header");
    parse_header(std::string(reinterpret_cast<const char*>(b + off), hlen), a);
    return off + hlen;
}

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) fail("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    if (n && !f.read(reinterpret_cast<char*>(buf.data()), n)) fail("cannot read " + path);
    return buf;
}

struct Member {
    std::string name;
    uint16_t method;
    uint64_t csize, usize, local_off;
};

// [off, off + len) lies inside the archive (no wrap-around for hostile 64-bit fields)
inline bool inside(uint64_t off, uint64_t len, size_t size) {
    return off <= size && len <= size - off;
}

std::vector<Member> central_directory(const uint8_t* b, size_t size) {
    if (size < 22) fail("file too small for a zip archive");
    // end of central directory record: scan backwards for its signature
    size_t eocd = std::string::npos;
    const size_t lo = size > 22 + 65535 ? size - 22 - 65535 : 0;
    for (size_t i = size - 22 + 1; i-- > lo;) {
        if (rd32(b + i) == 0x06054b50u) {
            eocd = i;
            break;
        }
    }
    if (eocd == std::string::npos) fail("zip end-of-central-directory not found");
    uint64_t n_entries = rd16(b + eocd + 10);
    uint64_t cd_off = rd32(b + eocd + 16);
    if (eocd >= 20 && rd32(b + eocd - 20) == 0x07064b50u) {  // ZIP64 locator
        const uint64_t z64 = rd64(b + eocd - 20 + 8);
        if (!inside(z64, 56, size) || rd32(b + z64) != 0x06064b50u) fail("bad zip64 record");
        n_entries = rd64(b + z64 + 32);
        cd_off = rd64(b + z64 + 48);
    }
    std::vector<Member> out;
    uint64_t p = cd_off;
    for (uint64_t e = 0; e < n_entries; ++e) {
        if (!inside(p, 46, size) || rd32(b + p) != 0x02014b50u) fail("bad central directory entry");
        Member m;
        m.method = rd16(b + p + 10);
        m.csize = rd32(b + p + 20);
        m.usize = rd32(b + p + 24);
        const uint16_t fn = rd16(b + p + 28), ex = rd16(b + p + 30), cm = rd16(b + p + 32);
        m.local_off = rd32(b + p + 42);
        if (!inside(p + 46, (uint64_t)fn + ex + cm, size)) fail("central directory entry is truncated");
        m.name.assign(reinterpret_cast<const char*>(b + p + 46), fn);
        // ZIP64 extended information: only the saturated fields are present, in order
        uint64_t q = p + 46 + fn;
        const uint64_t qend = q + ex;
        while (q + 4 <= qend) {
            const uint16_t id = rd16(b + q), len = rd16(b + q + 2);
            if (q + 4 + len > qend) fail("bad extra field in the central directory");
            if (id == 0x0001) {
                uint64_t r = q + 4;
                const uint64_t rend = q + 4 + len;
                auto take = [&](uint64_t& field) {
                    if (r + 8 > rend) fail("zip64 extra field is too short");
                    field = rd64(b + r);
                    r += 8;
                };
                if (m.usize == 0xFFFFFFFFu) take(m.usize);
                if (m.csize == 0xFFFFFFFFu) take(m.csize);
                if (m.local_off == 0xFFFFFFFFu) take(m.local_off);
            }
            q += 4 + len;
        }
        out.push_back(std::move(m));
        p += 46 + fn + ex + cm;
    }
    return out;
}

NpyArray load_member(const uint8_t* b, size_t size, const Member& m,
                     const std::shared_ptr<void>& mapping) {
    if (!inside(m.local_off, 30, size) || rd32(b + m.local_off) != 0x04034b50u)
        fail("bad local header for " + m.name);
    const uint16_t fn = rd16(b + m.local_off + 26), ex = rd16(b + m.local_off + 28);
    const uint64_t data_off = m.local_off + 30 + fn + ex;
    if (!inside(data_off, m.csize, size)) fail("member " + m.name + " exceeds the archive");
    NpyArray a;
    if (m.method == 0) {  // stored
        const size_t pre = parse_npy_preamble(b + data_off, (size_t)m.csize, a);
        const size_t want = a.num_vals * a.word_size;
        if (pre + want > m.csize) fail("member " + m.name + " is truncated");
        if (mapping && want >= (1u << 16)) {  // big stored member of a mapped file: view it
            a.view = b + data_off + pre;
            a.view_bytes = want;
            a.mapping = mapping;
        } else {
            a.data_holder.assign(b + data_off + pre, b + data_off + pre + want);
        }
    } else if (m.method == 8) {  // deflate: inflate the whole member, then strip the preamble
        // deflate cannot expand by more than ~1032:1; a larger claim is a corrupt header
        if (m.usize / 1040 > m.csize + 64) fail("member " + m.name + " claims an impossible size");
        std::vector<uint8_t> raw((size_t)m.usize);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) fail("inflateInit2 failed");
        // zlib counts in 32-bit uInt: feed and drain in bounded slices
        uint64_t in_done = 0, out_done = 0;
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            if (zs.avail_in == 0 && in_done < m.csize) {
                const uint64_t n = std::min<uint64_t>(m.csize - in_done, 1u << 30);
                zs.next_in = const_cast<Bytef*>(b + data_off + in_done);
                zs.avail_in = (uInt)n;
                in_done += n;
            }
            if (zs.avail_out == 0 && out_done < m.usize) {
                const uint64_t n = std::min<uint64_t>(m.usize - out_done, 1u << 30);
                zs.next_out = raw.data() + out_done;
                zs.avail_out = (uInt)n;
                out_done += n;
            }
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                fail("inflate failed for " + m.name);
            }
            if (rc == Z_OK && zs.avail_in == 0 && in_done >= m.csize && zs.avail_out != 0) break;
        }
        inflateEnd(&zs);
        const size_t pre = parse_npy_preamble(raw.data(), raw.size(), a);
        const size_t want = a.num_vals * a.word_size;
        if (pre + want > raw.size()) fail("member " + m.name + " is truncated");
        raw.erase(raw.begin(), raw.begin() + (long)pre);
        raw.resize(want);
        a.data_holder = std::move(raw);
    } else {
        fail("unsupported zip compression method for " + m.name);
    }
    return a;
}

}  // namespace

double NpyArray::as_double(size_t i) const {
    if (i >= num_vals) throw std::out_of_range("npy index");
    const uint8_t* p = bytes() + i * word_size;
    switch (kind) {
        case 'f':
            if (word_size == 4) { float v; std::memcpy(&v, p, 4); return v; }
            if (word_size == 8) { double v; std::memcpy(&v, p, 8); return v; }
            break;
        case 'i':
            if (word_size == 1) return (int8_t)p[0];
            if (word_size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
            if (word_size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
            if (word_size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
            break;
        case 'u':
        case 'b':
            if (word_size == 1) return p[0];
            if (word_size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
            if (word_size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
            if (word_size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
            break;
        default:
            break;
    }
    throw std::runtime_error("npz: array is not numeric");
}

std::string NpyArray::as_string() const {
    std::string s;
    const uint8_t* d = bytes();
    const size_t n = num_bytes();
    if (kind == 'U') {  // UCS4 little-endian: keep the low byte of each code point
        for (size_t i = 0; i + 3 < n; i += 4)
            if (d[i]) s.push_back((char)d[i]);
    } else {
        for (size_t i = 0; i < n; ++i)
            if (d[i]) s.push_back((char)d[i]);
    }
    return s;
}

NpyArray npy_parse(const uint8_t* bytes, size_t size) {
    NpyArray a;
    const size_t pre = parse_npy_preamble(bytes, size, a);
    const size_t want = a.num_vals * a.word_size;
    if (pre + want > size) fail("npy stream is truncated");
    a.data_holder.assign(bytes + pre, bytes + pre + want);
    return a;
}

NpyArray npy_load(const std::string& path) {
    const std::vector<uint8_t> buf = read_file(path);
    return npy_parse(buf.data(), buf.size());
}

namespace {
NpzFile load_all(const uint8_t* bytes, size_t size, const std::shared_ptr<void>& mapping) {
    // A deflate stream is sequential, but the members are independent: the big ones
    // (child, data / quant_map, sigma, ...) inflate concurrently, one thread each.
    NpzFile out;
    const std::vector<Member> members = central_directory(bytes, size);
    auto key_of = [](const Member& m) {
        std::string key = m.name;
        if (key.size() > 4 && key.compare(key.size() - 4, 4, ".npy") == 0) key.resize(key.size() - 4);
        return key;
    };
    std::vector<std::pair<size_t, std::future<NpyArray>>> pending;
    for (size_t i = 0; i < members.size(); ++i) {
        const Member& m = members[i];
        if (m.method == 8 && m.csize >= (1u << 20))
            pending.emplace_back(i, std::async(std::launch::async, [&, i] {
                                     return load_member(bytes, size, members[i], mapping);
                                 }));
        else
            out.emplace(key_of(m), load_member(bytes, size, m, mapping));
    }
    std::exception_ptr err;
    for (auto& p : pending) {  // join everything before rethrowing
        try {
            out.emplace(key_of(members[p.first]), p.second.get());
        } catch (...) {
            if (!err) err = std::current_exception();
        }
    }
    if (err) std::rethrow_exception(err);
    return out;
}
}  // namespace

NpzFile npz_load_mem(const uint8_t* bytes, size_t size) { return load_all(bytes, size, nullptr); }

// The file is memory-mapped: stored (np.savez) members become zero-copy views, deflated
// (np.savez_compressed) members are inflated straight out of the mapping.
NpzFile npz_load(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) fail("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        fail("cannot stat " + path);
    }
    const size_t size = (size_t)st.st_size;
    void* addr = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (addr == MAP_FAILED) {  // fall back to reading the file
        const std::vector<uint8_t> buf = read_file(path);
        return npz_load_mem(buf.data(), buf.size());
    }
    madvise(addr, size, MADV_SEQUENTIAL);
    std::shared_ptr<void> mapping(addr, [size](void* p) { munmap(p, size); });
    return load_all(static_cast<const uint8_t*>(addr), size, mapping);
}

}  // namespace internal
}  // namespace volrend
