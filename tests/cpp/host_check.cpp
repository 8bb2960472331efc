// host_check -- exercises the C++ host layer without a device; driven by
// tests/test_host_cpp.py which compares the printed facts with numpy.
#include <cstdio>
#include <cstring>
#include <string>

#include "volrend/internal/imwrite.hpp"
#include "volrend/internal/opts.hpp"
#include "volrend/n3tree.hpp"

using namespace volrend;

static unsigned long long checksum(const uint8_t* p, size_t n) {
    unsigned long long h = 1469598103934665603ull;  // FNV-1a
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char* argv[]) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    try {
        if (cmd == "npz" && argc >= 3) {
            internal::NpzFile z = internal::npz_load(argv[2]);
            for (auto& kv : z) {
                printf("%s kind=%c word=%zu shape=", kv.first.c_str(), kv.second.kind,
                       kv.second.word_size);
                for (size_t s : kv.second.shape) printf("%zu,", s);
                printf(" fnv=%llu\n", checksum(kv.second.bytes(), kv.second.num_bytes()));
            }
        } else if (cmd == "tree" && argc >= 3) {
            N3Tree::upload_on_open = false;
            N3Tree t(argv[2]);
            printf("N=%d capacity=%d data_dim=%d format=%s loaded=%d\n", t.N, t.capacity,
                   t.data_dim, t.data_format.to_string().c_str(), (int)t.is_data_loaded());
            printf("scale=%.9g,%.9g,%.9g offset=%.9g,%.9g,%.9g\n", t.scale[0], t.scale[1],
                   t.scale[2], t.offset[0], t.offset[1], t.offset[2]);
            printf("ndc=%d %.9g %.9g %.9g\n", (int)t.use_ndc, t.ndc_width, t.ndc_height,
                   t.ndc_focal);
            printf("child_fnv=%llu data_fnv=%llu extra_bytes=%zu\n",
                   checksum(t.child_.bytes(), t.child_.num_bytes()),
                   checksum(t.data_.bytes(), t.data_.num_bytes()),
                   t.extra_.num_bytes());
            if (t.N > 0) {
                const auto u = t.unpack_index(t.pack_index(3, 1, 0, 1));
                printf("pack=%d,%d,%d,%d\n", std::get<0>(u), std::get<1>(u), std::get<2>(u),
                       std::get<3>(u));
            }
        } else if (cmd == "png" && argc >= 5) {
            const int w = atoi(argv[3]), h = atoi(argv[4]);
            std::string buf((size_t)w * h * 4, '\0');
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    uint8_t* p = reinterpret_cast<uint8_t*>(&buf[((size_t)y * w + x) * 4]);
                    p[0] = (uint8_t)(x * 3);
                    p[1] = (uint8_t)(y * 5);
                    p[2] = (uint8_t)(x ^ y);
                    p[3] = 255;
                }
            return internal::write_png_file(argv[2], reinterpret_cast<const uint8_t*>(buf.data()),
                                            w, h) ? 0 : 1;
        } else if (cmd == "opts") {
            internal::Options o("host_check", "test");
            internal::add_common_opts(o);
            o.add("write_images", 'o', false, "", "");
            o.add("reverse_yz", 'r', true, "", "");
            o.add("scale", 0, false, "1.0", "");
            o.parse(argc - 1, argv + 1);
            const RenderOptions r = internal::render_options_from_args(o);
            printf("file=%s w=%d h=%d fx=%g bg=%g step=%g stop=%g sigma=%g out=%s r=%d scale=%g gpu=%d\n",
                   o.str("file").c_str(), o.as_int("width"), o.as_int("height"), o.as_float("fx"),
                   r.background_brightness, r.step_size, r.stop_thresh, r.sigma_thresh,
                   o.str("write_images").c_str(), (int)o.as_bool("reverse_yz"),
                   o.as_float("scale"), o.as_int("gpu"));
            for (const std::string& u : o.unmatched()) printf("unmatched=%s\n", u.c_str());
        } else {
            return 2;
        }
    } catch (const std::exception& e) {
        printf("EXCEPTION %s\n", e.what());
        return 3;
    }
    return 0;
}
