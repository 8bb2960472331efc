// opts.cpp -- command-line parsing for the volrend executables (flags of the reference's
// src/opts.cpp:7-66 and main_headless.cpp:85-97).
#include "volrend/internal/opts.hpp"

#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace volrend {
namespace internal {

Options::Options(std::string program, std::string description)
    : program_(std::move(program)), description_(std::move(description)) {}

void Options::add(const std::string& long_name, char short_name, bool is_flag,
                  const std::string& default_value, const std::string& help) {
    specs_.push_back({long_name, short_name, is_flag, default_value, help});
    if (!is_flag) values_[long_name] = default_value;
}

void Options::parse(int argc, char* argv[]) {
    bool have_file = false;
    auto find_long = [&](const std::string& n) -> const OptSpec* {
        for (const OptSpec& s : specs_)
            if (s.long_name == n) return &s;
        return nullptr;
    };
    auto find_short = [&](char c) -> const OptSpec* {
        for (const OptSpec& s : specs_)
            if (s.short_name == c) return &s;
        return nullptr;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const OptSpec* spec = nullptr;
        std::string inline_value;
        bool has_inline = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2);
            const size_t eq = name.find('=');
            if (eq != std::string::npos) {
                inline_value = name.substr(eq + 1);
                name.resize(eq);
                has_inline = true;
            }
            spec = find_long(name);
            if (!spec) {  // allow_unrecognised_options
                unmatched_.push_back(a);
                continue;
            }
        } else if (a.size() == 2 && a[0] == '-' && a[1] != '-' &&
                   !(a[1] >= '0' && a[1] <= '9')) {
            spec = find_short(a[1]);
            if (!spec) {
                unmatched_.push_back(a);
                continue;
            }
        } else {
            if (!have_file) {
                values_["file"] = a;  // first positional
                counts_["file"]++;
                have_file = true;
            } else {
                unmatched_.push_back(a);
            }
            continue;
        }
        counts_[spec->long_name]++;
        if (spec->is_flag) {
            values_[spec->long_name] = has_inline ? inline_value : "true";
        } else if (has_inline) {
            values_[spec->long_name] = inline_value;
        } else {
            if (i + 1 >= argc)
                throw std::runtime_error("option --" + spec->long_name + " needs a value");
            values_[spec->long_name] = argv[++i];
        }
    }
}

size_t Options::count(const std::string& name) const {
    auto it = counts_.find(name);
    return it == counts_.end() ? 0 : it->second;
}
std::string Options::str(const std::string& name) const {
    auto it = values_.find(name);
    return it == values_.end() ? std::string() : it->second;
}
int Options::as_int(const std::string& name) const { return std::atoi(str(name).c_str()); }
float Options::as_float(const std::string& name) const {
    return std::strtof(str(name).c_str(), nullptr);
}
bool Options::as_bool(const std::string& name) const {
    const std::string v = str(name);
    return v == "true" || v == "1";
}

std::string Options::help() const {
    std::ostringstream os;
    os << description_ << "\nUsage:\n  " << program_ << " [OPTION...] npz_file [c2w_txt_4x4...]\n\n";
    for (const OptSpec& s : specs_) {
        os << "  ";
        if (s.short_name) os << '-' << s.short_name << ", ";
        else os << "    ";
        os << "--" << s.long_name;
        if (!s.is_flag) os << " arg";
        os << "\t" << s.help;
        if (!s.is_flag && !s.default_value.empty()) os << " (default: " << s.default_value << ")";
        os << "\n";
    }
    return os.str();
}

void add_common_opts(Options& o) {
    o.add("file", 0, false, "", "npz file storing octree data");
    o.add("draw", 0, false, "", "npz drawlist file");
    o.add("gpu", 0, false, "-1", "device id (defaults to the current one)");
    o.add("width", 'w', false, "800", "image width");
    o.add("height", 'h', false, "800", "image height");
    o.add("fx", 0, false, "-1.0", "focal length in x direction; -1 = 1111 or default for NDC");
    o.add("fy", 0, false, "-1.0", "focal length in y direction; -1 = use fx");
    o.add("bg", 0, false, "1.0", "background brightness 0-1");
    o.add("step_size", 's', false, "1e-4", "step size epsilon added to computed cube size");
    o.add("stop_thresh", 'e', false, "1e-2", "early stopping threshold (on remaining intensity)");
    o.add("sigma_thresh", 'a', false, "1e-2", "sigma threshold (skip cells with < sigma)");
    o.add("help", 0, true, "", "Print this help message");
}

void parse_options(Options& options, int argc, char* argv[]) {
    options.parse(argc, argv);
    if (options.count("help")) {
        printf("%s\n", options.help().c_str());
        std::exit(0);
    }
}

RenderOptions render_options_from_args(const Options& args) {
    RenderOptions options;
    options.background_brightness = args.as_float("bg");
    options.step_size = args.as_float("step_size");
    options.stop_thresh = args.as_float("stop_thresh");
    options.sigma_thresh = args.as_float("sigma_thresh");
    return options;
}

}  // namespace internal
}  // namespace volrend
