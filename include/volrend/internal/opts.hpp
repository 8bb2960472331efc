// Command-line options shared by the volrend executables: the flags of the reference's
// src/opts.cpp:7-31 (+ main_headless.cpp:85-97), parsed by a small built-in parser.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "volrend/render_options.hpp"

namespace volrend {
namespace internal {

struct OptSpec {
    std::string long_name;   // "width"
    char short_name;         // 'w' or 0
    bool is_flag;            // no value
    std::string default_value;
    std::string help;
};

class Options {
   public:
    Options(std::string program, std::string description);
    void add(const std::string& long_name, char short_name, bool is_flag,
             const std::string& default_value, const std::string& help);
    // Throws std::runtime_error on a malformed command line.  The first bare argument is
    // the positional "file"; further bare arguments and unknown --options land in
    // unmatched() (the pose files), as with cxxopts' allow_unrecognised_options.
    void parse(int argc, char* argv[]);
    size_t count(const std::string& name) const;
    std::string str(const std::string& name) const;
    int as_int(const std::string& name) const;
    float as_float(const std::string& name) const;
    bool as_bool(const std::string& name) const;
    const std::vector<std::string>& unmatched() const { return unmatched_; }
    std::string help() const;

   private:
    std::string program_, description_;
    std::vector<OptSpec> specs_;
    std::map<std::string, std::string> values_;
    std::map<std::string, size_t> counts_;
    std::vector<std::string> unmatched_;
};

// file, draw, gpu, w/width, h/height, fx, fy, bg, s/step_size, e/stop_thresh,
// a/sigma_thresh, help  (src/opts.cpp:9-29)
void add_common_opts(Options& options);
// prints help and exits on --help, like the reference's parse_options
void parse_options(Options& options, int argc, char* argv[]);
// src/opts.cpp:44-66
RenderOptions render_options_from_args(const Options& args);

}  // namespace internal
}  // namespace volrend
