// volrend::DataFormat -- same type and semantics as the reference's
// include/volrend/data_format.hpp:8-25 (parse: src/n3tree.cpp:55-78).
#pragma once
#include <string>

#include "volrend/common.hpp"

namespace volrend {

struct DataFormat {
    enum {
        RGBA,  // Simply stores rgba
        SH,
        SG,
        ASG,
        _COUNT,
    } format = RGBA;

    // SH/SG/ASG dimension per channel
    int basis_dim = -1;

    // Parse a string like 'SH16', 'SG25'
    void parse(const std::string& str);

    // Convert to string
    std::string to_string() const;
};

}  // namespace volrend
