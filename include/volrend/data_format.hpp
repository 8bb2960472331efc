// volrend::DataFormat: what a leaf record holds (reference include/volrend/data_format.hpp:8-25;
// the "SH16" / "SG25" / "RGBA" strings of tree.npz are parsed as src/n3tree.cpp:55-78 does).
#pragma once
#include <string>

#include "volrend/common.hpp"

namespace volrend {

struct DataFormat {
    enum Kind { RGBA, SH, SG, ASG, _COUNT };  // RGBA: colour stored directly; the others: a basis
    Kind format = RGBA;
    int basis_dim = -1;  // basis functions per colour channel (-1 for RGBA)

    void parse(const std::string& str);  // "SH16" -> {SH, 16}; anything unknown -> RGBA
    std::string to_string() const;
};

}  // namespace volrend
