// volrend/common.hpp -- build configuration of the MI355X-native volrend host layer.
// Counterpart of the reference's cmake-generated common.hpp (common.hpp.in): the
// device backend here is always the HIP library behind include/volrend_hip.h.
#pragma once
#define VOLREND_VERSION_MAJOR 0
#define VOLREND_VERSION_MINOR 1
#define VOLREND_VERSION_PATCH 0
#define VOLREND_HIP 1
