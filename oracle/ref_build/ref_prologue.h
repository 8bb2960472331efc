/* Prepended to the reference translation unit (see Makefile).  Routes the
 * reference's expf() calls to the oracle's deterministic expf so that the
 * host build of the reference and the C restatement are bit-comparable;
 * build with -DVR_REF_LIBM_EXPF to keep glibc's expf instead. */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include "../vr_detmath.h"
#ifndef VR_REF_LIBM_EXPF
#define expf vr_det_expf
#endif
