// j40_amd/csrc/tables.hpp -- float constant tables shared by the host code and uploaded to the device
#pragma once
namespace j40hip {
const float *half_secants();   // [256]: [(1 << n) + k] = 1 / (2 cos((k + 1/2) pi / 2^(n+1))), reference float values (j40.h:5690)
const float *lf2llf_scales();  // [64] (j40.h:5739)
const float *afv_basis();      // [256] (j40.h:6108)
const float *srgb_u8_thresholds();   // [SRGB_TABLE_FLOATS]: see srgb_u8_from_thresholds (device/idct_dev.h)
}
