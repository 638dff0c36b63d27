// j40_amd/csrc/device/kernels.h -- launch entry points of device/kernels.hip and device/modular_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "plan.h"

namespace j40hip {

void upload_constant_tables(const float *half_secants, const float *afv_basis, hipStream_t stream);
void launch_hf_entropy(const DevPlan &plan, const HfLaunchInfo &info, int32_t first_group, int32_t num_groups, hipStream_t stream);
void launch_vardct_class(const DevPlan &plan, int32_t dctsel, const DevVarblock *list, int32_t count, float *large_scratch, uint8_t *rgba, size_t stride, hipStream_t stream);

} // namespace j40hip
