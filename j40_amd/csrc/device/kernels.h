// j40_amd/csrc/device/kernels.h -- launch entry points of device/kernels.hip and device/modular_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "plan.h"
#include "restore_dev.h"

namespace j40hip {

void upload_constant_tables(const float *half_secants, const float *afv_basis, const float *srgb_thr, hipStream_t stream);
void launch_hf_entropy(const DevPlan &plan, const HfLaunchInfo &info, int32_t first_group, int32_t num_groups, hipStream_t stream);
// the single-image path's two phases (runtime.hip): is the fast entropy kernel the one this frame gets; that kernel over the groups
// order[first .. first + count); the block_events entries of the groups order[0 .. k) copied from `shadow` into the plan's table
bool hf_entropy_fast_path(const DevPlan &plan, const HfLaunchInfo &info);
void launch_hf_entropy_fast_ordered(const DevPlan &plan, const HfLaunchInfo &info, const uint32_t *order, int32_t first, int32_t count, hipStream_t stream);
void launch_merge_block_events(const DevPlan &plan, const uint32_t *order, int32_t k, const uint32_t *shadow, hipStream_t stream);
// the rectangles of the groups order[0 .. k) out of the device image into a host image the device can write (pinned)
void launch_store_group_rects(const uint32_t *order, int32_t k, int32_t gcolumns, int32_t shift, int32_t width, int32_t height, const uint8_t *src, uint8_t *dst_host_mapped, size_t stride_bytes, hipStream_t stream);
uint32_t hf_lanes_lds_bytes(const HfLaunchInfo &info);
void launch_hf_entropy_lanes(const DevPlan *plans, const HfLaneWork *work, int32_t num_work, bool tables_in_lds, uint32_t lds_bytes, hipStream_t stream);
void launch_hf_lanes(const DevPlan *plans, const HfLaneWork *work, int32_t num_work, int32_t waves_per_wg, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started = nullptr, hipEvent_t stopped = nullptr, uint32_t *queue = nullptr);
void launch_vardct_class(const DevPlan &plan, int32_t dctsel, const DevVarblock *list, int32_t count, float *large_scratch, uint8_t *rgba, size_t stride, hipStream_t stream);


// every frame of a batch: one persistent launch per class of transforms, spread over `nside` side streams that fork from and join
// `stream` (nside = 0: all on `stream`). tile_prefix_dev: K2_NUM_BATCH_LAUNCHES * (nframes + 1) ints of scratch; totals_dev:
// K2_NUM_BATCH_LAUNCHES ints (out: tiles per launch); grids: workgroups per launch, from k2_batch_grids
enum { K2_NUM_BATCH_LAUNCHES = 16, K2_LARGE_WGS = 256 };
void k2_batch_grids(const int32_t *last_totals, size_t cells_total, int32_t nframes, int32_t wg_slots, int32_t *grids);
void launch_vardct_batch(const K2Frame *frames_dev, int32_t nframes, int32_t *tile_prefix_dev, int32_t *totals_dev, const int32_t *grids, float *large_scratch, hipStream_t stream, hipStream_t *side, int nside, hipEvent_t fork, hipEvent_t *side_done);
void launch_vardct_frame(const DevPlan &plan, const int32_t *class_start, const DevVarblock *sorted, float *large_scratch, uint8_t *rgba, size_t stride, hipStream_t stream);

// the restoration filters (device/restore_kernels.h, restore_dev.h): the pixel kernels with the samples left in XYB (three float planes of
// `stride` bytes per row, one behind the other), the reciprocal-sigma plane, Gaborish + the edge-preserving filter's steps between
// `xyb` and `tmp` (returns where the result lies), the colour tail on planes
void launch_vardct_frame_xyb(const DevPlan &plan, const int32_t *class_start, const DevVarblock *sorted, float *large_scratch, float *xyb, size_t stride, hipStream_t stream);
void launch_epf_sigma(const DevPlan &plan, int32_t num_lf_groups, const int16_t *sharpness, const RestoreParams &p, float *sigma, uint32_t *sharp_or, hipStream_t stream);
void launch_epf_sigma_cells(const int16_t *sharpness, const float *hfmul_inv, const RestoreParams &p, float *sigma, uint32_t *sharp_or, hipStream_t stream);
float *launch_restoration(float *xyb, float *tmp, size_t pitch, const RestoreParams &p, bool gab, int32_t epf_iters, const float *sigma, hipStream_t stream);
void launch_xyb_to_rgba(const float *xyb, size_t pitch, const DevFrame *frame_dev, int32_t width, int32_t height, uint8_t *rgba, size_t stride_bytes, hipStream_t stream);

// LfGroup tail on the device (device/lf_tail_kernels.hip)
void upload_lf_tail_tables(const float *half_secants, const float *lf2llf, hipStream_t stream);
void launch_lf_tail(const DevPlan &plan, int32_t num_lf_groups, int32_t max_cells, size_t cells, float *lfs, const DevVarblock *sorted, int32_t count, int32_t first_large, int32_t smooth,
		const float inv_m_lf[3], hipStream_t stream);

void launch_lf_tail_batch(const DevPlan *plans, const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nframes, int32_t nlf, int32_t max_lf_cells, size_t max_frame_cells, hipStream_t stream);
// the LF-dependent half of the plan of every frame of a batch (device/plan_kernels.hip)
void launch_plan_build(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nframes, int32_t nlf, int32_t max_lf_cells, hipStream_t stream);
void launch_clear_block_events(const DevPlan *plans, const DevPlanBuild *builds, int32_t nframes, size_t max_frame_cells, hipStream_t stream);
void launch_plan_verdict(const DevPlanBuild *builds, const DevPlan *plans, int32_t nframes, hipStream_t stream);

void launch_kat_srgb_u8(const float *v, size_t n, uint8_t *out, hipStream_t stream);

// Modular path (device/modular_kernels.hip)
void launch_modular_sections(const DevModPlan &plan, int32_t first_section, int32_t num_sections, const ModLaunchInfo &info, hipStream_t stream);
void launch_lf_groups(const DevLfTask *tasks, int32_t num_tasks, hipStream_t stream);
// one LfGroup section per lane (lf_lanes_dev.h): pack_lf_waves deals the sections of the sets (host copies) to wavefronts and returns
// the LDS a wavefront needs at most; launch_lf_lanes takes the device copies of both arrays
#include <vector>
uint32_t pack_lf_waves(const DevLfLaneSet *sets_host, int32_t num_sets, std::vector<DevLfWave> *waves, const std::vector<int32_t> *only = nullptr);
void launch_lf_lanes(const DevLfLaneSet *sets, const DevLfWave *waves, int32_t num_waves, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started = nullptr, hipEvent_t stopped = nullptr);
// the same with tree, alias tables and row windows in LDS (lf_rows_dev.h, k_lf_rows): pack_lf_row_waves returns 0 when some frame's
// tables do not fit a wavefront's LDS -- with `oversized`, those frames alone are listed for k_lf_lanes --; lf_rows_enabled: J40HIP_LF_KERNEL != "lanes"
bool lf_rows_enabled();
uint32_t pack_lf_row_waves(const DevLfLaneSet *sets_host, int32_t num_sets, std::vector<DevLfWave> *waves, std::vector<int32_t> *oversized = nullptr);
void launch_lf_rows(const DevLfLaneSet *sets, const DevLfWave *waves, int32_t num_waves, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started = nullptr, hipEvent_t stopped = nullptr);
void launch_modular_quad(const DevModPlan &plan, int32_t first_section, int32_t num_sections, int32_t spec_idx, uint32_t table_span, int32_t max_width, hipStream_t stream);
void launch_modular_split(const DevModPlan &plan, int32_t first_section, int32_t num_sections, const ModLaunchInfo &info, hipStream_t stream);
void launch_modular_coop(const DevModPlan &plan, int32_t first_section, int32_t num_sections, int32_t max_width, hipStream_t stream);
void launch_section_inverse_rcts(const DevModPlan &plan, int32_t first_section, int32_t num_sections, hipStream_t stream);
void launch_paste_plane(const int16_t *src, int32_t w, int32_t h, int16_t *dst, int32_t dst_stride, hipStream_t stream);
void launch_inverse_rct(int16_t *a, int16_t *b, int16_t *c, size_t n, int32_t type7, hipStream_t stream);
void launch_inverse_palette_plain(const int16_t *idx, const int16_t *palrow, int16_t *dst, size_t n, int32_t i, int32_t nb_colours, int32_t bpp, hipStream_t stream);
void launch_inverse_palette_predicted(const int16_t *idx, const int16_t *pal, int32_t pal_stride, int16_t *const *dst_dev, int32_t num_c, int32_t width, int32_t height,
		int32_t nb_colours, int32_t nb_deltas, int32_t d_pred, int32_t bpp, const int8_t *wpp_dev, int32_t *wp_scratch, uint32_t *status, hipStream_t stream);
void launch_inverse_squeeze(const int16_t *avg, const int16_t *res, int16_t *out, int32_t aw, int32_t ah, int32_t rw, int32_t rh, bool horizontal, hipStream_t stream);
void launch_pack_planes_rect(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t plane_width, int32_t x0, int32_t y0, int32_t rw, int32_t rh, int32_t bpp, uint8_t *rgba, size_t stride, hipStream_t stream);
void launch_pack_planes(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t width, int32_t height, int32_t bpp, uint8_t *rgba, size_t stride, hipStream_t stream);

} // namespace j40hip
