// j40_amd/csrc/plan_front.hpp -- the part of a VarDCT frame's plan that does NOT depend on the LfGroup sections, for the pipeline
// (device/async.hip): there the LfGroup streams are decoded and the rest of the plan is built on the device (device/plan_dev.h),
// so the host must not wait for them. Two pieces:
//
//   StaticTables   dequantisation weights of all 17 parameter sets and the coefficient orders of all 13 shapes -- a function of
//                  HfGlobal's matrix / order encodings alone (j40__load_dq_matrix, j40.h:4828; j40__natural_order + the coded
//                  permutations, j40.h:4980, 5460). The reference loads the ones a frame's varblocks use (j40.h:7694-7732); which
//                  those are is known only after the LfGroups, so all are loaded -- once per distinct encoding: the device copy is
//                  cached and shared by every frame with the same encoding (nearly always "all default": one bit in the stream).
//   FrontPlan      frame constants, entropy tables of the coefficient streams, TOC sections, event regions, LfGroup geometry,
//                  the constants of the device-side plan build, and the tables k_lf_groups needs.
#pragma once
#include "plan_build.hpp"

namespace j40hip {

struct StaticTables {
	std::vector<uint8_t> key;          // what the tables are a function of, byte for byte
	std::vector<float> pool_f32;       // dequantisation weights, planar per channel; for single-pass frames also in scan order
	std::vector<uint16_t> pool_u16;    // coefficient orders
	uint32_t order_off[11 * 13 * 3], dq_off[17], dq_size[17], dq_scan_off[17];
	uint32_t dq_error[17];             // what loading matrix i raised (0: nothing); it counts only if a varblock uses the matrix
};

void static_tables_key(const Frame &fr, std::vector<uint8_t> *key);
void build_static_tables(const Frame &fr, StaticTables *out);

struct FrontPlan {
	DevFrame frame;
	std::vector<uint8_t> pool_u8;
	std::vector<int32_t> pool_i32;
	std::vector<uint64_t> pool_u64;
	std::vector<DevCluster> clusters;
	std::vector<DevCodeSpec> coeff_specs;
	uint32_t block_ctx_map_off = 0;
	std::vector<DevLfGroup> lf_groups;       // geometry, bases (vb_base = cell_base: room for one varblock per cell), mult_lf; nb_varblocks = 0
	std::vector<DevSection> sections;
	std::vector<uint32_t> ev_range; size_t ev_capacity = 0;
	std::vector<uint32_t> lf_section_off;
	std::vector<uint32_t> lane_order;        // DevPlan::lane_order: the groups by decreasing section bytes (summed over the passes)
	size_t cells = 0, c64s = 0;
	int32_t max_lf_cells = 0;                // cells of the largest LfGroup
	HfLaunchInfo hf;
	uint32_t lz_window_size = 0;
	DevPlanBuild build;                      // constants only; the runtime fills in the pointers
	bool lf_smooth = false; float inv_m_lf[3] = {0.0f, 0.0f, 0.0f};
	// the global MA tree and code spec in the tables of k_lf_lanes (device/lf_lanes_dev.h); lf_device = false: the kernel cannot take
	// them (prefix codes, LZ77, weighted predictor, previous-channel properties, tables beyond its LDS budget): the host decodes the LfGroups
	bool lf_device = false;
	std::vector<DevTreeNode> lf_tree; std::vector<uint8_t> lf_ctx_map; std::vector<uint32_t> lf_cfg; std::vector<uint64_t> lf_alias;
	int32_t lf_log_alpha = 0; uint32_t lf_uses = 0, lf_lds_bytes = 0;
	void reset() {
		pool_u8.clear(); pool_i32.clear(); pool_u64.clear(); clusters.clear(); coeff_specs.clear(); lf_groups.clear(); sections.clear(); ev_range.clear();
		lf_section_off.clear(); lane_order.clear(); lf_alias.clear(); lf_tree.clear(); lf_ctx_map.clear(); lf_cfg.clear(); block_ctx_map_off = 0; ev_capacity = 0; cells = c64s = 0; max_lf_cells = 0; lz_window_size = 0; lf_smooth = lf_device = false;
	}
};

// returns 0, or "TODO" for frames the pipeline's device-side plan build does not take (the caller then uses the host path)
uint32_t build_front_plan(const Frame &fr, const StaticTables &st, size_t cs_size, const std::vector<int32_t> &extra_prec, bool want_lf_device, FrontPlan *out);

} // namespace j40hip
