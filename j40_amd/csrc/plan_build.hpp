// j40_amd/csrc/plan_build.hpp -- flattens a parsed Frame into the pointer-free plan the kernels read
// (device/plan.h). Pure host code: runtime.hip uploads the arrays to HBM; tests/hostsim points a
// DevPlan at them directly to single-step the device functions on the CPU.
#pragma once
#include "frame.hpp"
#include "device/plan.h"

namespace j40hip {

struct HostPlan {
	DevFrame frame;
	std::vector<uint8_t> codestream;   // padded copy
	std::vector<uint8_t> pool_u8;
	std::vector<uint16_t> pool_u16;
	std::vector<int32_t> pool_i32;
	std::vector<uint64_t> pool_u64;
	std::vector<float> pool_f32;
	std::vector<DevCluster> clusters;
	std::vector<DevCodeSpec> coeff_specs;
	std::vector<DevLfGroup> lf_groups;
	std::vector<DevSection> sections;
	std::vector<DevGroupBlock> group_blocks;
	std::vector<uint32_t> group_block_start;
	uint32_t block_ctx_map_off = 0;
	std::vector<int32_t> blocks, vb_coeffoff_qfidx;
	std::vector<uint8_t> lfindices;
	std::vector<float> llf[3], vb_hfmul_inv;
	std::vector<int16_t> xfromy, bfromy;
	// the LfGroup tail left to the device (Frame::defer_lf_tail): decoded LF integers per cell (channels X, Y, B) instead of `llf`
	bool lf_tail_pending = false, lf_smooth = false;
	std::vector<int16_t> lfraw[3];
	float inv_m_lf[3] = {0.0f, 0.0f, 0.0f};
	std::vector<DevVarblock> vb_sorted;   // by DctSelect
	bool force_dense = false;             // in: dense coefficient planes even for single-pass frames (fallback after ERR_EVOF)
	std::vector<uint32_t> ev_range;       // sparse coefficients: [2 * group] first / end event of the group's region
	size_t ev_capacity = 0;               // events in total
	int32_t class_start[28];
	size_t coeff_floats = 0;
	uint32_t lz_window_size = 0;          // 0: no LZ77 in any coefficient code spec
	int32_t max_large = 0;
	HfLaunchInfo hf;                     // sizes that shape K1's LDS layout                // most varblocks of one 128/256-sized transform type
	// empties the plan but keeps the vectors' storage: a worker thread builds one plan after the other into the same object, and
	// allocating (and page-faulting in) ~35 MB of vectors per 8K frame was a third of the plan build
	void reset() {
		codestream.clear(); pool_u8.clear(); pool_u16.clear(); pool_i32.clear(); pool_u64.clear(); pool_f32.clear(); clusters.clear(); coeff_specs.clear();
		lf_groups.clear(); sections.clear(); group_block_start.clear();
		for (int c = 0; c < 3; ++c) inv_m_lf[c] = 0.0f;
		xfromy.clear(); bfromy.clear(); ev_range.clear();
		// (vb_sorted, group_blocks, blocks, lfindices, the LF planes and the per-varblock arrays keep their sizes: build_vardct_plan sizes
		// them to the frame and overwrites every element -- a thread that decodes frame after frame of one size does not zero 10 MB a frame)
		block_ctx_map_off = 0; lf_tail_pending = lf_smooth = force_dense = false; ev_capacity = 0; coeff_floats = 0; lz_window_size = 0; max_large = 0;
	}
};

bool build_lf_coop(const Frame &fr, DevCoopTree *tree, std::vector<uint64_t> *alias, int32_t *log_alpha_size);   // see plan_build.cpp

// returns 0 or a 4-char error code ("TODO" for frame kinds the hot path does not cover)
// threads: how many may work on the frame-wide arrays together (1: the calling thread alone; the arrays are the same either way)
uint32_t build_vardct_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, HostPlan *out, int threads = 1);

// single-pass frames keep HF coefficients in scan order on the device (DevFrame::scan_order_coeffs);
// this rewrites LF group `gg`, channel c in place into the canonical layout the reference uses
// (coeffs[order[i]], j40.h:6989) -- for stage dumps / parity tests

// ---- Modular frames ----
struct HostModPlan {
	DevModFrame frame;
	std::vector<uint8_t> codestream;
	std::vector<uint8_t> pool_u8;
	std::vector<int32_t> pool_i32;
	std::vector<uint64_t> pool_u64;
	std::vector<DevCluster> clusters;
	std::vector<DevCodeSpec> specs;                       // [0] global, then per-section own specs
	std::vector<DevTreeNode> tree;                        // global tree, then the sections' own trees
	std::vector<CodeSpec> host_specs;                     // the parsed form of `specs`, same order (for the oracle's view)
	int32_t max_tree_nodes = 0, max_num_dist = 0, max_clusters = 0; uint32_t max_table_bytes = 0;   // over the specs / trees, for the kernel's LDS layout
	bool any_lz77 = false, any_wp = false;
	std::vector<DevModSection> sections;                  // LfGlobal's channel data, then num_passes * sections_per_pass group sections, pass-major
	int32_t num_passes = 1, sections_per_pass = 0;
	std::vector<int32_t> local_rct;                       // {begin_c, rct_type} pairs, DevModSection::local_off / local_count
	// sections that list a palette of their own decode into a sub-image of their own (DevModSection::sub_off): its coded channels,
	// its transforms (undone on the sub-image, then the channels are pasted over the section's rectangle, j40.h:7030-7032)
	struct SubImage {
		int32_t section, first_plane, num_planes;
		std::vector<Transform> transforms;
		int8_t wp[12];
		bool paste;                                       // false: an earlier pass of a multi-pass frame, decoded for its status only
	};
	std::vector<SubImage> sub_images;
	std::vector<int32_t> sub_w, sub_h, sub_meta;          // the sub-images' planes, SubImage::first_plane ...
	std::vector<int32_t> plane_w, plane_h, plane_meta;   // coded channels
	std::vector<DevChanRect> chan_rects;                  // sections of frames with channels of different sizes (DevModSection::chan_off)
	std::vector<DevCoopTree> coop_trees;                  // DevModSection::coop_idx
	int32_t coop_width = 0, coop_sections = 0;            // widest channel / number of the sections k_modular_coop takes
	int32_t quad_sections = 0, quad_spec = 0, quad_width = 0;   // ... of those, the ones k_modular_quad takes four to a wavefront
	// sections decoded in two passes by modular_split.hip (position-only MA tree): how many, their residual tokens in all, their widest
	// channel, the most channels one of them codes
	int32_t split_sections = 0, split_width = 0, split_channels = 0; size_t split_samples = 0;
	std::vector<Transform> transforms;                    // global transforms in coded order
	int32_t alpha_channel = -1;                           // index (after inverse transforms) of the first alpha extra channel
	uint32_t lz_window_size = 0;
};

// parses every pass-group section's Modular header on the host (a few bits each) and lays out the
// device plan; returns 0 or a 4-char code ("TODO": local transforms / layouts the
// reference itself refuses)
uint32_t build_modular_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, HostModPlan *out);

// VarDCT frames with extra channels: the Modular sub-images behind the HF coefficients of every pass-group section, laid out for
// K3 (plan_build.cpp); sections: those whose coefficients decoded, section_of[i] = their index in the frame
uint32_t build_trailer_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, const uint32_t *end_bits, const uint32_t *k1_status, HostModPlan *hp,
		std::vector<std::pair<int32_t, uint32_t>> *trailer_errors, std::vector<int32_t> *section_of);

// pieces of build_vardct_plan shared with the pipeline's front plan (plan_front.cpp)
void fill_frame_constants(const Frame &fr, DevFrame *out);
void fill_sections(const Frame &fr, std::vector<DevSection> *sections);
bool fill_event_ranges(const std::vector<DevSection> &sections, int32_t num_groups, bool sparse, std::vector<uint32_t> *ev_range, size_t *ev_capacity);
void fill_hf_launch_info(const std::vector<DevCodeSpec> &coeff_specs, uint32_t block_ctx_size, size_t coeff_floats, HfLaunchInfo *out);

// fills a DevCodeSpec and appends its tables to the pools
void flatten_code_spec(const CodeSpec &spec, std::vector<uint8_t> &u8, std::vector<int32_t> &i32, std::vector<uint64_t> &u64, std::vector<DevCluster> &clusters, DevCodeSpec *out);

} // namespace j40hip
