// j40_amd/csrc/capi.hpp -- the object behind the opaque j40hip_frame handle
#pragma once
#include "../../include/j40hip.h"
#include "frame.hpp"
#include "plan_build.hpp"

struct j40hip_device_state;  // defined in device/runtime.hip

struct j40hip_frame {
	const uint8_t *cs = nullptr;     // codestream bytes (inside the caller's buffer, or cs_storage)
	size_t cs_size = 0;
	std::vector<uint8_t> cs_storage;
	bool bare_codestream = false;    // the input was the codestream itself, no container around it
	bool from_view = false;          // built by j40hip_frame_from_vardct_view: no global MA tree / code spec, so the extra channels'
	                                 // sub-images behind the coefficients cannot be validated (runtime.hip: validate_trailers)
	int container_stray_tail = 0;    // container input: 1..7 bytes behind the last box (not enough for a box header)
	j40hip::Frame frame;
	j40hip_device_state *dev = nullptr;
	bool force_dense = false;        // upload with dense coefficient planes (set after a decode ran out of event space, ERR_EVOF)
	int restoration = -1;            // the restoration filters (j40hip_frame_set_restoration): -1 as J40HIP_RESTORATION says, 0 off, 1 on, 2 as j40's routines stand
	int threads = 1;                 // what the frame was parsed with: the plan build at upload may use as many (plan_build.cpp)
	// backing storage of the plan views (include/j40hip.h)
	struct Views {
		std::vector<std::vector<j40hip_cluster_view>> clusters;
		std::vector<j40hip_codespec_view> specs;
		std::vector<j40hip::CodeSpec> host_specs;
		std::vector<j40hip_lf_group_view> lf_groups;
		std::vector<j40hip_section_view> sections;
		std::vector<std::vector<float>> dq;
		std::vector<j40hip_tree_node> tree;
		std::vector<int32_t> ch_w, ch_h, ch_meta;
		std::vector<j40hip_transform_view> transforms;
		std::vector<j40hip_modular_section_view> mod_sections;
		std::vector<int32_t> local_rct, sub_w, sub_h, sub_meta, chan_rects;
		std::vector<j40hip_transform_view> sub_transforms;
		std::vector<std::vector<int32_t>> vb_coeffoff_qfidx;
		std::vector<std::vector<float>> vb_hfmul_inv;
	} views;
};

extern "C" j40hip_frame *j40hip_frame_parse_with(const void *buf, size_t size, int threads, uint32_t flags, j40hip::LfDeviceDecoder lf_decoder, void *lf_ctx, uint32_t *err);

// implemented next to the kernels; a no-op when nothing was uploaded
extern "C" void j40hip_release_device(j40hip_frame *f);

// internal to the library (api.cpp <-> the device side): the process-wide serving pipeline of a device (device/pipeline.hip) and
// the pool of pinned host planes the public API hands out as image pixels (device/runtime.hip)
extern "C" j40hip_pipeline *j40hip_serve_pipeline(int device, uint32_t *err);
extern "C" void j40hip_serve_shutdown(void);
// CPUs' worth of time the process may use: the visible CPUs, or the cgroup's quota (v2 cpu.max, v1 cfs_quota_us) when that is less.
// A process that runs into its quota has ALL its threads throttled, the HIP runtime's included: thread counts are sized from this.
extern "C" int j40hip_cpu_quota();
extern "C" void *j40hip_pinned_acquire(size_t bytes);
extern "C" void j40hip_pinned_release(void *ptr, size_t bytes);
