// j40_amd/csrc/device/plan_kernels.hip -- the LF-dependent half of a VarDCT frame's plan, built on the device for every frame of
// a batch at once (replaces, in the pipeline, frame.cpp's lf_group_finish -- the reference's
// j40__hf_metadata placement loop, j40.h:6634-6701, and LF index, j40.h:6566-6570 -- and plan_build.cpp's work lists). The
// arithmetic lives in plan_dev.h, which tests/hostsim compiles for the CPU and checks against the host path array by array.
//
//   k_plan_place     one LfGroup per LANE: the placement is a serial walk (where a block goes depends on every block before it),
//                    64 LfGroups side by side per wavefront, their column state interleaved in LDS. A batch of 256 8K frames is
//                    3072 LfGroups = 48 wavefronts for a few milliseconds: latency, not throughput; the other kernels of other
//                    batches fill the machine meanwhile
//   k_plan_scan      one lane per frame: where each group's block list and each (DctSelect, LfGroup)'s work items start
//   k_plan_emit      one lane per varblock: its K1 record (block contexts from the LF index of its top-left cell) and K2 record
//   k_plan_verdict   one wavefront per frame, after the entropy kernel: the first failing section in file order
#include <hip/hip_runtime.h>
#include "plan_dev.h"
#include "kernels.h"

namespace j40hip {

__global__ void __launch_bounds__(64) k_plan_place(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nlf) {
	__shared__ uint16_t occ[256 * 64];
	__shared__ uint16_t grp[64 * 64];
	__shared__ uint32_t cls[28 * 64];
	const int32_t i = (int32_t) (blockIdx.x * 64 + threadIdx.x);
	if (i >= nlf) return;
	const DevBatchLf w = lfs[i];
	plan_place_lf_group(builds[w.frame], w.lfg, occ + threadIdx.x, grp + threadIdx.x, cls + threadIdx.x, 64);
}

__global__ void __launch_bounds__(64) k_plan_scan(const DevPlanBuild *builds, int32_t nframes) {
	const int32_t f = (int32_t) (blockIdx.x * 64 + threadIdx.x);
	if (f < nframes) plan_scan_frame(builds[f]);
}

__global__ void __launch_bounds__(256) k_plan_emit(const DevPlanBuild *builds, const DevBatchLf *lfs) {
	const DevBatchLf w = lfs[blockIdx.y];
	const DevPlanBuild &pb = builds[w.frame];
	const int32_t v = (int32_t) (blockIdx.x * 256 + threadIdx.x);
	if (v < pb.lf_slots[w.lfg].placed) plan_emit_varblock(pb, w.lfg, v);
}

__global__ void __launch_bounds__(64) k_plan_verdict(const DevPlanBuild *builds, const DevPlan *plans) {
	const DevPlanBuild &pb = builds[blockIdx.x];
	const DevPlan &plan = plans[blockIdx.x];
	const int32_t lane = threadIdx.x, nsec = plan.frame->num_passes * pb.num_groups;
	uint64_t best = ~(uint64_t) 0;
	uint32_t flags = 0, used = 0;
	for (int32_t g = lane; g < pb.num_lf_groups; g += 64) {
		const DevLfSlot sl = pb.lf_slots[g];
		if (sl.status == (uint32_t) ERR_LFFB) flags |= 1u;
		used |= sl.dct_used;
		const uint64_t k = plan_verdict_key(sl.status, pb.lf_section_off[g]);
		best = k < best ? k : best;
	}
	for (int32_t i = lane; i < nsec; i += 64) {
		const uint32_t st = plan.status[i];
		if (st == (uint32_t) ERR_EVOF) flags |= 2u;
		const uint64_t k = plan_verdict_key(st, plan.sections[i].byte_off);
		best = k < best ? k : best;
	}
	for (int d = 32; d >= 1; d >>= 1) {
		const uint64_t o = __shfl_xor(best, d);
		best = o < best ? o : best;
		flags |= __shfl_xor(flags, d); used |= __shfl_xor(used, d);
	}
	if (lane == 0) { pb.verdict[0] = best == ~(uint64_t) 0 ? 0u : (uint32_t) best; pb.verdict[1] = flags; pb.verdict[2] = used; pb.verdict[3] = (uint32_t) pb.class_start[27]; }
}

void launch_plan_build(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nframes, int32_t nlf, int32_t max_lf_cells, hipStream_t stream) {
	if (nframes <= 0 || nlf <= 0) return;
	hipLaunchKernelGGL(k_plan_place, dim3((unsigned) ((nlf + 63) / 64)), dim3(64), 0, stream, builds, lfs, nlf);
	hipLaunchKernelGGL(k_plan_scan, dim3((unsigned) ((nframes + 63) / 64)), dim3(64), 0, stream, builds, nframes);
	hipLaunchKernelGGL(k_plan_emit, dim3((unsigned) ((max_lf_cells + 255) / 256), (unsigned) nlf), dim3(256), 0, stream, builds, lfs);
}

void launch_plan_verdict(const DevPlanBuild *builds, const DevPlan *plans, int32_t nframes, hipStream_t stream) {
	if (nframes > 0) hipLaunchKernelGGL(k_plan_verdict, dim3((unsigned) nframes), dim3(64), 0, stream, builds, plans);
}

} // namespace j40hip
