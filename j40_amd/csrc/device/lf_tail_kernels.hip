// j40_amd/csrc/device/lf_tail_kernels.hip -- the tail of every LfGroup on the device (the part that is not
// an entropy-coded stream): dequantisation of the LF samples (j40__lf_quant, j40.h:6544-6590), adaptive LF smoothing
// (j40__smooth_lf, j40.h:6492-6540) and the LLF coefficients of every varblock (j40.h:6668-6683: the LF samples under the block
// through a forward DCT scaled for its place in the block's coefficient array, j40__forward_dct2d_scaled_for_llf, j40.h:5944).
// The host hands over the decoded integers; these kernels run once per frame on the upload stream and fill DevPlan::llf, which
// the coefficients -> pixels kernels read as before. Float arithmetic in the reference's order (-ffp-contract=off).
//
//   k_lf_dequant_smooth   one lane per 8x8 cell: a 3x3 stencil over the UNSMOOTHED dequantised samples of its LfGroup
//                         (edge cells pass through), three channels at once because the blend factor is shared (j40.h:6517-6529)
//   k_llf_small           one lane per varblock of up to 4x4 cells (every transform up to 32x32): the block's samples and the
//                         transform live in registers (fully unrolled, constant indices)
//   k_llf_large           one 64-lane workgroup per varblock with a 64-, 128- or 256-sized side: columns in parallel through LDS
#include <hip/hip_runtime.h>
#include "hf_dev.h"
#include "kernels.h"

namespace j40hip {

__constant__ float c_tail_half_secants[256];
__constant__ float c_tail_lf2llf[64];

void upload_lf_tail_tables(const float *half_secants, const float *lf2llf, hipStream_t stream) {
	(void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tail_half_secants), half_secants, sizeof(float) * 256, 0, hipMemcpyHostToDevice, stream);
	(void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tail_lf2llf), lf2llf, sizeof(float) * 64, 0, hipMemcpyHostToDevice, stream);
}

__device__ __forceinline__ void lf_dequant_smooth_cell(const DevPlan &plan, const DevLfGroup &gg, int32_t i, float *out0, float *out1, float *out2, int32_t smooth, float inv0, float inv1, float inv2) {
	const int32_t w8 = gg.width8, h8 = gg.height8;
	if (i >= w8 * h8) return;
	const int32_t y = i / w8, x = i - y * w8;
	const size_t at = (size_t) gg.cell_base + (size_t) i;
	float *out[3] = {out0, out1, out2};
	const float inv_m_lf[3] = {inv0, inv1, inv2};
	const bool edge = !smooth || w8 < 3 || h8 < 3 || y == 0 || x == 0 || y == h8 - 1 || x == w8 - 1;
	if (edge) {
		for (int c = 0; c < 3; ++c) out[c][at] = (float) plan.lfraw[c][at] * gg.mult_lf[c];
		return;
	}
	const float W0 = 0.05226273532324128f, W1 = 0.20345139757231578f, W2 = 0.0334829185968739f;
	float wa[3], centre[3], gap = 0.5f;
	for (int c = 0; c < 3; ++c) {
		const int16_t *p = plan.lfraw[c] + at;
		const float m = gg.mult_lf[c];
		const float n0 = (float) p[-w8 - 1] * m, n1 = (float) p[-w8] * m, n2 = (float) p[-w8 + 1] * m;
		const float l0 = (float) p[-1] * m, l1 = (float) p[0] * m, l2 = (float) p[1] * m;
		const float s0 = (float) p[w8 - 1] * m, s1 = (float) p[w8] * m, s2 = (float) p[w8 + 1] * m;
		wa[c] = (n0 * W2 + n1 * W1 + n2 * W2) + (l0 * W1 + l1 * W0 + l2 * W1) + (s0 * W2 + s1 * W1 + s2 * W2);
		centre[c] = l1;
		const float diff = fabsf(wa[c] - l1) * inv_m_lf[c];
		if (gap < diff) gap = diff;
	}
	gap = 3.0f - 4.0f * gap;
	gap = 0.0f > gap ? 0.0f : gap;
	for (int c = 0; c < 3; ++c) out[c][at] = (wa[c] - centre[c]) * gap + centre[c];
}

__global__ void __launch_bounds__(256) k_lf_dequant_smooth(DevPlan plan, float *out0, float *out1, float *out2, int32_t smooth, float inv0, float inv1, float inv2) {
	lf_dequant_smooth_cell(plan, plan.lf_groups[blockIdx.y], (int32_t) (blockIdx.x * blockDim.x + threadIdx.x), out0, out1, out2, smooth, inv0, inv1, inv2);
}
// every LfGroup of a batch (plan_kernels.hip: DevBatchLf)
__global__ void __launch_bounds__(256) k_lf_dequant_smooth_batch(const DevPlan *plans, const DevPlanBuild *builds, const DevBatchLf *lfs) {
	const DevBatchLf w = lfs[blockIdx.y];
	const DevPlan &plan = plans[w.frame];
	const DevPlanBuild &pb = builds[w.frame];
	float *lfs0 = pb.lf_scratch;
	lf_dequant_smooth_cell(plan, plan.lf_groups[w.lfg], (int32_t) (blockIdx.x * blockDim.x + threadIdx.x), lfs0, lfs0 + pb.cells, lfs0 + 2 * (size_t) pb.cells, pb.lf_smooth, pb.inv_m_lf[0], pb.inv_m_lf[1], pb.inv_m_lf[2]);
}

// forward DCT of length 1 << T over elements S apart, in the reference's order (j40__forward_dct_core, j40.h:5760-5800):
// the result lands in `out`, `in` is used as workspace
template <int T, int S> struct FwdDct {
	static __device__ __forceinline__ void run(float *out, float *in) {
		constexpr int N = 1 << T;
#pragma unroll
		for (int i = 0; i < N / 2; ++i) {
			const float x = in[i * S], y = in[(N - i - 1) * S];
			out[i * S] = x + y;
			out[(N / 2 + i) * S] = (x - y) * c_tail_half_secants[N / 2 + i];
		}
		FwdDct<T - 1, S>::run(in, out);
		FwdDct<T - 1, S>::run(in + N / 2 * S, out + N / 2 * S);
#pragma unroll
		for (int i = 0; i < N / 2; ++i) out[i * 2 * S] = in[i * S];
		out[S] = 1.4142135623730951f * in[N / 2 * S] + in[(N / 2 + 1) * S];
#pragma unroll
		for (int i = 1; i < N / 2 - 1; ++i) out[(i * 2 + 1) * S] = in[(N / 2 + i) * S] + in[(N / 2 + i + 1) * S];
		out[(N - 1) * S] = in[(N - 1) * S];
	}
};
template <int S> struct FwdDct<1, S> { static __device__ __forceinline__ void run(float *out, float *in) { const float x = in[0], y = in[S]; out[0] = x + y; out[S] = x - y; } };
template <int S> struct FwdDct<0, S> { static __device__ __forceinline__ void run(float *out, float *in) { out[0] = in[0]; } };

// the LLF coefficients of one varblock of (1 << LR) x (1 << LC) cells; buf: its samples row-major, result in place
template <int LR, int LC> __device__ __forceinline__ void llf_block(float *buf) {
	constexpr int R = 1 << LR, C = 1 << LC;
	float tmp[R * C];
#pragma unroll
	for (int r = 0; r < C; ++r) FwdDct<LR, C>::run(tmp + r, buf + r);                 // along the rows' direction, per column
#pragma unroll
	for (int y = 0; y < R; ++y)
#pragma unroll
		for (int x = 0; x < C; ++x) buf[x * R + y] = tmp[y * C + x];
#pragma unroll
	for (int r = 0; r < R; ++r) FwdDct<LC, R>::run(tmp + r, buf + r);                 // tmp is [C][R]
#pragma unroll
	for (int y = 0; y < C; ++y)
#pragma unroll
		for (int x = 0; x < R; ++x) tmp[y * R + x] *= c_tail_lf2llf[R + x] * c_tail_lf2llf[C + y];
	if (LC > LR) {
#pragma unroll
		for (int y = 0; y < C; ++y)
#pragma unroll
			for (int x = 0; x < R; ++x) buf[x * C + y] = tmp[y * R + x];
	} else {
#pragma unroll
		for (int k = 0; k < R * C; ++k) buf[k] = tmp[k];
	}
}

template <int LR, int LC> __device__ __forceinline__ void llf_small_case(const float *lf, size_t cell, int32_t w8, float *llf) {
	constexpr int R = 1 << LR, C = 1 << LC;
	float buf[R * C];
#pragma unroll
	for (int i = 0; i < R; ++i)
#pragma unroll
		for (int j = 0; j < C; ++j) buf[i * C + j] = lf[cell + (size_t) i * (size_t) w8 + (size_t) j];
	if (R * C > 1) llf_block<LR, LC>(buf);
#pragma unroll
	for (int k = 0; k < R * C; ++k) llf[k] = buf[k];
}

// blocks with a side of 64 or more are left to k_llf_large
__device__ __forceinline__ void llf_small_one(const DevPlan &plan, const DevVarblock &vb, const float *lf0, const float *lf1, const float *lf2, float *llf0, float *llf1, float *llf2) {
	const int32_t log_rows = DEV_DCT_SELECT[vb.dctsel][0], log_columns = DEV_DCT_SELECT[vb.dctsel][1];
	if (log_rows > 5 || log_columns > 5) return;
	const DevLfGroup gg = plan.lf_groups[(uint32_t) vb.pad[0] | ((uint32_t) vb.pad[1] << 8) | ((uint32_t) vb.pad[2] << 16)];
	const size_t cell = (size_t) gg.cell_base + (size_t) ((vb.py - gg.top) >> 3) * (size_t) gg.width8 + (size_t) ((vb.px - gg.left) >> 3);
	const float *lf[3] = {lf0, lf1, lf2};
	float *llf[3] = {llf0, llf1, llf2};
	for (int c = 0; c < 3; ++c) {
		float *dst = llf[c] + vb.llf_base;
		switch ((log_rows - 3) * 3 + (log_columns - 3)) {
		case 0: llf_small_case<0, 0>(lf[c], cell, gg.width8, dst); break;
		case 1: llf_small_case<0, 1>(lf[c], cell, gg.width8, dst); break;
		case 2: llf_small_case<0, 2>(lf[c], cell, gg.width8, dst); break;
		case 3: llf_small_case<1, 0>(lf[c], cell, gg.width8, dst); break;
		case 4: llf_small_case<1, 1>(lf[c], cell, gg.width8, dst); break;
		case 5: llf_small_case<1, 2>(lf[c], cell, gg.width8, dst); break;
		case 6: llf_small_case<2, 0>(lf[c], cell, gg.width8, dst); break;
		case 7: llf_small_case<2, 1>(lf[c], cell, gg.width8, dst); break;
		default: llf_small_case<2, 2>(lf[c], cell, gg.width8, dst); break;
		}
	}
}

// list: the frame's varblocks (any order)
__global__ void __launch_bounds__(256) k_llf_small(DevPlan plan, const DevVarblock *list, int32_t count, const float *lf0, const float *lf1, const float *lf2, float *llf0, float *llf1, float *llf2) {
	const int32_t v = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
	if (v >= count) return;
	llf_small_one(plan, list[v], lf0, lf1, lf2, llf0, llf1, llf2);
}
// every frame of a batch (blockIdx.y); the number of varblocks is the device's to know (DevPlanBuild::class_start[27])
__global__ void __launch_bounds__(256) k_llf_small_batch(const DevPlan *plans, const DevPlanBuild *builds) {
	const DevPlan &plan = plans[blockIdx.y];
	const DevPlanBuild &pb = builds[blockIdx.y];
	const int32_t v = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
	if (v >= pb.class_start[27]) return;
	const float *lfs = pb.lf_scratch;
	llf_small_one(plan, pb.vb_sorted[v], lfs, lfs + pb.cells, lfs + 2 * (size_t) pb.cells, const_cast<float *>(plan.llf[0]), const_cast<float *>(plan.llf[1]), const_cast<float *>(plan.llf[2]));
}

// run-time form of FwdDct for the large blocks (same arithmetic; elements `stride` apart, in LDS)
__device__ void fwd_dct_rt(float *out, float *in, int32_t t, int32_t stride) {
	const int32_t N = 1 << t;
	if (t == 0) { out[0] = in[0]; return; }
	if (t == 1) { const float x = in[0], y = in[stride]; out[0] = x + y; out[stride] = x - y; return; }
	for (int32_t i = 0; i < N / 2; ++i) {
		const float x = in[i * stride], y = in[(N - i - 1) * stride];
		out[i * stride] = x + y;
		out[(N / 2 + i) * stride] = (x - y) * c_tail_half_secants[N / 2 + i];
	}
	fwd_dct_rt(in, out, t - 1, stride);
	fwd_dct_rt(in + N / 2 * stride, out + N / 2 * stride, t - 1, stride);
	for (int32_t i = 0; i < N / 2; ++i) out[i * 2 * stride] = in[i * stride];
	out[stride] = 1.4142135623730951f * in[N / 2 * stride] + in[(N / 2 + 1) * stride];
	for (int32_t i = 1; i < N / 2 - 1; ++i) out[(i * 2 + 1) * stride] = in[(N / 2 + i) * stride] + in[(N / 2 + i + 1) * stride];
	out[(N - 1) * stride] = in[(N - 1) * stride];
}

__device__ void llf_large_one(const DevPlan &plan, const DevVarblock &vb, const float *lf0, const float *lf1, const float *lf2, float *llf0, float *llf1, float *llf2, float *buf, float *tmp, int32_t lane) {
	const int32_t lr = DEV_DCT_SELECT[vb.dctsel][0] - 3, lc = DEV_DCT_SELECT[vb.dctsel][1] - 3, R = 1 << lr, C = 1 << lc;
	const DevLfGroup gg = plan.lf_groups[(uint32_t) vb.pad[0] | ((uint32_t) vb.pad[1] << 8) | ((uint32_t) vb.pad[2] << 16)];
	const size_t cell = (size_t) gg.cell_base + (size_t) ((vb.py - gg.top) >> 3) * (size_t) gg.width8 + (size_t) ((vb.px - gg.left) >> 3);
	const float *lf[3] = {lf0, lf1, lf2};
	float *llf[3] = {llf0, llf1, llf2};
	for (int c = 0; c < 3; ++c) {
		for (int32_t k = lane; k < R * C; k += 64) buf[k] = lf[c][cell + (size_t) (k / C) * (size_t) gg.width8 + (size_t) (k % C)];
		__syncthreads();
		for (int32_t r = lane; r < C; r += 64) fwd_dct_rt(tmp + r, buf + r, lr, C);
		__syncthreads();
		for (int32_t k = lane; k < R * C; k += 64) { const int32_t y = k / C, x = k % C; buf[x * R + y] = tmp[y * C + x]; }
		__syncthreads();
		for (int32_t r = lane; r < R; r += 64) fwd_dct_rt(tmp + r, buf + r, lc, R);
		__syncthreads();
		float *dst = llf[c] + vb.llf_base;
		for (int32_t k = lane; k < R * C; k += 64) {
			const int32_t y = k / R, x = k % R;   // tmp is [C][R]
			const float v = tmp[y * R + x] * (c_tail_lf2llf[R + x] * c_tail_lf2llf[C + y]);
			dst[lc > lr ? x * C + y : k] = v;
		}
		__syncthreads();
	}
}

// list: the varblocks of the classes with a 64-, 128- or 256-sized side (contiguous in the sorted list)
__global__ void __launch_bounds__(64) k_llf_large(DevPlan plan, const DevVarblock *list, int32_t count, const float *lf0, const float *lf1, const float *lf2, float *llf0, float *llf1, float *llf2) {
	__shared__ float buf[1024], tmp[1024];
	if ((int32_t) blockIdx.x >= count) return;
	llf_large_one(plan, list[blockIdx.x], lf0, lf1, lf2, llf0, llf1, llf2, buf, tmp, threadIdx.x);
}
// every frame of a batch (blockIdx.y): the workgroups of a frame share out its large blocks, however many the device found
__global__ void __launch_bounds__(64) k_llf_large_batch(const DevPlan *plans, const DevPlanBuild *builds) {
	__shared__ float buf[1024], tmp[1024];
	const DevPlan &plan = plans[blockIdx.y];
	const DevPlanBuild &pb = builds[blockIdx.y];
	const float *lfs = pb.lf_scratch;
	for (int32_t v = pb.class_start[18] + (int32_t) blockIdx.x; v < pb.class_start[27]; v += (int32_t) gridDim.x)
		llf_large_one(plan, pb.vb_sorted[v], lfs, lfs + pb.cells, lfs + 2 * (size_t) pb.cells, const_cast<float *>(plan.llf[0]), const_cast<float *>(plan.llf[1]), const_cast<float *>(plan.llf[2]), buf, tmp, threadIdx.x);
}

// the whole tail of one frame on `stream`: lfs = scratch of three planes of `cells` floats (dequantised + smoothed samples)
void launch_lf_tail(const DevPlan &plan, int32_t num_lf_groups, int32_t max_cells, size_t cells, float *lfs, const DevVarblock *sorted, int32_t count, int32_t first_large, int32_t smooth,
		const float inv_m_lf[3], hipStream_t stream) {
	if (num_lf_groups <= 0 || max_cells <= 0) return;
	hipLaunchKernelGGL(k_lf_dequant_smooth, dim3((unsigned) ((max_cells + 255) / 256), (unsigned) num_lf_groups), dim3(256), 0, stream, plan, lfs, lfs + cells, lfs + 2 * cells, smooth, inv_m_lf[0], inv_m_lf[1], inv_m_lf[2]);
	float *llf0 = const_cast<float *>(plan.llf[0]), *llf1 = const_cast<float *>(plan.llf[1]), *llf2 = const_cast<float *>(plan.llf[2]);
	if (count > 0) hipLaunchKernelGGL(k_llf_small, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, stream, plan, sorted, count, lfs, lfs + cells, lfs + 2 * cells, llf0, llf1, llf2);
	if (count > first_large) hipLaunchKernelGGL(k_llf_large, dim3((unsigned) (count - first_large)), dim3(64), 0, stream, plan, sorted + first_large, count - first_large, lfs, lfs + cells, lfs + 2 * cells, llf0, llf1, llf2);
}

// the tails of every frame of a batch: nlf LfGroups in all, the largest of max_lf_cells cells; the largest frame of max_frame_cells
void launch_lf_tail_batch(const DevPlan *plans, const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nframes, int32_t nlf, int32_t max_lf_cells, size_t max_frame_cells, hipStream_t stream) {
	if (nframes <= 0 || nlf <= 0 || max_lf_cells <= 0) return;
	hipLaunchKernelGGL(k_lf_dequant_smooth_batch, dim3((unsigned) ((max_lf_cells + 255) / 256), (unsigned) nlf), dim3(256), 0, stream, plans, builds, lfs);
	hipLaunchKernelGGL(k_llf_small_batch, dim3((unsigned) ((max_frame_cells + 255) / 256), (unsigned) nframes), dim3(256), 0, stream, plans, builds);
	hipLaunchKernelGGL(k_llf_large_batch, dim3(32, (unsigned) nframes), dim3(64), 0, stream, plans, builds);
}

} // namespace j40hip
