// j40_amd/csrc/device/restore_kernels.h -- the restoration filters' kernels; part of kernels.hip's translation unit
// (they share its constant tables: the colour tail's threshold table). The arithmetic is restore_dev.h's, per output sample.
//
//   k_epf_sigma       one lane per 8x8 cell of an LfGroup: reciprocal sigma from the sharpness map and the covering varblock's HfMul
//                     (j40__epf_recip_sigmas, j40.h:7374), into a frame-wide plane; the OR of all sharpness values for "shrp"
//   k_gaborish        one lane per sample and channel plane: 3x3 smoothing (j40__gaborish, j40.h:7271), out of place
//   k_epf<STEP>       a workgroup per tile of 32 x 16 samples staged in LDS with its halo, all three channels: the step's weighted sum
//                     (j40__epf_step, j40.h:7427), out of place;
//                     cells with a negative reciprocal sigma are copied through
//   k_xyb_to_rgba     the fused kernels' colour tail on the filtered planes: XYB -> linear -> sRGB -> u8x4 (j40.h:7204-7240, 7910-7962)
//
// Roofline: HBM. Algorithmic bytes per sample: Gaborish 12 read + 12 written, an EPF step the same (+ 4 per cell of sigma), the colour
// tail 12 read + 4 written; the neighbours a sample reads are its workgroup's or its neighbours' lines and come out of the caches.
// (An EPF step reads up to 3 x 12 x 5 x 2 samples per output sample: a workgroup's tile of 32 x 16 samples waits in LDS with its halo.)
#pragma once

struct XybPlanes { const float *p[3]; size_t pitch; };   // pitch in floats
struct XybPlanesOut { float *p[3]; size_t pitch; };

__global__ void __launch_bounds__(256) k_epf_sigma(const DevLfGroup *lf_groups, const int32_t *blocks, const float *vb_hfmul_inv, const int16_t *sharpness, RestoreParams p, float *sigma, uint32_t *sharp_or) {
	const DevLfGroup g = lf_groups[blockIdx.y];
	const int32_t cells = g.width8 * g.height8;
	uint32_t acc = 0;
	for (int32_t i = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x); i < cells; i += (int32_t) (gridDim.x * blockDim.x)) {
		const int32_t y8 = i / g.width8, x8 = i - y8 * g.width8;
		const int32_t sh = sharpness[g.cell_base + i];
		acc |= (uint32_t) (uint16_t) sh;
		const float inv = vb_hfmul_inv[g.vb_base + (blocks[g.cell_base + i] & 0xfffff)];
		sigma[(size_t) (g.top / 8 + y8) * (size_t) p.w8 + (size_t) (g.left / 8 + x8)] = epf_recip_sigma(p, sh, inv);
	}
	if (acc) atomicOr(sharp_or, acc);
}

__global__ void __launch_bounds__(256) k_gaborish(XybPlanes in, XybPlanesOut out, RestoreParams p) {
	const int32_t x = (int32_t) (blockIdx.x * 64 + (threadIdx.x & 63)), y = (int32_t) (blockIdx.y * 4 + (threadIdx.x >> 6)), c = (int32_t) blockIdx.z;
	if (x >= p.width || y >= p.height) return;
	const float *base = in.p[c];
	const float *n = base + (size_t) (y > 0 ? y - 1 : 0) * in.pitch, *l = base + (size_t) y * in.pitch, *s = base + (size_t) (y + 1 < p.height ? y + 1 : y) * in.pitch;
	out.p[c][(size_t) y * out.pitch + (size_t) x] = gaborish_sample(n, l, s, x, p.width, p.gab_w[c][0], p.gab_w[c][1], p.gab_w[c][2]);
}

// A workgroup's tile of the step's input with a halo of three samples all round (step 0's distances reach three rows up and down, its
// taps two columns sideways), the three channels, filled MIRRORED at the picture's edges: epf_sample then reads LDS by plain offset.
enum { EPF_TW = 32, EPF_TH = 16, EPF_HALO = 3, EPF_LW = EPF_TW + 2 * EPF_HALO, EPF_LH = EPF_TH + 2 * EPF_HALO };
struct EpfTile {
	const float *t; int32_t x0, y0;   // the tile's first sample (the halo's corner is at x0 - 3, y0 - 3)
	__device__ __forceinline__ float operator()(int32_t c, int32_t x, int32_t y) const { return t[(c * EPF_LH + (y - y0 + EPF_HALO)) * EPF_LW + (x - x0 + EPF_HALO)]; }
};
template <int STEP>
__global__ void __launch_bounds__(256) k_epf(XybPlanes in, XybPlanesOut out, const float *sigma, RestoreParams p) {
	__shared__ float tile[3 * EPF_LH * EPF_LW];
	const int32_t x0 = (int32_t) blockIdx.x * EPF_TW, y0 = (int32_t) blockIdx.y * EPF_TH;
	for (int32_t i = (int32_t) threadIdx.x; i < 3 * EPF_LH * EPF_LW; i += 256) {
		const int32_t c = i / (EPF_LH * EPF_LW), r = i - c * (EPF_LH * EPF_LW), ly = r / EPF_LW, lx = r - ly * EPF_LW;
		tile[i] = in.p[c][(size_t) restore_mirror(y0 + ly - EPF_HALO, p.height) * in.pitch + (size_t) restore_mirror(x0 + lx - EPF_HALO, p.width)];
	}
	__syncthreads();
	const EpfTile acc = {tile, x0, y0};
#pragma unroll
	for (int32_t k = 0; k < EPF_TW * EPF_TH / 256; ++k) {
		const int32_t i = (int32_t) threadIdx.x + 256 * k, x = x0 + (i & (EPF_TW - 1)), y = y0 + i / EPF_TW;
		if (x >= p.width || y >= p.height) continue;
		const float rs = sigma[(size_t) (y >> 3) * (size_t) p.w8 + (size_t) (x >> 3)];
		float v[3];
		if (rs < 0.0f) { for (int c = 0; c < 3; ++c) v[c] = acc(c, x, y); }   // the cell keeps its samples (j40.h:7521)
		else epf_sample<STEP>(acc, p, x, y, rs, v);
		for (int c = 0; c < 3; ++c) out.p[c][(size_t) y * out.pitch + (size_t) x] = v[c];
	}
}

__global__ void __launch_bounds__(256) k_xyb_to_rgba(XybPlanes in, const DevFrame *frame, int32_t width, int32_t height, uint8_t *rgba, size_t stride_bytes) {
	J40_STAGE_SRGB_THRESHOLDS(f);
	const ColourConsts cc = load_colour_consts(*frame);
	__syncthreads();
	const int32_t x = (int32_t) (blockIdx.x * 64 + (threadIdx.x & 63)), y = (int32_t) (blockIdx.y * 4 + (threadIdx.x >> 6));
	if (x >= width || y >= height) return;
	const size_t i = (size_t) y * in.pitch + (size_t) x;
	const uint32_t px = xyb_to_rgba8(in.p[0][i], in.p[1][i], in.p[2][i], cc, srgb_thr);
	__builtin_nontemporal_store(px, (uint32_t *) (rgba + (size_t) y * stride_bytes + (size_t) x * 4));
}

// `xyb`: three planes of width x height floats, `pitch` floats per row, one behind the other (what launch_vardct_frame_xyb wrote);
// `tmp`: as much again. Runs Gaborish (when p.gab_w[0][0] != 0 ... `gab`), then `epf_iters` steps; returns where the result lies (xyb or tmp).
float *launch_restoration(float *xyb, float *tmp, size_t pitch, const RestoreParams &p, bool gab, int32_t epf_iters, const float *sigma, hipStream_t stream) {
	const size_t plane = pitch * (size_t) p.height;
	float *cur = xyb, *other = tmp;
	auto planes_in = [&](const float *b) { XybPlanes q; for (int c = 0; c < 3; ++c) q.p[c] = b + (size_t) c * plane; q.pitch = pitch; return q; };
	auto planes_out = [&](float *b) { XybPlanesOut q; for (int c = 0; c < 3; ++c) q.p[c] = b + (size_t) c * plane; q.pitch = pitch; return q; };
	if (gab) {
		hipLaunchKernelGGL(k_gaborish, dim3((unsigned) ((p.width + 63) / 64), (unsigned) ((p.height + 3) / 4), 3), dim3(256), 0, stream, planes_in(cur), planes_out(other), p);
		std::swap(cur, other);
	}
	const dim3 grid((unsigned) ((p.width + EPF_TW - 1) / EPF_TW), (unsigned) ((p.height + EPF_TH - 1) / EPF_TH));
	if (epf_iters >= 3) { hipLaunchKernelGGL(k_epf<0>, grid, dim3(256), 0, stream, planes_in(cur), planes_out(other), sigma, p); std::swap(cur, other); }
	if (epf_iters >= 1) { hipLaunchKernelGGL(k_epf<1>, grid, dim3(256), 0, stream, planes_in(cur), planes_out(other), sigma, p); std::swap(cur, other); }
	if (epf_iters >= 2) { hipLaunchKernelGGL(k_epf<2>, grid, dim3(256), 0, stream, planes_in(cur), planes_out(other), sigma, p); std::swap(cur, other); }
	return cur;
}
void launch_epf_sigma(const DevPlan &plan, int32_t num_lf_groups, const int16_t *sharpness, const RestoreParams &p, float *sigma, uint32_t *sharp_or, hipStream_t stream) {
	hipLaunchKernelGGL(k_epf_sigma, dim3(64, (unsigned) num_lf_groups), dim3(256), 0, stream, plan.lf_groups, plan.blocks, plan.vb_hfmul_inv, sharpness, p, sigma, sharp_or);
}
// the sigma plane of a picture given cell by cell (known-answer tests: no plan behind it)
__global__ void k_epf_sigma_cells(const int16_t *sharpness, const float *hfmul_inv, RestoreParams p, float *sigma, uint32_t *sharp_or) {
	const int32_t i = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= p.w8 * p.h8) return;
	if (sharpness[i] & ~7) atomicOr(sharp_or, (uint32_t) (uint16_t) sharpness[i]);
	sigma[i] = epf_recip_sigma(p, sharpness[i], hfmul_inv[i]);
}
void launch_epf_sigma_cells(const int16_t *sharpness, const float *hfmul_inv, const RestoreParams &p, float *sigma, uint32_t *sharp_or, hipStream_t stream) {
	hipLaunchKernelGGL(k_epf_sigma_cells, dim3((unsigned) ((p.w8 * p.h8 + 255) / 256)), dim3(256), 0, stream, sharpness, hfmul_inv, p, sigma, sharp_or);
}
void launch_xyb_to_rgba(const float *xyb, size_t pitch, const DevFrame *frame_dev, int32_t width, int32_t height, uint8_t *rgba, size_t stride_bytes, hipStream_t stream) {
	XybPlanes q; for (int c = 0; c < 3; ++c) q.p[c] = xyb + (size_t) c * pitch * (size_t) height; q.pitch = pitch;
	hipLaunchKernelGGL(k_xyb_to_rgba, dim3((unsigned) ((width + 63) / 64), (unsigned) ((height + 3) / 4)), dim3(256), 0, stream, q, frame_dev, width, height, rgba, stride_bytes);
}
