// j40_amd/csrc/device/modular_kernels.hip -- HIP kernels of the Modular path for gfx950.
//
//   k_modular_sections   K3: per-pixel MA-tree walk + prediction + entropy decode, one section
//                        (LfGlobal's channel data or one group rectangle) per wavefront
//                        (replaces j40__modular_channel, j40.h:4127, as driven by j40__lf_global,
//                        j40.h:6334, and j40__pass_group, j40.h:7024-7033)
//   k_inverse_rct        K4: inverse reversible colour transform, one lane per pixel (j40.h:4318)
//   k_inverse_palette*   K4: palette look-up (one lane per pixel) or, with delta prediction, the
//                        serial form (j40.h:4402)
//   k_pack_planes        K5: int16 planes -> clamped RGBA u8x4 (j40.h:7910)
//
// Integer work throughout: results are bit-exact with the reference.
#include <hip/hip_runtime.h>
#include "modular_dev.h"
#include "kernels.h"

namespace j40hip {

__global__ void __launch_bounds__(64) k_modular_sections(DevModPlan plan) {
	if (threadIdx.x != 0) return;
	const int32_t s = blockIdx.x;
	plan.status[s] = decode_modular_section(plan, s);
}

__global__ void __launch_bounds__(256) k_inverse_rct(int16_t *a, int16_t *b, int16_t *c, size_t n, int32_t type7) {
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		int16_t p0 = a[i], p1 = b[i], p2 = c[i];
		inverse_rct_pixel(type7, p0, p1, p2);
		a[i] = p0; b[i] = p1; c[i] = p2;
	}
}

// output channel i of a palette without delta prediction; dst may alias idx (last channel, in place)
__global__ void __launch_bounds__(256) k_inverse_palette_plain(const int16_t *idx, const int16_t *palrow, int16_t *dst, size_t n, int32_t i, int32_t nb_colours, int32_t bpp) {
	for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t) gridDim.x * blockDim.x)
		dst[k] = palette_value(idx[k], i, palrow, nb_colours, bpp);
}

// palette with delta prediction: a delta entry is added to a prediction from already reconstructed
// neighbours of the same output channel, which makes the channel strictly sequential (j40.h:4469-4476)
__global__ void __launch_bounds__(64) k_inverse_palette_predicted(const int16_t *idx, const int16_t *pal, int32_t pal_stride, int16_t *const *dst, int32_t num_c,
		int32_t width, int32_t height, int32_t nb_colours, int32_t nb_deltas, int32_t d_pred, int32_t bpp, const int8_t *wpp, int32_t *wp_scratch, uint32_t *status) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	ModWP wp;
	wp.on = d_pred == 6; wp.width = width; wp.errors = wp_scratch;
	wp.p1 = wpp[0]; wp.p2 = wpp[1];
	for (int i = 0; i < 5; ++i) wp.p3[i] = wpp[2 + i];
	for (int i = 0; i < 4; ++i) wp.w[i] = wpp[7 + i];
	uint32_t err = 0;
	for (int32_t i = 0; i < num_c; ++i) {
		const int16_t *palrow = nb_colours > 0 ? pal + (size_t) i * (size_t) pal_stride : nullptr;
		int16_t *out = dst[i];
		if (wp.on) for (int32_t k = 0; k < 2 * width * 5; ++k) wp.errors[k] = 0;
		for (int k = 0; k < 5; ++k) wp.pred[k] = 0;
		wp.trueerrw = wp.trueerrn = wp.trueerrnw = wp.trueerrne = 0;
		for (int32_t y = 0; y < height; ++y) {
			const int16_t *idxline = idx + (size_t) y * (size_t) width;
			int16_t *line = out + (size_t) y * (size_t) width;
			for (int32_t x = 0; x < width; ++x) {
				const int16_t index = idxline[x];
				const bool is_delta = index < nb_deltas;
				int16_t val = palette_value(index, i, palrow, nb_colours, bpp);
				const ModNeigh p = mod_neighbours(line, width, width, x, y);
				wp_before(wp, x, y, p);
				if (is_delta) val = (int16_t) (val + mod_predict(d_pred, wp, p, &err));
				wp_after(wp, x, y, val);
				line[x] = val;
			}
		}
	}
	if (err) *status = err;
}

__global__ void __launch_bounds__(256) k_pack_planes(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t width, int32_t height, int32_t bpp, uint8_t *rgba, size_t stride_bytes) {
	const size_t n = (size_t) width * (size_t) height;
	const int32_t opaque = (1 << bpp) - 1;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		const size_t y = i / (size_t) width, x = i - y * (size_t) width;
		*(uint32_t *) (rgba + y * stride_bytes + x * 4) = pack_rgba8(r[i], g[i], b[i], a ? a[i] : opaque, bpp);
	}
}

static unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned) (b < 1 ? 1 : b > 8192 ? 8192 : b); }

void launch_modular_sections(const DevModPlan &plan, int32_t num_sections, hipStream_t stream) {
	if (num_sections > 0) hipLaunchKernelGGL(k_modular_sections, dim3((unsigned) num_sections), dim3(64), 0, stream, plan);
}
void launch_inverse_rct(int16_t *a, int16_t *b, int16_t *c, size_t n, int32_t type7, hipStream_t stream) {
	if (n) hipLaunchKernelGGL(k_inverse_rct, dim3(grid_for(n)), dim3(256), 0, stream, a, b, c, n, type7);
}
void launch_inverse_palette_plain(const int16_t *idx, const int16_t *palrow, int16_t *dst, size_t n, int32_t i, int32_t nb_colours, int32_t bpp, hipStream_t stream) {
	if (n) hipLaunchKernelGGL(k_inverse_palette_plain, dim3(grid_for(n)), dim3(256), 0, stream, idx, palrow, dst, n, i, nb_colours, bpp);
}
void launch_inverse_palette_predicted(const int16_t *idx, const int16_t *pal, int32_t pal_stride, int16_t *const *dst_dev, int32_t num_c, int32_t width, int32_t height,
		int32_t nb_colours, int32_t nb_deltas, int32_t d_pred, int32_t bpp, const int8_t *wpp_dev, int32_t *wp_scratch, uint32_t *status, hipStream_t stream) {
	hipLaunchKernelGGL(k_inverse_palette_predicted, dim3(1), dim3(64), 0, stream, idx, pal, pal_stride, dst_dev, num_c, width, height, nb_colours, nb_deltas, d_pred, bpp, wpp_dev, wp_scratch, status);
}
void launch_pack_planes(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t width, int32_t height, int32_t bpp, uint8_t *rgba, size_t stride, hipStream_t stream) {
	hipLaunchKernelGGL(k_pack_planes, dim3(grid_for((size_t) width * (size_t) height)), dim3(256), 0, stream, r, g, b, a, width, height, bpp, rgba, stride);
}

} // namespace j40hip
