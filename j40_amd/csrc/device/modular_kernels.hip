// j40_amd/csrc/device/modular_kernels.hip -- HIP kernels of the Modular path for gfx950.
//
//   k_modular_sections   K3: per-pixel MA-tree walk + prediction + entropy decode, one section
//                        (LfGlobal's channel data or one group rectangle) per wavefront
//                        (replaces j40__modular_channel, j40.h:4127, as driven by j40__lf_global,
//                        j40.h:6334, and j40__pass_group, j40.h:7024-7033)
//   k_inverse_rct        K4: inverse reversible colour transform, one lane per pixel (j40.h:4318)
//   k_inverse_palette*   K4: palette look-up (one lane per pixel) or, with delta prediction, the
//                        serial form (j40.h:4402)
//   k_unsqueeze_h/_v     K4: one inverse Squeeze step (ISO 18181-1; squeeze_dev.h): an average and a residual channel
//                        joined along rows / columns. The recurrence runs along the squeezed axis, lines across it are
//                        independent: one lane per line, rows staged through LDS so that HBM sees whole 128-byte pieces
//   k_pack_planes        K5: int16 planes -> clamped RGBA u8x4 (j40.h:7910)
//
// Integer work throughout: results are bit-exact with the reference.
#include <hip/hip_runtime.h>
#include "modular_dev.h"
#include "squeeze_dev.h"
#include "kernels.h"

namespace j40hip {

// One section per wavefront, one wavefront per workgroup. Every lane runs the same (wave-uniform) decoder, so the serial
// per-pixel loop lives on the scalar unit like k_hf_entropy's. IN_LDS: the MA tree, the code spec's tables, the three most
// recent rows of the channel being decoded and the weighted predictor's error rows all live in LDS (the usual case; the
// template keeps their address space static, a run-time choice would turn every access into a flat one), so the tree
// walk, the symbol decode and the neighbour fetch wait on LDS instead of L2 -- a sample that was just stored to the plane is
// not in the vector L1, and the previous pixel is the next one's W neighbour.
template <bool IN_LDS>
__global__ void __launch_bounds__(64) k_modular_sections(DevModPlan plan, int32_t first_section, int32_t rows_width, int32_t wp_width) {
	extern __shared__ __attribute__((aligned(16))) uint8_t mod_lds[];
	const int32_t lane = threadIdx.x;
	const int32_t s = first_section + (int32_t) blockIdx.x;
	const DevModSection &msec = plan.sections[s];
	if (msec.coop_idx >= 0 || msec.split) return;   // k_modular_coop's, modular_split.hip's
	if (msec.preset_status) { if (lane == 0) plan.status[s] = msec.preset_status; return; }
	const DevCodeSpec &spec = plan.spec[msec.spec_idx];
	ModTables t = mod_tables_in_hbm(plan, s);
	if (IN_LDS) {
		auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
		const int32_t tree_nodes = msec.tree_nodes;
		uint32_t off = 0;
		DevTreeNode *l_tree = (DevTreeNode *) mod_lds; off = align16((uint32_t) tree_nodes * (uint32_t) sizeof(DevTreeNode));
		uint32_t *l_map = (uint32_t *) (mod_lds + off); off = align16(off + (uint32_t) spec.num_dist + 4);
		DevCluster *l_clusters = (DevCluster *) (mod_lds + off); off = align16(off + (uint32_t) spec.num_clusters * (uint32_t) sizeof(DevCluster));
		uint8_t *l_tab = mod_lds + off; off = align16(off + (spec.use_prefix_code ? 4u : 8u) * spec.table_span);
		int32_t *l_rows = (int32_t *) (mod_lds + off); off = align16(off + 12u * (uint32_t) rows_width);   // rows_width = widest rectangle + 4 spare entries
		int32_t *l_wp = (int32_t *) (mod_lds + off);
		{ const uint4 *src = (const uint4 *) (plan.tree + msec.tree_off); uint4 *dst = (uint4 *) l_tree; for (int32_t i = lane; i < tree_nodes; i += 64) dst[i] = src[i]; }
		{ const uint32_t *src = (const uint32_t *) (plan.pool_u8 + spec.cluster_map_off); for (int32_t i = lane; i < (spec.num_dist + 3) / 4; i += 64) l_map[i] = src[i]; }   // 4-byte aligned table (plan_build.cpp)
		const DevCluster *csrc = plan.clusters + spec.cluster_off;
		const uint32_t base_off = csrc[0].table_off;
		for (int32_t i = lane; i < spec.num_clusters; i += 64) { DevCluster c = csrc[i]; c.table_off -= base_off; l_clusters[i] = c; }
		if (spec.use_prefix_code) { const int32_t *src = plan.pool_i32 + base_off; int32_t *dst = (int32_t *) l_tab; for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i]; }
		else { const uint64_t *src = plan.pool_u64 + base_off; uint64_t *dst = (uint64_t *) l_tab; for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i]; }
		t.tree = l_tree; t.cluster_map = (const uint8_t *) l_map; t.clusters = l_clusters; t.alias = (const uint64_t *) l_tab; t.prefix = (const int32_t *) l_tab;
		t.rows = l_rows; t.rows_width = rows_width;
		if (wp_width) { t.wp_errors = l_wp; t.wp_errors_width = wp_width; }
		__syncthreads();
	}
	const uint32_t err = decode_modular_section<true, IN_LDS>(plan, t, s);
	if (lane == 0) plan.status[s] = err;
}

// K3, throughput form: one section per LANE (64 sections per wavefront), the generic per-lane decoder with every table in
// HBM / L2. The lanes of a wavefront sit at different tree nodes and predictors, so the wavefront walks the union of their
// paths; what it buys is 64 sections per wavefront instead of one, i.e. frames with thousands of sections (or batches of
// LfGroup streams) fill the machine. Selected by the host when sections >> wave slots (launch_modular_sections).
__global__ void __launch_bounds__(64) k_modular_sections_lanes(DevModPlan plan, int32_t first_section, int32_t num_sections) {
	const int32_t s = first_section + (int32_t) (blockIdx.x * 64 + threadIdx.x);
	if (s >= first_section + num_sections) return;
	const DevModSection &msec = plan.sections[s];
	if (msec.coop_idx >= 0 || msec.split) return;
	if (msec.preset_status) { plan.status[s] = msec.preset_status; return; }
	const ModTables t = mod_tables_in_hbm(plan, s);
	plan.status[s] = decode_modular_section<false, false>(plan, t, s);
}

// one workgroup per section that lists transforms of its own; runs after K3 (kernel boundary = the planes are visible)
__global__ void __launch_bounds__(256) k_section_inverse_rcts(DevModPlan plan, int32_t first_section) {
	section_inverse_rcts(plan, first_section + (int32_t) blockIdx.x, (int32_t) threadIdx.x, 256);
}

// rows of a tightly packed sub-image plane over a rectangle of a frame plane (j40__combine_modular_from_pass_group, j40.h:3688)
__global__ void __launch_bounds__(256) k_paste_plane(const int16_t *src, int32_t w, size_t n, int16_t *dst, int32_t dst_stride) {
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		const size_t y = i / (size_t) w, x = i - y * (size_t) w;
		dst[y * (size_t) dst_stride + x] = src[i];
	}
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
		wp.blend_err_w = wp.blend_err_n = wp.blend_err_nw = wp.blend_err_ne = 0;
		for (int32_t y = 0; y < height; ++y) {
			const int16_t *idxline = idx + (size_t) y * (size_t) width;
			int16_t *line = out + (size_t) y * (size_t) width;
			for (int32_t x = 0; x < width; ++x) {
				const int16_t index = idxline[x];
				const bool is_delta = index < nb_deltas;
				int16_t val = palette_value(index, i, palrow, nb_colours, bpp);
				const ModNeigh p = mod_neighbours<false>(line, width, width, x, y);
				wp_before<false>(wp, x, y, p);
				if (is_delta) val = (int16_t) (val + mod_predict(d_pred, wp, p, &err));
				wp_after(wp, x, y, val);
				line[x] = val;
			}
		}
	}
	if (err) *status = err;
}

// horizontal step: out[y][2k], out[y][2k + 1] from avg[y][k], res[y][k]. One wavefront takes 64 rows; it walks along them in
// pieces of SQZ_CHUNK pairs: the pieces of all 64 rows are loaded with one coalesced 128-byte read per row into LDS (odd pitch
// in dwords: a lane walking its own row hits its own bank), each lane runs the recurrence over its row's piece, and the
// 2 * SQZ_CHUNK results per row go out as two coalesced 128-byte writes. `left` (the sample before the pair) stays in a register.
enum { SQZ_CHUNK = 64 };
__global__ void __launch_bounds__(64) k_unsqueeze_h(const int16_t *__restrict__ avg, const int16_t *__restrict__ res, int16_t *__restrict__ out, int32_t aw, int32_t ah, int32_t rw) {
	__shared__ int16_t s_avg[64][SQZ_CHUNK + 2], s_res[64][SQZ_CHUNK + 2], s_out[64][2 * SQZ_CHUNK + 2];
	const int32_t lane = threadIdx.x, y0 = (int32_t) blockIdx.x * 64, rows = min(64, ah - y0), ow = aw + rw;
	int32_t left = 0;
	for (int32_t x0 = 0; x0 < aw; x0 += SQZ_CHUNK) {
		for (int32_t r0 = 0; r0 < rows; r0 += 8) {   // eight rows' requests in flight before the first one is waited for
			int16_t va[8], vr[8], ve[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int32_t r = min(r0 + j, rows - 1);
				const size_t ra = (size_t) (y0 + r) * (size_t) aw, rr = (size_t) (y0 + r) * (size_t) rw;
				va[j] = x0 + lane < aw ? avg[ra + (size_t) (x0 + lane)] : (int16_t) 0;
				ve[j] = lane == 0 && x0 + SQZ_CHUNK < aw ? avg[ra + (size_t) (x0 + SQZ_CHUNK)] : (int16_t) 0;
				vr[j] = x0 + lane < rw ? res[rr + (size_t) (x0 + lane)] : (int16_t) 0;
			}
#pragma unroll
			for (int j = 0; j < 8; ++j) if (r0 + j < rows) {
				s_avg[r0 + j][lane] = va[j];
				if (lane == 0) s_avg[r0 + j][SQZ_CHUNK] = ve[j];
				s_res[r0 + j][lane] = vr[j];
			}
		}
		__syncthreads();
		if (lane < rows) {
			const int32_t n = min(SQZ_CHUNK, rw - x0);   // pairs in this piece (<= 0 behind the last residual)
			for (int32_t k = 0; k < n; ++k) {
				const int32_t a = s_avg[lane][k], next = x0 + k + 1 < aw ? (int32_t) s_avg[lane][k + 1] : a;
				int32_t p, q;
				unsqueeze_pair(a, s_res[lane][k], x0 + k > 0 ? left : a, next, &p, &q);
				s_out[lane][2 * k] = (int16_t) p; s_out[lane][2 * k + 1] = (int16_t) q;
				left = (int16_t) q;
			}
			if (aw > rw && aw - 1 >= x0 && aw - 1 < x0 + SQZ_CHUNK) s_out[lane][2 * (aw - 1 - x0)] = s_avg[lane][aw - 1 - x0];   // odd width: the last average passes through
		}
		__syncthreads();
		for (int32_t r = 0; r < rows; ++r) {   // (stores do not wait for one another)
			const size_t ro = (size_t) (y0 + r) * (size_t) ow + (size_t) (2 * x0);
			if (2 * x0 + lane < ow) out[ro + (size_t) lane] = s_out[r][lane];
			if (2 * x0 + 64 + lane < ow) out[ro + 64 + (size_t) lane] = s_out[r][64 + lane];
		}
		__syncthreads();
	}
}

// vertical step: one lane per column; consecutive lanes touch consecutive samples of a row, so every access is coalesced. The
// recurrence runs down the column; its inputs do not depend on it, so they are requested SQZ_AHEAD rows ahead of their use (a
// column of 8192 pairs would otherwise wait for memory 8192 times)
enum { SQZ_AHEAD = 8 };
__global__ void __launch_bounds__(256) k_unsqueeze_v(const int16_t *__restrict__ avg, const int16_t *__restrict__ res, int16_t *__restrict__ out, int32_t aw, int32_t ah, int32_t rh) {
	const int32_t x = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
	if (x >= aw) return;
	const size_t pitch = (size_t) aw;
	int32_t a_q[SQZ_AHEAD + 1], r_q[SQZ_AHEAD];   // a_q[j] = avg[k + j], r_q[j] = res[k + j] for the block of rows being processed
	int32_t left = 0;
	for (int32_t k0 = 0; k0 < rh; k0 += SQZ_AHEAD) {
#pragma unroll
		for (int j = 0; j <= SQZ_AHEAD; ++j) a_q[j] = k0 + j < ah ? (int32_t) avg[(size_t) (k0 + j) * pitch + (size_t) x] : 0;
#pragma unroll
		for (int j = 0; j < SQZ_AHEAD; ++j) r_q[j] = k0 + j < rh ? (int32_t) res[(size_t) (k0 + j) * pitch + (size_t) x] : 0;
#pragma unroll
		for (int j = 0; j < SQZ_AHEAD; ++j) {
			const int32_t k = k0 + j;
			if (k < rh) {
				const int32_t a = a_q[j], next = k + 1 < ah ? a_q[j + 1] : a;
				int32_t p, q;
				unsqueeze_pair(a, r_q[j], k > 0 ? left : a, next, &p, &q);
				out[(size_t) (2 * k) * pitch + (size_t) x] = (int16_t) p; out[(size_t) (2 * k + 1) * pitch + (size_t) x] = (int16_t) q;
				left = (int16_t) q;
			}
		}
	}
	if (ah > rh) out[(size_t) (2 * rh) * pitch + (size_t) x] = avg[(size_t) rh * pitch + (size_t) x];
}

__global__ void __launch_bounds__(256) k_pack_planes(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t width, int32_t height, int32_t bpp, uint8_t *rgba, size_t stride_bytes) {
	const size_t n = (size_t) width * (size_t) height;
	const int32_t opaque = (1 << bpp) - 1;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		const size_t y = i / (size_t) width, x = i - y * (size_t) width;
		*(uint32_t *) (rgba + y * stride_bytes + x * 4) = pack_rgba8(r[i], g[i], b[i], a ? a[i] : opaque, bpp);
	}
}

// the same over one rectangle of the picture (a group range's, sharded decodes): planes are `plane_width` samples wide
__global__ void __launch_bounds__(256) k_pack_planes_rect(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t plane_width, int32_t x0, int32_t y0, int32_t rw, int32_t rh, int32_t bpp, uint8_t *rgba, size_t stride_bytes) {
	const size_t n = (size_t) rw * (size_t) rh;
	const int32_t opaque = (1 << bpp) - 1;
	for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t) gridDim.x * blockDim.x) {
		const size_t y = k / (size_t) rw + (size_t) y0, x = k % (size_t) rw + (size_t) x0, i = y * (size_t) plane_width + x;
		*(uint32_t *) (rgba + y * stride_bytes + x * 4) = pack_rgba8(r[i], g[i], b[i], a ? a[i] : opaque, bpp);
	}
}

static unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned) (b < 1 ? 1 : b > 8192 ? 8192 : b); }

// sections [first_section, first_section + num_sections): the passes of a multi-pass frame are launched one after the other
void launch_modular_sections(const DevModPlan &plan, int32_t first_section, int32_t num_sections, const ModLaunchInfo &info, hipStream_t stream) {
	if (num_sections <= 0) return;
	// the sections the wave-cooperative kernel takes (modular_coop.hip), then the others
	if (info.quad_sections > 0) launch_modular_quad(plan, first_section, num_sections, info.quad_spec, info.quad_table_span, info.quad_width, stream);
	if (info.coop_width > 0 && info.quad_sections < info.coop_sections) launch_modular_coop(plan, first_section, num_sections, info.coop_width, stream);
	if (info.split_sections > 0) launch_modular_split(plan, first_section, num_sections, info, stream);   // (sections with a position-only tree: modular_split.hip)
	if (info.all_coop) return;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const uint32_t wp_bytes = info.uses_wp ? align16(40u * (uint32_t) info.max_width) : 0;
	const uint32_t lds = align16((uint32_t) info.num_tree_nodes * (uint32_t) sizeof(DevTreeNode)) + align16((uint32_t) info.num_dist + 4)
		+ align16((uint32_t) info.num_clusters * (uint32_t) sizeof(DevCluster)) + align16(info.table_bytes) + align16(12u * (uint32_t) (info.max_width + 4)) + wp_bytes + 64;
	static const int lanes_mode = [] { const char *e = getenv("J40HIP_K3_LANES"); return e ? atoi(e) : 0; }();
	if (lanes_mode == 1 || (lanes_mode == 2 && num_sections >= 1024)) {
		hipLaunchKernelGGL(k_modular_sections_lanes, dim3((unsigned) ((num_sections + 63) / 64)), dim3(64), 0, stream, plan, first_section, num_sections);
		return;
	}
	if (lds <= 156u * 1024u) {
		static bool configured = false;
		if (!configured) { (void) hipFuncSetAttribute((const void *) k_modular_sections<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
		hipLaunchKernelGGL(k_modular_sections<true>, dim3((unsigned) num_sections), dim3(64), lds, stream, plan, first_section, info.max_width + 4, info.uses_wp ? info.max_width : 0);
	} else {
		hipLaunchKernelGGL(k_modular_sections<false>, dim3((unsigned) num_sections), dim3(64), 0, stream, plan, first_section, 0, 0);
	}
}
void launch_section_inverse_rcts(const DevModPlan &plan, int32_t first_section, int32_t num_sections, hipStream_t stream) {
	if (num_sections > 0) hipLaunchKernelGGL(k_section_inverse_rcts, dim3((unsigned) num_sections), dim3(256), 0, stream, plan, first_section);
}
void launch_paste_plane(const int16_t *src, int32_t w, int32_t h, int16_t *dst, int32_t dst_stride, hipStream_t stream) {
	const size_t n = (size_t) w * (size_t) h;
	if (n) hipLaunchKernelGGL(k_paste_plane, dim3(grid_for(n)), dim3(256), 0, stream, src, w, n, dst, dst_stride);
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
// avg: aw x ah, res: rw x rh; horizontal: rh == ah, out is (aw + rw) x ah; vertical: rw == aw, out is aw x (ah + rh)
void launch_inverse_squeeze(const int16_t *avg, const int16_t *res, int16_t *out, int32_t aw, int32_t ah, int32_t rw, int32_t rh, bool horizontal, hipStream_t stream) {
	if (aw <= 0 || ah <= 0) return;
	if (horizontal) hipLaunchKernelGGL(k_unsqueeze_h, dim3((unsigned) ((ah + 63) / 64)), dim3(64), 0, stream, avg, res, out, aw, ah, rw);
	else hipLaunchKernelGGL(k_unsqueeze_v, dim3((unsigned) ((aw + 255) / 256)), dim3(256), 0, stream, avg, res, out, aw, ah, rh);
}
void launch_pack_planes(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t width, int32_t height, int32_t bpp, uint8_t *rgba, size_t stride, hipStream_t stream) {
	hipLaunchKernelGGL(k_pack_planes, dim3(grid_for((size_t) width * (size_t) height)), dim3(256), 0, stream, r, g, b, a, width, height, bpp, rgba, stride);
}

void launch_pack_planes_rect(const int16_t *r, const int16_t *g, const int16_t *b, const int16_t *a, int32_t plane_width, int32_t x0, int32_t y0, int32_t rw, int32_t rh, int32_t bpp, uint8_t *rgba, size_t stride, hipStream_t stream) {
	if (rw <= 0 || rh <= 0) return;
	hipLaunchKernelGGL(k_pack_planes_rect, dim3(grid_for((size_t) rw * (size_t) rh)), dim3(256), 0, stream, r, g, b, a, plane_width, x0, y0, rw, rh, bpp, rgba, stride);
}

} // namespace j40hip
