// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY. Compiles the *device* functions of the hot path
// (j40_amd/csrc/device/*_dev.h) for the CPU and runs them with the same orchestration as the HIP
// kernels, so their logic can be checked here (no GPU in the build container) against oracle/_ref
// before spending GPU minutes. Never linked into the product library.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../j40_amd/csrc/plan_build.hpp"
#include "../../j40_amd/csrc/tables.hpp"
#include "../../j40_amd/csrc/device/hf_dev.h"
#include <algorithm>
#include "../../j40_amd/csrc/device/hf_lanes_dev.h"
#include "../../j40_amd/csrc/device/hf_uni_dev.h"
#include "../../j40_amd/csrc/device/vardct_dev.h"
#include "../../j40_amd/csrc/device/special8_dev.h"
#include "../../j40_amd/csrc/device/large_dev.h"
#include "../../j40_amd/csrc/device/k2_iter_dev.h"
#include "../../j40_amd/csrc/device/modular_dev.h"
#include "../../j40_amd/csrc/device/squeeze_dev.h"

using namespace j40hip;

namespace {

void idct_sweeps_host(float *A, float *B, int32_t t, int32_t ncols, int32_t stride_k, int32_t stride_col, const float *hs) {
	// sequential model of idct_sweeps in kernels.hip (same level structure)
	const int32_t N = 1 << t, half = N >> 1;
	if (t == 0) { for (int32_t w = 0; w < ncols; ++w) B[w * stride_col] = A[w * stride_col]; return; }
	for (int32_t d = 0; d <= t - 2; ++d) {
		const float *src = (d & 1) ? B : A; float *dst = (d & 1) ? A : B;
		const int32_t n = N >> d, hn = n >> 1;
		for (int32_t w = 0; w < ncols * half; ++w) {
			const int32_t col = w % ncols, j = w / ncols, o = (j / hn) * n, i = j % hn;
			const float *s = src + col * stride_col; float *q = dst + col * stride_col;
			q[(o + i) * stride_k] = s[(o + 2 * i) * stride_k];
			q[(o + hn + i) * stride_k] = i == 0 ? J40_SQRT2F * s[(o + 1) * stride_k] : s[(o + 2 * i - 1) * stride_k] + s[(o + 2 * i + 1) * stride_k];
		}
	}
	{
		const int32_t d = t - 1;
		const float *src = (d & 1) ? B : A; float *dst = (d & 1) ? A : B;
		for (int32_t w = 0; w < ncols * half; ++w) {
			const int32_t col = w % ncols, o = (w / ncols) * 2;
			const float p = src[col * stride_col + o * stride_k], q = src[col * stride_col + (o + 1) * stride_k];
			dst[col * stride_col + o * stride_k] = p + q;
			dst[col * stride_col + (o + 1) * stride_k] = p - q;
		}
	}
	for (int32_t d = t - 2; d >= 0; --d) {
		const float *src = (d & 1) ? B : A; float *dst = (d & 1) ? A : B;
		const int32_t n = N >> d, hn = n >> 1;
		for (int32_t w = 0; w < ncols * half; ++w) {
			const int32_t col = w % ncols, j = w / ncols, o = (j / hn) * n, i = j % hn;
			const float *s = src + col * stride_col; float *q = dst + col * stride_col;
			const float x = s[(o + i) * stride_k], y = s[(o + hn + i) * stride_k];
			const float ym = y * hs[hn + i];
			q[(o + i) * stride_k] = x + ym;
			q[(o + n - 1 - i) * stride_k] = x - ym;
		}
	}
}

// large_dev.h's passes on the CPU: the calls of a workgroup's 256 lanes one after the other, in an order that changes from phase to
// phase (a phase in which one lane read what another wrote would show)
struct HostExec {
	int phase = 0;
	template <class F> void run(F f) { if (phase++ & 1) for (int tid = 255; tid >= 0; --tid) f(tid, 256); else for (int tid = 0; tid < 256; ++tid) f(tid, 256); }
};

template <int N> void idct_rows(float *tile, int rows, int pitch, const float *hs) {  // along c for each r
	for (int r = 0; r < rows; ++r) { float x[N]; for (int k = 0; k < N; ++k) x[k] = tile[r * pitch + k]; Idct1D<N>::run(x, hs); for (int k = 0; k < N; ++k) tile[r * pitch + k] = x[k]; }
}
template <int N> void idct_cols(float *tile, int cols, int pitch, const float *hs) {  // along r for each x
	for (int c = 0; c < cols; ++c) { float x[N]; for (int k = 0; k < N; ++k) x[k] = tile[k * pitch + c]; Idct1D<N>::run(x, hs); for (int k = 0; k < N; ++k) tile[k * pitch + c] = x[k]; }
}
void idct_rows_dyn(float *t, int n, int rows, int pitch, const float *hs) {
	switch (n) { case 8: idct_rows<8>(t, rows, pitch, hs); break; case 16: idct_rows<16>(t, rows, pitch, hs); break; case 32: idct_rows<32>(t, rows, pitch, hs); break; default: idct_rows<64>(t, rows, pitch, hs); }
}
void idct_cols_dyn(float *t, int n, int cols, int pitch, const float *hs) {
	switch (n) { case 8: idct_cols<8>(t, cols, pitch, hs); break; case 16: idct_cols<16>(t, cols, pitch, hs); break; case 32: idct_cols<32>(t, cols, pitch, hs); break; default: idct_cols<64>(t, cols, pitch, hs); }
}

} // namespace

// group range for the next hostsim_decode calls (mirrors j40hip_frame_set_group_range; count < 0: every group)
static int64_t g_first_group = 0, g_group_count = -1;

// Modular frames: K3 / K4 / K5 device functions with the runtime's orchestration (device/runtime.hip)
static uint32_t hostsim_decode_modular(const Frame &fr, const uint8_t *cs, size_t cs_size, uint8_t *rgba) {
	HostModPlan hp;
	if (uint32_t e = build_modular_plan(fr, cs, cs_size, &hp)) return e;
	const int32_t nch = hp.frame.num_channels;
	std::vector<std::vector<int16_t>> store((size_t) nch);
	DevModPlan plan;
	memset(&plan, 0, sizeof plan);
	plan.frame = &hp.frame; plan.codestream = hp.codestream.data(); plan.pool_u8 = hp.pool_u8.data(); plan.pool_i32 = hp.pool_i32.data(); plan.pool_u64 = hp.pool_u64.data();
	plan.clusters = hp.clusters.data(); plan.spec = hp.specs.data(); plan.tree = hp.tree.data(); plan.sections = hp.sections.data();
	struct Ref { int16_t *p; int32_t w, h; };
	std::vector<Ref> planes;
	std::vector<DevPlaneRef> refs((size_t) nch);
	for (int32_t c = 0; c < nch; ++c) {
		store[(size_t) c].assign((size_t) std::max(hp.plane_w[(size_t) c], 0) * (size_t) std::max(hp.plane_h[(size_t) c], 0) + 1, 0);
		refs[(size_t) c] = DevPlaneRef{store[(size_t) c].data(), hp.plane_w[(size_t) c], hp.plane_h[(size_t) c], hp.plane_meta[(size_t) c], 0};
		planes.push_back({refs[(size_t) c].ptr, refs[(size_t) c].w, refs[(size_t) c].h});
	}
	plan.planes = refs.data(); plan.chan_rects = hp.chan_rects.data();
	std::vector<std::vector<int16_t>> sub_store(hp.sub_w.size());
	std::vector<DevSubPlane> subp(hp.sub_w.size());
	for (size_t k = 0; k < hp.sub_w.size(); ++k) { sub_store[k].assign((size_t) hp.sub_w[k] * (size_t) hp.sub_h[k] + 1, 0); subp[k] = DevSubPlane{sub_store[k].data(), hp.sub_w[k], hp.sub_h[k], hp.sub_meta[k], 0}; }
	plan.sub_planes = subp.data();
	std::vector<int32_t> wps((size_t) hp.sections.size() * (size_t) (2 * hp.frame.max_width * 5) + 16), window(hp.lz_window_size ? (size_t) hp.sections.size() * hp.lz_window_size : 0);
	std::vector<uint32_t> status(hp.sections.size() + 1, 0);
	plan.wp_scratch = hp.frame.tree_uses_wp ? wps.data() : nullptr;
	plan.lz_window = window.empty() ? nullptr : window.data(); plan.lz_window_size = hp.lz_window_size; plan.status = status.data();
	std::vector<int32_t> ring(3 * ((size_t) hp.frame.max_width + 4) + 8, 0x7fff0000), wperr(10 * (size_t) hp.frame.max_width + 8);
	// a group range (hostsim_set_group_range, mirrors j40hip_frame_set_group_range for Modular frames): LfGlobal's section and the
	// range's groups of every pass are decoded, and only the range's rectangles are written
	const int32_t per_pass = hp.sections_per_pass, lead = hp.frame.num_sections - per_pass * hp.num_passes;
	const bool ranged = g_group_count >= 0 && !(g_first_group == 0 && g_group_count == (int64_t) fr.fh.num_groups);
	if (ranged) {
		if (per_pass != (int32_t) fr.fh.num_groups) return ERR_TODO;
		for (const Transform &t : hp.transforms) if (t.kind == Transform::SQUEEZE || (t.kind == Transform::PALETTE && t.nb_deltas > 0)) return ERR_TODO;
	}
	auto in_range = [&](int32_t sct) { if (!ranged || sct < lead) return true; const int32_t g = (sct - lead) % per_pass; return g >= g_first_group && g < g_first_group + g_group_count; };
	for (int32_t sct = 0; sct < hp.frame.num_sections; ++sct) {
		if (!in_range(sct)) continue;
		if (hp.sections[(size_t) sct].preset_status) { status[(size_t) sct] = hp.sections[(size_t) sct].preset_status; continue; }
		ModTables mt = mod_tables_in_hbm(plan, sct);
		mt.rows = ring.data(); mt.rows_width = hp.frame.max_width + 4; mt.wp_errors = wperr.data(); mt.wp_errors_width = hp.frame.max_width;   // as the kernel lays them out in LDS
		status[(size_t) sct] = (sct & 1) ? decode_modular_section<false, true>(plan, mt, sct) : decode_modular_section<false, false>(plan, mt, sct);   // both neighbour sources
	}
	{   // like j40hip_frame_status: the reference reports the first failing section in file order (j40.h:5608)
		uint32_t first = 0, first_off = 0xffffffffu;
		for (size_t i = 0; i < hp.sections.size(); ++i) if (status[i] && hp.sections[i].byte_off < first_off) { first = status[i]; first_off = hp.sections[i].byte_off; }
		if (first) return first;
		if (status.back()) return status.back();
	}
	plan.local_rct = hp.local_rct.data();
	for (int32_t sct = 0; sct < hp.frame.num_sections; ++sct) if (in_range(sct)) for (int32_t lane = 0; lane < 3; ++lane) section_inverse_rcts(plan, sct, lane, 3);
	static const uint8_t PERM[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
	std::vector<std::vector<int16_t>> extra;
	extra.reserve(1024);
	// undoes `trs` last to first on the image `planes` (the frame, or the sub-image of a section with a palette of its own)
	auto undo = [&](std::vector<Ref> &planes, const std::vector<Transform> &trs, const int8_t *wpb) -> uint32_t {
	for (size_t ti = trs.size(); ti-- > 0; ) {
		const Transform &t = trs[ti];
		if (t.kind == Transform::RCT) {
			Ref c[3] = {planes[(size_t) t.begin_c], planes[(size_t) t.begin_c + 1], planes[(size_t) t.begin_c + 2]};
			const size_t n = (size_t) c[0].w * (size_t) c[0].h;
			for (size_t i = 0; i < n; ++i) inverse_rct_pixel(t.rct_type % 7, c[0].p[i], c[1].p[i], c[2].p[i]);
			for (int i = 0; i < 3; ++i) planes[(size_t) (t.begin_c + PERM[t.rct_type / 7][i])] = c[i];
		} else if (t.kind == Transform::SQUEEZE) {   // as the runtime schedules it: a new plane per squeezed channel (device/runtime.hip)
			const int32_t nc = (int32_t) planes.size(), end_c = t.begin_c + t.num_c, offset = t.in_place ? end_c : nc - t.num_c;
			for (int32_t c = t.begin_c; c < end_c; ++c) {
				const Ref avg = planes[(size_t) c], res = planes[(size_t) (offset + c - t.begin_c)];
				Ref out = {nullptr, t.horizontal ? avg.w + res.w : avg.w, t.horizontal ? avg.h : avg.h + res.h};
				extra.emplace_back((size_t) std::max(out.w, 0) * (size_t) std::max(out.h, 0) + 1, 0); out.p = extra.back().data();
				if (t.horizontal) for (int32_t y = 0; y < out.h; ++y)
					unsqueeze_line(avg.p + (size_t) y * (size_t) avg.w, 1, res.w > 0 ? res.p + (size_t) y * (size_t) res.w : avg.p, 1, avg.w, res.w, out.p + (size_t) y * (size_t) out.w, 1);
				else for (int32_t x = 0; x < out.w; ++x)
					unsqueeze_line(avg.p + x, avg.w, res.h > 0 ? res.p + x : avg.p, res.w, avg.h, res.h, out.p + x, out.w);
				planes[(size_t) c] = out;
			}
			planes.erase(planes.begin() + offset, planes.begin() + offset + t.num_c);
		} else {
			const int32_t first = t.begin_c + 1;
			const Ref idx = planes[(size_t) first], pal = planes[0];
			const size_t n = (size_t) idx.w * (size_t) idx.h;
			std::vector<Ref> outs;
			for (int32_t i = 0; i < t.num_c - 1; ++i) { extra.emplace_back(n + 1, 0); outs.push_back({extra.back().data(), idx.w, idx.h}); }
			outs.push_back(idx);
			ModWP wp;
			wp.on = t.nb_deltas > 0 && t.d_pred == 6; wp.width = idx.w; std::vector<int32_t> errs((size_t) 2 * (size_t) idx.w * 5 + 16, 0); wp.errors = errs.data();
			wp.p1 = wpb[0]; wp.p2 = wpb[1]; for (int i = 0; i < 5; ++i) wp.p3[i] = wpb[2 + i]; for (int i = 0; i < 4; ++i) wp.w[i] = wpb[7 + i];
			uint32_t err = 0;
			for (int32_t i = 0; i < t.num_c; ++i) {
				const int16_t *palrow = t.nb_colours > 0 ? pal.p + (size_t) i * (size_t) pal.w : nullptr;
				std::fill(errs.begin(), errs.end(), 0);
				for (int k = 0; k < 5; ++k) wp.pred[k] = 0;
				wp.blend_err_w = wp.blend_err_n = wp.blend_err_nw = wp.blend_err_ne = 0;
				for (int32_t y = 0; y < idx.h; ++y) for (int32_t x = 0; x < idx.w; ++x) {
					const int16_t index = idx.p[(size_t) y * (size_t) idx.w + (size_t) x];
					int16_t val = palette_value(index, i, palrow, t.nb_colours, fr.im.bpp);
					if (t.nb_deltas > 0) {
						int16_t *line = outs[(size_t) i].p + (size_t) y * (size_t) idx.w;
						const ModNeigh p = mod_neighbours(line, idx.w, idx.w, x, y);
						wp_before(wp, x, y, p);
						if (index < t.nb_deltas) val = (int16_t) (val + mod_predict(t.d_pred, wp, p, &err));
						wp_after(wp, x, y, val);
					}
					outs[(size_t) i].p[(size_t) y * (size_t) idx.w + (size_t) x] = val;
				}
			}
			if (err) return err;
			std::vector<Ref> next(planes.begin() + 1, planes.begin() + first);
			next.insert(next.end(), outs.begin(), outs.end());
			next.insert(next.end(), planes.begin() + first + 1, planes.end());
			planes.swap(next);
		}
	}
	return 0;
	};
	// sections with a palette of their own: their sub-image's transforms, then the paste over the section's rectangle (j40.h:7030-7032)
	for (const HostModPlan::SubImage &si : hp.sub_images) if (si.paste) {
		std::vector<Ref> sp;
		for (int32_t k = 0; k < si.num_planes; ++k) sp.push_back({subp[(size_t) (si.first_plane + k)].ptr, hp.sub_w[(size_t) (si.first_plane + k)], hp.sub_h[(size_t) (si.first_plane + k)]});
		if (uint32_t e = undo(sp, si.transforms, si.wp)) return e;
		const DevModSection &sec = hp.sections[(size_t) si.section];
		for (size_t c = 0; c < sp.size(); ++c) {
			const Ref &dst = planes[(size_t) sec.first_channel + c];
			for (int32_t y = 0; y < sp[c].h; ++y) memcpy(dst.p + (size_t) (sec.gy + y) * (size_t) dst.w + (size_t) sec.gx, sp[c].p + (size_t) y * (size_t) sp[c].w, sizeof(int16_t) * (size_t) sp[c].w);
		}
	}
	{ int8_t gwp[12]; const WPParams &gp = fr.gmodular.wp; gwp[0] = gp.p1; gwp[1] = gp.p2; for (int i = 0; i < 5; ++i) gwp[2 + i] = gp.p3[i]; for (int i = 0; i < 4; ++i) gwp[7 + i] = gp.w[i]; gwp[11] = 0;
	  if (uint32_t e = undo(planes, hp.transforms, gwp)) return e; }
	const int32_t W = hp.frame.width, H = hp.frame.height;
	if (planes.size() < 3) return ERR_TODO;
	const int32_t opaque = (1 << fr.im.bpp) - 1;
	int32_t rects[3][4] = {{0, 0, W, H}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	const int nr = ranged ? group_range_rects(g_first_group, g_group_count, W, H, fr.fh.group_size_shift, rects) : 1;
	for (int k = 0; k < nr; ++k) for (int32_t y = rects[k][1]; y < rects[k][3]; ++y) for (int32_t x = rects[k][0]; x < rects[k][2]; ++x) {
		const size_t i = (size_t) y * (size_t) W + (size_t) x;
		const uint32_t px = pack_rgba8(planes[0].p[i], planes[1].p[i], planes[2].p[i], hp.alpha_channel >= 0 ? planes[(size_t) hp.alpha_channel].p[i] : opaque, fr.im.bpp);
		memcpy(rgba + i * 4, &px, 4);
	}
	return 0;
}

extern "C" __attribute__((visibility("default"))) void hostsim_set_group_range(int64_t first, int64_t count) { g_first_group = first; g_group_count = count; }

// decodes a stream with the device functions on the CPU.
//   rgba: width*height*4 bytes; coeffs_out (optional): 3 arrays of total_cells*64 floats concatenated
// returns 0 or the first error code
extern "C" __attribute__((visibility("default"))) uint32_t hostsim_decode(const uint8_t *buf, size_t size, uint8_t *rgba, float *coeffs_out, int only_entropy) {
	Frame fr;
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	HostPlan hp;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		parse_frame(cs, cs_size, &fr, 1);
	} catch (const DecodeError &e) { return e.code; }
	if (fr.fh.is_modular) return hostsim_decode_modular(fr, cs, cs_size, rgba);
	if (uint32_t e = build_vardct_plan(fr, cs, cs_size, &hp)) return e;
	std::vector<float> coeff_store(3 * hp.coeff_floats, 0.0f);   // one allocation, plane c at c * coeff_floats (as on the device)
	float *coeffs[3] = {coeff_store.data(), coeff_store.data() + hp.coeff_floats, coeff_store.data() + 2 * hp.coeff_floats};
	std::vector<int8_t> nonzeros((size_t) hp.frame.num_groups * 32 * 32 * 3);
	std::vector<uint32_t> status(hp.sections.size(), 0);
	std::vector<int32_t> window(hp.lz_window_size ? (size_t) hp.frame.num_groups * hp.lz_window_size : 0);
	DevPlan plan;
	memset(&plan, 0, sizeof plan);
	plan.frame = &hp.frame; plan.codestream = hp.codestream.data();
	plan.pool_u8 = hp.pool_u8.data(); plan.pool_u16 = hp.pool_u16.data(); plan.pool_i32 = hp.pool_i32.data(); plan.pool_u64 = hp.pool_u64.data(); plan.pool_f32 = hp.pool_f32.data();
	plan.clusters = hp.clusters.data(); plan.coeff_specs = hp.coeff_specs.data(); plan.lf_groups = hp.lf_groups.data(); plan.sections = hp.sections.data();
	plan.block_ctx_map_off = hp.block_ctx_map_off;
	plan.group_blocks = hp.group_blocks.data(); plan.group_block_start = hp.group_block_start.data();
	plan.blocks = hp.blocks.data(); plan.lfindices = hp.lfindices.data();
	for (int c = 0; c < 3; ++c) { plan.llf[c] = hp.llf[c].data(); plan.coeffs[c] = coeffs[c]; }
	plan.coeff_stride = (uint32_t) hp.coeff_floats;
	// sparse coefficients (single-pass frames): event lists + per-block table, the table cleared like the runtime does
	std::vector<CoeffEvent> events(hp.ev_capacity + 1);
	std::vector<uint32_t> block_events(4 * hp.group_blocks.size() + 4, 0);
	if (hp.frame.sparse_coeffs) { plan.events = events.data(); plan.ev_range = hp.ev_range.data(); plan.block_events = block_events.data(); }
	plan.vb_coeffoff_qfidx = hp.vb_coeffoff_qfidx.data(); plan.vb_hfmul_inv = hp.vb_hfmul_inv.data();
	plan.xfromy = hp.xfromy.data(); plan.bfromy = hp.bfromy.data();
	plan.nonzeros = nonzeros.data(); plan.status = status.data();
	plan.lz_window = window.empty() ? nullptr : window.data(); plan.lz_window_size = hp.lz_window_size;
	std::vector<uint32_t> end_bits(status.size(), 0);
	plan.section_end_bit = hp.frame.sections_have_trailer ? end_bits.data() : nullptr;

	if (only_entropy & 16) {   // bit 4: the latency kernel's fast path (hf_uni_dev.h: k_hf_entropy_fast), one section after the other
		const DevFrame &df = hp.frame;
		if (!hp.hf.lanes_fast || !df.sparse_coeffs || df.num_passes != 1 || hp.hf.max_clusters > 64) return ERR_TODO;
		const DevCodeSpec &spec = hp.coeff_specs[0];
		UniTables t;
		t.ctx_map = plan.pool_u8 + spec.cluster_map_off; t.alias = plan.pool_u64 + plan.clusters[spec.cluster_off].table_off;
		t.log_alpha = spec.log_alpha_size; t.log_bucket = 12 - spec.log_alpha_size; t.num_dist = spec.num_dist;
		const int32_t *csrc = plan.pool_i32 + spec.lane_cfg_off;
		lr_fill(t.cfg, [&](int32_t i) { return i < spec.num_clusters ? csrc[i] : 0; });
		lr_fill(t.nnz2, [&](int32_t i) { return (int32_t) DEV_NNZ_CTX2[i]; });
		lr_fill(t.freq2, [&](int32_t i) { return (int32_t) DEV_FREQ_CTX2[i]; });
		lr_fill(t.dct, [&](int32_t i) { return i < 27 ? (int32_t) DEV_DCT_SELECT[i][0] | ((int32_t) DEV_DCT_SELECT[i][1] << 8) | ((int32_t) DEV_DCT_SELECT[i][2] << 16) : 0; });
		for (int32_t g = 0; g < df.num_groups; ++g) {
			HfTables h;
			memset(&h, 0, sizeof h);
			h.blocks = plan.group_blocks + plan.group_block_start[g];
			h.nblocks = (int32_t) (plan.group_block_start[g + 1] - plan.group_block_start[g]);
			h.nonzeros = plan.nonzeros + (size_t) g * (32 * 32 * 3);
			h.block_first = plan.group_block_start[g]; h.ev_first = plan.ev_range[2 * g]; h.ev_end = plan.ev_range[2 * g + 1];
			status[(size_t) g] = decode_hf_section_fast<false>(plan, df, t, h, plan.sections[g]);
		}
	} else
	if (only_entropy & 4) {   // bit 2: the throughput kernel's fast path (hf_lanes_dev.h), tables laid out as the kernel stages them
		if (!hp.hf.lanes_fast) return ERR_TODO;
		const DevFrame &df = hp.frame;
		LaneFrame lf = {df.nb_block_ctx, df.num_hf_presets, df.preset_bits, df.check_section_end, df.single_declared_end, df.order_off};
		LaneGlobals G = {plan.codestream, (const uint32_t *) plan.group_blocks, plan.coeffs[0], plan.events, plan.block_events, plan.pool_u16, plan.coeff_stride};
		std::vector<int8_t> cols(3 * 32);
		std::vector<uint32_t> ring((size_t) HF_LANE_RING_SLOTS + 1, 0xdeadbeefu);   // the lane's event ring (slots one word apart)
		std::vector<uint32_t> dct(27);
		for (int d = 0; d < 27; ++d) dct[(size_t) d] = (uint32_t) DEV_DCT_SELECT[d][0] | ((uint32_t) DEV_DCT_SELECT[d][1] << 8) | ((uint32_t) DEV_DCT_SELECT[d][2] << 16);
		for (int32_t pass = 0; pass < df.num_passes; ++pass) {
			const DevCodeSpec &spec = hp.coeff_specs[(size_t) pass];
			LaneTables t;
			t.ctx_map = plan.pool_u8 + spec.cluster_map_off; t.cluster_cfg = (const uint32_t *) (plan.pool_i32 + spec.lane_cfg_off);
			t.alias = plan.pool_u64 + plan.clusters[spec.cluster_off].table_off;
			t.nnz_ctx2 = DEV_NNZ_CTX2; t.freq_ctx2 = DEV_FREQ_CTX2; t.dct_info = dct.data();
			t.log_alpha = spec.log_alpha_size; t.log_bucket = 12 - spec.log_alpha_size;
			if (only_entropy & 8) {
				// bit 3: ONE lane takes every section of the pass, the largest first -- the form in which k_hf_lanes' lanes take a further
				// section from their frame's queue when they have finished one (decode_hf_sections_lane, column state carried over)
				struct Queue {
					const DevPlan &plan; const HostPlan &hp; int32_t pass; std::vector<int32_t> order; size_t at; int32_t cur; std::vector<uint32_t> &status;
					bool next(LaneSection &S) {
						if (at >= order.size()) return false;
						cur = order[at++];
						const DevSection &sec = plan.sections[pass * hp.frame.num_groups + cur];
						S.start_bit = 8u * sec.byte_off + sec.bit_off; S.end_bit = 8u * (sec.byte_off + sec.size);
						S.cell_base = (uint32_t) plan.lf_groups[sec.ggidx].cell_base;
						S.block_first = plan.group_block_start[cur]; S.nblocks = (int32_t) (plan.group_block_start[cur + 1] - S.block_first);
						S.ev_first = hp.frame.sparse_coeffs ? hp.ev_range[2 * (size_t) cur] : 0; S.ev_end = hp.frame.sparse_coeffs ? hp.ev_range[2 * (size_t) cur + 1] : 0;
						return true;
					}
					void done(uint32_t st, uint32_t) { status[(size_t) (pass * hp.frame.num_groups + cur)] = st; }
				} q{plan, hp, pass, {}, 0, 0, status};
				for (int32_t g = 0; g < df.num_groups; ++g) q.order.push_back(g);
				std::stable_sort(q.order.begin(), q.order.end(), [&](int32_t a, int32_t b) { return plan.sections[pass * df.num_groups + a].size > plan.sections[pass * df.num_groups + b].size; });
				if (df.sparse_coeffs) decode_hf_sections_lane<true>(lf, t, G, q, cols.data(), 1, pass, ring.data(), 1);
				else decode_hf_sections_lane<false>(lf, t, G, q, cols.data(), 1, pass);
			} else
			for (int32_t g = 0; g < df.num_groups; ++g) {
				const DevSection &sec = plan.sections[pass * df.num_groups + g];
				const uint32_t b0 = plan.group_block_start[g], b1 = plan.group_block_start[g + 1];
				const uint32_t cell_base = (uint32_t) plan.lf_groups[sec.ggidx].cell_base;
				status[(size_t) (pass * df.num_groups + g)] = df.sparse_coeffs ? decode_hf_section_lane<true>(lf, t, G, sec, cell_base, b0, (int32_t) (b1 - b0), hp.ev_range[2 * (size_t) g], hp.ev_range[2 * (size_t) g + 1], cols.data(), 1, pass, nullptr, ring.data(), 1)
					: decode_hf_section_lane<false>(lf, t, G, sec, cell_base, b0, (int32_t) (b1 - b0), 0, 0, cols.data(), 1, pass);
			}
		}
	} else
	for (int32_t g = 0; g < hp.frame.num_groups; ++g) {
		if (g_group_count >= 0 && (g < g_first_group || g >= g_first_group + g_group_count)) continue;
		decode_hf_group(plan, g, (only_entropy & 2) != 0);  // bit 1: flat (per-lane) decoder
	}
	if (coeffs_out) for (int c = 0; c < 3; ++c) {   // canonical layout, as the reference keeps them
		float *dst = coeffs_out + (size_t) c * hp.coeff_floats;
		memcpy(dst, coeffs[c], sizeof(float) * hp.coeff_floats);
		if (hp.frame.sparse_coeffs) for (const DevVarblock &vb : hp.vb_sorted) {
			const uint32_t *be = block_events.data() + 4 * (size_t) vb.blk;
			const uint32_t skip = c == 1 ? 0 : c == 0 ? be[1] : be[1] + be[2], n = be[c == 1 ? 1 : c == 0 ? 2 : 3];
			const std::vector<int32_t> &order = fr.orders[0][DCT_SELECT[vb.dctsel].order_idx][(size_t) c];
			for (uint32_t e = 0; e < n; ++e) { const CoeffEvent &ev = events[be[0] + skip + e]; dst[(size_t) vb.coeff_base + (size_t) order[coeff_event_pos(ev)]] = (float) coeff_event_value(ev); }
		}
	}
	// the Modular sub-images behind the coefficients (VarDCT frames with extra channels), as runtime.hip's validate_trailers does
	if (hp.frame.sections_have_trailer && !(only_entropy & 4) && g_group_count < 0) {
		HostModPlan tp;
		std::vector<std::pair<int32_t, uint32_t>> header_errors;
		std::vector<int32_t> section_of;
		if (uint32_t e = build_trailer_plan(fr, hp.codestream.data(), hp.codestream.size() - 16, end_bits.data(), status.data(), &tp, &header_errors, &section_of)) return e;
		DevModPlan mp;
		memset(&mp, 0, sizeof mp);
		mp.frame = &tp.frame; mp.codestream = hp.codestream.data(); mp.pool_u8 = tp.pool_u8.data(); mp.pool_i32 = tp.pool_i32.data(); mp.pool_u64 = tp.pool_u64.data();
		mp.clusters = tp.clusters.data(); mp.spec = tp.specs.data(); mp.tree = tp.tree.data(); mp.sections = tp.sections.data();
		std::vector<std::vector<int16_t>> store(tp.sub_w.size());
		std::vector<DevSubPlane> subp(tp.sub_w.size());
		for (size_t k = 0; k < subp.size(); ++k) { store[k].assign((size_t) tp.sub_w[k] * (size_t) tp.sub_h[k] + 1, 0); subp[k] = DevSubPlane{store[k].data(), tp.sub_w[k], tp.sub_h[k], tp.sub_meta[k], 0}; }
		mp.sub_planes = subp.data();
		std::vector<int32_t> wps(tp.sections.size() * (size_t) (2 * tp.frame.max_width * 5) + 16), win(tp.lz_window_size ? tp.sections.size() * tp.lz_window_size : 0);
		std::vector<uint32_t> tstatus(tp.sections.size() + 1, 0);
		mp.wp_scratch = tp.frame.tree_uses_wp ? wps.data() : nullptr;
		mp.lz_window = win.empty() ? nullptr : win.data(); mp.lz_window_size = tp.lz_window_size; mp.status = tstatus.data();
		for (int32_t i = 0; i < tp.frame.num_sections; ++i) {
			const ModTables mt = mod_tables_in_hbm(mp, i);
			if (const uint32_t e = decode_modular_section<false, false>(mp, mt, i)) status[(size_t) section_of[(size_t) i]] = e;
		}
		for (const auto &e : header_errors) status[(size_t) e.first] = e.second;
	}
	{
		uint32_t first = 0, first_off = 0xffffffffu;
		for (size_t i = 0; i < status.size(); ++i) if (status[i] && hp.sections[i].byte_off < first_off) { first = status[i]; first_off = hp.sections[i].byte_off; }
		if (first) return first;
	}
	if (only_entropy & 1) return 0;

	const float *hs = half_secants(), *afv = afv_basis();
	const DevFrame &f = hp.frame;
	const size_t stride = (size_t) f.width * 4;
	std::vector<float> A(3 * 65536), B(3 * 65536), scratch(SP8_TILE), scratch2(SP8_TILE);
	for (const DevVarblock &vb : hp.vb_sorted) {
		if (g_group_count >= 0) {   // sharded decode: varblocks of the selected groups only
			const int64_t gid = ((int64_t) vb.py >> fr.fh.group_size_shift) * fr.fh.gcolumns + ((int64_t) vb.px >> fr.fh.group_size_shift);
			if (gid < g_first_group || gid >= g_first_group + g_group_count) continue;
		}
		const int log_rows = DEV_DCT_SELECT[vb.dctsel][0], log_columns = DEV_DCT_SELECT[vb.dctsel][1];
		const int R = 1 << log_rows, C = 1 << log_columns, sz = R * C;
		const int long_side = R > C ? R : C, vh8 = (R < C ? R : C) / 8, vw8 = long_side / 8;
		static const int8_t PARAM[27] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
		const float *dq = plan.pool_f32 + f.dq_off[PARAM[vb.dctsel]];
		const VbGeom g = varblock_geometry(plan, vb);
		const bool special = (vb.dctsel >= 1 && vb.dctsel <= 3) || (vb.dctsel >= 12 && vb.dctsel <= 17);
		const bool large = log_rows > 6 || log_columns > 6;
		const int P = special ? SP8_PITCH : large ? C : C + 1;   // (the specials' tiles: rows 9 apart, special8_dev.h)
		// k_vardct_large's whole block (large_dev.h: where the tile lives, the channel-at-a-time scatter of the 128x128 tiles, the recursion's
		// top levels over 64-point sub-vectors in registers), lane by lane. HOSTSIM_LARGE_SWEEPS=1: the model of round 3's kernel instead
		// (the tile in the scratch, every butterfly level as a sweep), which must give the same bits
		if (large && getenv("HOSTSIM_LARGE_SWEEPS") == nullptr) {
			HostExec ex;
			std::vector<float> panels(2 * LARGE_PANEL_FLOATS, -1.0f);
			const LargeSamples S = large_block(ex, plan, vb, g, panels.data(), A.data(), B.data(), hs);
			for (int y = 0; y < g.effh; ++y) for (int x = 0; x < g.effw; ++x) {
				const uint32_t px = xyb_to_rgba8(S.p[0][y * S.pitch[0] + x], S.p[1][y * S.pitch[1] + x], S.p[2][y * S.pitch[2] + x], f, srgb_u8_thresholds());
				memcpy(rgba + (size_t) (g.py + y) * stride + (size_t) (g.px + x) * 4, &px, 4);
			}
			continue;
		}
		if (f.sparse_coeffs) {   // the pixel kernels' way: zeroed tiles, scattered events, LLF corner, chroma-from-luma in place
			for (int ch = 0; ch < 3; ++ch) std::fill(A.begin() + (size_t) ch * 65536, A.begin() + (size_t) ch * 65536 + std::min<size_t>(65536, (size_t) R * (size_t) P), 0.0f);
			const TileMap map = {R, C, P, special ? 1 : 0};
			const uint16_t *order = plan.pool_u16 + f.order_off[DEV_DCT_SELECT[vb.dctsel][2] * 3];
			const uint32_t *be = plan.block_events + 4 * (size_t) vb.blk;
			if (vb.blk & 1) {   // both forms the kernels use
				tile_scatter_events(plan, g, be, order, plan.pool_f32 + f.dq_scan_off[PARAM[vb.dctsel]], sz, map, A.data(), 65536, f.quant_bias, f.quant_bias_num, 0, 1);
				tile_fill_llf(plan, g, long_side, vh8, vw8, map, A.data(), 65536, f.kx_lf, f.kb_lf, 0, 1);
			} else {
				const uint32_t be1[1][4] = {{be[0], be[1], be[2], be[3]}}, prefix[2] = {0, be[1] + be[2] + be[3]};
				for (int lane = 0; lane < 3; ++lane) tiles_scatter_events<1>(plan, &g, be1, prefix, order, plan.pool_f32 + f.dq_scan_off[PARAM[vb.dctsel]], (const uint32_t *) nullptr, sz, map, A.data(), 0, 65536, f.quant_bias, f.quant_bias_num, lane, 3);
				tiles_fill_llf(plan, &g, 1, long_side, vh8, vw8, map, A.data(), 0, 65536, f.kx_lf, f.kb_lf, 0, 1);
			}
		} else for (int i = 0; i < sz; ++i) {
			float v[3];
			load_coeff3(plan, g, dq, sz, i, long_side, vh8, vw8, v);
			int r, c;
			if (special) { r = i / 8; c = i % 8; }
			else { r = C > R ? i / C : i % R; c = C > R ? i % C : i / R; }
			for (int ch = 0; ch < 3; ++ch) A[(size_t) ch * 65536 + r * P + c] = v[ch];
		}
		for (int ch = 0; ch < 3; ++ch) {
			float *t = A.data() + (size_t) ch * 65536;
			if (special) {   // the cooperative form (k_vardct_special): two phases of eight lanes; here one lane after the other, in an
				// order that changes with the block so that a dependence between the lanes of a phase would show
				// (out of place here: the kernel's lanes run in lockstep and work in place -- every phase loads all it needs before it stores)
				for (int l = 0; l < 8; ++l) special8_phase0(vb.dctsel, (vb.blk & 1) ? 7 - l : l, (const float *) t, scratch.data(), hs, afv, false);
				for (int l = 0; l < 8; ++l) special8_phase1(vb.dctsel, (vb.blk & 2) ? 7 - l : l, (const float *) scratch.data(), scratch2.data(), hs, false);
				memcpy(t, scratch2.data(), sizeof(float) * SP8_TILE);
			}
			else if (large) {
				idct_sweeps_host(t, B.data() + (size_t) ch * 65536, log_columns, R, 1, C, hs);
				idct_sweeps_host(B.data() + (size_t) ch * 65536, t, log_rows, C, C, 1, hs);
			} else { idct_rows_dyn(t, C, R, P, hs); idct_cols_dyn(t, R, C, P, hs); }
		}
		for (int y = 0; y < g.effh; ++y) for (int x = 0; x < g.effw; ++x) {
			const uint32_t px = xyb_to_rgba8(A[y * P + x], A[65536 + y * P + x], A[2 * 65536 + y * P + x], f, srgb_u8_thresholds());
			memcpy(rgba + (size_t) (g.py + y) * stride + (size_t) (g.px + x) * 4, &px, 4);
		}
	}
	return 0;
}

// known-answer hook: one 8x8 special transform (DctSelect 1-3, 12-17) through the cooperative device functions; `order` permutes
// the lanes of both phases (0: ascending, 1: descending, 2: odd lanes first)
extern "C" __attribute__((visibility("default"))) void hostsim_special8(int dctsel, float *tile64, int order) {
	float in[SP8_TILE], mid[SP8_TILE], out[SP8_TILE];
	for (int k = 0; k < SP8_TILE; ++k) { in[k] = 0.0f; mid[k] = -12345.0f; out[k] = -54321.0f; }   // both phases must rewrite every value
	for (int k = 0; k < 64; ++k) in[SP8(k)] = tile64[k];
	auto lane_of = [order](int l) { return order == 0 ? l : order == 1 ? 7 - l : (l < 4 ? 2 * l + 1 : 2 * (l - 4)); };
	for (int l = 0; l < 8; ++l) special8_phase0(dctsel, lane_of(l), (const float *) in, mid, half_secants(), afv_basis(), false);
	for (int l = 0; l < 8; ++l) special8_phase1(dctsel, lane_of(l), (const float *) mid, out, half_secants(), false);
	for (int k = 0; k < 64; ++k) tile64[k] = out[SP8(k)];
}

// the same with both phases IN PLACE on one tile, the eight lanes of a phase in lockstep as a wavefront runs them: a lane's loads
// see the tile as it was before the phase (every phase loads all it needs before it stores anything), its stores land in the tile
extern "C" __attribute__((visibility("default"))) void hostsim_special8_in_place(int dctsel, float *tile64) {
	float tile[SP8_TILE], before[SP8_TILE];
	for (int k = 0; k < SP8_TILE; ++k) tile[k] = 0.0f;
	for (int k = 0; k < 64; ++k) tile[SP8(k)] = tile64[k];
	memcpy(before, tile, sizeof tile);
	for (int l = 0; l < 8; ++l) special8_phase0(dctsel, l, (const float *) before, tile, half_secants(), afv_basis(), true);
	memcpy(before, tile, sizeof tile);
	for (int l = 0; l < 8; ++l) special8_phase1(dctsel, l, (const float *) before, tile, half_secants(), true);
	for (int k = 0; k < 64; ++k) tile64[k] = tile[SP8(k)];
}

// sweeps float bit patterns [first, last] with stride `step` through pow_1_over_2p4 and counts results that
// differ from (float) pow((double) x, (double) (1.0f / 2.4f)); *worst receives one offending input
extern "C" __attribute__((visibility("default"))) uint64_t hostsim_pow_sweep(uint32_t first, uint32_t last, uint32_t step, float *worst) {
	uint64_t bad = 0;
	const double P = (double) (1.0f / 2.4f);
	for (uint64_t u = first; u <= last; u += step) {
		const uint32_t bits = (uint32_t) u;
		float x; memcpy(&x, &bits, 4);
		const float got = pow_1_over_2p4(x), expect = (float) pow((double) x, P);
		if (memcmp(&got, &expect, 4) != 0 && !(got != got && expect != expect)) { ++bad; if (worst) *worst = x; }
	}
	return bad;
}

// the table path at every place where it could be off by one: each threshold, each bucket edge, their neighbouring floats,
// and a few values outside (0, 1)
extern "C" __attribute__((visibility("default"))) uint64_t hostsim_srgb_u8_edges() {
	const float *thr = srgb_u8_thresholds();
	const double P = (double) (1.0f / 2.4f);
	std::vector<float> probe = {0.0f, -0.0f, -1e-30f, -0.5f, -8.99f, 1e-30f, 1.0f, 1.5f, 100.0f, 49999.0f};
	auto around = [&](uint32_t bits) { for (int d = -2; d <= 2; ++d) { const uint32_t u = bits + (uint32_t) d; float x; memcpy(&x, &u, 4); probe.push_back(x); } };
	for (int k = 1; k <= 255; ++k) { uint32_t u; memcpy(&u, &thr[k], 4); around(u); }
	for (uint32_t b = SRGB_BUCKET_LO - 2; b <= SRGB_BUCKET_HI + 2; ++b) around(b << 16);
	uint64_t bad = 0;
	for (float v : probe) {
		const float t = v <= 0.0031308f ? 12.92f * v : 1.055f * (float) pow((double) v, P) - 0.055f;
		int32_t px = f32_to_i16_x86(255.0f * t + 0.5f);
		px = px < 0 ? 0 : px > 255 ? 255 : px;
		bad += srgb_u8_from_thresholds(v, thr) != px;
	}
	return bad;
}

// the same for the 8-bit sample the renderer finally stores (what parity is judged on)
extern "C" __attribute__((visibility("default"))) uint64_t hostsim_srgb_u8_sweep(uint32_t first, uint32_t last, uint32_t step) {
	uint64_t bad = 0;
	const double P = (double) (1.0f / 2.4f);
	for (uint64_t u = first; u <= last; u += step) {
		const uint32_t bits = (uint32_t) u;
		float v; memcpy(&v, &bits, 4);
		const float a = srgb_transfer(v);
		const float b = v <= 0.0031308f ? 12.92f * v : 1.055f * (float) pow((double) v, P) - 0.055f;
		int32_t pa = f32_to_i16_x86(255.0f * a + 0.5f), pb = f32_to_i16_x86(255.0f * b + 0.5f);
		pa = pa < 0 ? 0 : pa > 255 ? 255 : pa; pb = pb < 0 ? 0 : pb > 255 ? 255 : pb;
		bad += pa != pb;
		if (v > -9.0f && v < 50000.0f) bad += srgb_u8_from_thresholds(v, srgb_u8_thresholds()) != pb;   // the table path of the pixel kernels
	}
	return bad;
}

// the DevCoopTree of every section k_modular_coop takes (plan_build.cpp, assign_coop), checked against a walk of the MA tree
// it was built from: for `trials` random property vectors per tree, the leaf the masks select must carry the walk's leaf.
// returns the number of trees checked, or -(1 + mismatches)
extern "C" __attribute__((visibility("default"))) int32_t hostsim_coop_check(const uint8_t *buf, size_t size, uint32_t seed, int32_t trials) {
	Frame fr;
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	HostModPlan hp;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		parse_frame(cs, cs_size, &fr, 1);
	} catch (const DecodeError &e) { return 0; }
	if (!fr.fh.is_modular || build_modular_plan(fr, cs, cs_size, &hp)) return 0;
	int32_t checked = 0, bad = 0;
	std::vector<char> seen(hp.coop_trees.size(), 0);
	auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
	for (const DevModSection &s : hp.sections) {
		if (s.coop_idx < 0 || seen[(size_t) s.coop_idx]) continue;
		seen[(size_t) s.coop_idx] = 1; ++checked;
		const DevCoopTree &t = hp.coop_trees[(size_t) s.coop_idx];
		const DevTreeNode *tree = hp.tree.data() + s.tree_off;
		const DevCodeSpec &sp = hp.specs[(size_t) s.spec_idx];
		for (int32_t k = 0; k < trials; ++k) {
			int32_t props[15];
			for (int q = 0; q < 15; ++q) {
				// values around the thresholds the tree tests, so that both sides of every branch are taken
				const int32_t pick = t.num_nodes ? t.node_thr[rnd() % (uint32_t) t.num_nodes] : 0;
				props[q] = pick + (int32_t) (rnd() % 5) - 2;
				if (rnd() % 4 == 0) props[q] = (int32_t) (rnd() % 65536) - 32768;
			}
			const DevTreeNode *n = tree;
			while (n->prop >= 0) n += props[n->prop] > n->value ? n->a : n->b;
			uint64_t outcomes = 0;
			for (int i = 0; i < 64; ++i) if (t.node_prop[i] >= 0 && props[t.node_prop[i]] > t.node_thr[i]) outcomes |= (uint64_t) 1 << i;
			int32_t leaf = -1, matches = 0;
			for (int i = 0; i < 64; ++i) {
				const uint64_t mask = (uint64_t) t.mask_lo[i] | (uint64_t) t.mask_hi[i] << 32, want = (uint64_t) t.want_lo[i] | (uint64_t) t.want_hi[i] << 32;
				if ((outcomes & mask) == want) { ++matches; if (leaf < 0) leaf = i; }
			}
			const DevCluster &cl = hp.clusters[(size_t) sp.cluster_off + (size_t) hp.pool_u8[sp.cluster_map_off + (uint32_t) n->value]];
			const bool same = matches == 1 && leaf >= 0 && (int32_t) (t.leaf_a[leaf] & 15) == -1 - n->prop && t.leaf_off[leaf] == n->a && t.leaf_mul[leaf] == n->b
				&& t.leaf_tab[leaf] == cl.table_off && ((t.leaf_a[leaf] >> 4) & 0xfff) == cl.cfg && (int32_t) (t.leaf_a[leaf] >> 16) == cl.max_token;
			bad += !same;
		}
	}
	return bad ? -(1 + bad) : checked;
}

// ---- the host glue around the device's LfGroup decoder (frame.cpp: parse_frame's lf_decoder branch, lf_group_finish), exercised
// without a GPU: a stand-in decoder that does on the host what k_lf_groups does on the device -- start at the bit the task names,
// decode the LF image, check the final rANS state, read the varblock count and the second header (plain ones only: anything else
// reports 'lffb'), decode the HF metadata -- and hands back planes in the kernel's layout. The frame parsed through it must
// equal the frame parsed by read_lf_group. mode 1: the stand-in answers 'lffb' for every second section (host fallback).
namespace {
struct FakeLfStore { std::vector<std::vector<int16_t>> planes; int mode = 0; int calls = 0; };
bool fake_lf_decoder(void *ctx, const Frame &f, const uint8_t *cs, size_t cs_size, std::vector<LfDeviceTask> &tasks) {
	FakeLfStore &st = *(FakeLfStore *) ctx;
	++st.calls;
	st.planes.assign(tasks.size(), std::vector<int16_t>());
	for (size_t i = 0; i < tasks.size(); ++i) {
		LfDeviceTask &t = tasks[i];
		if (t.byte_off + t.size > cs_size) { t.status = ERR_SHRT; continue; }
		if (st.mode == 1 && (i & 1)) { t.status = (uint32_t) ERR_LFFB; continue; }
		const size_t cells = (size_t) t.w8 * (size_t) t.h8, c64 = (size_t) t.w64 * (size_t) t.h64;
		std::vector<int16_t> &out = st.planes[i];
		try {
			BitReader br(cs + t.byte_off, t.size);
			br.skip_bits((int64_t) t.bit_off);
			Modular a; a.bpp = f.im.bpp; a.use_global_tree = true; a.tree = &f.global_tree; a.codespec = &f.global_codespec;
			a.channel.assign(3, Plane());
			for (Plane &p : a.channel) { p.width = t.w8; p.height = t.h8; }
			allocate_modular(&a);
			{ CodeState code(a.codespec); for (int32_t c = 0; c < 3; ++c) decode_modular_channel(br, a, code, c, t.sidx0); finish_code(br, code); }
			t.nb_varblocks = (int32_t) br.u(t.nbvb_bits) + 1;
			if (br.u(4) != 3u || (size_t) t.nb_varblocks > cells) { t.status = (uint32_t) ERR_LFFB; continue; }
			Modular b; b.bpp = f.im.bpp; b.use_global_tree = true; b.tree = &f.global_tree; b.codespec = &f.global_codespec;
			b.channel.assign(4, Plane());
			b.channel[0].width = b.channel[1].width = t.w64; b.channel[0].height = b.channel[1].height = t.h64;
			b.channel[2].width = t.nb_varblocks; b.channel[2].height = 2;
			b.channel[3].width = t.w8; b.channel[3].height = t.h8;
			allocate_modular(&b);
			{ CodeState code(b.codespec); for (int32_t c = 0; c < 4; ++c) decode_modular_channel(br, b, code, c, t.sidx2); finish_code(br, code); }
			out.resize(3 * cells + 2 * c64 + 2 * (size_t) t.nb_varblocks + cells);
			int16_t *p = out.data();
			for (int c = 0; c < 3; ++c) { memcpy(p, a.channel[(size_t) c].px.data(), cells * 2); t.lf[c] = p; p += cells; }
			memcpy(p, b.channel[0].px.data(), c64 * 2); t.xfromy = p; p += c64;
			memcpy(p, b.channel[1].px.data(), c64 * 2); t.bfromy = p; p += c64;
			memcpy(p, b.channel[2].px.data(), 2 * (size_t) t.nb_varblocks * 2); t.info0 = p; t.info1 = p + t.nb_varblocks; p += 2 * (size_t) t.nb_varblocks;
			memcpy(p, b.channel[3].px.data(), cells * 2);
			t.status = 0;
		} catch (const DecodeError &e) { t.status = e.code; }
	}
	return true;
}
}

// returns 0 when both parses agree (or fail with the same code), else a non-zero description: 1 error codes differ, 2 the stand-in
// was not called, 3 LF groups differ; *err_out = the plain parse's error code
extern "C" __attribute__((visibility("default"))) int32_t hostsim_lf_decoder_glue(const uint8_t *buf, size_t size, int mode, uint32_t *err_out) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	uint32_t e1 = 0, e2 = 0;
	Frame plain, hooked;
	FakeLfStore store; store.mode = mode;
	plain.defer_lf_tail = hooked.defer_lf_tail = true;
	try { extract_codestream(buf, size, &cs, &cs_size, &storage); } catch (const DecodeError &e) { if (err_out) *err_out = e.code; return 0; }
	try { parse_frame(cs, cs_size, &plain, 1); } catch (const DecodeError &e) { e1 = e.code; }
	hooked.lf_decoder = fake_lf_decoder; hooked.lf_decoder_ctx = &store;
	try { parse_frame(cs, cs_size, &hooked, 1); } catch (const DecodeError &e) { e2 = e.code; }
	if (err_out) *err_out = e1;
	if (e1 != e2) return 1;
	if (e1) return 0;
	if (plain.fh.is_modular || plain.toc.single) return 0;   // (those never reach the decoder)
	if (!store.calls || !hooked.lf_decoded_on_device) return 2;
	for (size_t g = 0; g < plain.lf_groups.size(); ++g) {
		const LfGroup &a = plain.lf_groups[g], &b = hooked.lf_groups[g];
		bool same = a.blocks == b.blocks && a.lfindices == b.lfindices && a.xfromy == b.xfromy && a.bfromy == b.bfromy && a.varblocks.size() == b.varblocks.size();
		for (int c = 0; same && c < 3; ++c) same = a.lfraw[c] == b.lfraw[c] && a.mult_lf[c] == b.mult_lf[c];
		for (size_t v = 0; same && v < a.varblocks.size(); ++v)
			same = a.varblocks[v].coeffoff_qfidx == b.varblocks[v].coeffoff_qfidx && a.varblocks[v].hfmul_inv == b.varblocks[v].hfmul_inv && a.varblocks[v].x8 == b.varblocks[v].x8 && a.varblocks[v].y8 == b.varblocks[v].y8 && a.varblocks[v].dctsel == b.varblocks[v].dctsel;
		if (!same) return 3;
	}
	return plain.dct_select_used == hooked.dct_select_used && plain.order_used == hooked.order_used ? 0 : 3;
}

// ---- the device-side plan build (device/plan_dev.h) on the CPU, against plan_build.cpp ----
#include "../../j40_amd/csrc/plan_front.hpp"
#include "../../j40_amd/csrc/device/plan_dev.h"

// Runs the pipeline's path on the CPU: front parse, the LfGroup sections' raw planes (host decoder), then the three device
// functions of plan_dev.h in the kernels' orchestration; compares every array they produce with what parse_frame +
// build_vardct_plan produce. Returns 0 when all agree, -1 when the frame is not one the front plan takes (nothing compared),
// otherwise the number of the first check that failed; *err_out = the full parse's error code (on error nothing is compared
// but the LfGroup statuses: the first failing LfGroup in file order must carry that code).
extern "C" __attribute__((visibility("default"))) int32_t hostsim_device_plan_check(const uint8_t *buf, size_t size, uint32_t *err_out) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	uint32_t e_full = 0;
	Frame full, front;
	full.defer_lf_tail = true;
	try { extract_codestream(buf, size, &cs, &cs_size, &storage); } catch (const DecodeError &e) { if (err_out) *err_out = e.code; return -1; }
	try { parse_frame(cs, cs_size, &full, 1); } catch (const DecodeError &e) { e_full = e.code; }
	if (err_out) *err_out = e_full;
	std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
	try { if (!parse_frame_front(cs, cs_size, &front, &tasks, &extra_prec, &plain)) return -1; }
	catch (const DecodeError &e) { return e.code == e_full ? -1 : 1; }   // an error in front of the LfGroups: both report it
	StaticTables st;
	build_static_tables(front, &st);
	FrontPlan fp;
	if (build_front_plan(front, st, cs_size, extra_prec, true, &fp)) return -1;
	const size_t ngg = front.lf_groups.size(), cells = fp.cells;
	// raw planes, frame-wide arrays as the runtime lays them out
	std::vector<int16_t> lfraw[3], xfromy(fp.c64s), bfromy(fp.c64s), vbinfo(2 * cells + 2);
	for (int c = 0; c < 3; ++c) lfraw[c].assign(cells, 0);
	std::vector<DevLfSlot> slots(ngg);
	for (size_t g = 0; g < ngg; ++g) {
		const DevLfGroup &d = fp.lf_groups[g];
		DevLfSlot &sl = slots[g];
		memset(&sl, 0, sizeof sl);
		try {
			BitReader sr(cs + front.toc.lf_groups[g].offset, front.toc.lf_groups[g].size);
			LfRaw raw;
			read_lf_group_raw(sr, front, front.lf_groups[g], &raw);
			static const int XYB_FROM_STREAM[3] = {1, 0, 2};
			for (int c = 0; c < 3; ++c) memcpy(lfraw[c].data() + d.cell_base, raw.lf[XYB_FROM_STREAM[c]].data(), raw.lf[0].size() * 2);
			memcpy(xfromy.data() + d.c64_base, raw.xfromy.data(), raw.xfromy.size() * 2); memcpy(bfromy.data() + d.c64_base, raw.bfromy.data(), raw.bfromy.size() * 2);
			if ((size_t) raw.nb_varblocks > (size_t) d.width8 * (size_t) d.height8) sl.status = ERR_VBLK;   // (more varblocks than cells: the placement would say so)
			else memcpy(vbinfo.data() + 2 * (size_t) d.cell_base, raw.info.data(), raw.info.size() * 2);
			sl.nb_varblocks = raw.nb_varblocks;
		} catch (const DecodeError &e) { sl.status = e.code; }
	}
	std::vector<DevVbRec> recs(cells + 1);
	std::vector<uint32_t> group_count((size_t) fp.build.num_groups, 0), group_block_start((size_t) fp.build.num_groups + 1, 0), class_count(ngg * 28, 0);
	std::vector<DevGroupBlock> group_blocks(cells + 1);
	std::vector<DevVarblock> vb_sorted(cells + 1);
	int32_t class_start[28];
	DevPlanBuild pb = fp.build;
	pb.pool_u8 = fp.pool_u8.data(); pb.lf_groups = fp.lf_groups.data(); pb.lf_slots = slots.data();
	for (int c = 0; c < 3; ++c) pb.lfraw[c] = lfraw[c].data();
	pb.xfromy = xfromy.data(); pb.bfromy = bfromy.data(); pb.vbinfo = vbinfo.data(); pb.vb_recs = recs.data();
	pb.group_count = group_count.data(); pb.group_block_start = group_block_start.data(); pb.class_count = class_count.data(); pb.class_start = class_start;
	pb.group_blocks = group_blocks.data(); pb.vb_sorted = vb_sorted.data(); pb.lf_section_off = fp.lf_section_off.data();
	{   // k_plan_place: 64 LfGroups side by side, scratch interleaved
		std::vector<uint16_t> occ(256 * 64), grp(64 * 64); std::vector<uint32_t> cls(28 * 64);
		for (size_t g = 0; g < ngg; ++g) plan_place_lf_group(pb, (int32_t) g, occ.data() + g % 64, grp.data() + g % 64, cls.data() + g % 64, 64);
	}
	plan_scan_frame(pb);
	for (size_t g = 0; g < ngg; ++g) for (int32_t v = 0; v < slots[g].placed; ++v) plan_emit_varblock(pb, (int32_t) g, v);
	// the verdict over the LfGroup sections
	uint64_t best = ~(uint64_t) 0;
	for (size_t g = 0; g < ngg; ++g) best = std::min(best, plan_verdict_key(slots[g].status, fp.lf_section_off[g]));
	const uint32_t lf_err = best == ~(uint64_t) 0 ? 0 : (uint32_t) best;
	if (e_full) return lf_err == e_full ? 0 : 2;
	if (lf_err) return 3;
	HostPlan hp;
	if (build_vardct_plan(full, cs, cs_size, &hp)) return -1;
	if (hp.group_block_start != group_block_start) return 4;
	if (memcmp(hp.group_blocks.data(), group_blocks.data(), sizeof(DevGroupBlock) * hp.group_blocks.size()) != 0) return 5;
	if (memcmp(hp.class_start, class_start, sizeof class_start) != 0) return 6;
	if (memcmp(hp.vb_sorted.data(), vb_sorted.data(), sizeof(DevVarblock) * hp.vb_sorted.size()) != 0) return 7;
	for (size_t g = 0; g < ngg; ++g) {
		const DevLfGroup &a = hp.lf_groups[g], &b = fp.lf_groups[g];
		if (a.nb_varblocks != b.nb_varblocks || a.cell_base != b.cell_base || a.c64_base != b.c64_base || memcmp(a.mult_lf, b.mult_lf, sizeof a.mult_lf) != 0 || a.width8 != b.width8 || a.height != b.height) return 8;
		if (slots[g].dct_used == 0) return 9;
	}
	for (int c = 0; c < 3; ++c) if (hp.lfraw[c] != lfraw[c]) return 10;
	// tables: every table the host path loaded must be in the static set with the same contents
	const DevFrame &fa = hp.frame, &fb = fp.frame;
	for (int i = 0; i < 17; ++i) if (fa.dq_off[i] != 0xffffffffu) {
		if (fb.dq_off[i] == 0xffffffffu || fa.dq_size[i] != fb.dq_size[i] || memcmp(hp.pool_f32.data() + fa.dq_off[i], st.pool_f32.data() + fb.dq_off[i], sizeof(float) * 3 * fa.dq_size[i]) != 0) return 11;
		if ((fa.dq_scan_off[i] != 0xffffffffu) != (fb.dq_scan_off[i] != 0xffffffffu)) return 12;
		if (fa.dq_scan_off[i] != 0xffffffffu && memcmp(hp.pool_f32.data() + fa.dq_scan_off[i], st.pool_f32.data() + fb.dq_scan_off[i], sizeof(float) * 3 * fa.dq_size[i]) != 0) return 13;
	}
	for (int i = 0; i < 11 * 13 * 3; ++i) if (fa.order_off[i] != 0xffffffffu) {
		const int o = (i / 3) % 13; const size_t n = (size_t) 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1]);
		if (fb.order_off[i] == 0xffffffffu || memcmp(hp.pool_u16.data() + fa.order_off[i], st.pool_u16.data() + fb.order_off[i], 2 * n) != 0) return 14;
	}
	{   // everything else of DevFrame
		DevFrame x = fa, y = fb;
		memset(x.order_off, 0, sizeof x.order_off); memset(y.order_off, 0, sizeof y.order_off); memset(x.dq_off, 0, sizeof x.dq_off); memset(y.dq_off, 0, sizeof y.dq_off);
		memset(x.dq_size, 0, sizeof x.dq_size); memset(y.dq_size, 0, sizeof y.dq_size); memset(x.dq_scan_off, 0, sizeof x.dq_scan_off); memset(y.dq_scan_off, 0, sizeof y.dq_scan_off);
		if (memcmp(&x, &y, sizeof x) != 0) return 15;
	}
	if (hp.ev_range != fp.ev_range || hp.sections.size() != fp.sections.size() || memcmp(hp.sections.data(), fp.sections.data(), sizeof(DevSection) * hp.sections.size()) != 0) return 16;
	return 0;
}

// timing aid (tools/front_timing.py): the host stage of the pipeline on this CPU, stage by stage; out_ms[0] front parse, [1] static
// tables key + front plan, [2] the LfGroup streams (host decoder), [3] full parse_frame for comparison
#include <chrono>
extern "C" __attribute__((visibility("default"))) int32_t hostsim_front_timing(const uint8_t *buf, size_t size, int32_t iters, double *out_ms) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	try { extract_codestream(buf, size, &cs, &cs_size, &storage); } catch (const DecodeError &) { return -1; }
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	for (int i = 0; i < 4; ++i) out_ms[i] = 0;
	StaticTables st; bool have_st = false;
	FrontPlan fp;
	for (int32_t it = 0; it < iters; ++it) {
		try {
			Frame fr;
			std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
			double t0 = now();
			if (!parse_frame_front(cs, cs_size, &fr, &tasks, &extra_prec, &plain)) return -1;
			double t1 = now();
			std::vector<uint8_t> key; static_tables_key(fr, &key);
			if (!have_st) { build_static_tables(fr, &st); have_st = true; t1 = now(); }
			if (build_front_plan(fr, st, cs_size, extra_prec, true, &fp)) return -1;
			double t2 = now();
			for (size_t g = 0; g < fr.lf_groups.size(); ++g) { BitReader sr(cs + fr.toc.lf_groups[g].offset, fr.toc.lf_groups[g].size); LfRaw raw; read_lf_group_raw(sr, fr, fr.lf_groups[g], &raw); }
			double t3 = now();
			Frame full; full.defer_lf_tail = true;
			parse_frame(cs, cs_size, &full, 1);
			double t4 = now();
			out_ms[0] += t1 - t0; out_ms[1] += t2 - t1; out_ms[2] += t3 - t2; out_ms[3] += t4 - t3;
		} catch (const DecodeError &) { return -1; }
	}
	for (int i = 0; i < 4; ++i) out_ms[i] /= iters;
	return 0;
}

// The persistent pixel kernels' walk over their tiles (k2_iter_dev.h), every workgroup of a launch one after the other: `counts[f]` =
// varblocks of the launch's class in frame f (frames without any among them), tiles of `per_wg` varblocks, `grid` workgroups.
// Every tile of every frame must be taken exactly once with its first varblock, the frame's list / count / output bound when --
// and only when -- a run enters a frame. Returns 0 or the number of the check that failed.
extern "C" __attribute__((visibility("default"))) int32_t hostsim_k2_runs_check(const int32_t *counts, int32_t nframes, int32_t per_wg, int32_t grid) {
	std::vector<K2Frame> frames((size_t) nframes);
	std::vector<DevVarblock> store(16);
	std::vector<int32_t> prefix((size_t) nframes + 1, 0);
	const int32_t class_a = 4, class_b = 7;   // (a class spanning several DctSelect values, like the specials')
	for (int32_t f = 0; f < nframes; ++f) {
		memset(&frames[(size_t) f], 0, sizeof(K2Frame));
		K2Frame &fr = frames[(size_t) f];
		fr.sorted = store.data() + f % 7; fr.rgba = (uint8_t *) store.data() + 1000 * f; fr.stride = 4096 + (size_t) f;
		for (int d = 0; d < 28; ++d) fr.class_start[d] = d <= class_a ? 11 * f : d < class_b ? 11 * f + counts[f] / 2 : 11 * f + counts[f];
		prefix[(size_t) f + 1] = prefix[(size_t) f] + (counts[f] + per_wg - 1) / per_wg;   // as k_k2_tiles lays it out
	}
	std::vector<int32_t> taken((size_t) prefix[(size_t) nframes], 0);
	for (int32_t block = 0; block < grid; ++block) {
		const DevVarblock *list = nullptr; int32_t count = -1; uint8_t *rgba = nullptr; size_t stride = 0;
		int32_t last_frame = -1;
		for (K2Iter it = k2_run_begin(prefix.data(), nframes, block, grid); ; ) {
			int32_t frame, first; bool entered;
			if (!k2_run_bind(it, frames.data(), prefix.data(), class_a, class_b, per_wg, list, count, rgba, stride, frame, first, entered)) break;
			if (frame < 0 || frame >= nframes) return 1;
			if (entered != (frame != last_frame)) return 2;
			last_frame = frame;
			const K2Frame &fr = frames[(size_t) frame];
			if (list != fr.sorted + fr.class_start[class_a] || count != counts[frame] || rgba != fr.rgba || stride != fr.stride) return 3;
			if (first < 0 || first >= count || first % per_wg != 0) return 4;
			const int32_t tile = prefix[(size_t) frame] + first / per_wg;
			if (tile >= prefix[(size_t) frame + 1]) return 5;
			++taken[(size_t) tile];
		}
	}
	for (int32_t t : taken) if (t != 1) return 6;
	return 0;
}

// build_vardct_plan with a team of `threads` threads against the calling thread alone: every frame-wide array byte for byte.
// returns 0, a negative number when the frame does not parse, or the number of the first array that differs
extern "C" __attribute__((visibility("default"))) int32_t hostsim_plan_threads_check(const uint8_t *buf, size_t size, int32_t threads) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	Frame fr;
	try { extract_codestream(buf, size, &cs, &cs_size, &storage); parse_frame(cs, cs_size, &fr, 1); } catch (const DecodeError &) { return -1; }
	HostPlan a, b;
	const uint32_t ea = build_vardct_plan(fr, cs, cs_size, &a, 1), eb = build_vardct_plan(fr, cs, cs_size, &b, threads);
	if (ea != eb) return 1;
	if (ea) return -2;
	auto same = [](const auto &x, const auto &y) { return x.size() == y.size() && (x.empty() || memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0); };
	if (!same(a.lf_groups, b.lf_groups)) return 2;
	if (!same(a.blocks, b.blocks) || !same(a.lfindices, b.lfindices)) return 3;
	for (int c = 0; c < 3; ++c) if (!same(a.llf[c], b.llf[c]) || !same(a.lfraw[c], b.lfraw[c])) return 4;
	if (!same(a.xfromy, b.xfromy) || !same(a.bfromy, b.bfromy)) return 5;
	if (!same(a.vb_coeffoff_qfidx, b.vb_coeffoff_qfidx) || !same(a.vb_hfmul_inv, b.vb_hfmul_inv)) return 6;
	if (!same(a.group_block_start, b.group_block_start) || !same(a.group_blocks, b.group_blocks)) return 7;
	if (memcmp(a.class_start, b.class_start, sizeof a.class_start) != 0 || !same(a.vb_sorted, b.vb_sorted)) return 8;
	if (!same(a.sections, b.sections) || !same(a.ev_range, b.ev_range) || a.ev_capacity != b.ev_capacity || !same(a.codestream, b.codestream)) return 9;
	if (a.vb_sorted.empty() || a.group_blocks.size() != a.vb_sorted.size()) return 10;
	return 0;
}

// the order in which k_hf_lanes' lanes take a frame's groups (FrontPlan::lane_order) and the bytes of each group's sections, summed
// over the passes; returns the number of groups (or -1)
extern "C" __attribute__((visibility("default"))) int32_t hostsim_lane_order(const uint8_t *buf, size_t size, uint32_t *order, uint64_t *bytes, int32_t capacity) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		Frame fr;
		std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
		if (!parse_frame_front(cs, cs_size, &fr, &tasks, &extra_prec, &plain)) return -1;
		StaticTables st; build_static_tables(fr, &st);
		FrontPlan fp;
		if (build_front_plan(fr, st, cs_size, extra_prec, true, &fp)) return -1;
		const int32_t ng = (int32_t) fr.fh.num_groups;
		if ((int32_t) fp.lane_order.size() != ng || ng > capacity) return -1;
		for (int32_t g = 0; g < ng; ++g) {
			order[g] = fp.lane_order[(size_t) g];
			bytes[g] = 0;
			for (int32_t p = 0; p < fr.fh.num_passes; ++p) bytes[g] += fp.sections[(size_t) p * (size_t) ng + (size_t) g].size;
		}
		return ng;
	} catch (const DecodeError &) { return -1; }
}

// ---- the lane decoder of the LfGroup sections (device/lf_lanes_dev.h) on the CPU, against the host decoder ----
#include "../../j40_amd/csrc/device/lf_lanes_dev.h"

// Every LfGroup section of the stream through lf_lane_step (one lane after the other, tables laid out as k_lf_lanes stages them)
// and through read_lf_group_raw: same status, same planes. Returns 0 when all agree, -1 when the frame is not one the lane decoder
// takes, else 10 * section + what differed (1 status, 2 varblock count, 3-5 LF planes, 6 x-from-y, 7 b-from-y, 8 varblock info).
// *sections = how many were compared, *failed = how many of them ended in an error (on both sides).
extern "C" __attribute__((visibility("default"))) int32_t hostsim_lf_lanes_check(const uint8_t *buf, size_t size, int32_t *sections, int32_t *failed) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	Frame fr;
	std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
	if (sections) *sections = 0;
	if (failed) *failed = 0;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		if (!parse_frame_front(cs, cs_size, &fr, &tasks, &extra_prec, &plain)) return -1;
	} catch (const DecodeError &) { return -1; }
	if (!plain) return -1;
	StaticTables st;
	build_static_tables(fr, &st);
	FrontPlan fp;
	if (build_front_plan(fr, st, cs_size, extra_prec, true, &fp) || !fp.lf_device) return -1;
	std::vector<uint8_t> padded(cs, cs + cs_size);
	padded.resize(cs_size + 32, 0);
	LaneTables T;
	memset(&T, 0, sizeof T);
	T.ctx_map = fp.lf_ctx_map.data(); T.cluster_cfg = fp.lf_cfg.data(); T.alias = fp.lf_alias.data(); T.log_alpha = fp.lf_log_alpha; T.log_bucket = 12 - fp.lf_log_alpha;
	LfLaneFrame F;
	F.tree = fp.lf_tree.data(); F.uses = fp.lf_uses;
	for (size_t g = 0; g < fr.lf_groups.size(); ++g) {
		const LfGroup &gg = fr.lf_groups[g];
		const size_t cells = (size_t) gg.width8 * (size_t) gg.height8, c64 = (size_t) gg.width64 * (size_t) gg.height64;
		std::vector<int16_t> lf[3], xfy(c64 + 1), bfy(c64 + 1), info(2 * cells + 2), sharp(cells + 1);
		for (int c = 0; c < 3; ++c) lf[c].assign(cells + 1, 0);
		DevLfResult res = {0, 0, 0, 0};
		DevLfTask t;
		memset(&t, 0, sizeof t);
		t.codestream = padded.data(); t.byte_off = (uint32_t) tasks[g].byte_off; t.size = (uint32_t) tasks[g].size; t.bit_off = tasks[g].bit_off;
		t.w8 = gg.width8; t.h8 = gg.height8; t.w64 = gg.width64; t.h64 = gg.height64; t.sidx0 = tasks[g].sidx0; t.sidx2 = tasks[g].sidx2; t.nbvb_bits = tasks[g].nbvb_bits;
		for (int c = 0; c < 3; ++c) t.lf[c] = lf[c].data();
		t.xfromy = xfy.data(); t.bfromy = bfy.data(); t.info = info.data(); t.sharp = sharp.data(); t.info_capacity = (uint32_t) (2 * cells); t.result = &res;
		LfLane L;
		lf_lane_init(L, t);
		while (!lf_lane_done(L)) lf_lane_step(L, t, F, T);
		uint32_t host_err = 0;
		LfRaw raw;
		try { BitReader sr(cs + fr.toc.lf_groups[g].offset, fr.toc.lf_groups[g].size); read_lf_group_raw(sr, fr, gg, &raw); } catch (const DecodeError &e) { host_err = e.code; }
		if (sections) ++*sections;
		if (L.err == (uint32_t) ERR_LFFB) continue;   // (the host decodes such a section: nothing to compare)
		if (L.err != host_err) return (int32_t) (10 * g + 1);
		if (host_err) { if (failed) ++*failed; continue; }
		if (L.nb_varblocks != raw.nb_varblocks) return (int32_t) (10 * g + 2);
		for (int c = 0; c < 3; ++c) if (memcmp(lf[c].data(), raw.lf[c].data(), cells * 2) != 0) return (int32_t) (10 * g + 3 + c);
		if (memcmp(xfy.data(), raw.xfromy.data(), c64 * 2) != 0) return (int32_t) (10 * g + 6);
		if (memcmp(bfy.data(), raw.bfromy.data(), c64 * 2) != 0) return (int32_t) (10 * g + 7);
		if (memcmp(info.data(), raw.info.data(), raw.info.size() * 2) != 0) return (int32_t) (10 * g + 8);
	}
	return 0;
}

// ---- the same sections through the row-window form of the lane decoder (device/lf_rows_dev.h, k_lf_rows) ----
#include "../../j40_amd/csrc/device/lf_rows_dev.h"

// As hostsim_lf_lanes_check, for lf_row_step: tables staged as k_lf_rows stages them (leaves carrying cluster and configuration),
// `lanes` sections decoded in lockstep (one step each per turn, as a wavefront's lanes run) with their windows side by side at
// LF_ROW_PITCH, completed pieces copied out between the steps. `win` caps the row length served from the window for the test's
// purposes only through the stream's own sizes (LF_ROW_WIN is a compile-time constant); frames wider than 256 cells per LfGroup
// do not exist, the varblock-info channel exercises the wide path.
static int64_t plain_steps = 0, general_steps = 0, need_seen[32], raw_seen = 0, deferred_sections = 0;
static int32_t general_only = 0;
// (how many samples of the last checks went through the straight-line step / the general one; mode 1: the general step only)
extern "C" __attribute__((visibility("default"))) void hostsim_lf_rows_counts(int64_t *plain, int64_t *general, int32_t reset, int32_t mode) {
	if (plain) *plain = plain_steps;
	if (general) *general = general_steps;
	if (reset) plain_steps = general_steps = 0;
	general_only = mode;
}
// (how often each combination of needs -- LF_NEED_* -- was what the stepped lanes asked for, since the library was loaded)
extern "C" __attribute__((visibility("default"))) void hostsim_lf_rows_needs_seen(int64_t *out32) { for (int i = 0; i < 32; ++i) out32[i] = need_seen[i]; }
// (how many channels the checks' lanes left as residuals, since the library was loaded)
extern "C" __attribute__((visibility("default"))) int64_t hostsim_lf_rows_raw_channels() { return raw_seen; }
// (how many sections ended "lffb" because a run of straight-line steps ran into an error of the stream -- each checked to be one the host's decoder reports too)
extern "C" __attribute__((visibility("default"))) int64_t hostsim_lf_rows_deferred_sections() { return deferred_sections; }
// The fast entries by themselves (lf_rows_fast_entry + the symbol arithmetic of lf_row_step_plain_for) against lane_symbol_in_cluster on
// random alias entries, hybrid-integer configurations (every split_exp / msb / lsb the format allows, max_token anywhere), states and
// bit windows -- what the generator's streams, which use few configurations, do not reach. For every draw without an error: the same
// value, the same next state, the same number of bits taken; "iovf" draws must come out as the entry's flag (and LF_FAST_IOVF_BASE).
// Returns the number of draws compared, -(draw + 1) at the first difference.
extern "C" __attribute__((visibility("default"))) int64_t hostsim_lf_fast_entry_check(uint32_t seed, int32_t draws) {
	uint64_t rng = 0x9e3779b97f4a7c15ull ^ ((uint64_t) seed << 17);
	auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
	int64_t compared = 0;
	for (int32_t k = 0; k < draws; ++k) {
		const int32_t log_alpha = 5 + (int32_t) (next() % 4), log_bucket = 12 - log_alpha;
		const uint32_t split_exp = (uint32_t) (next() % 16);
		uint32_t msb = 0, lsb = 0;
		if (split_exp) { msb = (uint32_t) (next() % (split_exp + 1)); lsb = (uint32_t) (next() % (split_exp - msb + 1)); }
		const uint32_t max_token = (next() & 3) ? (uint32_t) (next() % 300) : 0xfffffu;
		const uint32_t cfg = split_exp | (msb << 4) | (lsb << 8) | (max_token << 12);
		const uint32_t bucket = (uint32_t) (next() % (1u << log_alpha)), cutoff = (uint32_t) (next() % ((1u << log_bucket) + 1));
		const uint32_t off_r = (uint32_t) (next() & 0xfff), tok_r = (uint32_t) (next() & 0xff), d_r = 1 + (uint32_t) (next() % 4096), d_l = 1 + (uint32_t) (next() % 4096);
		const uint64_t e = (uint64_t) cutoff | ((uint64_t) off_r << 8) | ((uint64_t) tok_r << 20) | ((uint64_t) d_r << 28) | ((uint64_t) d_l << 41);
		// the cluster's widest token decides whether the straight-line step takes the cluster at all (lf_rows_leaf_word)
		if ((uint32_t) lf_rows_leaf_word(0, cfg) & (uint32_t) LF_LEAF_WIDE) continue;
		std::vector<uint64_t> alias((size_t) 1 << log_alpha, 0);
		alias[bucket] = e;
		uint32_t state = 0x10000u + (uint32_t) (next() % 0xfff00000u);
		state = (state & ~0xfffu) | (bucket << log_bucket) | (uint32_t) (next() % (1u << log_bucket));
		uint32_t words[6];
		for (uint32_t &w : words) w = (uint32_t) next();
		LaneBits b;
		lane_bits_init(b, (const uint8_t *) words, 0);
		lane_bits_refill(b);
		LaneBits b2 = b;
		uint32_t state_a = state, err_a = 0;
		const int32_t v_a = lane_symbol_in_cluster<true>(b, state_a, alias.data(), log_alpha, log_bucket, 0u, cfg, 0xffffffffu, &err_a);
		// the step's arithmetic
		const LfFastQuad q = lf_rows_fast_entry(e, bucket, cfg);
		const uint32_t pos = state & ((1u << log_bucket) - 1u);
		const bool aliased = pos >= (q[0] & 0xffu);
		const uint32_t w = aliased ? q[1] : q[0] >> 8, base = aliased ? q[3] : q[2];
		uint32_t state_b = (w & 0x1fffu) * (state >> 12) + (w >> 19) + pos;
		const bool renorm = state_b < (1u << 16);
		const uint32_t window = (uint32_t) b2.bits, skip = renorm ? 16u : 0u, extra = (w >> 13) & 31u;
		state_b = renorm ? lf_renorm_word(state_b, window) : state_b;
		const uint32_t mid = lf_bfe(window, skip, extra), taken = skip + extra;
		const int32_t v_b = (int32_t) (base + (mid << lsb));
		if (err_a == (uint32_t) ERR_IOVF) {
			if (!(w & (uint32_t) LF_FAST_IOVF) || base != (uint32_t) LF_FAST_IOVF_BASE) return -(int64_t) (k + 1);
			++compared;
			continue;
		}
		if (err_a || (w & (uint32_t) LF_FAST_IOVF)) return -(int64_t) (k + 1);
		if (v_a != v_b || state_a != state_b || (uint32_t) (b2.nbits - b.nbits) != taken) return -(int64_t) (k + 1);
		++compared;
	}
	return compared;
}

extern "C" __attribute__((visibility("default"))) int32_t hostsim_lf_rows_check(const uint8_t *buf, size_t size, int32_t lanes, int32_t *sections, int32_t *failed) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	Frame fr;
	std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
	if (sections) *sections = 0;
	if (failed) *failed = 0;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		if (!parse_frame_front(cs, cs_size, &fr, &tasks, &extra_prec, &plain)) return -1;
	} catch (const DecodeError &) { return -1; }
	if (!plain) return -1;
	StaticTables st;
	build_static_tables(fr, &st);
	FrontPlan fp;
	if (build_front_plan(fr, st, cs_size, extra_prec, true, &fp) || !fp.lf_device) return -1;
	std::vector<uint8_t> padded(cs, cs + cs_size);
	padded.resize(cs_size + LF_CODESTREAM_PAD, 0);
	// the staged tree
	std::vector<DevTreeNode> tree = fp.lf_tree;
	for (DevTreeNode &n : tree) if (n.prop < 0) { const uint32_t cl = fp.lf_ctx_map[(size_t) n.value]; n.value = lf_rows_leaf_word(cl, fp.lf_cfg[cl]); }
	std::vector<LfFastQuad> fast(fp.lf_alias.size());   // (k_lf_rows' staging: lf_decode.hip)
	for (size_t i = 0; i < fast.size(); ++i) fast[i] = lf_rows_fast_entry(fp.lf_alias[i], (uint32_t) i & ((1u << fp.lf_log_alpha) - 1u), fp.lf_cfg[i >> fp.lf_log_alpha]);
	LfRowTables T;
	T.tree = tree.data(); T.fast = (const uint32_t *) fast.data(); T.alias = fp.lf_alias.data(); T.log_alpha = fp.lf_log_alpha; T.log_bucket = 12 - fp.lf_log_alpha; T.uses = fp.lf_uses;
	// (as the kernel runs by default: leaf-only channels are left as residuals and predicted afterwards; mode bit 2: every channel predicted
	// by its lane)
	if (!(general_only & 4)) T.uses |= (uint32_t) LF_USES_RAW;
	if (lanes < 1) lanes = 1;
	if (lanes > 64) lanes = 64;
	struct Out { std::vector<int16_t> lf[3], xfy, bfy, info, sharp; DevLfResult res; DevLfTask t; };
	const size_t ng = fr.lf_groups.size();
	for (size_t g0 = 0; g0 < ng; g0 += (size_t) lanes) {
		const size_t n = std::min((size_t) lanes, ng - g0);
		std::vector<Out> out(n);
		std::vector<LfRowLane> L(n);
		std::vector<int16_t> wins((size_t) LF_ROW_PITCH * n, (int16_t) 0x5a5a);
		for (size_t k = 0; k < n; ++k) {
			const size_t g = g0 + k;
			const LfGroup &gg = fr.lf_groups[g];
			const size_t cells = (size_t) gg.width8 * (size_t) gg.height8, c64 = (size_t) gg.width64 * (size_t) gg.height64;
			Out &o = out[k];
			for (int c = 0; c < 3; ++c) o.lf[c].assign(cells + 1, 0);
			o.xfy.assign(c64 + 1, 0); o.bfy.assign(c64 + 1, 0); o.info.assign(2 * cells + 2, 0); o.sharp.assign(cells + 1, 0);
			o.res = DevLfResult{0, 0, 0, 0};
			DevLfTask &t = o.t;
			memset(&t, 0, sizeof t);
			t.codestream = padded.data(); t.byte_off = (uint32_t) tasks[g].byte_off; t.size = (uint32_t) tasks[g].size; t.bit_off = tasks[g].bit_off;
			t.w8 = gg.width8; t.h8 = gg.height8; t.w64 = gg.width64; t.h64 = gg.height64; t.sidx0 = tasks[g].sidx0; t.sidx2 = tasks[g].sidx2; t.nbvb_bits = tasks[g].nbvb_bits;
			for (int c = 0; c < 3; ++c) t.lf[c] = o.lf[c].data();
			t.xfromy = o.xfy.data(); t.bfromy = o.bfy.data(); t.info = o.info.data(); t.sharp = o.sharp.data(); t.info_capacity = (uint32_t) (2 * cells); t.result = &o.res;
			lf_row_init(L[k], t, wins.data() + (size_t) LF_ROW_PITCH * k);
		}
		for (bool any = true; any; ) {
			any = false;
			// (the kernel's dispatch: the straight-line step where the lane's sample allows it, in the instantiation that covers what the
			// stepped lanes ask of it together, as the kernel works that out after every general step)
			uint32_t need = 0;
			for (size_t k = 0; k < n; ++k) need |= lf_plain_needs(L[k]);
			++need_seen[need & 31];
			for (size_t k = 0; k < n; ++k) {
				LfRowLane &A = L[k];
				if (lf_row_done(A)) continue;
				any = true;
				if (A.plain_left > 0 && !(general_only & 1)) { lf_row_step_plain_needs(A, T, need); ++plain_steps; }
				else { if (general_only & 1) A.in_run = false; lf_row_step(A, out[k].t, T); ++general_steps; }   // (mode bit 0: every sample through the general step, also inside what would be a run)
			}
			for (size_t k = 0; k < n; ++k) if (L[k].flush_n > 0) lf_row_flush_serial(L[k]);
		}
		for (size_t k = 0; k < n; ++k) {
			const size_t g = g0 + k;
			const LfGroup &gg = fr.lf_groups[g];
			const size_t cells = (size_t) gg.width8 * (size_t) gg.height8, c64 = (size_t) gg.width64 * (size_t) gg.height64;
			uint32_t host_err = 0;
			LfRaw raw;
			try { BitReader sr(cs + fr.toc.lf_groups[g].offset, fr.toc.lf_groups[g].size); read_lf_group_raw(sr, fr, gg, &raw); } catch (const DecodeError &e) { host_err = e.code; }
			if (sections) ++*sections;
			for (int c = 0; c < 7; ++c) if ((L[k].raw_mask >> (4 * c)) & 15u) ++raw_seen;
			const uint32_t status = lf_predict_section_serial(out[k].t, L[k].err, L[k].nb_varblocks, L[k].raw_mask, L[k].stopped_at);   // k_lf_predict's part
			if (status == (uint32_t) ERR_LFFB) {
				// the lane gave the section up: over a form it does not take, a residual too wide for the plane -- or over what a run of
				// straight-line steps ran into (lf_row_deferred), and then the stream has an error, which the host's decoder names
				if (L[k].deferred_real) { if (!host_err) return (int32_t) (10 * g + 1); ++deferred_sections; if (failed) ++*failed; }
				continue;
			}
			if (status != host_err) return (int32_t) (10 * g + 1);
			if (host_err) { if (failed) ++*failed; continue; }
			if (L[k].nb_varblocks != raw.nb_varblocks) return (int32_t) (10 * g + 2);
			for (int c = 0; c < 3; ++c) if (memcmp(out[k].lf[c].data(), raw.lf[c].data(), cells * 2) != 0) return (int32_t) (10 * g + 3 + c);
			if (memcmp(out[k].xfy.data(), raw.xfromy.data(), c64 * 2) != 0) return (int32_t) (10 * g + 6);
			if (memcmp(out[k].bfy.data(), raw.bfromy.data(), c64 * 2) != 0) return (int32_t) (10 * g + 7);
			if (memcmp(out[k].info.data(), raw.info.data(), raw.info.size() * 2) != 0) return (int32_t) (10 * g + 8);
			for (size_t i = cells; i < out[k].lf[0].size(); ++i) if (out[k].lf[0][i] || out[k].lf[1][i] || out[k].lf[2][i] || out[k].sharp[i]) return (int32_t) (10 * g + 9);   // (nothing written past a plane)
		}
	}
	return 0;
}

// The wavefront's schedule as k_lf_rows runs it (lf_decode.hip: runs of plain steps until some live lane leaves its run, then the general
// step for those lanes), over `copies` x the frame's sections side by side: how many wave-level plain iterations and general-step
// rounds a wavefront goes through, and per combination of needs how many plain iterations -- what the kernel's duration is made of
// (MEASUREMENT AID for tools/lf_rows_schedule.py; decodes into scratch planes, checks nothing).
// out: [0] plain iterations, [1] general rounds, [2] samples of the longest lane, [3] samples of all lanes, [4 + need] plain iterations by need
extern "C" __attribute__((visibility("default"))) int32_t hostsim_lf_rows_schedule(const uint8_t *buf, size_t size, int32_t copies, int64_t *out36) {
	const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
	Frame fr;
	std::vector<LfDeviceTask> tasks; std::vector<int32_t> extra_prec; bool plain = true;
	try {
		extract_codestream(buf, size, &cs, &cs_size, &storage);
		if (!parse_frame_front(cs, cs_size, &fr, &tasks, &extra_prec, &plain)) return -1;
	} catch (const DecodeError &) { return -1; }
	if (!plain) return -1;
	StaticTables st;
	build_static_tables(fr, &st);
	FrontPlan fp;
	if (build_front_plan(fr, st, cs_size, extra_prec, true, &fp) || !fp.lf_device) return -1;
	std::vector<uint8_t> padded(cs, cs + cs_size);
	padded.resize(cs_size + LF_CODESTREAM_PAD, 0);
	std::vector<DevTreeNode> tree = fp.lf_tree;
	for (DevTreeNode &n : tree) if (n.prop < 0) { const uint32_t cl = fp.lf_ctx_map[(size_t) n.value]; n.value = lf_rows_leaf_word(cl, fp.lf_cfg[cl]); }
	std::vector<LfFastQuad> fast(fp.lf_alias.size());   // (k_lf_rows' staging: lf_decode.hip)
	for (size_t i = 0; i < fast.size(); ++i) fast[i] = lf_rows_fast_entry(fp.lf_alias[i], (uint32_t) i & ((1u << fp.lf_log_alpha) - 1u), fp.lf_cfg[i >> fp.lf_log_alpha]);
	LfRowTables T;
	T.tree = tree.data(); T.fast = (const uint32_t *) fast.data(); T.alias = fp.lf_alias.data(); T.log_alpha = fp.lf_log_alpha; T.log_bucket = 12 - fp.lf_log_alpha; T.uses = fp.lf_uses;
	if (copies > 0) T.uses |= (uint32_t) LF_USES_RAW;   // (copies < 0: every channel predicted by its lane)
	copies = copies < 0 ? -copies : copies;
	struct Out { std::vector<int16_t> lf[3], xfy, bfy, info, sharp; DevLfResult res; DevLfTask t; };
	const size_t ng = fr.lf_groups.size(), n = std::min((size_t) 64, ng * (size_t) std::max(copies, 1));
	std::vector<Out> out(n);
	std::vector<LfRowLane> L(n);
	std::vector<int16_t> wins((size_t) LF_ROW_PITCH * n, 0);
	for (size_t k = 0; k < n; ++k) {
		const size_t g = k % ng;
		const LfGroup &gg = fr.lf_groups[g];
		const size_t cells = (size_t) gg.width8 * (size_t) gg.height8, c64 = (size_t) gg.width64 * (size_t) gg.height64;
		Out &o = out[k];
		for (int c = 0; c < 3; ++c) o.lf[c].assign(cells + 1, 0);
		o.xfy.assign(c64 + 1, 0); o.bfy.assign(c64 + 1, 0); o.info.assign(2 * cells + 2, 0); o.sharp.assign(cells + 1, 0);
		o.res = DevLfResult{0, 0, 0, 0};
		DevLfTask &t = o.t;
		memset(&t, 0, sizeof t);
		t.codestream = padded.data(); t.byte_off = (uint32_t) tasks[g].byte_off; t.size = (uint32_t) tasks[g].size; t.bit_off = tasks[g].bit_off;
		t.w8 = gg.width8; t.h8 = gg.height8; t.w64 = gg.width64; t.h64 = gg.height64; t.sidx0 = tasks[g].sidx0; t.sidx2 = tasks[g].sidx2; t.nbvb_bits = tasks[g].nbvb_bits;
		for (int c = 0; c < 3; ++c) t.lf[c] = o.lf[c].data();
		t.xfromy = o.xfy.data(); t.bfromy = o.bfy.data(); t.info = o.info.data(); t.sharp = o.sharp.data(); t.info_capacity = (uint32_t) (2 * cells); t.result = &o.res;
		lf_row_init(L[k], t, wins.data() + (size_t) LF_ROW_PITCH * k);
	}
	for (int i = 0; i < 36; ++i) out36[i] = 0;
	std::vector<int64_t> samples(n, 0);
	uint32_t need = LF_NEED_ALL;
	for (;;) {
		for (;;) {   // lf_row_run_plain_for
			bool any_plain = false;
			for (size_t k = 0; k < n; ++k) any_plain |= L[k].plain_left > 0;
			if (!any_plain) break;
			++out36[0]; ++out36[4 + (need & 31)];
			for (size_t k = 0; k < n; ++k) if (L[k].plain_left > 0) { lf_row_step_plain_needs(L[k], T, need); ++samples[k]; }
			bool leave = false;
			for (size_t k = 0; k < n; ++k) leave |= !(L[k].plain_left > 0) && L[k].live;
			if (leave) break;
		}
		++out36[1];
		for (size_t k = 0; k < n; ++k) if (!(L[k].plain_left > 0)) { const bool was = L[k].live && !L[k].setup && !L[k].in_run; lf_row_step(L[k], out[k].t, T); if (was) ++samples[k]; }
		for (size_t k = 0; k < n; ++k) if (L[k].flush_n > 0) lf_row_flush_serial(L[k]);
		bool live = false;
		for (size_t k = 0; k < n; ++k) live |= L[k].live;
		if (!live) break;
		need = 0;
		for (size_t k = 0; k < n; ++k) need |= lf_plain_needs(L[k]);
	}
	for (size_t k = 0; k < n; ++k) { out36[2] = std::max(out36[2], samples[k]); out36[3] += samples[k]; }
	return (int32_t) n;
}

// ---- known-answer hooks for the inverse Squeeze step (device/squeeze_dev.h), tests/test_squeeze.py ----
extern "C" __attribute__((visibility("default"))) int32_t hostsim_squeeze_tendency(int32_t B, int32_t a, int32_t n) { return squeeze_tendency(B, a, n); }
extern "C" __attribute__((visibility("default"))) void hostsim_unsqueeze_line(const int16_t *avg, int32_t n_avg, const int16_t *res, int32_t n_res, int16_t *out) {
	unsqueeze_line(avg, 1, res, 1, n_avg, n_res, out, 1);
}

// ---- the device memory cache's bookkeeping (device/block_cache.hpp) on the CPU: a backend with a byte budget stands in for
// hipMalloc / hipFree, the driver below is runtime.hip's cache_acquire / cache_release / cache_trim line for line ----
#include "../../j40_amd/csrc/device/block_cache.hpp"
#include <map>
#include <random>
// Random acquire / release / trim sequences; checks after every step: no two live blocks overlap, every live block lies inside a
// live backend allocation, the cache's byte count equals the sum of its idle blocks, a slab is only freed with all blocks idle,
// trim leaves nothing idle but blocks of partly used slabs, an allocation that fails with idle blocks around succeeds after the
// trim. Returns 0, or the number of the first check that failed.
extern "C" __attribute__((visibility("default"))) int32_t hostsim_block_cache_selftest(uint32_t seed, int32_t steps, uint64_t budget, uint64_t limit) {
	using j40hip_rt::BlockCacheCore;
	BlockCacheCore cache;
	std::map<uintptr_t, size_t> backend;   // live backend allocations: base -> bytes (addresses are made up: nothing is dereferenced)
	uint64_t used = 0; uintptr_t next_addr = (uintptr_t) 1 << 40;
	auto b_alloc = [&](size_t bytes) -> void * { if (used + bytes > budget) return nullptr; used += bytes; const uintptr_t a = next_addr; next_addr += (bytes + 4095) & ~(uintptr_t) 4095; backend[a] = bytes; return (void *) a; };
	auto b_free = [&](void *q) -> bool { auto it = backend.find((uintptr_t) q); if (it == backend.end()) return false; used -= it->second; backend.erase(it); return true; };
	struct Live { void *p; size_t got; };
	std::vector<Live> live;
	bool bad_free = false;
	auto trim = [&] { std::vector<void *> gone; cache.trim(&gone); for (void *q : gone) if (!b_free(q)) bad_free = true; };
	auto acquire = [&](size_t bytes, size_t *got) -> void * {
		bool clean = false;
		bytes = BlockCacheCore::size_class(bytes);
		if (void *q = cache.take(bytes, got, &clean)) return q;
		if (BlockCacheCore::slab_class(bytes)) {
			const int n = BlockCacheCore::slab_blocks(bytes);
			if (void *q = b_alloc(bytes * (size_t) n)) { cache.adopt_slab(q, bytes, n); *got = bytes; return q; }
		}
		void *q = b_alloc(bytes);
		if (!q) { trim(); q = b_alloc(bytes); }
		if (q) *got = bytes;
		return q;
	};
	auto release = [&](const Live &l) { void *gone = nullptr; cache.give(l.p, l.got, false, (size_t) limit, &gone); if (gone && !b_free(gone)) bad_free = true; };
	auto check = [&]() -> int32_t {
		if (bad_free) return 1;   // something was freed that the backend never handed out (or twice)
		size_t sum = 0;
		for (const auto &b : cache.idle) sum += b.bytes;
		if (sum != cache.idle_bytes) return 2;
		std::vector<std::pair<uintptr_t, uintptr_t>> spans;
		for (const Live &l : live) spans.push_back({(uintptr_t) l.p, (uintptr_t) l.p + l.got});
		for (const auto &b : cache.idle) spans.push_back({(uintptr_t) b.ptr, (uintptr_t) b.ptr + b.bytes});
		std::sort(spans.begin(), spans.end());
		for (size_t i = 1; i < spans.size(); ++i) if (spans[i].first < spans[i - 1].second) return 3;   // two blocks overlap
		for (const auto &sp : spans) {   // inside a live backend allocation
			auto it = backend.upper_bound(sp.first);
			if (it == backend.begin()) return 4;
			--it;
			if (sp.second > it->first + it->second) return 4;
		}
		for (const auto &sl : cache.slabs) if (sl.base) {   // a slab's idle count is what the idle list says
			int n = 0;
			for (const auto &b : cache.idle) if (b.slab >= 0 && &cache.slabs[(size_t) b.slab] == &sl) ++n;
			if (n != sl.idle || sl.idle > sl.total) return 5;
			if (!backend.count((uintptr_t) sl.base)) return 6;
		}
		return 0;
	};
	std::mt19937_64 rng(seed);
	static const size_t SIZES[] = {4096, 100000, 300000, 1 << 20, 12 << 20, 13 << 20, 211 << 20, 220 << 20, (size_t) 600 << 20};
	for (int32_t s = 0; s < steps; ++s) {
		const uint64_t r = rng();
		if ((r & 7) < 4 || live.empty()) {
			size_t got = 0;
			const size_t want = SIZES[(r >> 8) % (sizeof SIZES / sizeof SIZES[0])] + (size_t) ((r >> 16) & 0xfff);
			const bool idle_before = !cache.idle.empty();
			void *q = acquire(want, &got);
			if (q) { if (got < want) return 7; live.push_back({q, got}); }
			else if (idle_before && used + BlockCacheCore::size_class(want) <= budget) return 8;   // refused although trimming would have made room
		} else if ((r & 7) < 7) {
			const size_t i = (size_t) ((r >> 8) % live.size());
			release(live[i]);
			live[i] = live.back(); live.pop_back();
		} else {
			trim();
			for (const auto &b : cache.idle) if (b.slab < 0 || cache.slabs[(size_t) b.slab].idle == cache.slabs[(size_t) b.slab].total) return 9;
		}
		if (const int32_t e = check()) return e;
	}
	for (const Live &l : live) release(l);
	live.clear();
	trim();
	if (const int32_t e = check()) return e;
	if (!cache.idle.empty() || cache.idle_bytes != 0 || used != 0 || !backend.empty()) return 10;   // everything went back
	return 0;
}

// ---- the restoration filters' device functions (device/restore_dev.h) run sample by sample on the CPU, out of place like the kernels
// (restore_kernels.h): three planes [3][h][w] in place; sharpness / hfmul_inv per 8x8 cell; params24 in ref_stage_restoration's order
// (gab.enabled, gab.weights[3][2], epf.iters, sharp_lut[8], channel_scale[3], quant_mul, pass0, pass2, border_sad_mul, sigma_for_modular);
// mode 1: the filters as intended, 2: as j40's routines stand. Returns 0 or the routines' own 4-char complaint.
#include "../../j40_amd/csrc/device/restore_dev.h"
namespace {
struct HostPlanes { const float *p[3]; size_t pitch; float operator()(int32_t c, int32_t x, int32_t y) const { return p[c][(size_t) y * pitch + (size_t) x]; } };
template <int STEP> void hostsim_epf_step(const j40hip::RestoreParams &p, const std::vector<float> &sigma, std::vector<float> &a, std::vector<float> &b) {
	const size_t plane = (size_t) p.width * (size_t) p.height;
	const HostPlanes inside = {{a.data(), a.data() + plane, a.data() + 2 * plane}, (size_t) p.width};
	const j40hip::EpfMirrored<HostPlanes> in = {inside, p.width, p.height};
	for (int32_t y = 0; y < p.height; ++y) for (int32_t x = 0; x < p.width; ++x) {
		const float rs = sigma[(size_t) (y >> 3) * (size_t) p.w8 + (size_t) (x >> 3)];
		float v[3];
		if (rs < 0.0f) for (int c = 0; c < 3; ++c) v[c] = in(c, x, y);
		else j40hip::epf_sample<STEP>(in, p, x, y, rs, v);
		for (int c = 0; c < 3; ++c) b[(size_t) c * plane + (size_t) y * (size_t) p.width + (size_t) x] = v[c];
	}
	a.swap(b);
}
}
extern "C" __attribute__((visibility("default"))) uint32_t hostsim_restoration(float *xyb, int32_t w, int32_t h, const int16_t *sharpness, const float *hfmul_inv, const float *params24, int32_t mode, float *sigma_out) {
	using namespace j40hip;
	auto nonzero = [](float x) { return std::isfinite(x) && std::fabs(x) >= 1e-8f; };
	auto e4 = [](const char *s) { return ((uint32_t) (uint8_t) s[0] << 24) | ((uint32_t) (uint8_t) s[1] << 16) | ((uint32_t) (uint8_t) s[2] << 8) | (uint32_t) (uint8_t) s[3]; };
	RestoreParams p;
	memset(&p, 0, sizeof p);
	p.width = w; p.height = h; p.w8 = (w + 7) / 8; p.h8 = (h + 7) / 8; p.quirk = mode == 2;
	const bool gab = params24[0] != 0.0f; const int iters = (int) params24[7];
	if (gab) for (int c = 0; c < 3; ++c) {
		const float w1 = params24[1 + c * 2], w2 = params24[2 + c * 2], wsum = 1.0f + w1 * 4 + w2 * 4;
		if (!nonzero(wsum)) return e4("gab0");
		p.gab_w[c][0] = 1.0f / wsum; p.gab_w[c][1] = w1 / wsum; p.gab_w[c][2] = w2 / wsum;
	}
	if (iters > 0) {
		for (int i = 0; i < 8; ++i) { const float q = params24[19] * params24[8 + i]; if (!nonzero(q)) return e4("epf0"); p.inv_quant_sharp_lut[i] = 1.0f / q; }
		const float scale[3] = {params24[20], 1.0f, params24[21]};
		for (int k = 0; k < 3; ++k) { p.sigma_scale[k] = scale[k] * 1.9330952441687859f; p.border_scale[k] = p.sigma_scale[k] * params24[22]; }
		for (int c = 0; c < 3; ++c) p.channel_scale[c] = params24[16 + c];
	}
	const size_t plane = (size_t) w * (size_t) h, cells = (size_t) p.w8 * (size_t) p.h8;
	std::vector<float> a(xyb, xyb + 3 * plane), b(3 * plane), sigma(cells);
	if (iters > 0) {
		uint16_t ub = 0;
		for (size_t i = 0; i < cells; ++i) ub |= (uint16_t) sharpness[i];
		if (!(ub < 8)) return e4("shrp");
		for (size_t i = 0; i < cells; ++i) sigma[i] = epf_recip_sigma(p, sharpness[i], hfmul_inv[i]);
		if (sigma_out) memcpy(sigma_out, sigma.data(), cells * 4);
	}
	if (gab) {
		if (w < 2) return e4("TODO");
		for (int c = 0; c < 3; ++c) for (int32_t y = 0; y < h; ++y) {
			const float *base = a.data() + (size_t) c * plane, *n = base + (size_t) (y > 0 ? y - 1 : 0) * (size_t) w, *l = base + (size_t) y * (size_t) w, *s = base + (size_t) (y + 1 < h ? y + 1 : y) * (size_t) w;
			for (int32_t x = 0; x < w; ++x) b[(size_t) c * plane + (size_t) y * (size_t) w + (size_t) x] = gaborish_sample(n, l, s, x, w, p.gab_w[c][0], p.gab_w[c][1], p.gab_w[c][2]);
		}
		a.swap(b);
	}
	if (iters >= 3) hostsim_epf_step<0>(p, sigma, a, b);
	if (iters >= 1) hostsim_epf_step<1>(p, sigma, a, b);
	if (iters >= 2) hostsim_epf_step<2>(p, sigma, a, b);
	memcpy(xyb, a.data(), 3 * plane * 4);
	return 0;
}
