// j40_amd/csrc/capi_host.cpp -- host half of the thin C-ABI (include/j40hip.h): parse + stage accessors
#include <cstdio>
#include <cstring>
#include <thread>
#include <type_traits>
#include "capi.hpp"
#include "tables.hpp"

using namespace j40hip;

extern "C" {

j40hip_frame *j40hip_frame_parse_ex(const void *buf, size_t size, int threads, uint32_t flags, uint32_t *err) { return j40hip_frame_parse_with(buf, size, threads, flags, nullptr, nullptr, err); }

// (internal; j40hip_frame_parse_on in device/runtime.hip passes the device's LfGroup decoder)
j40hip_frame *j40hip_frame_parse_with(const void *buf, size_t size, int threads, uint32_t flags, LfDeviceDecoder lf_decoder, void *lf_ctx, uint32_t *err) {
	j40hip_frame *h = new j40hip_frame();
	uint32_t code = 0;
	try {
		h->frame.defer_lf_tail = (flags & 1u) != 0;
		h->frame.lf_decoder = lf_decoder; h->frame.lf_decoder_ctx = lf_ctx;
		extract_codestream((const uint8_t *) buf, size, &h->cs, &h->cs_size, &h->cs_storage, &h->container_stray_tail);
		h->bare_codestream = h->cs == (const uint8_t *) buf && h->cs_size == size;
		parse_frame(h->cs, h->cs_size, &h->frame, threads);
		h->threads = threads < 1 ? 1 : threads > 16 ? 16 : threads;
	} catch (const DecodeError &e) { code = e.code; }
	catch (const std::bad_alloc &) { code = E4("!mem"); }
	h->frame.lf_decoder = nullptr; h->frame.lf_decoder_ctx = nullptr;   // (the context lives on the caller's stack)
	if (err) *err = code;
	if (code) { delete h; return nullptr; }
	return h;
}

j40hip_frame *j40hip_frame_parse(const void *buf, size_t size, int threads, uint32_t *err) { return j40hip_frame_parse_ex(buf, size, threads, 0, err); }

j40hip_frame *j40hip_frame_parse_streamed(const void *buf, size_t size, int threads, uint32_t flags, j40hip_need_bytes need, j40hip_have_bytes have, void *ctx, uint32_t *err) {
	if (!need || !have) return j40hip_frame_parse_ex(buf, size, threads, flags, err);
	need(ctx, size < 2 ? size : 2);
	const uint8_t *p = (const uint8_t *) buf;
	if (size < 2 || !(p[0] == 0xff && p[1] == 0x0a)) {   // a container (or nothing a decoder knows): its boxes are walked once all of it is there
		need(ctx, size);
		return j40hip_frame_parse_ex(buf, size, threads, flags, err);
	}
	j40hip_frame *h = new j40hip_frame();
	uint32_t code = 0;
	try {
		h->frame.defer_lf_tail = (flags & 1u) != 0;
		h->frame.need_bytes = need; h->frame.have_bytes = have; h->frame.need_ctx = ctx;
		h->cs = p; h->cs_size = size; h->bare_codestream = true;
		parse_frame(h->cs, h->cs_size, &h->frame, threads);
		h->threads = threads < 1 ? 1 : threads > 16 ? 16 : threads;
	} catch (const DecodeError &e) { code = e.code; }
	catch (const std::bad_alloc &) { code = E4("!mem"); }
	catch (const std::exception &) { code = E4("!mem"); }   // (std::system_error from a worker thread that could not start: nothing may cross the C boundary)
	h->frame.need_bytes = nullptr; h->frame.have_bytes = nullptr; h->frame.need_ctx = nullptr;   // (the source lives with the caller)
	if (err) *err = code;
	if (code) { delete h; return nullptr; }
	return h;
}

// The seam for a host that has done its own parsing (a patched j40: INTEGRATION.md): a frame handle built from the plan view
// instead of from a bitstream. Everything is copied except the codestream, which must outlive the handle.
j40hip_frame *j40hip_frame_from_vardct_view(const j40hip_vardct_view *v, uint32_t *err) {
	uint32_t code = 0;
	j40hip_frame *h = new j40hip_frame();
	try {
		if (!v || !v->codestream || v->num_passes < 1 || v->num_passes > 11 || v->num_groups < 1 || v->num_lf_groups < 1 || v->width < 1 || v->height < 1) J40HIP_RAISE("rnge");
		// kernels read the bit window up to 16 bytes past a section: keep a padded copy
		h->cs_storage.assign(v->codestream, v->codestream + v->codestream_size);
		h->cs_storage.resize(v->codestream_size + 16, 0);
		h->cs = h->cs_storage.data(); h->cs_size = v->codestream_size;
		h->from_view = true;
		Frame &f = h->frame;
		f.im.width = v->width; f.im.height = v->height; f.im.bpp = v->bpp; f.im.exp_bits = 0; f.im.xyb_encoded = true; f.im.grey = false;
		f.im.intensity_target = v->intensity_target; f.im.quant_bias_num = v->quant_bias_num;
		for (int i = 0; i < 3; ++i) { f.im.opsin_bias[i] = v->opsin_bias[i]; f.im.quant_bias[i] = v->quant_bias[i]; for (int j = 0; j < 3; ++j) f.im.opsin_inv_mat[i][j] = v->opsin_inv_mat[3 * i + j]; }
		FrameHeader &fh = f.fh;
		fh.is_modular = false; fh.do_ycbcr = false; fh.group_size_shift = 8; fh.num_passes = v->num_passes;
		fh.x_qm_scale = v->x_qm_scale; fh.b_qm_scale = v->b_qm_scale; fh.width = v->width; fh.height = v->height;
		fh.gcolumns = (v->width + 255) / 256; fh.grows = (v->height + 255) / 256; fh.ggcolumns = (v->width + 2047) / 2048; fh.ggrows = (v->height + 2047) / 2048;
		fh.num_groups = (int64_t) fh.gcolumns * fh.grows; fh.num_lf_groups = (int64_t) fh.ggcolumns * fh.ggrows;
		if (fh.num_groups != v->num_groups || fh.num_lf_groups != v->num_lf_groups) J40HIP_RAISE("rnge");
		f.global_scale = v->global_scale; f.nb_block_ctx = v->nb_block_ctx; f.nb_qf_thr = v->nb_qf_thr; for (int i = 0; i < 3; ++i) f.nb_lf_thr[i] = v->nb_lf_thr[i];
		f.block_ctx_map.assign(v->block_ctx_map, v->block_ctx_map + v->block_ctx_size);
		f.inv_colour_factor = v->inv_colour_factor; f.base_corr_x = v->base_corr_x; f.base_corr_b = v->base_corr_b; f.x_factor_lf = v->x_factor_lf; f.b_factor_lf = v->b_factor_lf;
		f.num_hf_presets = v->num_hf_presets;
		f.num_gm_channels = 0; f.gmodular.channel.resize(v->sections_have_trailer ? 1 : 0);
		for (int32_t p = 0; p < v->num_passes; ++p) {
			const j40hip_codespec_view &sv = v->coeff_specs[p];
			CodeSpec &cs = f.coeff_codespec[p];
			cs.num_dist = sv.num_dist; cs.num_clusters = sv.num_clusters; cs.lz77_enabled = sv.lz77_enabled != 0; cs.use_prefix_code = sv.use_prefix_code != 0;
			cs.min_symbol = sv.min_symbol; cs.min_length = sv.min_length; cs.log_alpha_size = sv.log_alpha_size;
			cs.lz_len_cfg.split_exp = sv.lz_len_split_exp; cs.lz_len_cfg.msb_in_token = sv.lz_len_msb; cs.lz_len_cfg.lsb_in_token = sv.lz_len_lsb;
			cs.cluster_map.assign(sv.cluster_map, sv.cluster_map + sv.num_dist + (sv.lz77_enabled ? 1 : 0));
			cs.clusters.resize((size_t) sv.num_clusters);
			for (int32_t c = 0; c < sv.num_clusters; ++c) {
				const j40hip_cluster_view &cv = sv.clusters[c];
				Cluster &cl = cs.clusters[(size_t) c];
				cl.cfg.split_exp = cv.split_exp; cl.cfg.msb_in_token = cv.msb_in_token; cl.cfg.lsb_in_token = cv.lsb_in_token;
				if (cs.use_prefix_code) cl.lengths.assign(cv.lengths, cv.lengths + cv.alphabet_size);
				else cl.D.assign(cv.D, cv.D + ((size_t) 1 << sv.log_alpha_size));
			}
			finish_code_spec_tables(&cs);
			for (int o = 0; o < 13; ++o) for (int c = 0; c < 3; ++c) if (const int32_t *ord = v->orders[(p * 13 + o) * 3 + c])
				f.orders[p][o][c].assign(ord, ord + ((size_t) 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1])));
		}
		for (int i = 0; i < 17; ++i) if (v->dq_matrix[i]) {
			DqMatrix &dq = f.dq_matrix[i];
			dq.params.resize((size_t) v->dq_size[i]);
			for (int32_t k = 0; k < v->dq_size[i]; ++k) for (int c = 0; c < 3; ++c) dq.params[(size_t) k][(size_t) c] = v->dq_matrix[i][3 * k + c];
			dq.loaded = true;
		}
		f.lf_groups.resize((size_t) v->num_lf_groups);
		for (int32_t g = 0; g < v->num_lf_groups; ++g) {
			const j40hip_lf_group_view &gv = v->lf_groups[g];
			LfGroup &gg = f.lf_groups[(size_t) g];
			gg.idx = g; gg.left = gv.left; gg.top = gv.top; gg.width = gv.width; gg.height = gv.height; gg.width8 = gv.width8; gg.height8 = gv.height8; gg.width64 = gv.width64; gg.height64 = gv.height64;
			const size_t cells = (size_t) gv.width8 * (size_t) gv.height8, c64 = (size_t) gv.width64 * (size_t) gv.height64;
			gg.blocks.assign(gv.blocks, gv.blocks + cells); gg.lfindices.assign(gv.lfindices, gv.lfindices + cells);
			for (int c = 0; c < 3; ++c) gg.llfcoeffs[c].assign(gv.llfcoeffs[c], gv.llfcoeffs[c] + cells);
			gg.xfromy.assign(gv.xfromy, gv.xfromy + c64); gg.bfromy.assign(gv.bfromy, gv.bfromy + c64);
			gg.varblocks.assign((size_t) gv.nb_varblocks, VarblockInfo{0, 0.0f, 0, 0, 0});
			for (int32_t y = 0; y < gv.height8; ++y) for (int32_t x = 0; x < gv.width8; ++x) {   // a varblock's top-left cell names its transform
				const int32_t cell = gv.blocks[(size_t) y * (size_t) gv.width8 + (size_t) x], sel = cell >> 20, vi = cell & 0xfffff;
				if (sel < 2) continue;
				if (vi >= gv.nb_varblocks || sel - 2 >= 27) J40HIP_RAISE("rnge");
				gg.varblocks[(size_t) vi] = VarblockInfo{gv.coeffoff_qfidx[vi], gv.hfmul_inv[vi], x, y, sel - 2};
			}
			gg.loaded = true;
		}
		const size_t nsec = (size_t) v->num_passes * (size_t) v->num_groups;
		f.toc.single = nsec == 1 && v->sections[0].bit_off != 0;
		f.toc.pass_groups.resize(nsec);
		for (size_t i = 0; i < nsec; ++i) { f.toc.pass_groups[i].offset = v->sections[i].byte_off; f.toc.pass_groups[i].size = v->sections[i].size; }
		if (f.toc.single) { f.toc.single_section = f.toc.pass_groups[0]; f.single_pass_group_bitpos = v->sections[0].bit_off; f.toc.single_declared_end = v->single_declared_end; }
	} catch (const DecodeError &e) { code = e.code; }
	catch (const std::bad_alloc &) { code = E4("!mem"); }
	if (err) *err = code;
	if (code) { delete h; return nullptr; }
	return h;
}

uint32_t j40hip_frame_after_frame_status(const j40hip_frame *h) {
	if (!h) return 0;
	const size_t end = h->frame.toc.end_offset;
	if (!h->bare_codestream) {
		// container: the reference asks for the next box when it looks behind the frame, and a header cut short is `shrt`
		// (j40__box_header); like above it only looks while its main buffer still covers the end of the frame
		// A frame of several sections whose TOC adds up to more than the boxes hold: the reference seeks to the frame's end after the
		// last section (j40__end_of_frame, j40.h:7894) and runs out of boxes -- `shrt` (a bare codestream lets that seek pass)
		if (!h->frame.toc.single && end > h->cs_size) return E4("shrt");
		return h->container_stray_tail && (h->frame.toc.single || end < 0x10000) ? E4("shrt") : 0;
	}
	if (end >= h->cs_size) return 0;
	if (h->frame.toc.single) return E4("excs");
	return end < std::min<size_t>(h->cs_size, 0x10000) ? E4("excs") : 0;
}

void j40hip_frame_free(j40hip_frame *f) {
	if (!f) return;
	j40hip_release_device(f);
	delete f;
}

void j40hip_frame_info(const j40hip_frame *h, int64_t *out) {
	const Frame &f = h->frame;
	int i = 0;
	out[i++] = f.fh.width; out[i++] = f.fh.height; out[i++] = f.fh.is_modular;
	out[i++] = f.fh.num_lf_groups; out[i++] = f.fh.num_groups; out[i++] = f.fh.num_passes;
	out[i++] = f.nb_block_ctx; out[i++] = (int64_t) f.block_ctx_map.size(); out[i++] = f.num_hf_presets;
	out[i++] = f.global_scale; out[i++] = f.quant_lf; out[i++] = f.fh.x_qm_scale; out[i++] = f.fh.b_qm_scale;
	out[i++] = f.nb_qf_thr; out[i++] = f.nb_lf_thr[0]; out[i++] = f.nb_lf_thr[1]; out[i++] = f.nb_lf_thr[2];
	out[i++] = f.fh.group_size_shift; out[i++] = f.im.bpp; out[i++] = (int64_t) f.im.ec.size(); out[i++] = f.im.xyb_encoded;
}

size_t j40hip_frame_codestream_size(const j40hip_frame *h) { return h->cs_size; }
int64_t j40hip_frame_num_sections(const j40hip_frame *h) { return h->frame.toc.single ? 1 : (int64_t) h->frame.toc.pass_groups.size(); }
// bytes of every pass-group section as the TOC lists them (j40.h:5529), pass-major; returns how many
int64_t j40hip_frame_section_sizes(const j40hip_frame *h, int64_t *out) {
	if (h->frame.toc.single) { if (out) out[0] = (int64_t) h->frame.toc.single_section.size; return 1; }
	const std::vector<Section> &pg = h->frame.toc.pass_groups;
	if (out) for (size_t i = 0; i < pg.size(); ++i) out[i] = (int64_t) pg[i].size;
	return (int64_t) pg.size();
}

void j40hip_frame_lf_group_info(const j40hip_frame *h, int64_t gg, int32_t *out) {
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	out[0] = g.left; out[1] = g.top; out[2] = g.width; out[3] = g.height; out[4] = g.width8; out[5] = g.height8;
	out[6] = g.width64; out[7] = g.height64; out[8] = (int32_t) g.varblocks.size();
}

int j40hip_frame_lf_group_plane(const j40hip_frame *h, int64_t gg, int which, void *out) {
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	switch (which) {
	case 0: memcpy(out, g.blocks.data(), g.blocks.size() * 4); return 0;
	case 1: memcpy(out, g.lfindices.data(), g.lfindices.size()); return 0;
	case 2: memcpy(out, g.xfromy.data(), g.xfromy.size() * 2); return 0;
	case 3: memcpy(out, g.bfromy.data(), g.bfromy.size() * 2); return 0;
	}
	return -1;
}

void j40hip_frame_varblocks(const j40hip_frame *h, int64_t gg, int32_t *coeffoff_qfidx, float *hfmul_inv) {
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	for (size_t i = 0; i < g.varblocks.size(); ++i) { coeffoff_qfidx[i] = g.varblocks[i].coeffoff_qfidx; hfmul_inv[i] = g.varblocks[i].hfmul_inv; }
}

void j40hip_frame_llf(const j40hip_frame *h, int64_t gg, int c, float *out) {
	finish_lf_tail(&const_cast<j40hip_frame *>(h)->frame);   // (frames parsed with the tail deferred: compute it here for whoever asks)
	const LfGroup &g = h->frame.lf_groups[(size_t) gg];
	memcpy(out, g.llfcoeffs[c].data(), g.llfcoeffs[c].size() * 4);
}

int32_t j40hip_frame_dq_matrix(const j40hip_frame *h, int idx, float *out) {
	const DqMatrix &dq = h->frame.dq_matrix[idx];
	if (!dq.loaded) return 0;
	for (size_t i = 0; i < dq.params.size(); ++i) for (int c = 0; c < 3; ++c) out[i * 3 + (size_t) c] = dq.params[i][(size_t) c];
	return (int32_t) dq.params.size();
}

int32_t j40hip_frame_order(const j40hip_frame *h, int pass, int idx, int c, int32_t *out) {
	const std::vector<int32_t> &o = h->frame.orders[pass][idx][c];
	memcpy(out, o.data(), o.size() * 4);
	return (int32_t) o.size();
}

int32_t j40hip_frame_block_ctx_map(const j40hip_frame *h, uint8_t *out) {
	memcpy(out, h->frame.block_ctx_map.data(), h->frame.block_ctx_map.size());
	return (int32_t) h->frame.block_ctx_map.size();
}

int j40hip_frame_global_plane(const j40hip_frame *h, int c, int16_t *out, int32_t *w, int32_t *hh) {
	const Modular &m = h->frame.gmodular;
	if (c < 0 || c >= (int) m.channel.size()) return -1;
	const Plane &p = m.channel[(size_t) c];
	*w = p.width; *hh = p.height;
	if (out && !p.px.empty()) memcpy(out, p.px.data(), p.px.size() * 2);
	return 0;
}

static void fill_codespec_view(j40hip_frame *h, const CodeSpec &spec, j40hip_codespec_view *v) {
	h->views.clusters.emplace_back();
	std::vector<j40hip_cluster_view> &cl = h->views.clusters.back();
	for (const Cluster &c : spec.clusters) {
		j40hip_cluster_view cv;
		cv.split_exp = c.cfg.split_exp; cv.msb_in_token = c.cfg.msb_in_token; cv.lsb_in_token = c.cfg.lsb_in_token;
		cv.D = c.D.empty() ? nullptr : c.D.data(); cv.lengths = c.lengths.empty() ? nullptr : c.lengths.data(); cv.alphabet_size = (int32_t) c.lengths.size();
		cl.push_back(cv);
	}
	v->num_dist = spec.num_dist; v->num_clusters = spec.num_clusters; v->lz77_enabled = spec.lz77_enabled; v->use_prefix_code = spec.use_prefix_code;
	v->min_symbol = spec.min_symbol; v->min_length = spec.min_length; v->log_alpha_size = spec.log_alpha_size;
	v->lz_len_split_exp = spec.lz_len_cfg.split_exp; v->lz_len_msb = spec.lz_len_cfg.msb_in_token; v->lz_len_lsb = spec.lz_len_cfg.lsb_in_token;
	v->cluster_map = spec.cluster_map.data(); v->clusters = cl.data();
}

uint32_t j40hip_frame_vardct_view(j40hip_frame *h, j40hip_vardct_view *v) {
	finish_lf_tail(&h->frame);   // (frames parsed with the tail deferred: the view carries LLF coefficients)
	const Frame &f = h->frame;
	if (f.fh.is_modular || f.im.grey || f.fh.do_ycbcr || f.im.bpp < 8 || f.im.exp_bits) return E4("TODO");
	memset(v, 0, sizeof *v);
	h->views = j40hip_frame::Views();
	h->views.clusters.reserve(16);
	v->width = f.fh.width; v->height = f.fh.height; v->num_passes = f.fh.num_passes; v->num_groups = (int32_t) f.fh.num_groups; v->num_lf_groups = (int32_t) f.fh.num_lf_groups;
	v->nb_block_ctx = f.nb_block_ctx; v->nb_qf_thr = f.nb_qf_thr; for (int i = 0; i < 3; ++i) v->nb_lf_thr[i] = f.nb_lf_thr[i];
	v->num_hf_presets = f.num_hf_presets; v->bpp = f.im.bpp;
	v->sections_have_trailer = (int32_t) f.gmodular.channel.size() > f.num_gm_channels;
	v->check_section_end = f.toc.single && !v->sections_have_trailer;
	v->single_declared_end = (uint32_t) f.toc.single_declared_end;
	v->global_scale = f.global_scale; v->x_qm_scale = f.fh.x_qm_scale; v->b_qm_scale = f.fh.b_qm_scale; v->x_factor_lf = f.x_factor_lf; v->b_factor_lf = f.b_factor_lf;
	for (int i = 0; i < 3; ++i) { v->quant_bias[i] = f.im.quant_bias[i]; v->opsin_bias[i] = f.im.opsin_bias[i]; for (int j = 0; j < 3; ++j) v->opsin_inv_mat[i * 3 + j] = f.im.opsin_inv_mat[i][j]; }
	v->quant_bias_num = f.im.quant_bias_num; v->base_corr_x = f.base_corr_x; v->base_corr_b = f.base_corr_b; v->inv_colour_factor = f.inv_colour_factor;
	v->intensity_target = f.im.intensity_target;
	v->codestream = h->cs; v->codestream_size = h->cs_size;
	v->block_ctx_map = f.block_ctx_map.data(); v->block_ctx_size = (int32_t) f.block_ctx_map.size();
	h->views.specs.assign((size_t) f.fh.num_passes, j40hip_codespec_view());
	for (int32_t p = 0; p < f.fh.num_passes; ++p) fill_codespec_view(h, f.coeff_codespec[p], &h->views.specs[(size_t) p]);
	v->coeff_specs = h->views.specs.data();
	for (int32_t p = 0; p < f.fh.num_passes; ++p) for (int o = 0; o < 13; ++o) for (int c = 0; c < 3; ++c)
		v->orders[(p * 13 + o) * 3 + c] = f.orders[p][o][c].empty() ? nullptr : f.orders[p][o][c].data();
	h->views.dq.assign(17, {});
	for (int i = 0; i < 17; ++i) if (f.dq_matrix[i].loaded) {
		for (const auto &w : f.dq_matrix[i].params) for (int c = 0; c < 3; ++c) h->views.dq[(size_t) i].push_back(w[(size_t) c]);
		v->dq_matrix[i] = h->views.dq[(size_t) i].data(); v->dq_size[i] = (int32_t) f.dq_matrix[i].params.size();
	}
	for (const LfGroup &g : f.lf_groups) {
		j40hip_lf_group_view gv;
		memset(&gv, 0, sizeof gv);   // (its padding too: the view travels byte for byte in j40hip_frame_lf_bundle's blob)
		gv.left = g.left; gv.top = g.top; gv.width = g.width; gv.height = g.height; gv.width8 = g.width8; gv.height8 = g.height8; gv.width64 = g.width64; gv.height64 = g.height64;
		gv.nb_varblocks = (int32_t) g.varblocks.size(); gv.blocks = g.blocks.data(); gv.lfindices = g.lfindices.data();
		for (int c = 0; c < 3; ++c) gv.llfcoeffs[c] = g.llfcoeffs[c].data();
		gv.coeffoff_qfidx = nullptr; gv.hfmul_inv = nullptr; gv.xfromy = g.xfromy.data(); gv.bfromy = g.bfromy.data();
		h->views.lf_groups.push_back(gv);
	}
	h->views.vb_coeffoff_qfidx.assign(f.lf_groups.size(), {}); h->views.vb_hfmul_inv.assign(f.lf_groups.size(), {});
	for (size_t g = 0; g < f.lf_groups.size(); ++g) {
		for (const VarblockInfo &vb : f.lf_groups[g].varblocks) { h->views.vb_coeffoff_qfidx[g].push_back(vb.coeffoff_qfidx); h->views.vb_hfmul_inv[g].push_back(vb.hfmul_inv); }
		h->views.lf_groups[g].coeffoff_qfidx = h->views.vb_coeffoff_qfidx[g].data(); h->views.lf_groups[g].hfmul_inv = h->views.vb_hfmul_inv[g].data();
	}
	v->lf_groups = h->views.lf_groups.data();
	const int32_t ng = (int32_t) f.fh.num_groups;
	for (int32_t p = 0; p < f.fh.num_passes; ++p) for (int32_t g = 0; g < ng; ++g) {
		const GroupInfo gi = group_info(f.fh, g);
		j40hip_section_view s;
		if (f.toc.single) { s.byte_off = (uint32_t) f.toc.single_section.offset; s.size = (uint32_t) f.toc.single_section.size; s.bit_off = (uint32_t) f.single_pass_group_bitpos; }
		else { const Section &sec = f.toc.pass_groups[(size_t) p * (size_t) ng + (size_t) g]; s.byte_off = (uint32_t) sec.offset; s.size = (uint32_t) sec.size; s.bit_off = 0; }
		s.ggidx = gi.ggidx; s.gx_in_gg = gi.gx_in_gg; s.gy_in_gg = gi.gy_in_gg; s.gw = gi.gw; s.gh = gi.gh;
		h->views.sections.push_back(s);
	}
	v->sections = h->views.sections.data();
	return 0;
}

// ---- the LF bundle as one relocatable blob (what rank 0 broadcasts when the other ranks are not to parse the
// stream themselves): the VarDCT plan view with every pointer replaced by its byte offset from the start of the blob ----
}   // extern "C"
namespace {
struct BundleWriter {
	std::vector<uint8_t> bytes;
	template <typename T> const T *put(const T *p, size_t n) {   // returns the OFFSET, typed as a pointer
		if (!p || !n) return nullptr;
		bytes.resize((bytes.size() + 15) & ~(size_t) 15);
		const size_t off = bytes.size();
		bytes.insert(bytes.end(), (const uint8_t *) p, (const uint8_t *) p + n * sizeof(T));
		return (const T *) (uintptr_t) off;
	}
	template <typename T> T *at(const T *off) { return (T *) (bytes.data() + (uintptr_t) off); }
};
enum : uint32_t { BUNDLE_MAGIC = 0x424c344au };   // "J4LB"
struct BundleHeader { uint32_t magic, version; uint64_t size; j40hip_vardct_view view; };
static size_t order_size(int o) { return (size_t) 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1]); }
// offset -> pointer, bounds checked
template <typename T> static void bundle_fix(const T *&p, size_t n, uint8_t *base, size_t size) {
	const size_t off = (size_t) (uintptr_t) p;
	if (!off) { p = nullptr; return; }
	if (off >= size || n > (size - off) / sizeof(T)) J40HIP_RAISE("rnge");
	p = (const T *) (base + off);
}
}
extern "C" {

size_t j40hip_frame_lf_bundle(j40hip_frame *h, void *out, size_t capacity, uint32_t *err) {
	uint32_t code = 0; size_t need = 0;
	try {
		j40hip_vardct_view v;
		if ((code = j40hip_frame_vardct_view(h, &v)) != 0) { if (err) *err = code; return 0; }
		BundleWriter w;
		w.bytes.resize(sizeof(BundleHeader));
		j40hip_vardct_view o = v;
		o.codestream = w.put(v.codestream, v.codestream_size);
		o.block_ctx_map = w.put(v.block_ctx_map, (size_t) v.block_ctx_size);
		{
			std::vector<j40hip_codespec_view> specs(v.coeff_specs, v.coeff_specs + v.num_passes);
			for (j40hip_codespec_view &sv : specs) {
				std::vector<j40hip_cluster_view> cl(sv.clusters, sv.clusters + sv.num_clusters);
				for (j40hip_cluster_view &c : cl) { c.D = w.put(c.D, c.D && !sv.use_prefix_code ? (size_t) 1 << sv.log_alpha_size : 0); c.lengths = w.put(c.lengths, c.lengths ? (size_t) c.alphabet_size : 0); }
				sv.cluster_map = w.put(sv.cluster_map, (size_t) sv.num_dist + (sv.lz77_enabled ? 1 : 0));
				sv.clusters = w.put(cl.data(), cl.size());
			}
			o.coeff_specs = w.put(specs.data(), specs.size());
		}
		for (int p = 0; p < 11; ++p) for (int q = 0; q < 13; ++q) for (int c = 0; c < 3; ++c) { const int i = (p * 13 + q) * 3 + c; o.orders[i] = w.put(v.orders[i], v.orders[i] ? order_size(q) : 0); }
		for (int i = 0; i < 17; ++i) o.dq_matrix[i] = w.put(v.dq_matrix[i], v.dq_matrix[i] ? (size_t) v.dq_size[i] * 3 : 0);
		{
			std::vector<j40hip_lf_group_view> groups(v.lf_groups, v.lf_groups + v.num_lf_groups);
			for (j40hip_lf_group_view &g : groups) {
				const size_t cells = (size_t) g.width8 * (size_t) g.height8, c64 = (size_t) g.width64 * (size_t) g.height64;
				g.blocks = w.put(g.blocks, cells); g.lfindices = w.put(g.lfindices, cells);
				for (int c = 0; c < 3; ++c) g.llfcoeffs[c] = w.put(g.llfcoeffs[c], cells);
				g.coeffoff_qfidx = w.put(g.coeffoff_qfidx, (size_t) g.nb_varblocks); g.hfmul_inv = w.put(g.hfmul_inv, (size_t) g.nb_varblocks);
				g.xfromy = w.put(g.xfromy, c64); g.bfromy = w.put(g.bfromy, c64);
			}
			o.lf_groups = w.put(groups.data(), groups.size());
		}
		o.sections = w.put(v.sections, (size_t) v.num_passes * (size_t) v.num_groups);
		BundleHeader hd; memset(&hd, 0, sizeof hd);
		hd.magic = BUNDLE_MAGIC; hd.version = 1; hd.size = w.bytes.size(); hd.view = o;
		memcpy(w.bytes.data(), &hd, sizeof hd);
		need = w.bytes.size();
		if (out && capacity >= need) memcpy(out, w.bytes.data(), need);
	} catch (const std::exception &) { code = E4("!mem"); need = 0; }
	if (err) *err = code;
	return need;
}

j40hip_frame *j40hip_frame_from_lf_bundle(const void *blob, size_t size, uint32_t *err) {
	uint32_t code = 0;
	try {
		BundleHeader hd;
		if (!blob || size < sizeof hd) J40HIP_RAISE("rnge");
		memcpy(&hd, blob, sizeof hd);
		if (hd.magic != BUNDLE_MAGIC || hd.version != 1 || hd.size != size) J40HIP_RAISE("rnge");
		std::vector<uint8_t> bytes((const uint8_t *) blob, (const uint8_t *) blob + size);
		uint8_t *base = bytes.data();
		auto fix = [&](auto &p, size_t n) { bundle_fix(p, n, base, size); };
		j40hip_vardct_view &v = ((BundleHeader *) base)->view;
		if (v.num_passes < 1 || v.num_passes > 11 || v.num_lf_groups < 1 || v.num_groups < 1 || v.block_ctx_size < 0) J40HIP_RAISE("rnge");
		fix(v.codestream, v.codestream_size); fix(v.block_ctx_map, (size_t) v.block_ctx_size);
		fix(v.coeff_specs, (size_t) v.num_passes);
		if (!v.coeff_specs) J40HIP_RAISE("rnge");
		for (int32_t p = 0; p < v.num_passes; ++p) {
			j40hip_codespec_view &sv = const_cast<j40hip_codespec_view &>(v.coeff_specs[p]);
			if (sv.num_clusters < 1 || sv.num_clusters > 256 || sv.num_dist < 1 || (!sv.use_prefix_code && (sv.log_alpha_size < 5 || sv.log_alpha_size > 8))) J40HIP_RAISE("rnge");
			fix(sv.cluster_map, (size_t) sv.num_dist + (sv.lz77_enabled ? 1 : 0)); fix(sv.clusters, (size_t) sv.num_clusters);
			if (!sv.cluster_map || !sv.clusters) J40HIP_RAISE("rnge");
			for (int32_t c = 0; c < sv.num_clusters; ++c) {
				j40hip_cluster_view &cv = const_cast<j40hip_cluster_view &>(sv.clusters[c]);
				fix(cv.D, sv.use_prefix_code ? 0 : (size_t) 1 << sv.log_alpha_size); fix(cv.lengths, (size_t) (cv.alphabet_size > 0 ? cv.alphabet_size : 0));
			}
		}
		for (int p = 0; p < 11; ++p) for (int q = 0; q < 13; ++q) for (int c = 0; c < 3; ++c) fix(v.orders[(p * 13 + q) * 3 + c], order_size(q));
		for (int i = 0; i < 17; ++i) fix(v.dq_matrix[i], (size_t) (v.dq_size[i] > 0 ? v.dq_size[i] : 0) * 3);
		fix(v.lf_groups, (size_t) v.num_lf_groups);
		if (!v.lf_groups) J40HIP_RAISE("rnge");
		for (int32_t g = 0; g < v.num_lf_groups; ++g) {
			j40hip_lf_group_view &gv = const_cast<j40hip_lf_group_view &>(v.lf_groups[g]);
			if (gv.width8 < 1 || gv.height8 < 1 || gv.width64 < 1 || gv.height64 < 1 || gv.nb_varblocks < 1) J40HIP_RAISE("rnge");
			const size_t cells = (size_t) gv.width8 * (size_t) gv.height8, c64 = (size_t) gv.width64 * (size_t) gv.height64;
			fix(gv.blocks, cells); fix(gv.lfindices, cells);
			for (int c = 0; c < 3; ++c) fix(gv.llfcoeffs[c], cells);
			fix(gv.coeffoff_qfidx, (size_t) gv.nb_varblocks); fix(gv.hfmul_inv, (size_t) gv.nb_varblocks);
			fix(gv.xfromy, c64); fix(gv.bfromy, c64);
			if (!gv.blocks || !gv.lfindices || !gv.llfcoeffs[0] || !gv.llfcoeffs[1] || !gv.llfcoeffs[2] || !gv.coeffoff_qfidx || !gv.hfmul_inv || !gv.xfromy || !gv.bfromy) J40HIP_RAISE("rnge");
		}
		fix(v.sections, (size_t) v.num_passes * (size_t) v.num_groups);
		if (!v.sections || !v.codestream || !v.block_ctx_map) J40HIP_RAISE("rnge");
		return j40hip_frame_from_vardct_view(&v, err);   // copies everything
	} catch (const DecodeError &e) { code = e.code; }
	catch (const std::exception &) { code = E4("!mem"); }
	if (err) *err = code;
	return nullptr;
}

// Modular frames: how many of the plan's sections the wave-cooperative kernel takes (modular_coop.hip), out of how many; -1: no plan
int32_t j40hip_frame_coop_sections(j40hip_frame *h, int32_t *total) {
	try {
		HostModPlan hp;
		if (build_modular_plan(h->frame, h->cs, h->cs_size, &hp)) return -1;
		if (total) *total = (int32_t) hp.sections.size();
		return hp.coop_sections;
	} catch (const std::exception &) { return -1; }
}

// ... and how many of those share wavefronts four at a time (modular_quad.hip); -1: no plan
int32_t j40hip_frame_quad_sections(j40hip_frame *h) {
	try {
		HostModPlan hp;
		if (build_modular_plan(h->frame, h->cs, h->cs_size, &hp)) return -1;
		return hp.quad_sections;
	} catch (const std::exception &) { return -1; }
}

// ... and how many sections the two-pass decoder takes (modular_split.hip: position-only MA trees); -1: no plan
int32_t j40hip_frame_split_sections(j40hip_frame *h) {
	try {
		HostModPlan hp;
		if (build_modular_plan(h->frame, h->cs, h->cs_size, &hp)) return -1;
		return hp.split_sections;
	} catch (const std::exception &) { return -1; }
}

uint32_t j40hip_frame_modular_view(j40hip_frame *h, j40hip_modular_view *v) {
	HostModPlan hp;
	if (uint32_t e = build_modular_plan(h->frame, h->cs, h->cs_size, &hp)) return e;
	finish_lf_tail(&h->frame);
	const Frame &f = h->frame;
	memset(v, 0, sizeof *v);
	h->views = j40hip_frame::Views();
	h->views.clusters.reserve(hp.host_specs.size() + 1);
	v->width = f.fh.width; v->height = f.fh.height; v->bpp = f.im.bpp; v->num_channels = hp.frame.num_channels; v->num_sections = hp.frame.num_sections;
	v->alpha_channel = hp.alpha_channel;
	v->check_section_end = hp.frame.check_section_end; v->single_declared_end = hp.frame.single_declared_end;
	v->codestream = h->cs; v->codestream_size = h->cs_size;
	h->views.specs.assign(hp.host_specs.size(), j40hip_codespec_view());
	h->views.host_specs = hp.host_specs;   // the views point into these
	for (size_t i = 0; i < hp.host_specs.size(); ++i) fill_codespec_view(h, h->views.host_specs[i], &h->views.specs[i]);
	v->codespec = h->views.specs.data(); v->num_codespecs = (int32_t) h->views.specs.size();
	for (const DevTreeNode &n : hp.tree) h->views.tree.push_back(j40hip_tree_node{n.prop, n.value, n.a, n.b});
	v->tree = h->views.tree.data(); v->num_tree_nodes = (int32_t) h->views.tree.size();
	h->views.ch_w = hp.plane_w; h->views.ch_h = hp.plane_h; h->views.ch_meta = hp.plane_meta;
	v->channel_w = h->views.ch_w.data(); v->channel_h = h->views.ch_h.data(); v->channel_meta = h->views.ch_meta.data();
	for (const Transform &t : hp.transforms) h->views.transforms.push_back(j40hip_transform_view{(int32_t) t.kind, t.begin_c, t.rct_type, t.num_c, t.nb_colours, t.nb_deltas, t.d_pred, t.horizontal ? 1 : 0, t.in_place ? 1 : 0});
	v->transforms = h->views.transforms.data(); v->num_transforms = (int32_t) h->views.transforms.size();
	for (const DevModSection &s : hp.sections) {
		j40hip_modular_section_view sv;
		sv.byte_off = s.byte_off; sv.size = s.size; sv.bit_off = s.bit_off; sv.gx = s.gx; sv.gy = s.gy; sv.gw = s.gw; sv.gh = s.gh; sv.sidx = s.sidx;
		sv.first_channel = s.first_channel; sv.num_channels = s.num_channels; memcpy(sv.wp, s.wp, 12);
		sv.tree_off = s.tree_off; sv.tree_nodes = s.tree_nodes; sv.spec_idx = s.spec_idx; sv.local_off = s.local_off; sv.local_count = s.local_count; sv.sub_off = s.sub_off; sv.sub_tr_off = sv.sub_tr_count = sv.sub_paste = 0; sv.preset_status = s.preset_status; sv.chan_off = s.chan_off; sv.dist_mult_p1 = s.dist_mult_p1;
		h->views.mod_sections.push_back(sv);
	}
	v->sections = h->views.mod_sections.data();
	h->views.local_rct = hp.local_rct; v->local_rct = h->views.local_rct.data();
	for (const DevChanRect &r : hp.chan_rects) for (int32_t x : {r.plane, r.x0, r.y0, r.w, r.h, r.shifts}) h->views.chan_rects.push_back(x);
	v->chan_rects = h->views.chan_rects.data();
	h->views.sub_w = hp.sub_w; h->views.sub_h = hp.sub_h; h->views.sub_meta = hp.sub_meta;
	v->sub_w = h->views.sub_w.data(); v->sub_h = h->views.sub_h.data(); v->sub_meta = h->views.sub_meta.data();
	for (const HostModPlan::SubImage &si : hp.sub_images) {
		j40hip_modular_section_view &sv = h->views.mod_sections[(size_t) si.section];
		sv.sub_tr_off = (int32_t) h->views.sub_transforms.size(); sv.sub_tr_count = (int32_t) si.transforms.size(); sv.sub_paste = si.paste;
		for (const Transform &t : si.transforms) h->views.sub_transforms.push_back(j40hip_transform_view{(int32_t) t.kind, t.begin_c, t.rct_type, t.num_c, t.nb_colours, t.nb_deltas, t.d_pred, t.horizontal ? 1 : 0, t.in_place ? 1 : 0});
	}
	v->sub_transforms = h->views.sub_transforms.data();
	{ const WPParams &wp = f.gmodular.wp; v->global_wp[0] = wp.p1; v->global_wp[1] = wp.p2; for (int i = 0; i < 5; ++i) v->global_wp[2 + i] = wp.p3[i]; for (int i = 0; i < 4; ++i) v->global_wp[7 + i] = wp.w[i]; v->global_wp[11] = 0; }
	return 0;
}

int32_t j40hip_kat_natural_order(int32_t log_rows, int32_t log_columns, int32_t *out) {
	std::vector<int32_t> o;
	natural_order(log_rows, log_columns, &o);
	memcpy(out, o.data(), o.size() * 4);
	return (int32_t) o.size();
}

int32_t j40hip_kat_library_dq_matrix(int idx, float *out) {
	DqMatrix dq;
	try { load_dq_matrix(idx, &dq); } catch (const DecodeError &) { return 0; }
	for (size_t i = 0; i < dq.params.size(); ++i) for (int c = 0; c < 3; ++c) out[i * 3 + (size_t) c] = dq.params[i][(size_t) c];
	return (int32_t) dq.params.size();
}

void j40hip_kat_forward_llf(float *buf, int32_t log_rows, int32_t log_columns) {
	float scratch[1024];
	forward_dct2d_scaled_for_llf(buf, scratch, log_rows, log_columns);
}

float j40hip_kat_half_secant(int i) { return half_secants()[i]; }
float j40hip_kat_lf2llf_scale(int i) { return lf2llf_scales()[i]; }

} // extern "C"

extern "C" __attribute__((visibility("default"))) int j40hip_cpu_quota() {
	unsigned hw = std::thread::hardware_concurrency();
	int n = hw ? (int) hw : 4;
	if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2
		char q[64] = {0}; long long period = 0;
		// (rounded UP and at least 1: a quota of half a CPU is one thread's worth, not "no quota" -- which used to read as every visible CPU)
		if (fscanf(fp, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0 && atoll(q) > 0) { const long long c = std::max<long long>(1, (atoll(q) + period - 1) / period); if (c < n) n = (int) c; }
		fclose(fp);
		return n;
	}
	long long quota = -1, period = 0;   // cgroup v1
	if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fp, "%lld", &quota) != 1) quota = -1; fclose(fp); }
	if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
	if (quota > 0 && period > 0) { const long long c = std::max<long long>(1, (quota + period - 1) / period); if (c < n) n = (int) c; }
	return n;
}
