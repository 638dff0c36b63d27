// j40_amd/csrc/capi_host.cpp -- host half of the thin C-ABI (include/j40hip.h): parse + stage accessors
#include "capi.hpp"
#include "tables.hpp"

using namespace j40hip;

extern "C" {

j40hip_frame *j40hip_frame_parse(const void *buf, size_t size, int threads, uint32_t *err) {
	j40hip_frame *h = new j40hip_frame();
	uint32_t code = 0;
	try {
		extract_codestream((const uint8_t *) buf, size, &h->cs, &h->cs_size, &h->cs_storage);
		parse_frame(h->cs, h->cs_size, &h->frame, threads);
	} catch (const DecodeError &e) { code = e.code; }
	catch (const std::bad_alloc &) { code = E4("!mem"); }
	if (err) *err = code;
	if (code) { delete h; return nullptr; }
	return h;
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
