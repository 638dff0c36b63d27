// j40_amd/csrc/plan_front.cpp -- see plan_front.hpp
#include "plan_front.hpp"
#include <algorithm>
#include <cmath>

namespace j40hip {

template <typename T> static void key_put(std::vector<uint8_t> *k, const T &v) { const uint8_t *p = (const uint8_t *) &v; k->insert(k->end(), p, p + sizeof(T)); }

void static_tables_key(const Frame &fr, std::vector<uint8_t> *key) {
	key->clear();
	key_put(key, (int32_t) fr.fh.num_passes);
	for (int i = 0; i < 17; ++i) {
		const DqMatrix &dq = fr.dq_matrix[i];
		key_put(key, dq.mode); key_put(key, dq.n); key_put(key, dq.m); key_put(key, (uint32_t) dq.params.size());
		for (const auto &w : dq.params) for (int c = 0; c < 3; ++c) key_put(key, w[(size_t) c]);
	}
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) for (int o = 0; o < 13; ++o) for (int c = 0; c < 3; ++c) {
		const uint8_t has = fr.order_has_lehmer[p][o][c] ? 1 : 0;
		key_put(key, has);
		if (!has) continue;
		const std::vector<int32_t> &l = fr.order_lehmer[p][o][c];
		key_put(key, (uint32_t) l.size());
		for (int32_t x : l) key_put(key, x);
	}
}

void build_static_tables(const Frame &fr, StaticTables *st) {
	static_tables_key(fr, &st->key);
	st->pool_f32.clear(); st->pool_u16.clear();
	for (size_t i = 0; i < 11 * 13 * 3; ++i) st->order_off[i] = 0xffffffffu;
	std::vector<int32_t> order[13][3];   // pass 0's, for the scan-order weights
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) for (int o = 0; o < 13; ++o) {
		const int32_t skip = 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1] - 6);
		for (int c = 0; c < 3; ++c) {
			std::vector<int32_t> ord;
			natural_order(LOG_ORDER_SIZE[o][0], LOG_ORDER_SIZE[o][1], &ord);
			if (fr.order_has_lehmer[p][o][c]) apply_permutation(ord.data() + skip, fr.order_lehmer[p][o][c]);   // j40.h:7711-7732
			st->order_off[(p * 13 + o) * 3 + c] = (uint32_t) st->pool_u16.size();
			for (int32_t v : ord) st->pool_u16.push_back((uint16_t) v);
			if (p == 0) order[o][c].swap(ord);
		}
	}
	static const int8_t ORDER_OF_PARAM[17] = {0, 1, 1, 1, 2, 3, 4, 5, 6, 1, 1, 7, 8, 9, 10, 11, 12};
	for (int i = 0; i < 17; ++i) {
		st->dq_off[i] = st->dq_scan_off[i] = 0xffffffffu; st->dq_size[i] = 0; st->dq_error[i] = 0;
		DqMatrix dq = fr.dq_matrix[i];
		try { load_dq_matrix(i, &dq); } catch (const DecodeError &e) { st->dq_error[i] = e.code; continue; }
		const size_t n = dq.params.size();
		st->dq_off[i] = (uint32_t) st->pool_f32.size(); st->dq_size[i] = (uint32_t) n;
		st->pool_f32.resize(st->pool_f32.size() + 3 * n);
		float *planar = st->pool_f32.data() + st->dq_off[i];
		for (size_t k = 0; k < n; ++k) for (int c = 0; c < 3; ++c) planar[(size_t) c * n + k] = dq.params[k][(size_t) c];
		if (fr.fh.num_passes != 1) continue;
		bool have = true;
		for (int c = 0; c < 3; ++c) have = have && order[ORDER_OF_PARAM[i]][c].size() == n;
		if (!have) continue;
		st->dq_scan_off[i] = (uint32_t) st->pool_f32.size();
		st->pool_f32.resize(st->pool_f32.size() + 3 * n);
		planar = st->pool_f32.data() + st->dq_off[i];
		float *scan = st->pool_f32.data() + st->dq_scan_off[i];
		for (int c = 0; c < 3; ++c) { const std::vector<int32_t> &ord = order[ORDER_OF_PARAM[i]][c]; for (size_t k = 0; k < n; ++k) scan[(size_t) c * n + k] = planar[(size_t) c * n + (size_t) ord[k]]; }
	}
}

// the global MA tree and code spec as k_lf_lanes wants them; false when it cannot take them
static bool build_lf_lanes(const Frame &fr, FrontPlan *fp) {
	const CodeSpec &spec = fr.global_codespec;
	if (fr.global_tree.empty() || fr.global_tree.size() > 1024 || spec.use_prefix_code || spec.lz77_enabled || spec.num_clusters < 1 || spec.num_clusters > 256) return false;
	if (spec.log_alpha_size < 5 || spec.log_alpha_size > 8) return false;
	for (const Cluster &c : spec.clusters) if (c.alias.size() != ((size_t) 1 << spec.log_alpha_size)) return false;
	fp->lf_uses = 0;
	fp->lf_tree.clear();
	for (const TreeNode &n : fr.global_tree) {
		if (n.prop >= 0) {
			if (n.prop > 14) return false;   // the weighted predictor's error, previous channels
			if (n.prop == 12) fp->lf_uses |= 1u;
			if (n.prop == 13) fp->lf_uses |= 4u;
		} else {
			const int32_t pred = -1 - n.prop;
			if (pred == 6 || pred > 13) return false;
			if (n.value < 0 || n.value >= spec.num_dist || (size_t) n.value >= spec.cluster_map.size()) return false;
			if (pred == 7 || pred == 12) fp->lf_uses |= 1u;
			if (pred == 13) fp->lf_uses |= 1u | 2u | 4u;
		}
		fp->lf_tree.push_back(DevTreeNode{n.prop, n.value, n.a, n.b});
	}
	fp->lf_ctx_map.assign(spec.cluster_map.begin(), spec.cluster_map.begin() + spec.num_dist);
	fp->lf_cfg.clear(); fp->lf_alias.clear();
	for (const Cluster &c : spec.clusters) {
		// The largest token the cluster's tables can produce: bucket i yields i below its cutoff and its alias symbol from there on
		// (AnsEntry, entropy.hpp). max_token is clamped to it -- `token > max_token` (j40.h:2316) is unchanged for every token that can
		// come out -- so that the lane decoder can tell from the configuration word how many extra bits a symbol may ask for at most
		// (lf_rows_dev.h: lf_rows_leaf_word).
		const uint32_t bucket = 4096u >> spec.log_alpha_size;
		int32_t top = 0;
		for (size_t i = 0; i < c.alias.size(); ++i) {
			const uint32_t lo = (uint32_t) c.alias[i], cutoff = lo & 0xff;
			if (cutoff > 0) top = std::max(top, (int32_t) i);
			if (cutoff < bucket) top = std::max(top, (int32_t) ((lo >> 20) & 0xff));
		}
		fp->lf_cfg.push_back(c.cfg.packed() | ((uint32_t) std::min(c.cfg.max_token, top) << 12));
		fp->lf_alias.insert(fp->lf_alias.end(), c.alias.begin(), c.alias.end());
	}
	fp->lf_log_alpha = spec.log_alpha_size;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	// (the alias tables stay in global memory unless J40HIP_LF_ALIAS_LDS asks for them in LDS: lf_decode.hip)
	fp->lf_lds_bytes = align16(16u * (uint32_t) fp->lf_tree.size()) + align16((uint32_t) spec.num_dist) + align16(4u * (uint32_t) spec.num_clusters);
	static const bool alias_lds = [] { const char *e = getenv("J40HIP_LF_ALIAS_LDS"); return e && atoi(e) != 0; }();
	return fp->lf_lds_bytes + (alias_lds ? 8u * ((uint32_t) spec.num_clusters << spec.log_alpha_size) : 0u) <= 56u * 1024u;
}

uint32_t build_front_plan(const Frame &fr, const StaticTables &st, size_t cs_size, const std::vector<int32_t> &extra_prec, bool want_lf_device, FrontPlan *fp) {
	// what build_vardct_plan refuses, plus what the device-side plan build leaves to the host path: a single section (the LfGroup
	// is then not a section of its own), Modular sub-images behind the coefficients (extra channels), groups other than 256 x 256
	if (cs_size + 16 >= ((size_t) 1 << 29)) return ERR_TODO;
	if (fr.fh.is_modular || fr.toc.single || fr.im.grey || fr.fh.do_ycbcr || fr.im.bpp < 8 || fr.im.exp_bits) return ERR_TODO;
	if ((int32_t) fr.gmodular.channel.size() > fr.num_gm_channels || fr.fh.group_size_shift != 8) return ERR_TODO;
	if (fr.lf_groups.size() >= ((size_t) 1 << 24) || extra_prec.size() != fr.lf_groups.size()) return ERR_TODO;
	fp->reset();
	DevFrame &df = fp->frame;
	fill_frame_constants(fr, &df);
	memcpy(df.order_off, st.order_off, sizeof df.order_off); memcpy(df.dq_off, st.dq_off, sizeof df.dq_off);
	memcpy(df.dq_size, st.dq_size, sizeof df.dq_size); memcpy(df.dq_scan_off, st.dq_scan_off, sizeof df.dq_scan_off);
	fp->coeff_specs.assign((size_t) fr.fh.num_passes, DevCodeSpec());
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) flatten_code_spec(fr.coeff_codespec[p], fp->pool_u8, fp->pool_i32, fp->pool_u64, fp->clusters, &fp->coeff_specs[(size_t) p]);
	fp->block_ctx_map_off = (uint32_t) fp->pool_u8.size();
	fp->pool_u8.insert(fp->pool_u8.end(), fr.block_ctx_map.begin(), fr.block_ctx_map.end());
	fp->pool_u8.resize(fp->pool_u8.size() + 16, 0);
	fp->lf_groups.assign(fr.lf_groups.size(), DevLfGroup());
	for (size_t g = 0; g < fr.lf_groups.size(); ++g) {
		const LfGroup &gg = fr.lf_groups[g];
		DevLfGroup &d = fp->lf_groups[g];
		memset(&d, 0, sizeof d);
		d.left = gg.left; d.top = gg.top; d.width = gg.width; d.height = gg.height;
		d.width8 = gg.width8; d.height8 = gg.height8; d.width64 = gg.width64; d.height64 = gg.height64;
		d.cell_base = d.vb_base = (int32_t) fp->cells; d.c64_base = (int32_t) fp->c64s;
		for (int c = 0; c < 3; ++c) d.mult_lf[c] = fr.m_lf_scaled[c] / (float) (fr.global_scale * fr.quant_lf) * (float) (65536 >> extra_prec[g]);   // j40.h:6562
		fp->cells += (size_t) gg.width8 * (size_t) gg.height8; fp->c64s += (size_t) gg.width64 * (size_t) gg.height64;
		fp->max_lf_cells = std::max(fp->max_lf_cells, gg.width8 * gg.height8);
		if (gg.width8 > 256 || gg.height8 > 256) return ERR_TODO;
		fp->lf_section_off.push_back((uint32_t) fr.toc.lf_groups[g].offset);
	}
	if (fp->cells * 64 * 3 * sizeof(float) >= 0xffffffffull && fr.fh.num_passes != 1) return ERR_TODO;
	fp->lf_smooth = !fr.fh.skip_adapt_lf_smooth;
	for (int c = 0; c < 3; ++c) fp->inv_m_lf[c] = (float) (fr.global_scale * fr.quant_lf) / fr.m_lf_scaled[c] / 65536.0f;   // j40.h:6497
	fill_sections(fr, &fp->sections);
	{   // K1's lanes take the groups by decreasing section size: a wavefront runs as long as its longest section, and 64 sections
		// picked in raster order always hold one near the frame's longest (1.9 x the mean); 64 neighbours in size end together
		const int32_t ng = (int32_t) fr.fh.num_groups;
		std::vector<uint64_t> key((size_t) ng);
		for (int32_t g = 0; g < ng; ++g) {
			uint64_t bytes = 0;
			for (int32_t p = 0; p < fr.fh.num_passes; ++p) bytes += fp->sections[(size_t) p * (size_t) ng + (size_t) g].size;
			key[(size_t) g] = (std::min<uint64_t>(bytes, 0xffffffffu) << 32) | (uint32_t) (0xffffffffu - (uint32_t) g);   // (ties: the lower group first)
		}
		std::sort(key.begin(), key.end(), [](uint64_t a, uint64_t b) { return a > b; });
		fp->lane_order.resize((size_t) ng);
		for (int32_t k = 0; k < ng; ++k) fp->lane_order[(size_t) k] = 0xffffffffu - (uint32_t) key[(size_t) k];
	}
	const int32_t num_groups = (int32_t) fr.fh.num_groups;
	df.sparse_coeffs = fr.fh.num_passes == 1;
	if (!fill_event_ranges(fp->sections, num_groups, df.sparse_coeffs != 0, &fp->ev_range, &fp->ev_capacity)) df.sparse_coeffs = 0;
	bool any_lz77 = false;
	for (const DevCodeSpec &sp : fp->coeff_specs) any_lz77 |= sp.lz77_enabled != 0;
	fp->lz_window_size = any_lz77 ? 3 * 65536 + 3 * 1024 + 16 : 0;
	fill_hf_launch_info(fp->coeff_specs, (uint32_t) fr.block_ctx_map.size(), fp->cells * 64, &fp->hf);
	DevPlanBuild &pb = fp->build;
	memset(&pb, 0, sizeof pb);
	memcpy(pb.lf_thr, fr.lf_thr, sizeof pb.lf_thr); memcpy(pb.qf_thr, fr.qf_thr, sizeof pb.qf_thr);
	for (int c = 0; c < 3; ++c) pb.nb_lf_thr[c] = fr.nb_lf_thr[c];
	pb.nb_qf_thr = fr.nb_qf_thr;
	pb.num_lf_groups = (int32_t) fr.fh.num_lf_groups; pb.num_groups = num_groups; pb.gcolumns = fr.fh.gcolumns; pb.ggcolumns = fr.fh.ggcolumns;
	pb.lfidx_size = df.lfidx_size;
	pb.mult_base = df.mult_base; pb.base_corr_x = fr.base_corr_x; pb.base_corr_b = fr.base_corr_b; pb.inv_colour_factor = fr.inv_colour_factor;
	pb.block_ctx_map_off = fp->block_ctx_map_off;
	const bool lanes_ok = build_lf_lanes(fr, fp);   // (always: the tables' room is part of the frame's block whoever decodes the streams)
	fp->lf_device = want_lf_device && lanes_ok;
	return 0;
}

} // namespace j40hip
