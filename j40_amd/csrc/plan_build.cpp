// j40_amd/csrc/plan_build.cpp -- see plan_build.hpp
#include "plan_build.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

namespace j40hip {

namespace {
// a team of threads for build_vardct_plan's big loops: body(tid, team size, barrier) runs on the calling thread and team - 1 more;
// the body allocates nothing and throws nothing (a thread that left before a barrier would strand the others)
struct TeamBarrier {
	std::atomic<int> waiting{0}, generation{0}, team{1};
	void wait() {
		const int n = team.load(std::memory_order_acquire);
		if (n <= 1) return;
		const int g = generation.load(std::memory_order_acquire);
		if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { waiting.store(0, std::memory_order_relaxed); generation.fetch_add(1, std::memory_order_release); }
		else while (generation.load(std::memory_order_acquire) == g) std::this_thread::yield();
	}
};
template <class F> void run_team(int threads, F &body) {
	TeamBarrier bar;
	if (threads <= 1) { body(0, 1, bar); return; }
	std::vector<std::thread> pool;
	std::atomic<int> go{0};   // the team's size, once it is known how many threads could be started
	try {
		for (int t = 1; t < threads; ++t) pool.emplace_back([&, t] { int n; while ((n = go.load(std::memory_order_acquire)) == 0) std::this_thread::yield(); if (t < n) body(t, n, bar); });
	} catch (const std::exception &) { }   // (as many as there are)
	const int n = (int) pool.size() + 1;
	bar.team.store(n, std::memory_order_release);
	go.store(n, std::memory_order_release);
	body(0, n, bar);
	for (std::thread &t : pool) t.join();
}
} // namespace

template <typename T> static uint32_t push(std::vector<T> &pool, const T *p, size_t n) { uint32_t off = (uint32_t) pool.size(); pool.insert(pool.end(), p, p + n); return off; }

void flatten_code_spec(const CodeSpec &spec, std::vector<uint8_t> &u8, std::vector<int32_t> &i32, std::vector<uint64_t> &u64, std::vector<DevCluster> &clusters, DevCodeSpec *out) {
	out->num_dist = spec.num_dist; out->num_clusters = spec.num_clusters;
	out->lz77_enabled = spec.lz77_enabled; out->use_prefix_code = spec.use_prefix_code;
	out->min_symbol = spec.min_symbol; out->min_length = spec.min_length;
	out->log_alpha_size = spec.log_alpha_size;
	out->lz_len_cfg = spec.lz_len_cfg.packed(); out->lz_len_max_token = spec.lz_len_cfg.max_token;
	while (u8.size() & 3) u8.push_back(0);   // kernels stage the map with 32-bit copies
	out->cluster_map_off = push(u8, spec.cluster_map.data(), spec.cluster_map.size());
	while (u8.size() & 3) u8.push_back(0);
	out->cluster_off = (uint32_t) clusters.size();
	const size_t span0 = spec.use_prefix_code ? i32.size() : u64.size();
	for (const Cluster &c : spec.clusters) {
		DevCluster d;
		d.cfg = c.cfg.packed(); d.max_token = c.cfg.max_token;
		d.fast_len = (int16_t) c.fast_len; d.max_len = (int16_t) c.max_len;
		d.table_off = spec.use_prefix_code ? push(i32, c.table.data(), c.table.size()) : push(u64, c.alias.data(), c.alias.size());
		clusters.push_back(d);
	}
	out->table_span = (uint32_t) ((spec.use_prefix_code ? i32.size() : u64.size()) - span0);
	out->lane_cfg_off = 0xffffffffu;
	bool uniform = !spec.use_prefix_code && !spec.lz77_enabled && spec.num_clusters <= 256;
	for (const Cluster &c : spec.clusters) uniform = uniform && c.alias.size() == ((size_t) 1 << spec.log_alpha_size);
	if (uniform) {   // hf_lanes_dev.h; tokens are < 256, so clamping max_token keeps `token > max_token` intact
		std::vector<int32_t> cfg;
		for (const Cluster &c : spec.clusters) cfg.push_back((int32_t) (c.cfg.packed() | ((uint32_t) std::min(c.cfg.max_token, 0xfffff) << 12)));
		out->lane_cfg_off = push(i32, cfg.data(), cfg.size());
	}
}

// one DevSection per (pass, group) TOC section, pass-major
void fill_sections(const Frame &fr, std::vector<DevSection> *sections) {
	const int32_t num_groups = (int32_t) fr.fh.num_groups;
	sections->assign((size_t) fr.fh.num_passes * (size_t) num_groups, DevSection());
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) for (int32_t g = 0; g < num_groups; ++g) {
		DevSection &d = (*sections)[(size_t) p * (size_t) num_groups + (size_t) g];
		const GroupInfo gi = group_info(fr.fh, g);
		if (fr.toc.single) {
			d.byte_off = (uint32_t) fr.toc.single_section.offset; d.size = (uint32_t) fr.toc.single_section.size; d.bit_off = (uint32_t) fr.single_pass_group_bitpos;
		} else {
			const Section &s0 = fr.toc.pass_groups[(size_t) p * (size_t) num_groups + (size_t) g];
			d.byte_off = (uint32_t) s0.offset; d.size = (uint32_t) s0.size; d.bit_off = 0;
		}
		d.ggidx = gi.ggidx; d.gx8 = gi.gx_in_gg / 8; d.gy8 = gi.gy_in_gg / 8;
		d.gw8 = ceil_div(gi.gw, 8); d.gh8 = ceil_div(gi.gh, 8);
		d.gx = fr.lf_groups[(size_t) gi.ggidx].left + gi.gx_in_gg; d.gy = fr.lf_groups[(size_t) gi.ggidx].top + gi.gy_in_gg; d.gw = gi.gw; d.gh = gi.gh;
	}
}

// sparse coefficients: every group's region of DevPlan::events, sized from its section (see build_vardct_plan). false: the frame
// needs more events than 32-bit indices reach (then dense planes)
bool fill_event_ranges(const std::vector<DevSection> &sections, int32_t num_groups, bool sparse, std::vector<uint32_t> *ev_range, size_t *ev_capacity) {
	ev_range->clear(); *ev_capacity = 0;
	if (!sparse) return true;
	for (int32_t g = 0; g < num_groups; ++g) {
		const DevSection &d = sections[(size_t) g];
		static const size_t per_byte = getenv("J40HIP_EVENTS_PER_BYTE") ? (size_t) atoi(getenv("J40HIP_EVENTS_PER_BYTE")) : 4;   // tests shrink it to reach the fallback
		const size_t worst = (size_t) d.gw8 * (size_t) d.gh8 * 64 * 3, cap = std::min(worst, (size_t) d.size * per_byte + 256);
		*ev_capacity = (*ev_capacity + 31) & ~(size_t) 31;   // regions start on a 128-byte line: the entropy kernel writes them in aligned pieces (hf_lanes_dev.h)
		ev_range->push_back((uint32_t) *ev_capacity);
		*ev_capacity += cap;
		ev_range->push_back((uint32_t) *ev_capacity);
	}
	if (*ev_capacity >= 0xffffffffull) { ev_range->clear(); *ev_capacity = 0; return false; }
	return true;
}

// K1's LDS budget
void fill_hf_launch_info(const std::vector<DevCodeSpec> &coeff_specs, uint32_t block_ctx_size, size_t coeff_floats, HfLaunchInfo *out) {
	HfLaunchInfo &hf = *out;
	hf.block_ctx_size = block_ctx_size; hf.max_num_dist = hf.max_clusters = hf.max_table_bytes = 0;
	for (const DevCodeSpec &sp : coeff_specs) {
		hf.max_num_dist = std::max<uint32_t>(hf.max_num_dist, (uint32_t) sp.num_dist);
		hf.max_clusters = std::max<uint32_t>(hf.max_clusters, (uint32_t) sp.num_clusters);
		hf.max_table_bytes = std::max<uint32_t>(hf.max_table_bytes, sp.table_span * (sp.use_prefix_code ? 4u : 8u));
	}
	{
		const uint32_t per_wave = 32 * 32 * 3 + 1024 * (uint32_t) sizeof(DevGroupBlock) + 16;
		const uint32_t fixed = hf.block_ctx_size + 256 + 64 + HF_WAVES * per_wave;
		hf.tables_fit_lds = fixed + hf.max_num_dist + hf.max_clusters * (uint32_t) sizeof(DevCluster) + hf.max_table_bytes + 64 <= 150u * 1024u;
	}
	{
		auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
		hf.lanes_fast = true; hf.lanes_lds_bytes = 0;
		for (const DevCodeSpec &sp : coeff_specs) {
			hf.lanes_fast = hf.lanes_fast && sp.lane_cfg_off != 0xffffffffu;
			const uint32_t n = 128 + 64 + 112 + align16((uint32_t) sp.num_dist) + align16(4u * (uint32_t) sp.num_clusters) + 8u * ((uint32_t) sp.num_clusters << sp.log_alpha_size);
			hf.lanes_lds_bytes = std::max(hf.lanes_lds_bytes, n);
		}
		hf.lanes_fast = hf.lanes_fast && hf.lanes_lds_bytes + 4u * HF_LANE_COLS_BYTES <= 156u * 1024u && coeff_floats * 3 * sizeof(float) < 0xffffffffull;
	}
}

// the frame-wide scalars of DevFrame (everything but the table offsets)
void fill_frame_constants(const Frame &fr, DevFrame *out) {
	DevFrame &df = *out;
	memset(&df, 0, sizeof df);
	df.width = fr.fh.width; df.height = fr.fh.height;
	df.num_passes = fr.fh.num_passes; df.num_groups = (int32_t) fr.fh.num_groups; df.num_lf_groups = (int32_t) fr.fh.num_lf_groups;
	df.nb_block_ctx = fr.nb_block_ctx; df.nb_qf_thr = fr.nb_qf_thr;
	df.lfidx_size = (fr.nb_lf_thr[0] + 1) * (fr.nb_lf_thr[1] + 1) * (fr.nb_lf_thr[2] + 1);
	df.num_hf_presets = fr.num_hf_presets; df.preset_bits = ceil_lg32((uint32_t) fr.num_hf_presets);
	df.bpp = fr.im.bpp;
	df.sections_have_trailer = (int32_t) fr.gmodular.channel.size() > fr.num_gm_channels;
	df.check_section_end = fr.toc.single && !df.sections_have_trailer;
	df.single_declared_end = (uint32_t) fr.toc.single_declared_end;
	for (int c = 0; c < 3; ++c) { df.quant_bias[c] = fr.im.quant_bias[c]; df.opsin_bias[c] = fr.im.opsin_bias[c]; df.cbrt_opsin_bias[c] = cbrtf(fr.im.opsin_bias[c]); }
	df.quant_bias_num = fr.im.quant_bias_num;
	static const float QM_SCALE[8] = {1.5625f, 1.25f, 1.0f, 0.8f, 0.64f, 0.512f, 0.4096f, 0.32768f};  // 0.8^(i-2), j40.h:7055
	df.mult_base = 65536.0f / (float) fr.global_scale;
	df.x_qm_mul = QM_SCALE[fr.fh.x_qm_scale]; df.b_qm_mul = QM_SCALE[fr.fh.b_qm_scale];
	df.kx_lf = fr.base_corr_x + (float) fr.x_factor_lf * fr.inv_colour_factor;   // j40.h:7115
	df.kb_lf = fr.base_corr_b + (float) fr.b_factor_lf * fr.inv_colour_factor;
	df.base_corr_x = fr.base_corr_x; df.base_corr_b = fr.base_corr_b; df.inv_colour_factor = fr.inv_colour_factor;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) df.opsin_inv_mat[i * 3 + j] = fr.im.opsin_inv_mat[i][j];
	df.itscale = 255.0f / fr.im.intensity_target;

}

uint32_t build_vardct_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, HostPlan *hp, int threads) {
	static const bool timing = getenv("J40HIP_PLAN_TIMING") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tb0 = timing ? now() : 0; double tbA = 0, tbB = 0, tbC = 0;
	if (cs_size + 16 >= ((size_t) 1 << 29)) return ERR_TODO;   // the kernels address the codestream with 32-bit BIT positions
	if (fr.fh.is_modular) return ERR_TODO;
	// same limits as j40.h:7867, 7917-7921. (A VarDCT frame of an image without xyb_encoded passes them: the reference
	// runs the XYB inverse with the default opsin matrix on it all the same, j40.h:7206-7233, and so does K2.)
	if (fr.im.grey || fr.fh.do_ycbcr) return ERR_TODO;
	if (fr.im.bpp < 8 || fr.im.exp_bits) return ERR_TODO;

	DevFrame &df = hp->frame;
	fill_frame_constants(fr, &df);
	hp->coeff_specs.assign((size_t) fr.fh.num_passes, DevCodeSpec());
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) flatten_code_spec(fr.coeff_codespec[p], hp->pool_u8, hp->pool_i32, hp->pool_u64, hp->clusters, &hp->coeff_specs[(size_t) p]);
	hp->block_ctx_map_off = push(hp->pool_u8, fr.block_ctx_map.data(), fr.block_ctx_map.size());
	hp->pool_u8.resize(hp->pool_u8.size() + 16, 0);
	for (size_t i = 0; i < 11 * 13 * 3; ++i) df.order_off[i] = 0xffffffffu;
	for (int32_t p = 0; p < fr.fh.num_passes; ++p) for (int o = 0; o < 13; ++o) for (int c = 0; c < 3; ++c) {
		const std::vector<int32_t> &ord = fr.orders[p][o][c];
		if (ord.empty()) continue;
		std::vector<uint16_t> tmp(ord.size());
		for (size_t i = 0; i < ord.size(); ++i) tmp[i] = (uint16_t) ord[i];
		df.order_off[(p * 13 + o) * 3 + c] = push(hp->pool_u16, tmp.data(), tmp.size());
	}
	for (int i = 0; i < 17; ++i) {
		df.dq_off[i] = 0xffffffffu; df.dq_size[i] = 0;
		const DqMatrix &dq = fr.dq_matrix[i];
		if (!dq.loaded) continue;
		const size_t n = dq.params.size();
		std::vector<float> planar(n * 3);
		for (size_t k = 0; k < n; ++k) for (int c = 0; c < 3; ++c) planar[(size_t) c * n + k] = dq.params[k][(size_t) c];
		df.dq_off[i] = push(hp->pool_f32, planar.data(), planar.size()); df.dq_size[i] = (uint32_t) n;
		// weights in scan order (DevFrame::dq_scan_off); every parameter set has one coefficient order
		static const int8_t ORDER_OF_PARAM[17] = {0, 1, 1, 1, 2, 3, 4, 5, 6, 1, 1, 7, 8, 9, 10, 11, 12};
		df.dq_scan_off[i] = 0xffffffffu;
		if (fr.fh.num_passes == 1) {
			std::vector<float> scan(n * 3);
			bool have = true;
			for (int c = 0; c < 3 && have; ++c) {
				const std::vector<int32_t> &ord = fr.orders[0][ORDER_OF_PARAM[i]][(size_t) c];
				have = ord.size() == n;
				for (size_t k = 0; k < n && have; ++k) scan[(size_t) c * n + k] = planar[(size_t) c * n + (size_t) ord[k]];
			}
			if (have) df.dq_scan_off[i] = push(hp->pool_f32, scan.data(), scan.size());
		}
	}

	// ---- the frame-wide arrays: LF bundle, K1's block lists, K2's work list. Every piece has its place before anything is written
	// (offsets from the LfGroups' and groups' sizes), so the pieces are written by a team of threads when the caller has them
	// (j40_next_frame's single-image path: 8 ms of one core per 8K frame otherwise, a quarter of the call) -- the same bytes either way.
	const size_t nlf = fr.lf_groups.size();
	const int32_t num_groups = (int32_t) fr.fh.num_groups;
	size_t nvb = 0, ncell = 0, nc64 = 0;
	hp->lf_groups.assign(nlf, DevLfGroup());
	bool any_tail = false, all_tail = true;
	for (size_t g = 0; g < nlf; ++g) {
		const LfGroup &gg = fr.lf_groups[g];
		DevLfGroup &d = hp->lf_groups[g];
		d.left = gg.left; d.top = gg.top; d.width = gg.width; d.height = gg.height;
		d.width8 = gg.width8; d.height8 = gg.height8; d.width64 = gg.width64; d.height64 = gg.height64;
		d.cell_base = (int32_t) ncell; d.vb_base = (int32_t) nvb; d.c64_base = (int32_t) nc64;
		d.nb_varblocks = (int32_t) gg.varblocks.size();
		for (int c = 0; c < 3; ++c) d.mult_lf[c] = gg.mult_lf[c];
		any_tail = any_tail || gg.tail_pending; all_tail = all_tail && gg.tail_pending;
		if (gg.lfindices.size() != gg.blocks.size() || gg.bfromy.size() != gg.xfromy.size()) return ERR_RNGE;   // (cannot happen: lf_group_finish sizes them alike)
		for (int c = 0; c < 3; ++c) if ((gg.tail_pending ? gg.lfraw[c].size() : gg.llfcoeffs[c].size()) != gg.blocks.size()) return ERR_RNGE;
		ncell += gg.blocks.size(); nvb += gg.varblocks.size(); nc64 += gg.xfromy.size();
	}
	hp->coeff_floats = ncell * 64;
	hp->lf_tail_pending = any_tail;
	if (any_tail) {   // the device computes the LLF coefficients from the decoded integers (lf_tail_kernels.hip)
		if (nlf >= ((size_t) 1 << 24)) return ERR_TODO;
		if (!all_tail) return ERR_TODO;   // (all or none)
		hp->lf_smooth = !fr.fh.skip_adapt_lf_smooth;
		for (int c = 0; c < 3; ++c) hp->inv_m_lf[c] = (float) (fr.global_scale * fr.quant_lf) / fr.m_lf_scaled[c] / 65536.0f;   // j40.h:6497
	}
	// (every element of these is written below; a reused plan object keeps last frame's storage, and its size when the frame's is the same)
	auto sized = [](auto &v, size_t n) { if (v.size() != n) v.resize(n); };
	sized(hp->blocks, ncell); sized(hp->lfindices, ncell);
	for (int c = 0; c < 3; ++c) { if (any_tail) { sized(hp->lfraw[c], ncell); hp->llf[c].clear(); } else { sized(hp->llf[c], ncell); hp->lfraw[c].clear(); } }
	hp->xfromy.resize(nc64); hp->bfromy.resize(nc64);
	sized(hp->vb_coeffoff_qfidx, nvb); sized(hp->vb_hfmul_inv, nvb);
	sized(hp->group_blocks, nvb);   // (every varblock is in exactly one group's list: its top-left cell's group)
	sized(hp->vb_sorted, nvb);
	fill_sections(fr, &hp->sections);
	hp->group_block_start.assign((size_t) num_groups + 1, 0);
	std::vector<std::vector<int32_t>> ordinal(nlf);   // [LF group][varblock] -> position in group_blocks
	for (size_t g = 0; g < nlf; ++g) ordinal[g].assign(fr.lf_groups[g].varblocks.size(), -1);
	// (an LfGroup's cells and varblocks go to the team in NCH pieces: an 8K frame has six full LfGroups and six slivers, and a thread
	// per LfGroup left half the team waiting for the other half)
	enum { NCH = 4 };
	std::vector<uint32_t> class_count(nlf * NCH * 28, 0), class_at(nlf * NCH * 28, 0), group_count((size_t) num_groups, 0);
	bool consistent = true;
	auto piece = [](size_t n, size_t ch, size_t *first, size_t *end) { *first = n * ch / NCH; *end = n * (ch + 1) / NCH; };

	const double tb1 = timing ? now() : 0;
	auto body = [&](int tid, int team, TeamBarrier &bar) {
		if (tid == team - 1) {   // (the thread with the smallest LfGroup of phase A: the codestream's padded copy, a third of a millisecond for 8K)
			hp->codestream.reserve(cs_size + 32);   // (assign + resize without it reallocates and copies the stream a second time)
			hp->codestream.assign(cs, cs + cs_size);
			hp->codestream.resize(cs_size + 32, 0);   // the lane decoders read up to three words past the position they stop at
		}
		// A: the LF bundle's pieces to their places; varblocks per (LfGroup piece, transform class); blocks per group
		for (size_t u = (size_t) tid; u < nlf * NCH; u += (size_t) team) {
			const size_t g = u / NCH, ch = u % NCH;
			const LfGroup &gg = fr.lf_groups[g];
			const DevLfGroup &d = hp->lf_groups[g];
			size_t c0, c1, q0, q1, v0, v1;
			piece(gg.blocks.size(), ch, &c0, &c1); piece(gg.xfromy.size(), ch, &q0, &q1); piece(gg.varblocks.size(), ch, &v0, &v1);
			std::copy(gg.blocks.begin() + (long) c0, gg.blocks.begin() + (long) c1, hp->blocks.begin() + d.cell_base + (long) c0);
			std::copy(gg.lfindices.begin() + (long) c0, gg.lfindices.begin() + (long) c1, hp->lfindices.begin() + d.cell_base + (long) c0);
			for (int c = 0; c < 3; ++c) {
				if (any_tail) std::copy(gg.lfraw[c].begin() + (long) c0, gg.lfraw[c].begin() + (long) c1, hp->lfraw[c].begin() + d.cell_base + (long) c0);
				else std::copy(gg.llfcoeffs[c].begin() + (long) c0, gg.llfcoeffs[c].begin() + (long) c1, hp->llf[c].begin() + d.cell_base + (long) c0);
			}
			std::copy(gg.xfromy.begin() + (long) q0, gg.xfromy.begin() + (long) q1, hp->xfromy.begin() + d.c64_base + (long) q0);
			std::copy(gg.bfromy.begin() + (long) q0, gg.bfromy.begin() + (long) q1, hp->bfromy.begin() + d.c64_base + (long) q0);
			uint32_t *cnt = class_count.data() + u * 28;
			for (size_t v = v0; v < v1; ++v) {
				const VarblockInfo &vb = gg.varblocks[v];
				hp->vb_coeffoff_qfidx[(size_t) d.vb_base + v] = vb.coeffoff_qfidx; hp->vb_hfmul_inv[(size_t) d.vb_base + v] = vb.hfmul_inv;
				++cnt[vb.dctsel >= 0 && vb.dctsel < 27 ? vb.dctsel : 27];
			}
		}
		for (int32_t g = tid; g < num_groups; g += team) {
			const DevSection &d = hp->sections[(size_t) g];
			const LfGroup &gg = fr.lf_groups[(size_t) d.ggidx];
			uint32_t n = 0;
			for (int32_t y8 = 0; y8 < d.gh8; ++y8) {
				const int32_t *row = gg.blocks.data() + (size_t) (d.gy8 + y8) * (size_t) gg.width8 + (size_t) d.gx8;
				for (int32_t x8 = 0; x8 < d.gw8; ++x8) n += (row[x8] >> 20) >= 2;
			}
			group_count[(size_t) g] = n;
		}
		bar.wait();
		if (timing && tid == 0) tbA = now();
		if (tid == 0) {   // where every group's list and every (LfGroup, class) run of records starts
			uint32_t at = 0;
			for (int32_t g = 0; g < num_groups; ++g) { hp->group_block_start[(size_t) g] = at; at += group_count[(size_t) g]; }
			hp->group_block_start[(size_t) num_groups] = at;
			consistent = at == nvb;
			uint32_t k = 0;
			for (int d = 0; d <= 27; ++d) {   // grouped by DctSelect, in (LF group, varblock) order within a class
				hp->class_start[d] = (int32_t) k;
				for (size_t u = 0; u < nlf * NCH; ++u) { class_at[u * 28 + (size_t) d] = k; k += class_count[u * 28 + (size_t) d]; }
			}
		}
		bar.wait();
		if (!consistent) return;
		// B: per-group block lists for K1, in the visiting order of j40__hf_coeffs
		for (int32_t g = tid; g < num_groups; g += team) {
			const DevSection &d = hp->sections[(size_t) g];
			const LfGroup &gg = fr.lf_groups[(size_t) d.ggidx];
			uint32_t at = hp->group_block_start[(size_t) g];
			const int32_t nb_qf1 = fr.nb_qf_thr + 1, lfidx_size = df.lfidx_size;
			for (int32_t y8 = 0; y8 < d.gh8; ++y8) for (int32_t x8 = 0; x8 < d.gw8; ++x8) {
				const size_t cell = (size_t) (d.gy8 + y8) * (size_t) gg.width8 + (size_t) (d.gx8 + x8);
				const int32_t blk = gg.blocks[cell];
				if ((blk >> 20) < 2) continue;
				DevGroupBlock gb;
				gb.coeffoff_qfidx = (uint32_t) gg.varblocks[(size_t) (blk & 0xfffff)].coeffoff_qfidx;
				gb.pos_dct = (uint16_t) ((y8 * 32 + x8) | (((blk >> 20) - 2) << 10));
				{   // block context per channel: block_ctx_map[(c_yxb * 13 + order) * (nb_qf_thr + 1) + qfidx) * lfidx_size + lfidx]
					const int32_t dctsel = (blk >> 20) - 2;
					const int32_t bctx0 = (DCT_SELECT[dctsel].order_idx * nb_qf1 + (int32_t) (gb.coeffoff_qfidx & 15u)) * lfidx_size + gg.lfindices[cell];
					uint32_t v = 0;
					for (int32_t c_yxb = 0; c_yxb < 3; ++c_yxb) v |= (uint32_t) (fr.block_ctx_map[(size_t) (bctx0 + 13 * nb_qf1 * lfidx_size * c_yxb)] & 15) << (4 * c_yxb);
					gb.bctx3 = (uint16_t) v;
				}
				ordinal[(size_t) d.ggidx][(size_t) (blk & 0xfffff)] = (int32_t) at;
				hp->group_blocks[at++] = gb;
			}
		}
		bar.wait();
		if (timing && tid == 0) tbB = now();
		// C: work lists for the coefficients -> pixels kernels: every record straight to its place (sorting a quarter of a million
		// 40-byte records afterwards was a third of this function)
		for (size_t u = (size_t) tid; u < nlf * NCH; u += (size_t) team) {
			const size_t g = u / NCH;
			const LfGroup &gg = fr.lf_groups[g];
			const DevLfGroup &d = hp->lf_groups[g];
			uint32_t *at = class_at.data() + u * 28;
			size_t v0, v1;
			piece(gg.varblocks.size(), u % NCH, &v0, &v1);
			for (size_t v = v0; v < v1; ++v) {
				const VarblockInfo &vb = gg.varblocks[v];
				DevVarblock dv;
				memset(&dv, 0, sizeof dv);
				const DctSelect &ds = DCT_SELECT[vb.dctsel];
				const int32_t coeffoff = vb.coeffoff_qfidx & ~15;
				dv.coeff_base = d.cell_base * 64 + coeffoff; dv.llf_base = d.cell_base + (coeffoff >> 6);
				dv.mult1 = df.mult_base * vb.hfmul_inv;
				const size_t c64 = (size_t) (vb.y8 / 8) * (size_t) gg.width64 + (size_t) (vb.x8 / 8);
				dv.kx_hf = fr.base_corr_x + fr.inv_colour_factor * (float) gg.xfromy[c64];   // j40.h:7138-7143, one factor per varblock
				dv.kb_hf = fr.base_corr_b + fr.inv_colour_factor * (float) gg.bfromy[c64];
				dv.px = gg.left + vb.x8 * 8; dv.py = gg.top + vb.y8 * 8;
				dv.effh = (uint16_t) std::min(gg.height - vb.y8 * 8, 1 << ds.log_rows); dv.effw = (uint16_t) std::min(gg.width - vb.x8 * 8, 1 << ds.log_columns);
				dv.dctsel = (uint8_t) vb.dctsel;
				dv.pad[0] = (uint8_t) g; dv.pad[1] = (uint8_t) (g >> 8); dv.pad[2] = (uint8_t) (g >> 16);
				dv.blk = ordinal[g][v];   // the block's ordinal in group_blocks / block_events
				hp->vb_sorted[at[vb.dctsel < 27 ? vb.dctsel : 27]++] = dv;
			}
		}
	};
	// (a team no larger than the work has pieces -- LfGroups in phase A, groups behind it -- and none for frames of a few groups:
	// starting eleven threads for a 64 x 64 image costs more than its plan)
	run_team(num_groups + (int32_t) nlf <= 8 ? 1 : std::min(threads, std::max((int32_t) nlf * NCH, num_groups)), body);
	tbC = timing ? now() : 0;
	if (!consistent) return ERR_RNGE;   // (a varblock without a top-left cell: the parse does not produce such frames, a caller's plan view may)

	// single-pass frames: sparse coefficients (DevPlan::events). A group's region is sized from its section: a non-zero
	// coefficient costs bits, 6 events per byte is far beyond what entropy coding reaches on real data; a section that still
	// overflows it fails with ERR_EVOF and the frame is decoded with dense planes instead (runtime.hip).
	df.sparse_coeffs = fr.fh.num_passes == 1 && !hp->force_dense;
	if (!fill_event_ranges(hp->sections, num_groups, df.sparse_coeffs != 0, &hp->ev_range, &hp->ev_capacity)) df.sparse_coeffs = 0;
	bool any_lz77 = false;
	for (const DevCodeSpec &sp : hp->coeff_specs) any_lz77 |= sp.lz77_enabled != 0;
	hp->lz_window_size = any_lz77 ? 3 * 65536 + 3 * 1024 + 16 : 0;  // bound on the integers one pass-group stream decodes
	fill_hf_launch_info(hp->coeff_specs, (uint32_t) fr.block_ctx_map.size(), hp->coeff_floats, &hp->hf);
	hp->max_large = 0;
	for (int d = 21; d < 27; ++d) hp->max_large = std::max(hp->max_large, hp->class_start[d + 1] - hp->class_start[d]);
	if (timing) fprintf(stderr, "[j40hip plan] before the team %.2f ms, phase A %.2f, B %.2f, C + join %.2f, after %.2f\n", tb1 - tb0, tbA - tb1, tbB - tbA, tbC - tbB, now() - tbC);
	return 0;
}

// the MA tree and the code spec a section decodes with: the global pair (flattened once, when the first header refers to it)
// or the section's own (use_global_tree = 0)
static void attach_tables(const Frame &fr, HostModPlan *hp, int32_t &global_spec, uint32_t &global_tree_off, const Modular &m, DevModSection *s) {
	const bool own = !m.use_global_tree;
	if (!own && global_spec >= 0) { s->tree_off = global_tree_off; s->tree_nodes = (int32_t) fr.global_tree.size(); s->spec_idx = global_spec; }
	else {
		const std::vector<TreeNode> &tree = *m.tree;
		s->tree_off = (uint32_t) hp->tree.size(); s->tree_nodes = (int32_t) tree.size(); s->spec_idx = (int32_t) hp->specs.size();
		for (const TreeNode &n : tree) hp->tree.push_back(DevTreeNode{n.prop, n.value, n.a, n.b});
		hp->specs.emplace_back(); hp->host_specs.push_back(*m.codespec);
		flatten_code_spec(*m.codespec, hp->pool_u8, hp->pool_i32, hp->pool_u64, hp->clusters, &hp->specs.back());
		if (!own) { global_spec = s->spec_idx; global_tree_off = s->tree_off; }
	}
	s->uses_wp = tree_uses_wp(*m.tree);
	const DevCodeSpec &sp = hp->specs[(size_t) s->spec_idx];
	hp->max_tree_nodes = std::max(hp->max_tree_nodes, s->tree_nodes);
	hp->max_num_dist = std::max(hp->max_num_dist, sp.num_dist); hp->max_clusters = std::max(hp->max_clusters, sp.num_clusters);
	hp->max_table_bytes = std::max(hp->max_table_bytes, sp.table_span * (sp.use_prefix_code ? 4u : 8u));
	hp->any_lz77 = hp->any_lz77 || sp.lz77_enabled; hp->any_wp = hp->any_wp || s->uses_wp;
}

// the MA tree at hp->tree[tree_off ...] with code spec `spec_idx` as a DevCoopTree, or false when k_modular_coop cannot take it
static bool build_coop_tree(const HostModPlan &hp, uint32_t tree_off, int32_t tree_nodes, int32_t spec_idx, DevCoopTree *out) {
	const DevCodeSpec &sp = hp.specs[(size_t) spec_idx];
	const CodeSpec &hs = hp.host_specs[(size_t) spec_idx];
	if (sp.use_prefix_code || sp.lz77_enabled || tree_nodes <= 0) return false;
	memset(out, 0, sizeof *out);
	for (int i = 0; i < 64; ++i) { out->node_prop[i] = -1; out->want_lo[i] = 1; }   // (mask 0, want 1: never reached)
	struct Item { int32_t node; uint64_t mask, want; };
	std::vector<Item> stack{{0, 0, 0}};
	int32_t nn = 0, nl = 0;
	while (!stack.empty()) {
		const Item it = stack.back(); stack.pop_back();
		if (it.node < 0 || it.node >= tree_nodes) return false;
		const DevTreeNode &n = hp.tree[(size_t) tree_off + (size_t) it.node];
		if (n.prop >= 0) {
			if (n.prop > 14 || nn >= 64) return false;
			const uint64_t bit = (uint64_t) 1 << nn;
			out->node_prop[nn] = n.prop; out->node_thr[nn] = n.value; out->used_props |= 1u << n.prop; ++nn;
			stack.push_back({it.node + n.b, it.mask | bit, it.want});
			stack.push_back({it.node + n.a, it.mask | bit, it.want | bit});
		} else {
			const int32_t predictor = -1 - n.prop;
			if (predictor == 6 || predictor > 13 || nl >= 64) return false;
			if (n.value < 0 || n.value >= sp.num_dist || (size_t) n.value >= hs.cluster_map.size()) return false;
			const DevCluster &cl = hp.clusters[(size_t) sp.cluster_off + (size_t) hs.cluster_map[(size_t) n.value]];
			if (cl.max_token < 0 || cl.max_token > 0xffff || cl.cfg > 0xfff) return false;
			out->mask_lo[nl] = (uint32_t) it.mask; out->mask_hi[nl] = (uint32_t) (it.mask >> 32);
			out->want_lo[nl] = (uint32_t) it.want; out->want_hi[nl] = (uint32_t) (it.want >> 32);
			out->leaf_a[nl] = (uint32_t) predictor | cl.cfg << 4 | (uint32_t) cl.max_token << 16;
			out->leaf_tab[nl] = cl.table_off; out->leaf_off[nl] = n.a; out->leaf_mul[nl] = n.b;
			++nl;
		}
	}
	out->num_nodes = nn; out->num_leaves = nl;
	return true;
}

// the global MA tree and code spec of a VarDCT frame laid out for the wave-cooperative decoder (device/lf_decode.hip); false when it
// cannot take them (prefix codes, LZ77, weighted predictor, previous-channel properties, more than 64 leaves)
bool build_lf_coop(const Frame &fr, DevCoopTree *tree, std::vector<uint64_t> *alias, int32_t *log_alpha_size) {
	if (fr.global_tree.empty()) return false;
	HostModPlan hp;
	for (const TreeNode &n : fr.global_tree) hp.tree.push_back(DevTreeNode{n.prop, n.value, n.a, n.b});
	hp.specs.emplace_back(); hp.host_specs.push_back(fr.global_codespec);
	flatten_code_spec(fr.global_codespec, hp.pool_u8, hp.pool_i32, hp.pool_u64, hp.clusters, &hp.specs.back());
	if (!build_coop_tree(hp, 0, (int32_t) hp.tree.size(), 0, tree)) return false;
	alias->swap(hp.pool_u64);
	*log_alpha_size = hp.specs[0].log_alpha_size;
	return true;
}

// which sections k_modular_coop decodes: those whose tree / code spec it can take and whose channels are not wider than its row
// buffers allow
static void assign_coop(HostModPlan *hp) {
	static const bool off = [] { const char *e = getenv("J40HIP_NO_COOP"); return e && atoi(e); }();
	std::vector<std::pair<uint64_t, int32_t>> known;   // (tree_off, spec_idx) -> coop tree or -1
	hp->coop_width = 0; hp->coop_sections = 0;
	std::vector<int32_t> section_width;   // widest channel of the sections k_modular_coop could take, -1 for the others
	for (DevModSection &s : hp->sections) {
		s.coop_idx = -1;
		if (off || s.preset_status) { section_width.push_back(-1); continue; }
		int32_t widest = 0;
		for (int32_t c = 0; c < s.num_channels; ++c) {
			int32_t w;
			if (s.sub_off >= 0) w = hp->sub_w[(size_t) (s.sub_off + c)];
			else if (s.chan_off >= 0) w = hp->chan_rects[(size_t) (s.chan_off + c)].w;
			else w = hp->plane_meta[(size_t) (s.first_channel + c)] ? hp->plane_w[(size_t) (s.first_channel + c)] : s.gw;
			widest = std::max(widest, w);
		}
		if (widest > 4096) { section_width.push_back(-1); continue; }
		const uint64_t key = (uint64_t) s.tree_off << 32 | (uint32_t) s.spec_idx;
		int32_t idx = -2;
		for (const auto &k : known) if (k.first == key) { idx = k.second; break; }
		if (idx == -2) {
			DevCoopTree t;
			idx = build_coop_tree(*hp, s.tree_off, s.tree_nodes, s.spec_idx, &t) ? (int32_t) hp->coop_trees.size() : -1;
			if (idx >= 0) hp->coop_trees.push_back(t);
			known.push_back({key, idx});
		}
		s.coop_idx = idx;
		if (idx >= 0) { hp->coop_width = std::max(hp->coop_width, widest); ++hp->coop_sections; }
		section_width.push_back(idx >= 0 ? widest : -1);
	}
	// Four sections to a wavefront (k_modular_quad): every instruction then serves four streams. They must share one code spec (its
	// alias tables are staged in LDS once per workgroup) -- the one most sections use. OFF unless J40HIP_QUAD_MIN=<sections> is set:
	// measured on 16384 x 16384 (4096 sections) it only ties k_modular_coop (222 vs 226 ms) -- one wavefront per SIMD is bound by the
	// latency of its own dependent chain; it needs >= 8192 sections (two wavefronts per SIMD) to pay. Kept as a tested variant.
	static const int32_t quad_min = [] { const char *e = getenv("J40HIP_QUAD_MIN"); return e ? atoi(e) : -1; }();
	hp->quad_sections = 0; hp->quad_spec = 0; hp->quad_width = 0;
	for (DevModSection &s : hp->sections) s.quad = 0;
	if (quad_min >= 0 && hp->coop_sections >= std::max(quad_min, 1)) {
		std::vector<int32_t> votes(hp->specs.size(), 0);
		for (const DevModSection &s : hp->sections) if (s.coop_idx >= 0) ++votes[(size_t) s.spec_idx];
		const int32_t spec = (int32_t) (std::max_element(votes.begin(), votes.end()) - votes.begin());
		const DevCodeSpec &sp = hp->specs[(size_t) spec];
		if (sp.table_span * 8u <= 32u * 1024u) {
			for (size_t i = 0; i < hp->sections.size(); ++i) {
				DevModSection &s = hp->sections[i];
				if (s.coop_idx < 0 || s.spec_idx != spec || section_width[i] > 512) continue;
				s.quad = 1; ++hp->quad_sections; hp->quad_width = std::max(hp->quad_width, section_width[i]);
			}
			hp->quad_spec = spec;
		}
	}
}

// after assign_split took sections away from k_modular_coop / k_modular_quad: their counts and widths again
static void recount_coop(HostModPlan *hp);
// Which sections modular_split.hip decodes in two passes (token parse, then prediction): those whose MA tree looks only at where a
// sample is -- properties 0-3 -- and predicts without the weighted predictor; fast lossless encoders write nothing else (one gradient
// leaf per channel). They leave k_modular_coop's list. J40HIP_NO_SPLIT=1: none (the one-pass kernels, for comparison).
static void assign_split(HostModPlan *hp) {
	static const bool off = [] { const char *e = getenv("J40HIP_NO_SPLIT"); return e && atoi(e); }();
	hp->split_sections = 0; hp->split_width = 0; hp->split_channels = 0; hp->split_samples = 0;
	std::vector<std::pair<uint64_t, int32_t>> known;   // (tree_off, tree_nodes) -> 0 no, 1 yes, 2 yes and it tests the column
	auto tree_kind = [&](uint32_t tree_off, int32_t tree_nodes, int32_t spec_idx) {
		if (tree_nodes <= 0) return 0;
		const DevCodeSpec &sp = hp->specs[(size_t) spec_idx];
		int32_t kind = 1;
		// every node reachable from the root: a branch on properties 0-3 with both children inside the tree, or a leaf with a predictor
		// the prediction pass has (0-13 but the weighted one) and a context the code spec knows
		std::vector<int32_t> stack{0}; std::vector<uint8_t> seen((size_t) tree_nodes, 0);
		while (!stack.empty()) {
			const int32_t at = stack.back(); stack.pop_back();
			if (at < 0 || at >= tree_nodes) return 0;
			if (seen[(size_t) at]) return 0;   // (not a tree)
			seen[(size_t) at] = 1;
			const DevTreeNode &n = hp->tree[(size_t) tree_off + (size_t) at];
			if (n.prop >= 0) {
				if (n.prop > 3) return 0;
				if (n.prop == 3) kind = 2;
				stack.push_back(at + n.a); stack.push_back(at + n.b);
			} else {
				const int32_t predictor = -1 - n.prop;
				if (predictor == 6 || predictor > 13) return 0;
				if (n.value < 0 || n.value >= sp.num_dist) return 0;
			}
		}
		return kind;
	};
	for (DevModSection &s : hp->sections) {
		s.split = 0; s.res_off = 0; s.res_count = 0;
		if (off || s.uses_wp) continue;
		int32_t widest = 0; size_t samples = 0;
		for (int32_t c = 0; c < s.num_channels; ++c) {
			int32_t w, h;
			if (s.sub_off >= 0) { w = hp->sub_w[(size_t) (s.sub_off + c)]; h = hp->sub_h[(size_t) (s.sub_off + c)]; }
			else if (s.chan_off >= 0) { w = hp->chan_rects[(size_t) (s.chan_off + c)].w; h = hp->chan_rects[(size_t) (s.chan_off + c)].h; }
			else if (hp->plane_meta[(size_t) (s.first_channel + c)]) { w = hp->plane_w[(size_t) (s.first_channel + c)]; h = hp->plane_h[(size_t) (s.first_channel + c)]; }
			else { w = s.gw; h = s.gh; }
			widest = std::max(widest, w);
			if (w > 0 && h > 0) samples += (size_t) w * (size_t) h;
		}
		if (widest > 8192 || samples == 0 || hp->split_samples + samples >= ((size_t) 1 << 32) - 64) continue;
		int32_t kind = -1;
		const uint64_t key = (uint64_t) s.tree_off << 32 | (uint32_t) s.spec_idx;
		for (const auto &k : known) if (k.first == key) { kind = k.second; break; }
		if (kind < 0) { kind = s.preset_status ? 0 : tree_kind(s.tree_off, s.tree_nodes, s.spec_idx); if (!s.preset_status) known.push_back({key, kind}); }
		if (!kind) continue;
		s.split = kind; s.coop_idx = -1; s.quad = 0;
		s.res_off = (uint32_t) hp->split_samples; s.res_count = (uint32_t) samples;
		hp->split_samples += (samples + 63) & ~(size_t) 63;
		++hp->split_sections; hp->split_width = std::max(hp->split_width, widest); hp->split_channels = std::max(hp->split_channels, s.num_channels);
	}
}

static void recount_coop(HostModPlan *hp) {
	if (!hp->split_sections) return;
	hp->coop_sections = 0; hp->quad_sections = 0; hp->coop_width = 0; hp->quad_width = 0;
	for (const DevModSection &s : hp->sections) {
		if (s.coop_idx < 0) continue;
		int32_t widest = 0;
		for (int32_t c = 0; c < s.num_channels; ++c) {
			int32_t w;
			if (s.sub_off >= 0) w = hp->sub_w[(size_t) (s.sub_off + c)];
			else if (s.chan_off >= 0) w = hp->chan_rects[(size_t) (s.chan_off + c)].w;
			else w = hp->plane_meta[(size_t) (s.first_channel + c)] ? hp->plane_w[(size_t) (s.first_channel + c)] : s.gw;
			widest = std::max(widest, w);
		}
		++hp->coop_sections; hp->coop_width = std::max(hp->coop_width, widest);
		if (s.quad) { ++hp->quad_sections; hp->quad_width = std::max(hp->quad_width, widest); }
	}
}

uint32_t build_modular_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, HostModPlan *hp) {
	if (!fr.fh.is_modular) return ERR_TODO;
	if (cs_size + 16 >= ((size_t) 1 << 29)) return ERR_TODO;   // the kernels address the codestream with 32-bit BIT positions
	const Modular &gm = fr.gmodular;
	const int32_t nch = (int32_t) gm.channel.size();
	if (nch > MOD_MAX_CHANNELS) return ERR_TODO;
	// what the reference's renderer accepts (j40.h:7917-7936)
	if (fr.im.bpp < 8 || fr.im.exp_bits || !fr.im.modular_16bit_buffers) return ERR_TODO;
	// a Modular frame flagged XYB or YCbCr gets no colour transform in the reference (j40.h:8209-8210 runs only the
	// Modular inverse transforms, j40.h:7910 renders the first three channels as they are): same here
	DevModFrame &df = hp->frame;
	memset(&df, 0, sizeof df);
	df.width = fr.fh.width; df.height = fr.fh.height; df.num_groups = (int32_t) fr.fh.num_groups; df.bpp = fr.im.bpp;
	df.num_channels = nch;
	df.check_section_end = fr.toc.single;
	df.single_declared_end = (uint32_t) fr.toc.single_declared_end;
	// trees and code specs: the global pair once (when a header refers to it), own pairs per section (use_global_tree = 0)
	int32_t global_spec = -1; uint32_t global_tree_off = 0;
	auto attach = [&](const Modular &m, DevModSection *s) { attach_tables(fr, hp, global_spec, global_tree_off, m, s); };
	hp->pool_u8.resize(hp->pool_u8.size() + 16, 0);
	hp->transforms = gm.transforms;
	int32_t max_width = 1;
	for (const Plane &p : gm.channel) { hp->plane_w.push_back(p.width); hp->plane_h.push_back(p.height); hp->plane_meta.push_back(p.vshift < 0); }
	auto wp_bytes = [](const WPParams &wp, int8_t *out) { out[0] = wp.p1; out[1] = wp.p2; for (int i = 0; i < 5; ++i) out[2 + i] = wp.p3[i]; for (int i = 0; i < 4; ++i) out[7 + i] = wp.w[i]; out[11] = 0; };
	bool shifted = false;
	for (const Transform &t : gm.transforms) shifted = shifted || t.kind == Transform::SQUEEZE;
	if (shifted) {
		// Squeeze leaves channels of different sizes and shifts: every section lists its channels as explicit rectangles, and
		// the channels are dealt out to the sections by their shifts (ISO 18181-1; the reference stops before this point with
		// "TODO", j40.h:3812, 6731-6737): LfGlobal codes the meta channels and the channels behind them that fit one group;
		// an LfGroup section the channels shifted by 3 or more in both directions, over its 8-groups-wide area; a pass-group
		// section the rest over its own area. A channel's rectangle is the area shifted down and clipped to the channel.
		if (!fr.gm_data_pending || fr.fh.num_passes != 1) return ERR_TODO;
		if (!gm.tree || gm.tree->empty() || !gm.codespec) return E4("mtre");
		const int32_t gdim = 1 << fr.fh.group_size_shift;
		auto add_rects = [&](DevModSection *s, const std::vector<int32_t> &chans, int32_t left, int32_t top, int32_t dim, bool whole, Modular *m) {
			s->chan_off = (int32_t) hp->chan_rects.size(); s->num_channels = 0;
			for (int32_t c : chans) {
				const Plane &p = gm.channel[(size_t) c];
				DevChanRect r; r.plane = c; r.shifts = (int32_t) (uint8_t) p.hshift | ((int32_t) (uint8_t) p.vshift << 8);
				if (whole || p.vshift < 0) { r.x0 = r.y0 = 0; r.w = p.width; r.h = p.height; }
				else {
					r.x0 = left >> p.hshift; r.y0 = top >> p.vshift;
					r.w = std::min(dim >> p.hshift, p.width - r.x0); r.h = std::min(dim >> p.vshift, p.height - r.y0);
					if (r.w <= 0 || r.h <= 0) continue;
				}
				hp->chan_rects.push_back(r); ++s->num_channels;
				max_width = std::max(max_width, r.w);
				if (m) { Plane q; q.width = r.w; q.height = r.h; q.hshift = p.hshift; q.vshift = p.vshift; m->channel.push_back(q); }
			}
		};
		{
			DevModSection s;
			memset(&s, 0, sizeof s);
			s.sub_off = -1;
			const Section &ls = fr.toc.single ? fr.toc.single_section : fr.toc.lf_global;
			s.byte_off = (uint32_t) ls.offset; s.size = (uint32_t) ls.size; s.bit_off = (uint32_t) fr.gm_data_bitpos;
			s.gw = fr.fh.width; s.gh = fr.fh.height; s.sidx = 0;
			s.dist_mult_p1 = gm.dist_mult + 1;
			std::vector<int32_t> chans;
			for (int32_t c = 0; c < fr.num_gm_channels; ++c) chans.push_back(c);
			add_rects(&s, chans, 0, 0, 0, true, nullptr);
			wp_bytes(gm.wp, s.wp);
			attach(gm, &s);
			hp->sections.push_back(s);
		}
		// one section of a group-like area: its own Modular header first (no transforms of its own are taken here)
		auto add_section = [&](const Section &ps, const std::vector<int32_t> &chans, int32_t left, int32_t top, int32_t dim, int64_t sidx) -> uint32_t {
			DevModSection s;
			memset(&s, 0, sizeof s);
			s.sub_off = -1;
			Modular m; m.bpp = fr.im.bpp;
			add_rects(&s, chans, left, top, dim, false, &m);
			if (s.num_channels == 0) return 0;   // nothing is coded for this area
			s.byte_off = (uint32_t) ps.offset; s.size = (uint32_t) ps.size; s.gx = left; s.gy = top; s.gw = s.gh = dim; s.sidx = (int32_t) sidx;
			BitReader br(cs + ps.offset, ps.size);
			try { read_modular_header(br, &fr.global_tree, &fr.global_codespec, &m); }
			catch (const DecodeError &e) { s.preset_status = e.code; s.num_channels = 0; hp->sections.push_back(s); return 0; }
			if (!m.transforms.empty()) return ERR_TODO;
			s.bit_off = (uint32_t) br.bit_position();
			wp_bytes(m.wp, s.wp);
			attach(m, &s);
			hp->sections.push_back(s);
			return 0;
		};
		if (!fr.toc.single && fr.num_gm_channels < nch) {
			std::vector<int32_t> lf_chans, hf_chans;
			for (int32_t c = fr.num_gm_channels; c < nch; ++c) {
				const Plane &p = gm.channel[(size_t) c];
				if (p.hshift < 0 || p.vshift < 0) return ERR_TODO;   // a meta channel behind an image channel
				(p.hshift >= 3 && p.vshift >= 3 ? lf_chans : hf_chans).push_back(c);
			}
			for (int64_t gg = 0; gg < fr.fh.num_lf_groups; ++gg) {
				const LfGroup &g = fr.lf_groups[(size_t) gg];
				if (uint32_t e = add_section(fr.toc.lf_groups[(size_t) gg], lf_chans, g.left, g.top, gdim * 8, 1 + fr.fh.num_lf_groups + gg)) return e;
			}
			const int32_t num_groups = (int32_t) fr.fh.num_groups;
			hp->num_passes = 1; hp->sections_per_pass = 0;   // (one launch decodes every section)
			for (int32_t g = 0; g < num_groups; ++g) {
				const GroupInfo gi = group_info(fr.fh, g);
				const LfGroup &gg = fr.lf_groups[(size_t) gi.ggidx];
				if (uint32_t e = add_section(fr.toc.pass_groups[(size_t) g], hf_chans, gg.left + gi.gx_in_gg, gg.top + gi.gy_in_gg, gdim, 1 + 3 * fr.fh.num_lf_groups + 17 + g)) return e;
			}
		}
	} else
	// LfGlobal's own channel data: every channel for single-group frames, only meta channels otherwise
	if (fr.gm_data_pending) {
		// (with zero channels this still validates the stream's final rANS state and the section end)
		DevModSection s;
		memset(&s, 0, sizeof s);
		s.sub_off = -1; s.chan_off = -1;
		const Section &ls = fr.toc.single ? fr.toc.single_section : fr.toc.lf_global;
		s.byte_off = (uint32_t) ls.offset; s.size = (uint32_t) ls.size; s.bit_off = (uint32_t) fr.gm_data_bitpos;
		s.gx = s.gy = 0; s.gw = fr.fh.width; s.gh = fr.fh.height; s.sidx = 0;
		s.first_channel = 0; s.num_channels = fr.num_gm_channels;
		s.dist_mult_p1 = gm.dist_mult + 1;   // the frame-wide image's multiplier (j40.h:3840-3844), not only that of the channels coded here
		wp_bytes(gm.wp, s.wp);
		if (!gm.tree || gm.tree->empty() || !gm.codespec) return E4("mtre");
		attach(gm, &s);
		hp->sections.push_back(s);
		for (int32_t c = 0; c < fr.num_gm_channels; ++c) max_width = std::max(max_width, hp->plane_meta[(size_t) c] ? hp->plane_w[(size_t) c] : fr.fh.width);
	} else return ERR_TODO;
	if (!shifted && !fr.toc.single && fr.num_gm_channels < nch) {
		const int32_t num_groups = (int32_t) fr.fh.num_groups;
		// every pass codes all channels of every group again (the reference's j40__pass_group ignores the passes' shift
		// ranges, j40.h:7025, 3702): decoded in order, the last pass stays
		hp->num_passes = fr.fh.num_passes; hp->sections_per_pass = num_groups;
		for (int32_t pass = 0; pass < fr.fh.num_passes; ++pass) for (int32_t g = 0; g < num_groups; ++g) {
			const Section &ps = fr.toc.pass_groups[(size_t) pass * (size_t) num_groups + (size_t) g];
			const GroupInfo gi = group_info(fr.fh, g);
			const LfGroup &gg = fr.lf_groups[(size_t) gi.ggidx];
			// the section starts with a Modular header for the group's sub-image (j40.h:7024-7026)
			Modular m; m.bpp = fr.im.bpp;
			for (int32_t c = fr.num_gm_channels; c < nch; ++c) { Plane p; p.width = gi.gw; p.height = gi.gh; m.channel.push_back(p); }
			BitReader br(cs + ps.offset, ps.size);
			DevModSection s;
			memset(&s, 0, sizeof s);
			s.sub_off = -1; s.chan_off = -1;
			try { read_modular_header(br, &fr.global_tree, &fr.global_codespec, &m); }
			catch (const DecodeError &e) {   // reported in its place among the sections (an earlier section's data may fail first)
				s.byte_off = (uint32_t) ps.offset; s.size = (uint32_t) ps.size; s.preset_status = e.code;
				hp->sections.push_back(s);
				continue;
			}
			// the group's own transforms. RCTs only: undone in place over the group's rectangle. With a palette the channel list of
			// the section differs from the frame's: it decodes into a sub-image of its own, the host undoes its transforms there
			s.local_off = (int32_t) (hp->local_rct.size() / 2);
			bool own_palette = false;
			for (const Transform &t : m.transforms) own_palette = own_palette || t.kind == Transform::PALETTE;
			if (own_palette) {
				HostModPlan::SubImage si;
				si.section = (int32_t) hp->sections.size(); si.first_plane = (int32_t) hp->sub_w.size(); si.num_planes = (int32_t) m.channel.size();
				si.transforms = m.transforms; si.paste = pass + 1 == fr.fh.num_passes;
				wp_bytes(m.wp, si.wp);
				for (const Plane &p : m.channel) { hp->sub_w.push_back(p.width); hp->sub_h.push_back(p.height); hp->sub_meta.push_back(p.vshift < 0); max_width = std::max(max_width, p.width); }
				hp->sub_images.push_back(si);
				s.sub_off = si.first_plane;
			} else for (const Transform &t : m.transforms) {
				if (t.kind != Transform::RCT || t.begin_c < 0 || t.begin_c + 3 > nch - fr.num_gm_channels) return ERR_TODO;
				if (pass + 1 < fr.fh.num_passes) continue;   // overwritten by the next pass before anything reads it
				hp->local_rct.push_back(t.begin_c); hp->local_rct.push_back(t.rct_type);
				++s.local_count;
			}
			s.byte_off = (uint32_t) ps.offset; s.size = (uint32_t) ps.size; s.bit_off = (uint32_t) br.bit_position();
			s.gx = gg.left + gi.gx_in_gg; s.gy = gg.top + gi.gy_in_gg; s.gw = gi.gw; s.gh = gi.gh;
			s.sidx = (int32_t) (1 + 3 * fr.fh.num_lf_groups + 17 + pass * num_groups + g);
			s.first_channel = fr.num_gm_channels; s.num_channels = own_palette ? (int32_t) m.channel.size() : nch - fr.num_gm_channels;
			wp_bytes(m.wp, s.wp);
			attach(m, &s);
			hp->sections.push_back(s);
			max_width = std::max(max_width, gi.gw);
		}
	}
	df.num_sections = (int32_t) hp->sections.size();
	df.max_width = max_width;
	// where the alpha channel ends up after the inverse transforms: colour channels are 0..2, extra
	// channels follow (j40.h:7923-7936); the transforms above preserve that tail order
	hp->alpha_channel = -1;
	for (size_t i = 0; i < fr.im.ec.size(); ++i) if (fr.im.ec[i].type == EC_ALPHA) {
		if (fr.im.ec[i].bpp != fr.im.bpp || fr.im.ec[i].exp_bits != fr.im.exp_bits || fr.im.ec[i].dim_shift || fr.im.ec[i].alpha_associated) return ERR_TODO;
		hp->alpha_channel = 3 + (int32_t) i;
		break;
	}
	hp->lz_window_size = 0;
	df.tree_uses_wp = hp->any_wp; df.num_tree_nodes = hp->max_tree_nodes;
	if (hp->any_lz77) {
		// integers decoded by one section: at most all samples of its rectangle in every channel
		size_t most = 0;
		for (const DevModSection &s : hp->sections) {
			size_t n = (size_t) s.num_channels * (size_t) s.gw * (size_t) s.gh;
			if (s.chan_off >= 0) { n = 0; for (int32_t c = 0; c < s.num_channels; ++c) n += (size_t) hp->chan_rects[(size_t) (s.chan_off + c)].w * (size_t) hp->chan_rects[(size_t) (s.chan_off + c)].h; }
			if (s.sub_off >= 0) { n = 0; for (int32_t c = 0; c < s.num_channels; ++c) n += (size_t) hp->sub_w[(size_t) (s.sub_off + c)] * (size_t) hp->sub_h[(size_t) (s.sub_off + c)]; }
			most = std::max(most, n);
		}
		hp->lz_window_size = (uint32_t) std::min<size_t>(most + 16, (size_t) 1 << 26);
	}
	hp->codestream.reserve(cs_size + 32);   // (assign + resize without it reallocates and copies the stream a second time)
	hp->codestream.assign(cs, cs + cs_size);
	hp->codestream.resize(cs_size + 32, 0);   // the lane decoders read up to three words past the position they stop at
	assign_coop(hp);
	assign_split(hp);
	recount_coop(hp);
	return 0;
}

// VarDCT frames with extra channels: after its HF coefficients every pass-group section carries the extra channels of the group
// as a Modular sub-image (j40.h:7024-7033). The reference decodes it and later drops it with the rest of the Modular image
// (j40__combine_vardct, j40.h:7868), so it never reaches the pixels, but damage in it is reported. The entropy kernel leaves the
// bit where each section's coefficients end; this lays out a Modular decode of what follows, into planes nobody reads.
// end_bits / k1_status: per section (pass-major), from the device. Sections that already failed are left out; a header that does
// not parse yields that section's error in trailer_errors (section index, code).
uint32_t build_trailer_plan(const Frame &fr, const uint8_t *cs, size_t cs_size, const uint32_t *end_bits, const uint32_t *k1_status, HostModPlan *hp,
		std::vector<std::pair<int32_t, uint32_t>> *trailer_errors, std::vector<int32_t> *section_of) {
	const Modular &gm = fr.gmodular;
	if (cs_size + 16 >= ((size_t) 1 << 29)) return ERR_TODO;   // the kernels address the codestream with 32-bit BIT positions
	const int32_t nch = (int32_t) gm.channel.size();
	if (nch <= fr.num_gm_channels || nch > MOD_MAX_CHANNELS) return ERR_TODO;
	DevModFrame &df = hp->frame;
	memset(&df, 0, sizeof df);
	df.width = fr.fh.width; df.height = fr.fh.height; df.num_groups = (int32_t) fr.fh.num_groups; df.bpp = fr.im.bpp; df.num_channels = 0;
	int32_t global_spec = -1, max_width = 1; uint32_t global_tree_off = 0;
	hp->pool_u8.resize(hp->pool_u8.size() + 16, 0);
	auto wp_bytes = [](const WPParams &wp, int8_t *out) { out[0] = wp.p1; out[1] = wp.p2; for (int i = 0; i < 5; ++i) out[2 + i] = wp.p3[i]; for (int i = 0; i < 4; ++i) out[7 + i] = wp.w[i]; out[11] = 0; };
	const int32_t num_groups = (int32_t) fr.fh.num_groups;
	for (int32_t pass = 0; pass < fr.fh.num_passes; ++pass) for (int32_t g = 0; g < num_groups; ++g) {
		const int32_t idx = pass * num_groups + g;
		if (k1_status[idx]) continue;
		const Section &ps = fr.toc.single ? fr.toc.single_section : fr.toc.pass_groups[(size_t) idx];
		const GroupInfo gi = group_info(fr.fh, g);
		Modular m; m.bpp = fr.im.bpp;
		for (int32_t c = fr.num_gm_channels; c < nch; ++c) { Plane p; p.width = gi.gw; p.height = gi.gh; m.channel.push_back(p); }
		const size_t start = (size_t) end_bits[idx];
		if (start < 8 * ps.offset || start > 8 * (ps.offset + ps.size)) { trailer_errors->push_back({idx, E4("shrt")}); continue; }
		BitReader br(cs + ps.offset, ps.size);
		try { br.skip_bits((int64_t) (start - 8 * ps.offset)); read_modular_header(br, &fr.global_tree, &fr.global_codespec, &m); }
		catch (const DecodeError &e) { trailer_errors->push_back({idx, e.code}); continue; }
		DevModSection s;
		memset(&s, 0, sizeof s);
		s.chan_off = -1;
		s.byte_off = (uint32_t) ps.offset; s.size = (uint32_t) ps.size; s.bit_off = (uint32_t) br.bit_position();
		s.gx = s.gy = 0; s.gw = gi.gw; s.gh = gi.gh;
		s.sidx = (int32_t) (1 + 3 * fr.fh.num_lf_groups + 17 + idx);
		s.first_channel = 0; s.num_channels = (int32_t) m.channel.size();
		s.sub_off = (int32_t) hp->sub_w.size();   // everything lands in planes of the section's own
		for (const Plane &p : m.channel) { hp->sub_w.push_back(p.width); hp->sub_h.push_back(p.height); hp->sub_meta.push_back(p.vshift < 0); max_width = std::max(max_width, p.width); }
		wp_bytes(m.wp, s.wp);
		attach_tables(fr, hp, global_spec, global_tree_off, m, &s);
		hp->sections.push_back(s);
		section_of->push_back(idx);
	}
	(void) cs_size;
	df.num_sections = (int32_t) hp->sections.size();
	df.max_width = max_width;
	df.tree_uses_wp = hp->any_wp; df.num_tree_nodes = hp->max_tree_nodes;
	hp->lz_window_size = 0;
	if (hp->any_lz77) {
		size_t most = 0;
		for (const DevModSection &s : hp->sections) { size_t n = 0; for (int32_t c = 0; c < s.num_channels; ++c) n += (size_t) hp->sub_w[(size_t) (s.sub_off + c)] * (size_t) hp->sub_h[(size_t) (s.sub_off + c)]; most = std::max(most, n); }
		hp->lz_window_size = (uint32_t) std::min<size_t>(most + 16, (size_t) 1 << 26);
	}
	assign_coop(hp);
	assign_split(hp);
	recount_coop(hp);
	return 0;
}

} // namespace j40hip
