// j40_amd/csrc/modular.cpp -- see modular.hpp
#include "modular.hpp"
#include "device/squeeze_dev.h"
#include "device/props_dev.h"
#include <algorithm>

namespace j40hip {

void read_tree(BitReader &br, int32_t max_tree_size, int32_t depth_limit, std::vector<TreeNode> *tree, CodeSpec *codespec) {  // j40.h:3461
	CodeSpec treespec;
	read_code_spec(br, 6, &treespec);
	CodeState code(&treespec);
	std::vector<TreeNode> t;
	int32_t ctx_id = 0, nodes_left = 1, depth = 0, nodes_upto_this_depth = 1;
	while (nodes_left-- > 0) {  // nodes arrive level by level
		if ((int32_t) t.size() == nodes_upto_this_depth) {
			J40HIP_SHOULD(++depth <= depth_limit, "tlim");
			nodes_upto_this_depth += nodes_left + 1;
		}
		int32_t prop = decode_symbol(br, code, 1, 0);
		TreeNode n;
		if (prop > 0) {
			n.prop = prop - 1;
			n.value = unpack_signed(decode_symbol(br, code, 0, 0));
			n.a = ++nodes_left;
			n.b = ++nodes_left;
		} else {
			int32_t predictor = decode_symbol(br, code, 2, 0);
			n.prop = -1 - predictor;
			n.value = ctx_id++;
			n.a = unpack_signed(decode_symbol(br, code, 3, 0));
			int32_t shift = decode_symbol(br, code, 4, 0);
			J40HIP_SHOULD(shift < 31, "tree");
			int32_t val = decode_symbol(br, code, 5, 0);
			J40HIP_SHOULD(((val + 1) >> (31 - shift)) == 0, "tree");
			n.b = (val + 1) << shift;
		}
		t.push_back(n);
		J40HIP_SHOULD((int64_t) t.size() + nodes_left <= max_tree_size, "tlim");
	}
	finish_code(br, code);
	read_code_spec(br, ctx_id, codespec);
	tree->swap(t);
}

bool tree_uses_wp(const std::vector<TreeNode> &tree) {  // j40.h:4142-4152
	for (const TreeNode &n : tree) if (n.prop == 15 || n.prop == -1 - 6) return true;
	return false;
}

static bool equal_sized(const Plane *b, const Plane *e) {  // j40.h:1094
	if (b >= e) return false;
	const Plane &c = *b;
	bool shift_should_match = c.vshift >= 0 && c.hshift >= 0;
	for (const Plane *p = b + 1; p < e; ++p) {
		if (c.width != p->width || c.height != p->height) return false;
		if (shift_should_match && (c.vshift != p->vshift || c.hshift != p->hshift)) return false;
	}
	return true;
}

void read_modular_header(BitReader &br, const std::vector<TreeNode> *global_tree, const CodeSpec *global_codespec, Modular *m) {  // j40.h:3717
	std::vector<Plane> &channel = m->channel;
	int32_t nb_meta = 0;
	m->use_global_tree = br.u(1);
	J40HIP_SHOULD(!m->use_global_tree || (global_tree && !global_tree->empty()), "mtre");
	if (!br.u(1)) {  // custom weighted-predictor parameters
		m->wp.p1 = (int8_t) br.u(5); m->wp.p2 = (int8_t) br.u(5);
		for (int i = 0; i < 5; ++i) m->wp.p3[i] = (int8_t) br.u(5);
		for (int i = 0; i < 4; ++i) m->wp.w[i] = (int8_t) br.u(4);
	}
	int32_t nb_transforms = br.u32(0, 0, 1, 0, 2, 4, 18, 8);
	J40HIP_SHOULD(nb_transforms <= 8, "xlim");  // level-5 limit, j40.h:1173
	for (int32_t i = 0; i < nb_transforms; ++i) {
		Transform tr;
		tr.kind = (Transform::Kind) br.u(2);
		int32_t num_channels = (int32_t) channel.size();
		switch (tr.kind) {
		case Transform::RCT: {
			tr.begin_c = br.u32(0, 3, 8, 6, 72, 10, 1096, 13);
			tr.rct_type = br.u32(6, 0, 0, 2, 2, 4, 10, 6);
			J40HIP_SHOULD(tr.rct_type < 42, "rctt");
			J40HIP_SHOULD(tr.begin_c + 3 <= num_channels, "rctc");
			J40HIP_SHOULD(tr.begin_c >= nb_meta || tr.begin_c + 3 <= nb_meta, "rctc");
			J40HIP_SHOULD(equal_sized(channel.data() + tr.begin_c, channel.data() + tr.begin_c + 3), "rtcd");
			break;
		}
		case Transform::PALETTE: {
			tr.begin_c = br.u32(0, 3, 8, 6, 72, 10, 1096, 13);
			tr.num_c = br.u32(1, 0, 3, 0, 4, 0, 1, 13);
			int32_t end_c = tr.begin_c + tr.num_c;
			tr.nb_colours = br.u32(0, 8, 256, 10, 1280, 12, 5376, 16);
			tr.nb_deltas = br.u32(0, 0, 1, 8, 257, 10, 1281, 16);
			tr.d_pred = (int32_t) br.u(4);
			J40HIP_SHOULD(tr.d_pred < 14, "palp");
			J40HIP_SHOULD(end_c <= num_channels, "palc");
			if (tr.begin_c < nb_meta) { J40HIP_SHOULD(end_c <= nb_meta, "palc"); nb_meta += 2 - tr.num_c; }
			else nb_meta += 1;
			J40HIP_SHOULD(equal_sized(channel.data() + tr.begin_c, channel.data() + end_c), "pald");
			Plane input = channel[(size_t) tr.begin_c], pal;
			pal.width = tr.nb_colours; pal.height = tr.num_c; pal.hshift = 0; pal.vshift = -1;
			std::vector<Plane> next;
			next.push_back(pal);
			next.insert(next.end(), channel.begin(), channel.begin() + tr.begin_c);
			next.push_back(input);
			next.insert(next.end(), channel.begin() + end_c, channel.end());
			channel.swap(next);
			break;
		}
		case Transform::SQUEEZE: {
			// The reference reads the parameters and then stops with "TODO" (j40.h:3794-3812); here the transform is carried
			// out (ISO 18181-1; device/squeeze_dev.h). Each step becomes a transform of its own, like the reference stores them.
			const int32_t num_sq = br.u32(0, 0, 1, 4, 9, 6, 41, 8);
			std::vector<Transform> steps;
			if (num_sq == 0) default_squeeze_steps(channel, nb_meta, &steps);
			else for (int32_t j = 0; j < num_sq; ++j) {
				Transform st; st.kind = Transform::SQUEEZE;
				st.horizontal = br.u(1) != 0; st.in_place = br.u(1) != 0;
				st.begin_c = br.u32(0, 3, 8, 6, 72, 10, 1096, 13); st.num_c = br.u32(1, 0, 2, 0, 3, 0, 4, 4);
				steps.push_back(st);
			}
			for (const Transform &st : steps) {
				const int32_t nc = (int32_t) channel.size(), end_c = st.begin_c + st.num_c;
				J40HIP_SHOULD(st.num_c >= 1 && end_c <= nc, "sqzc");
				if (st.begin_c < nb_meta) J40HIP_SHOULD(st.in_place && end_c <= nb_meta, "sqzc");   // meta channels: in place, not across the border
				J40HIP_SHOULD(nc + st.num_c <= 256, "xlim");
				// (libjxl's MetaSqueeze checks: nothing is squeezed past a shift of 30 or down from an empty channel -- the int8 shifts
				// would wrap into "meta channel" and the section layout shifts by them)
				for (int32_t c = st.begin_c; c < end_c; ++c) {
					const Plane &ch = channel[(size_t) c];
					J40HIP_SHOULD(ch.hshift <= 30 && ch.vshift <= 30 && ch.width > 0 && ch.height > 0, "sqzc");
				}
				apply_squeeze_meta(st, &channel, &nb_meta);
				m->transforms.push_back(st);
			}
			continue;   // (already recorded, step by step)
		}
		default: J40HIP_RAISE("xfm?");
		}
		m->transforms.push_back(tr);
	}
	J40HIP_SHOULD((int32_t) channel.size() <= 256, "xlim");
	if (m->use_global_tree) {
		m->tree = global_tree; m->codespec = global_codespec;
	} else {
		int64_t max_tree_size = 1024;
		for (const Plane &c : channel) max_tree_size += (int64_t) c.width * c.height;
		if (max_tree_size > (1 << 20)) max_tree_size = 1 << 20;
		read_tree(br, (int32_t) max_tree_size, 64, &m->own_tree, &m->own_codespec);
		m->tree = &m->own_tree; m->codespec = &m->own_codespec;
	}
	m->nb_meta_channels = nb_meta;
	m->dist_mult = 0;
	for (size_t i = (size_t) nb_meta; i < channel.size(); ++i) m->dist_mult = std::max(m->dist_mult, channel[i].width);
	m->dist_mult = std::min(m->dist_mult, 1 << 21);
}

void allocate_modular(Modular *m) { for (Plane &c : m->channel) c.allocate(); }

// ------------------------------------------------------------------------------------------------
// prediction (16-bit buffers: samples int16, intermediates int32; j40.h:3938-4125 with P = 16)

namespace {

struct Neigh { int32_t w, n, nw, ne, nn, nee, ww, nww; };

inline Neigh neighbours(const Plane &c, int32_t x, int32_t y) {  // j40.h:3965
	const int16_t *px = c.row(y); const int32_t stride = c.width, width = c.width;
	Neigh p;
	p.w = x > 0 ? px[x - 1] : y > 0 ? px[x - stride] : 0;
	p.n = y > 0 ? px[x - stride] : p.w;
	p.nw = x > 0 && y > 0 ? px[(x - 1) - stride] : p.w;
	p.ne = x + 1 < width && y > 0 ? px[(x + 1) - stride] : p.n;
	p.nn = y > 1 ? px[x - 2 * stride] : p.n;
	p.nee = x + 2 < width && y > 0 ? px[(x + 2) - stride] : p.ne;
	p.ww = x > 1 ? px[x - 2] : p.w;
	p.nww = x > 1 && y > 0 ? px[(x - 2) - stride] : p.ww;
	return p;
}

inline int32_t clamped_gradient(int32_t w, int32_t n, int32_t nw) {
	int32_t lo = std::min(w, n), hi = std::max(w, n);
	return std::min(std::max(lo, w + n - nw), hi);
}

inline int32_t div24(int32_t i) { return (int32_t) (((int64_t) 1 << 24) / (i + 1)); }  // J40__24DIVP1, j40.h:3905

struct WeightedPredictor {  // j40.h:3997-4119
	bool on = false;
	int32_t width = 0;
	WPParams params;
	std::vector<int32_t> errors;  // [2 * width][5]
	int32_t pred[5] = {0, 0, 0, 0, 0};
	int32_t trueerrw = 0, trueerrn = 0, trueerrnw = 0, trueerrne = 0;

	void init(const WPParams &p, int32_t w) { on = true; width = w; params = p; errors.assign((size_t) w * 10, 0); reset_scalars(); }
	void reset_scalars() { for (int i = 0; i < 5; ++i) pred[i] = 0; trueerrw = trueerrn = trueerrnw = trueerrne = 0; }
	void reset() { if (on) std::fill(errors.begin(), errors.end(), 0); reset_scalars(); }

	void before_predict(int32_t x, int32_t y, const Neigh &p) {
		if (!on) return;
		static const int32_t ZERO[5] = {0, 0, 0, 0, 0};
		const int32_t *err = errors.data() + (size_t) ((y & 1) ? width : 0) * 5;
		const int32_t *nerr = errors.data() + (size_t) ((y & 1) ? 0 : width) * 5;
		const int32_t *errw = x > 0 ? err + (x - 1) * 5 : ZERO;
		const int32_t *errn = y > 0 ? nerr + x * 5 : ZERO;
		const int32_t *errnw = x > 0 && y > 0 ? nerr + (x - 1) * 5 : errn;
		const int32_t *errne = x + 1 < width && y > 0 ? nerr + (x + 1) * 5 : errn;
		const int32_t *errww = x > 1 ? err + (x - 2) * 5 : ZERO;
		const int32_t *errw2 = x + 1 < width ? ZERO : errw;
		trueerrw = x > 0 ? err[(x - 1) * 5 + 4] : 0;
		trueerrn = y > 0 ? nerr[x * 5 + 4] : 0;
		trueerrnw = x > 0 && y > 0 ? nerr[(x - 1) * 5 + 4] : trueerrn;
		trueerrne = x + 1 < width && y > 0 ? nerr[(x + 1) * 5 + 4] : trueerrn;
		pred[0] = (p.w + p.ne - p.n) * 8;
		pred[1] = p.n * 8 - (((trueerrw + trueerrn + trueerrne) * params.p1) >> 5);
		pred[2] = p.w * 8 - (((trueerrw + trueerrn + trueerrnw) * params.p2) >> 5);
		pred[3] = p.n * 8 - ((trueerrnw * params.p3[0] + trueerrn * params.p3[1] + trueerrne * params.p3[2] +
			(p.nn - p.n) * 8 * params.p3[3] + (p.nw - p.w) * 8 * params.p3[4]) >> 5);
		int32_t w[4], wsum = 0, sum = 0;
		for (int i = 0; i < 4; ++i) {
			int32_t errsum = errn[i] + errw[i] + errnw[i] + errww[i] + errne[i] + errw2[i];
			int32_t shift = std::max(floor_lg32((uint32_t) errsum + 1) - 5, 0);
			w[i] = (int32_t) (4 + ((int64_t) params.w[i] * div24(errsum >> shift) >> shift));
		}
		int32_t logw = floor_lg32((uint32_t) (w[0] + w[1] + w[2] + w[3])) - 4;
		for (int i = 0; i < 4; ++i) { w[i] >>= logw; wsum += w[i]; sum += pred[i] * w[i]; }
		pred[4] = (int32_t) (((int64_t) sum + (wsum >> 1) - 1) * div24(wsum - 1) >> 24);
		if (((trueerrn ^ trueerrw) | (trueerrn ^ trueerrnw)) <= 0) {
			int32_t lo = std::min(p.w, std::min(p.n, p.ne)) * 8, hi = std::max(p.w, std::max(p.n, p.ne)) * 8;
			pred[4] = std::min(std::max(lo, pred[4]), hi);
		}
	}
	void after_predict(int32_t x, int32_t y, int32_t val) {
		if (!on) return;
		int32_t *e = errors.data() + ((size_t) ((y & 1) ? width : 0) + (size_t) x) * 5;
		for (int i = 0; i < 4; ++i) { int32_t d = pred[i] - val * 8; e[i] = ((d < 0 ? -d : d) + 3) >> 3; }
		e[4] = pred[4] - val * 8;
	}
};

inline int32_t predict(int32_t predictor, const WeightedPredictor &wp, const Neigh &p) {  // j40.h:4080
	switch (predictor) {
	case 0: return 0;
	case 1: return p.w;
	case 2: return p.n;
	case 3: return (p.w + p.n) / 2;
	case 4: return std::abs(p.n - p.nw) < std::abs(p.w - p.nw) ? p.w : p.n;
	case 5: return clamped_gradient(p.w, p.n, p.nw);
	case 6: return (wp.pred[4] + 3) >> 3;
	case 7: return p.ne;
	case 8: return p.nw;
	case 9: return p.ww;
	case 10: return (p.w + p.nw) / 2;
	case 11: return (p.n + p.nw) / 2;
	case 12: return (p.n + p.ne) / 2;
	case 13: return (6 * p.n - 2 * p.nn + 7 * p.w + p.ww + p.nee + 3 * p.ne + 8) / 16;
	default: J40HIP_RAISE("pred");
	}
}

} // namespace

// ---- the fast form of the per-pixel loop (same results as the general loop below) ----
// LfGroup sections are Modular streams of a quarter of a million samples each and the host parses them for every frame: with the
// pipeline bound by host CPU time their decode loop is worth three things the reference does not do:
//   * the MA tree is specialised per (channel, stream): nodes that test the channel index or the stream index (properties 0, 1)
//     are decided once, not once per sample (j40.h:4170-4171 evaluates them per sample) -- the LF trees encoders write split on
//     exactly those first, so what is left is often a single leaf;
//   * rANS symbols are decoded inline (no LZ77 state, no call per sample), the bit reader is topped up eight bytes at a time;
//   * interior samples read their neighbours straight from the rows; the edge rules (j40.h:3965-3990) run for edge samples only.
namespace {

// copies `tree` with every property-0 / property-1 node replaced by the child it leads to for this channel and stream
void specialise_tree(const TreeNode *tree, int32_t at, int32_t cidx, int32_t sidx, std::vector<TreeNode> *out) {
	for (;;) {   // static nodes: follow
		const TreeNode &n = tree[at];
		if (n.prop == 0) { at += cidx > n.value ? n.a : n.b; continue; }
		if (n.prop == 1) { at += sidx > n.value ? n.a : n.b; continue; }
		break;
	}
	const TreeNode &n = tree[at];
	const size_t me = out->size();
	out->push_back(n);
	if (n.prop < 0) return;
	(*out)[me].a = (int32_t) (out->size() - me);
	specialise_tree(tree, at + n.a, cidx, sidx, out);
	(*out)[me].b = (int32_t) (out->size() - me);
	specialise_tree(tree, at + n.b, cidx, sidx, out);
}

inline void refill8(BitReader &br) {   // >= 57 bits afterwards while eight bytes remain; else the byte-wise refill
	if (br.end - br.ptr >= 8) {
		uint64_t w;
		memcpy(&w, br.ptr, 8);
		const int take = (63 - br.nbits) >> 3;
		br.bits |= w << br.nbits;
		br.ptr += take; br.nbits += take * 8;
		if (br.nbits < 64) br.bits &= ((uint64_t) 1 << br.nbits) - 1;
	} else br.refill();
}
inline uint32_t take_bits(BitReader &br, int n) {   // == BitReader::u
	if (br.nbits < n) { refill8(br); if (br.nbits < n) J40HIP_RAISE("shrt"); }
	const uint32_t v = (uint32_t) (br.bits & (((uint64_t) 1 << n) - 1));
	br.bits >>= n; br.nbits -= n;
	return v;
}
// what one rANS symbol + hybrid integer of a cluster needs, as plain ints (HybridCfg's fields are bytes: behind a reference they
// are reloaded and sign-extended for every symbol, because a byte may alias anything the loop stores)
struct RansCluster {
	const AnsEntry *alias;
	int32_t split_exp, split, in_token, msb, lsb, max_token;
	explicit RansCluster(const Cluster &cl) : alias(cl.alias.data()), split_exp(cl.cfg.split_exp), split(1 << cl.cfg.split_exp), in_token(cl.cfg.msb_in_token + cl.cfg.lsb_in_token),
		msb(cl.cfg.msb_in_token), lsb(cl.cfg.lsb_in_token), max_token(cl.cfg.max_token) {}
};
// one rANS symbol + hybrid integer (ans_decode + hybrid_int of entropy.cpp; j40.h:2441, 2313)
inline int32_t rans_value(BitReader &br, uint32_t &state, int32_t log_bucket, const RansCluster &cl) {
	if (state == 0) { state = take_bits(br, 16); state |= take_bits(br, 16) << 16; }
	const uint32_t idx = state & 0xfff, i = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
	const AnsEntry e = cl.alias[i];
	const bool aliased = pos >= (uint32_t) (e & 0xff);
	const int32_t token = (int32_t) (aliased ? (uint32_t) (e >> 20) & 0xff : i);
	const uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
	state = d * (state >> 12) + offset + pos;
	if (state < (1u << 16)) state = (state << 16) | take_bits(br, 16);
	if (token < cl.split) return token;
	J40HIP_SHOULD(token <= cl.max_token, "iovf");
	const int32_t midbits = cl.split_exp - cl.in_token + ((token - cl.split) >> cl.in_token);
	const int32_t mid = (int32_t) take_bits(br, midbits);
	const int32_t top = 1 << cl.msb;
	const int32_t lo = token & ((1 << cl.lsb) - 1), hi = (token >> cl.lsb) & (top - 1);
	return ((top | hi) << (midbits + cl.lsb)) | ((mid << cl.lsb) | lo);
}

// channels whose specialised tree is a single leaf (the usual case for LF chroma and for the HF-metadata channels): no tree walk,
// and with the predictor a compile-time constant only the neighbours it uses are fetched
template <int PRED>
static void decode_single_leaf(BitReader &br_, Plane &c, uint32_t &state_, int32_t log_bucket, const Cluster &cluster, int32_t offset, int32_t multiplier) {
	const int32_t w = c.width, h = c.height;
	const RansCluster cl(cluster);
	// the reader and the rANS state live in locals for the duration (the stores into the plane cannot be proven not to alias them
	// otherwise, and every symbol would reload and store them); written back at the end -- after an error nobody reads them
	BitReader br = br_;
	uint32_t state = state_;
	// Predictors that look at W, N and NW only: away from the left edge and the first row the three are carried from sample to
	// sample (one load per sample: N), and none of the edge rules of j40.h:3965-3990 can apply
	constexpr bool WNNW = PRED == 0 || PRED == 1 || PRED == 2 || PRED == 3 || PRED == 4 || PRED == 5 || PRED == 8 || PRED == 10 || PRED == 11;
	for (int32_t y = 0; y < h; ++y) {
		int16_t *row = c.row(y);
		const int16_t *up = y > 0 ? c.row(y - 1) : row, *up2 = y > 1 ? c.row(y - 2) : up;
		int32_t x = 0;
		const int32_t edge_until = WNNW && y > 0 ? 1 : w;
		for (; x < edge_until; ++x) {
			// the neighbours with their fallbacks (j40.h:3965-3990); the ones PRED does not use fold away
			const int32_t pw = x > 0 ? row[x - 1] : y > 0 ? up[x] : 0;
			const int32_t pn = y > 0 ? up[x] : pw;
			const int32_t pnw = x > 0 && y > 0 ? up[x - 1] : pw;
			const int32_t pne = x + 1 < w && y > 0 ? up[x + 1] : pn;
			const int32_t pnn = y > 1 ? up2[x] : pn;
			const int32_t pnee = x + 2 < w && y > 0 ? up[x + 2] : pne;
			const int32_t pww = x > 1 ? row[x - 2] : pw;
			int32_t pred;
			switch (PRED) {
			case 0: pred = 0; break;
			case 1: pred = pw; break;
			case 2: pred = pn; break;
			case 3: pred = (pw + pn) / 2; break;
			case 4: pred = std::abs(pn - pnw) < std::abs(pw - pnw) ? pw : pn; break;
			case 5: pred = clamped_gradient(pw, pn, pnw); break;
			case 7: pred = pne; break;
			case 8: pred = pnw; break;
			case 9: pred = pww; break;
			case 10: pred = (pw + pnw) / 2; break;
			case 11: pred = (pn + pnw) / 2; break;
			case 12: pred = (pn + pne) / 2; break;
			default: pred = (6 * pn - 2 * pnn + 7 * pw + pww + pnee + 3 * pne + 8) / 16; break;   // 13
			}
			const int32_t v = unpack_signed(rans_value(br, state, log_bucket, cl)) * multiplier + offset + pred;
			J40HIP_SHOULD(-32768 <= v && v <= 32767, "povf");
			row[x] = (int16_t) v;
		}
		if (x < w) {   // (WNNW, y > 0, x = 1)
			int32_t pw = row[0], pnw = up[0];
			for (; x < w; ++x) {
				const int32_t pn = up[x];
				int32_t pred;
				switch (PRED) {
				case 0: pred = 0; break;
				case 1: pred = pw; break;
				case 2: pred = pn; break;
				case 3: pred = (pw + pn) / 2; break;
				case 4: pred = std::abs(pn - pnw) < std::abs(pw - pnw) ? pw : pn; break;
				case 5: pred = clamped_gradient(pw, pn, pnw); break;
				case 8: pred = pnw; break;
				case 10: pred = (pw + pnw) / 2; break;
				default: pred = (pn + pnw) / 2; break;   // 11
				}
				const int32_t v = unpack_signed(rans_value(br, state, log_bucket, cl)) * multiplier + offset + pred;
				J40HIP_SHOULD(-32768 <= v && v <= 32767, "povf");
				row[x] = (int16_t) v;
				pw = v; pnw = pn;
			}
		}
	}
	br_ = br; state_ = state;
}

// the fast loop: rANS without LZ77, no weighted predictor, no previous-channel properties in the specialised tree
bool decode_channel_fast(BitReader &br, Modular &m, CodeState &code, int32_t cidx, int64_t sidx) {
	const CodeSpec *spec = code.spec;
	if (spec->use_prefix_code || spec->lz77_enabled || code.num_to_copy > 0) return false;
	std::vector<TreeNode> tree;
	specialise_tree(m.tree->data(), 0, cidx, (int32_t) sidx, &tree);
	for (const TreeNode &n : tree) if (n.prop >= 15 || n.prop == -1 - 6 || n.prop < -1 - 13) return false;
	Plane &c = m.channel[(size_t) cidx];
	const int32_t w = c.width, h = c.height, log_bucket = 12 - spec->log_alpha_size;
	const TreeNode *root = tree.data();
	const bool single = root->prop < 0;
	const uint8_t *cluster_map = spec->cluster_map.data();
	const Cluster *clusters = spec->clusters.data();
	uint32_t state = code.ans_state;
	struct Restore { CodeState &code; uint32_t &state; ~Restore() { code.ans_state = state; } } restore{code, state};   // (also when an error unwinds)
	if (single) {
		const Cluster &cl = clusters[cluster_map[(size_t) root->value]];
		switch (-1 - root->prop) {
		case 0: decode_single_leaf<0>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 1: decode_single_leaf<1>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 2: decode_single_leaf<2>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 3: decode_single_leaf<3>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 4: decode_single_leaf<4>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 5: decode_single_leaf<5>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 7: decode_single_leaf<7>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 8: decode_single_leaf<8>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 9: decode_single_leaf<9>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 10: decode_single_leaf<10>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 11: decode_single_leaf<11>(br, c, state, log_bucket, cl, root->a, root->b); break;
		case 12: decode_single_leaf<12>(br, c, state, log_bucket, cl, root->a, root->b); break;
		default: decode_single_leaf<13>(br, c, state, log_bucket, cl, root->a, root->b); break;
		}
		return true;
	}
	std::vector<RansCluster> rcl;
	for (const Cluster &cl : spec->clusters) rcl.emplace_back(cl);
	const RansCluster *rclusters = rcl.data();
	BitReader lbr = br;   // (in locals for the duration, see decode_single_leaf)
	struct PutBack { BitReader &to, &from; ~PutBack() { to = from; } } put_back{br, lbr};
	// Trees that look at W, N and NW only -- properties y, x, |N|, |W|, N, W, W + N - NW, W - NW, NW - N; predictors of the same kind
	// (the LF trees of VarDCT encoders: gradient predictor, contexts from the local gradient): like decode_single_leaf, away from the
	// left edge and the first row the three neighbours are carried along, one load per sample, and no edge rule can apply
	bool wnnw = true;
	for (const TreeNode &n : tree) {
		if (n.prop >= 0) wnnw = wnnw && (n.prop <= 7 || (n.prop >= 9 && n.prop <= 11));
		else { const int32_t pr = -1 - n.prop; wnnw = wnnw && (pr <= 5 || pr == 8 || pr == 10 || pr == 11); }
	}
	if (wnnw) {
		for (int32_t y = 0; y < h; ++y) {
			int16_t *row = c.row(y);
			const int16_t *up = y > 0 ? c.row(y - 1) : row;
			int32_t pw = 0, pnw = 0;
			for (int32_t x = 0; x < w; ++x) {
				int32_t pn;
				if (x == 0 || y == 0) {   // the edge rules (j40.h:3965-3990)
					pw = x > 0 ? row[x - 1] : y > 0 ? up[x] : 0;
					pn = y > 0 ? up[x] : pw;
					pnw = x > 0 && y > 0 ? up[x - 1] : pw;
				} else pn = up[x];
				const TreeNode *n = root;
				while (n->prop >= 0) {
					int32_t val;
					switch (n->prop) {
					case 2: val = y; break;
					case 3: val = x; break;
					default: val = neighbour_property(n->prop, x, pw, pn, pnw, 0, 0, 0, 0); break;   // 4..7, 9..11 (the path's precondition): N, W, NW suffice
					}
					n += val > n->value ? n->a : n->b;
				}
				int32_t v = rans_value(lbr, state, log_bucket, rclusters[cluster_map[(size_t) n->value]]);
				v = unpack_signed(v) * n->b + n->a;
				switch (-1 - n->prop) {
				case 0: break;
				case 1: v += pw; break;
				case 2: v += pn; break;
				case 3: v += (pw + pn) / 2; break;
				case 4: v += std::abs(pn - pnw) < std::abs(pw - pnw) ? pw : pn; break;
				case 5: v += clamped_gradient(pw, pn, pnw); break;
				case 8: v += pnw; break;
				case 10: v += (pw + pnw) / 2; break;
				default: v += (pn + pnw) / 2; break;   // 11
				}
				J40HIP_SHOULD(-32768 <= v && v <= 32767, "povf");
				row[x] = (int16_t) v;
				pw = v; pnw = pn;
			}
		}
		return true;
	}
	for (int32_t y = 0; y < h; ++y) {
		int16_t *row = c.row(y);
		const int16_t *up = y > 0 ? c.row(y - 1) : row, *up2 = y > 1 ? c.row(y - 2) : up;
		for (int32_t x = 0; x < w; ++x) {
			Neigh p;
			if (x >= 2 && x + 2 < w) {   // away from the left and right edges: the fallbacks depend on the row only (j40.h:3965-3990)
				p.w = row[x - 1]; p.ww = row[x - 2];
				if (y >= 2) { p.n = up[x]; p.nw = up[x - 1]; p.ne = up[x + 1]; p.nn = up2[x]; p.nee = up[x + 2]; p.nww = up[x - 2]; }
				else if (y == 1) { p.n = up[x]; p.nw = up[x - 1]; p.ne = up[x + 1]; p.nn = p.n; p.nee = up[x + 2]; p.nww = up[x - 2]; }
				else { p.n = p.nw = p.ne = p.nn = p.nee = p.w; p.nww = p.ww; }   // first row (the varblock-info channel is two rows of thousands of samples)
			} else p = neighbours(c, x, y);
			const TreeNode *n = root;
			if (!single) while (n->prop >= 0) {
				int32_t val;
				switch (n->prop) {
				case 2: val = y; break;
				case 3: val = x; break;
				default: val = neighbour_property(n->prop, x, p.w, p.n, p.nw, p.ne, p.nn, p.ww, p.nww); break;   // 4..14 (device/props_dev.h)
				}
				n += val > n->value ? n->a : n->b;
			}
			int32_t v = rans_value(lbr, state, log_bucket, rclusters[cluster_map[(size_t) n->value]]);
			v = unpack_signed(v) * n->b + n->a;
			int32_t pred;
			switch (-1 - n->prop) {
			case 0: pred = 0; break;
			case 1: pred = p.w; break;
			case 2: pred = p.n; break;
			case 3: pred = (p.w + p.n) / 2; break;
			case 4: pred = std::abs(p.n - p.nw) < std::abs(p.w - p.nw) ? p.w : p.n; break;
			case 5: pred = clamped_gradient(p.w, p.n, p.nw); break;
			case 7: pred = p.ne; break;
			case 8: pred = p.nw; break;
			case 9: pred = p.ww; break;
			case 10: pred = (p.w + p.nw) / 2; break;
			case 11: pred = (p.n + p.nw) / 2; break;
			case 12: pred = (p.n + p.ne) / 2; break;
			default: pred = (6 * p.n - 2 * p.nn + 7 * p.w + p.ww + p.nee + 3 * p.ne + 8) / 16; break;   // 13
			}
			v += pred;
			J40HIP_SHOULD(-32768 <= v && v <= 32767, "povf");
			row[x] = (int16_t) v;
		}
	}
	return true;
}


} // namespace

void decode_modular_channel(BitReader &br, Modular &m, CodeState &code, int32_t cidx, int64_t sidx) {  // j40.h:4127
	if (!m.channel[(size_t) cidx].empty() && decode_channel_fast(br, m, code, cidx, sidx)) return;
	Plane &c = m.channel[(size_t) cidx];
	if (c.empty()) return;
	const TreeNode *tree = m.tree->data();
	WeightedPredictor wp;
	if (tree_uses_wp(*m.tree)) wp.init(m.wp, c.width);
	std::vector<int32_t> refc;  // earlier channels with identical geometry, nearest first (j40.h:4156-4165)
	for (int32_t i = cidx - 1; i >= 0; --i) {
		const Plane &r = m.channel[(size_t) i];
		if (c.width != r.width || c.height != r.height || c.hshift != r.hshift || c.vshift != r.vshift) continue;
		refc.push_back(i);
	}
	for (int32_t y = 0; y < c.height; ++y) {
		int16_t *out = c.row(y);
		for (int32_t x = 0; x < c.width; ++x) {
			const TreeNode *n = tree;
			Neigh p = neighbours(c, x, y);
			wp.before_predict(x, y, p);
			while (n->prop >= 0) {
				int32_t val;
				switch (n->prop) {
				case 0: val = cidx; break;
				case 1: val = (int32_t) sidx; break;
				case 2: val = y; break;
				case 3: val = x; break;
				case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14:
					val = neighbour_property(n->prop, x, p.w, p.n, p.nw, p.ne, p.nn, p.ww, p.nww); break;   // (device/props_dev.h)
				case 15:
					val = wp.trueerrw;
					if (std::abs(val) < std::abs(wp.trueerrn)) val = wp.trueerrn;
					if (std::abs(val) < std::abs(wp.trueerrnw)) val = wp.trueerrnw;
					if (std::abs(val) < std::abs(wp.trueerrne)) val = wp.trueerrne;
					break;
				default: {
					int32_t r = (n->prop - 16) / 4;
					J40HIP_SHOULD(r < (int32_t) refc.size(), "trec");
					const Plane &rc = m.channel[(size_t) refc[(size_t) r]];
					val = rc.row(y)[x];
					if (n->prop & 2) {
						int32_t rw = x > 0 ? rc.row(y)[x - 1] : 0;
						int32_t rn = y > 0 ? rc.row(y - 1)[x] : rw;
						int32_t rnw = x > 0 && y > 0 ? rc.row(y - 1)[x - 1] : rw;
						val -= clamped_gradient(rw, rn, rnw);
					}
					if (n->prop & 1) val = std::abs(val);
				} }
				n += val > n->value ? n->a : n->b;
			}
			int32_t v = decode_symbol(br, code, n->value, m.dist_mult);
			v = unpack_signed(v) * n->b + n->a;
			v += predict(-1 - n->prop, wp, p);
			J40HIP_SHOULD(-32768 <= v && v <= 32767, "povf");
			out[x] = (int16_t) v;
			wp.after_predict(x, y, v);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// inverse transforms

static void inverse_rct(Modular &m, const Transform &tr) {  // j40.h:4318
	static const uint8_t PERM[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
	Plane c[3];
	for (int i = 0; i < 3; ++i) c[i] = std::move(m.channel[(size_t) (tr.begin_c + i)]);
	if (!c[0].empty()) {
		const size_t n = c[0].px.size();
		int16_t *p0 = c[0].px.data(), *p1 = c[1].px.data(), *p2 = c[2].px.data();
		switch (tr.rct_type % 7) {
		case 0: break;
		case 1: for (size_t i = 0; i < n; ++i) p2[i] = (int16_t) (p2[i] + p0[i]); break;
		case 2: for (size_t i = 0; i < n; ++i) p2[i] = (int16_t) (p1[i] + p0[i]); break;
		case 3: for (size_t i = 0; i < n; ++i) { p1[i] = (int16_t) (p1[i] + p0[i]); p2[i] = (int16_t) (p2[i] + p0[i]); } break;
		case 4: for (size_t i = 0; i < n; ++i) { int16_t a = p0[i], b = p2[i]; p1[i] = (int16_t) (p1[i] + (int16_t) (a / 2 + b / 2 + (a & b & 1))); } break;
		case 5: for (size_t i = 0; i < n; ++i) { p1[i] = (int16_t) ((int32_t) p1[i] + p0[i] + (p2[i] >> 1)); p2[i] = (int16_t) (p2[i] + p0[i]); } break;
		case 6:
			for (size_t i = 0; i < n; ++i) {
				int32_t tmp = (int32_t) p0[i] - ((int32_t) p2[i] >> 1);
				int32_t q1 = (int32_t) p2[i] + tmp;
				int32_t q2 = tmp - ((int32_t) p1[i] >> 1);
				p0[i] = (int16_t) (q2 + p1[i]); p1[i] = (int16_t) q1; p2[i] = (int16_t) q2;
			}
			break;
		}
	}
	for (int i = 0; i < 3; ++i) m.channel[(size_t) (tr.begin_c + PERM[tr.rct_type / 7][i])] = std::move(c[i]);
}

// the spec's 72 palette delta triples; entry 2k is triple k, entry 2k + 1 its negation (j40.h:4275)
static const int16_t PALETTE_DELTA_BASE[72][3] = {
	{0, 0, 0}, {4, 4, 4}, {11, 0, 0}, {0, 0, -13}, {0, -12, 0}, {-10, -10, -10}, {-18, -18, -18}, {-27, -27, -27},
	{-18, -18, 0}, {0, 0, -32}, {-32, 0, 0}, {-37, -37, -37}, {0, -32, -32}, {24, 24, 45}, {50, 50, 50}, {-45, -24, -24},
	{-24, -45, -45}, {0, -24, -24}, {-34, -34, 0}, {-24, 0, -24}, {-45, -45, -24}, {64, 64, 64}, {-32, 0, -32}, {0, -32, 0},
	{-32, 0, 32}, {-24, -45, -24}, {45, 24, 45}, {24, -24, -45}, {-45, -24, 24}, {80, 80, 80}, {64, 0, 0}, {0, 0, -64},
	{0, -64, -64}, {-24, -24, 45}, {96, 96, 96}, {64, 64, 0}, {45, -24, -24}, {34, -34, 0}, {112, 112, 112}, {24, -45, -45},
	{45, 45, -24}, {0, -32, 32}, {24, -24, 45}, {0, 96, 96}, {45, -24, 24}, {24, -45, -24}, {-24, -45, 24}, {0, -64, 0},
	{96, 0, 0}, {128, 128, 128}, {64, 0, 64}, {144, 144, 144}, {96, 96, 0}, {-36, -36, 36}, {45, -24, -45}, {45, -45, -24},
	{0, 0, -96}, {0, 128, 128}, {0, 96, 0}, {45, 24, -45}, {-128, 0, 0}, {24, -45, 24}, {-45, 24, -45}, {64, 0, -64},
	{64, -64, -64}, {96, 0, 96}, {45, -45, 24}, {24, 45, -45}, {64, 64, -64}, {128, 128, 0}, {0, 0, -128}, {-24, 45, -45},
};
int16_t palette_delta(int32_t entry, int32_t channel) {
	int16_t v = PALETTE_DELTA_BASE[entry >> 1][channel];
	return (entry & 1) ? (int16_t) -v : v;
}

static void inverse_palette(Modular &m, const Transform &tr) {  // j40.h:4402
	const int32_t first = tr.begin_c + 1, bpp = m.bpp;
	const int32_t width = m.channel[(size_t) first].width, height = m.channel[(size_t) first].height;
	const bool use_pred = tr.nb_deltas > 0, use_wp = use_pred && tr.d_pred == 6;
	const bool index_empty = m.channel[(size_t) first].empty();
	// make room: the index channel ends up as the last restored colour channel
	for (int32_t i = 0; i < tr.num_c - 1; ++i) {
		Plane p; p.width = width; p.height = height;
		if (!index_empty) p.allocate(); else { p.width = p.height = 0; }
		m.channel.insert(m.channel.begin() + first, std::move(p));
	}
	const int32_t last = tr.begin_c + tr.num_c;
	WeightedPredictor wp;
	if (use_wp) wp.init(m.wp, width);
	if (!index_empty) for (int32_t i = 0; i < tr.num_c; ++i) {
		const int16_t *palp = tr.nb_colours > 0 ? m.channel[0].row(i) : nullptr;
		Plane &c = m.channel[(size_t) (first + i)];
		Plane &idxc = m.channel[(size_t) last];
		for (int32_t y = 0; y < height; ++y) {
			const int16_t *idxline = idxc.row(y);
			int16_t *line = c.row(y);
			for (int32_t x = 0; x < width; ++x) {
				int16_t idx = idxline[x], val;
				const bool is_delta = idx < tr.nb_deltas;
				if (idx < 0) {
					if (i < 3) {
						idx = (int16_t) (~idx % 143);
						val = palette_delta(idx + 1, i);
						if (bpp > 8) val = (int16_t) (val << (std::min(bpp, 24) - 8));
					} else val = 0;
				} else if (idx < tr.nb_colours) {
					val = palp[idx];
				} else {
					idx = (int16_t) (idx - tr.nb_colours);
					if (idx < 64) {
						val = (int16_t) ((i < 3 ? idx >> (2 * i) : 0) * (((int32_t) 1 << bpp) - 1) / 4 + ((int32_t) 1 << std::max(0, bpp - 3)));
					} else {
						val = (int16_t) (idx - 64);
						for (int32_t j = 0; j < i; ++j) val = (int16_t) (val / 5);
						val = (int16_t) ((val % 5) * ((1 << bpp) - 1) / 4);
					}
				}
				if (use_pred) {
					Neigh p = neighbours(c, x, y);
					wp.before_predict(x, y, p);
					if (is_delta) val = (int16_t) (val + predict(tr.d_pred, wp, p));
					wp.after_predict(x, y, val);
				}
				line[x] = val;
			}
		}
		wp.reset();
	}
	m.channel.erase(m.channel.begin());
}

// one Squeeze step undone: every squeezed channel is joined with its residual channel (device/squeeze_dev.h); the
// residual channels then leave the list. Empty planes (header-only images) only change their sizes.
static void inverse_squeeze(Modular &m, const Transform &tr) {
	const int32_t nc = (int32_t) m.channel.size(), end_c = tr.begin_c + tr.num_c;
	const int32_t offset = tr.in_place ? end_c : nc - tr.num_c;
	for (int32_t c = tr.begin_c; c < end_c; ++c) {
		Plane &avg = m.channel[(size_t) c]; const Plane &res = m.channel[(size_t) (offset + c - tr.begin_c)];
		Plane out;
		out.width = tr.horizontal ? avg.width + res.width : avg.width; out.height = tr.horizontal ? avg.height : avg.height + res.height;
		// (mirrors apply_squeeze_meta: only shifts that the forward step raised -- those of non-meta channels -- come down again)
		out.hshift = (int8_t) (avg.hshift - (tr.horizontal && avg.hshift > 0 ? 1 : 0)); out.vshift = (int8_t) (avg.vshift - (!tr.horizontal && avg.vshift > 0 ? 1 : 0));
		const bool allocated = avg.px.size() == (size_t) std::max(avg.width, 0) * (size_t) std::max(avg.height, 0) && res.px.size() == (size_t) std::max(res.width, 0) * (size_t) std::max(res.height, 0);
		if (allocated) {
			out.allocate();
			if (!out.empty()) {
				if (tr.horizontal) for (int32_t y = 0; y < out.height; ++y)
					unsqueeze_line(avg.row(y), 1, res.width > 0 ? res.row(y) : avg.row(y), 1, avg.width, res.width, out.row(y), 1);
				else for (int32_t x = 0; x < out.width; ++x)
					unsqueeze_line(avg.px.data() + x, avg.width, res.height > 0 ? res.px.data() + x : avg.px.data() + x, res.width, avg.height, res.height, out.px.data() + x, out.width);
			}
		}
		avg = std::move(out);
	}
	m.channel.erase(m.channel.begin() + offset, m.channel.begin() + offset + tr.num_c);
	if (tr.begin_c < m.nb_meta_channels) m.nb_meta_channels -= tr.num_c;
}

void apply_squeeze_meta(const Transform &tr, std::vector<Plane> *channel, int32_t *nb_meta) {
	const int32_t end_c = tr.begin_c + tr.num_c;
	const int32_t offset = tr.in_place ? end_c : (int32_t) channel->size();
	if (tr.begin_c < *nb_meta) *nb_meta += tr.num_c;
	for (int32_t c = tr.begin_c; c < end_c; ++c) {
		Plane &ch = (*channel)[(size_t) c];
		Plane res; res.hshift = ch.hshift; res.vshift = ch.vshift;
		if (tr.horizontal) {
			const int32_t w = ch.width; ch.width = (w + 1) / 2; res.width = w - ch.width; res.height = ch.height;
			if (ch.hshift >= 0) { ++ch.hshift; res.hshift = ch.hshift; }
		} else {
			const int32_t h = ch.height; ch.height = (h + 1) / 2; res.height = h - ch.height; res.width = ch.width;
			if (ch.vshift >= 0) { ++ch.vshift; res.vshift = ch.vshift; }
		}
		channel->insert(channel->begin() + offset + (c - tr.begin_c), res);
	}
}

void default_squeeze_steps(const std::vector<Plane> &channel, int32_t nb_meta, std::vector<Transform> *out) {
	const int32_t first = nb_meta, count = (int32_t) channel.size() - first;
	if (count <= 0) return;
	int32_t w = channel[(size_t) first].width, h = channel[(size_t) first].height;
	Transform st; st.kind = Transform::SQUEEZE;
	if (count > 2 && channel[(size_t) first + 1].width == w && channel[(size_t) first + 1].height == h) {
		// channels 1 and 2 are taken to be chroma: squeezed once in each direction first, residuals at the end of the list
		st.begin_c = first + 1; st.num_c = 2; st.in_place = false;
		st.horizontal = true; out->push_back(st);
		st.horizontal = false; out->push_back(st);
	}
	st.begin_c = first; st.num_c = count; st.in_place = true;
	if (h >= w && h > 8) { st.horizontal = false; out->push_back(st); h = (h + 1) / 2; }
	while (w > 8 || h > 8) {
		if (w > 8) { st.horizontal = true; out->push_back(st); w = (w + 1) / 2; }
		if (h > 8) { st.horizontal = false; out->push_back(st); h = (h + 1) / 2; }
	}
}

void inverse_transforms(Modular &m) {  // j40.h:4506
	if (m.channel.empty()) return;
	for (size_t i = m.transforms.size(); i-- > 0; ) {
		const Transform &tr = m.transforms[i];
		switch (tr.kind) {
		case Transform::RCT: inverse_rct(m, tr); break;
		case Transform::PALETTE: inverse_palette(m, tr); break;
		case Transform::SQUEEZE: inverse_squeeze(m, tr); break;
		default: J40HIP_RAISE("TODO");
		}
	}
}

void decode_modular_image(BitReader &br, const std::vector<TreeNode> *global_tree, const CodeSpec *global_codespec, int64_t sidx, Modular *m) {
	read_modular_header(br, global_tree, global_codespec, m);
	allocate_modular(m);
	CodeState code(m->codespec);
	for (int32_t c = 0; c < (int32_t) m->channel.size(); ++c) decode_modular_channel(br, *m, code, c, sidx);
	finish_code(br, code);
	inverse_transforms(*m);
}

} // namespace j40hip
