// j40_amd/csrc/device/modular_dev.h -- Modular hot path as per-lane device functions: per-pixel MA-tree
// walk + prediction + entropy decode of one group (K3), inverse RCT / Palette (K4), plane -> RGBA
// u8x4 pack (K5). Integer-exact restatement of j40__modular_channel16 (j40.h:4127-4240),
// j40__inverse_rct16 / j40__inverse_palette16 (j40.h:4318, 4402) and j40__render_to_u8x4_rgba (j40.h:7910).
#pragma once
#include "entropy_dev.h"
#include "props_dev.h"

namespace j40hip {

struct ModNeigh { int32_t w, n, nw, ne, nn, nee, ww, nww; };

// neighbours inside the group's own rectangle (x, y and width are group-local; j40.h:3965-3990)
template <bool UNI = false>
J40_DEV ModNeigh mod_neighbours(const int16_t *px /* row y, column 0 of the rectangle */, int32_t stride, int32_t width, int32_t x, int32_t y) {
	ModNeigh p;
	// every sample that exists is requested first (independent loads), the fallbacks are resolved afterwards
	const int32_t vw = x > 0 ? px[x - 1] : 0, vn = y > 0 ? px[x - stride] : 0, vnw = x > 0 && y > 0 ? px[(x - 1) - stride] : 0;
	const int32_t vne = x + 1 < width && y > 0 ? px[(x + 1) - stride] : 0, vnn = y > 1 ? px[x - 2 * stride] : 0;
	const int32_t vnee = x + 2 < width && y > 0 ? px[(x + 2) - stride] : 0, vww = x > 1 ? px[x - 2] : 0, vnww = x > 1 && y > 0 ? px[(x - 2) - stride] : 0;
	p.w = uni<UNI>(x > 0 ? vw : y > 0 ? vn : 0);
	p.n = uni<UNI>(y > 0 ? vn : p.w);
	p.nw = uni<UNI>(x > 0 && y > 0 ? vnw : p.w);
	p.ne = uni<UNI>(x + 1 < width && y > 0 ? vne : p.n);
	p.nn = uni<UNI>(y > 1 ? vnn : p.n);
	p.nee = uni<UNI>(x + 2 < width && y > 0 ? vnee : p.ne);
	p.ww = uni<UNI>(x > 1 ? vww : p.w);
	p.nww = uni<UNI>(x > 1 && y > 0 ? vnww : p.ww);
	return p;
}

J40_DEV int32_t mod_abs(int32_t v) { return v < 0 ? -v : v; }
J40_DEV int32_t mod_min(int32_t a, int32_t b) { return a < b ? a : b; }
J40_DEV int32_t mod_max(int32_t a, int32_t b) { return a > b ? a : b; }
J40_DEV int32_t mod_gradient(int32_t w, int32_t n, int32_t nw) { const int32_t lo = mod_min(w, n), hi = mod_max(w, n); return mod_min(mod_max(lo, w + n - nw), hi); }
J40_DEV int32_t mod_floor_lg(uint32_t x) { return 31 - __builtin_clz(x); }
J40_DEV int32_t mod_div24(int32_t i) { return (int32_t) (((int64_t) 1 << 24) / (i + 1)); }  // J40__24DIVP1, j40.h:3905

// Weighted ("self-correcting") predictor, ISO 18181-1 / j40.h:3997-4119. Four sub-predictors each propose a value; how much say
// each gets depends on how wrong it recently was around the sample (W, WW, N, NW, NE of the two most recent rows), and the blend's
// own signed error at W / N / NW / NE corrects three of the proposals. Everything is in 1/8 sample units. The arithmetic is
// normative (the result has to equal the reference's bit for bit); the decomposition below is this file's own.
// `errors` = [2 rows][width] cells of five ints: |error| of the four sub-predictors, then the blend's signed error; zero-initialised.
struct ModWP {
	int32_t on, width;
	int32_t p1, p2, p3[5], w[4];      // header parameters
	int32_t *errors;
	int32_t pred[5];                  // the four proposals and their blend
	int32_t blend_err_w, blend_err_n, blend_err_nw, blend_err_ne;   // the blend's signed error at the neighbours (property 15 looks at them too)
};

struct WpCell { int32_t v[5]; };
template <bool UNI> J40_DEV WpCell wp_cell(const int32_t *row, int32_t x, bool exists) {   // an absent neighbour counts as error-free
	WpCell c;
	for (int k = 0; k < 5; ++k) c.v[k] = uni<UNI>(exists ? row[x * 5 + k] : 0);
	return c;
}
// the say of a sub-predictor: 4 + w * 2^24 / (recent error + 1), the division done on the error's leading bits
J40_DEV int32_t wp_say(int32_t recent_error, int32_t w) {
	const int32_t drop = mod_max(mod_floor_lg((uint32_t) recent_error + 1) - 5, 0);
	return (int32_t) (4 + ((int64_t) w * mod_div24(recent_error >> drop) >> drop));
}

template <bool UNI = false>
J40_DEV void wp_before(ModWP &s, int32_t x, int32_t y, const ModNeigh &p) {
	if (!s.on) return;
	const int32_t *this_row = s.errors + (size_t) ((y & 1) ? s.width : 0) * 5, *row_above = s.errors + (size_t) ((y & 1) ? 0 : s.width) * 5;
	const bool has_left = x > 0, has_up = y > 0, has_right = x + 1 < s.width;
	const WpCell west = wp_cell<UNI>(this_row, x - 1, has_left), west2 = wp_cell<UNI>(this_row, x - 2, x > 1);
	const WpCell north = wp_cell<UNI>(row_above, x, has_up);
	WpCell north_west = wp_cell<UNI>(row_above, x - 1, has_left && has_up), north_east = wp_cell<UNI>(row_above, x + 1, has_right && has_up);
	if (!(has_left && has_up)) north_west = north;    // a missing diagonal neighbour is stood in for by the one above
	if (!(has_right && has_up)) north_east = north;
	s.blend_err_w = west.v[4]; s.blend_err_n = north.v[4]; s.blend_err_nw = north_west.v[4]; s.blend_err_ne = north_east.v[4];
	// the four proposals
	s.pred[0] = (p.w + p.ne - p.n) * 8;
	s.pred[1] = p.n * 8 - (((s.blend_err_w + s.blend_err_n + s.blend_err_ne) * s.p1) >> 5);
	s.pred[2] = p.w * 8 - (((s.blend_err_w + s.blend_err_n + s.blend_err_nw) * s.p2) >> 5);
	s.pred[3] = p.n * 8 - ((s.blend_err_nw * s.p3[0] + s.blend_err_n * s.p3[1] + s.blend_err_ne * s.p3[2] + (p.nn - p.n) * 8 * s.p3[3] + (p.nw - p.w) * 8 * s.p3[4]) >> 5);
	// their say: recent errors of the five neighbours, W once more in the last column (there is no NE to count)
	int32_t say[4], total = 0;
	for (int k = 0; k < 4; ++k) {
		const int32_t recent = north.v[k] + west.v[k] + north_west.v[k] + west2.v[k] + north_east.v[k] + (has_right ? 0 : west.v[k]);
		say[k] = wp_say(recent, s.w[k]);
		total += say[k];
	}
	const int32_t scale = mod_floor_lg((uint32_t) total) - 4;   // keeps the says within a few bits
	int32_t votes = 0, tally = 0;
	for (int k = 0; k < 4; ++k) { say[k] >>= scale; votes += say[k]; tally += s.pred[k] * say[k]; }
	s.pred[4] = (int32_t) (((int64_t) tally + (votes >> 1) - 1) * mod_div24(votes - 1) >> 24);
	// where the blend's errors at W, N and NW do not all point the same way, the blend stays within what W, N and NE span
	if (((s.blend_err_n ^ s.blend_err_w) | (s.blend_err_n ^ s.blend_err_nw)) <= 0) {
		const int32_t lo = mod_min(p.w, mod_min(p.n, p.ne)) * 8, hi = mod_max(p.w, mod_max(p.n, p.ne)) * 8;
		s.pred[4] = mod_min(mod_max(lo, s.pred[4]), hi);
	}
}

// the decoded sample is known: note how far off every proposal and the blend were
J40_DEV void wp_after(ModWP &s, int32_t x, int32_t y, int32_t val) {
	if (!s.on) return;
	int32_t *cell = s.errors + ((size_t) ((y & 1) ? s.width : 0) + (size_t) x) * 5;
	for (int k = 0; k < 4; ++k) cell[k] = (mod_abs(s.pred[k] - val * 8) + 3) >> 3;
	cell[4] = s.pred[4] - val * 8;
}

J40_DEV int32_t mod_predict(int32_t predictor, const ModWP &wp, const ModNeigh &p, uint32_t *err) {  // j40.h:4080
	switch (predictor) {
	case 0: return 0;
	case 1: return p.w;
	case 2: return p.n;
	case 3: return (p.w + p.n) / 2;
	case 4: return mod_abs(p.n - p.nw) < mod_abs(p.w - p.nw) ? p.w : p.n;
	case 5: return mod_gradient(p.w, p.n, p.nw);
	case 6: return (wp.pred[4] + 3) >> 3;
	case 7: return p.ne;
	case 8: return p.nw;
	case 9: return p.ww;
	case 10: return (p.w + p.nw) / 2;
	case 11: return (p.n + p.nw) / 2;
	case 12: return (p.n + p.ne) / 2;
	case 13: return (6 * p.n - 2 * p.nn + 7 * p.w + p.ww + p.nee + 3 * p.ne + 8) / 16;
	default: if (!*err) *err = ERR_PRED; return 0;
	}
}

// what the per-pixel loop keeps touching: the kernel stages these in LDS when they fit, else they point into HBM
struct ModTables {
	const DevTreeNode *tree;
	const DevCluster *clusters;      // of the global code spec
	const uint8_t *cluster_map;
	const uint64_t *alias;           // base DevCluster::table_off indexes
	const int32_t *prefix;
	int32_t *rows;                   // ring of three rows, [3][rows_width] int32, or nullptr: neighbours are read back from the plane
	int32_t rows_width;
	int32_t *wp_errors;              // weighted-predictor error rows [2 * width][5] in LDS, or nullptr: the HBM scratch
	int32_t wp_errors_width;
};
J40_DEV ModTables mod_tables_in_hbm(const DevModPlan &plan, int32_t g) {
	const DevModSection &sec = plan.sections[g];
	const DevCodeSpec &spec = plan.spec[sec.spec_idx];
	ModTables t = {plan.tree + sec.tree_off, plan.clusters + spec.cluster_off, plan.pool_u8 + spec.cluster_map_off, plan.pool_u64, plan.pool_i32, nullptr, 0, nullptr, 0};
	return t;
}

// K3: one pass-group section = every not-yet-decoded channel of the group's rectangle, one stream.
// UNI: every lane of the wavefront runs this on the same section (wave-uniform decoder state, see entropy_dev.h).
// RING: neighbours come from t.rows and the weighted predictor's rows from t.wp_errors (both sized for the widest rectangle of
// the frame by the caller), never from HBM -- a compile-time choice so that their address space stays static.
// where coded channel `cidx` of a section lives: top-left sample of its rectangle, row pitch, size, meta flag
struct ModChan { int16_t *base; int32_t stride, gw, gh, meta, shifts; };
J40_DEV ModChan mod_channel(const DevModPlan &plan, const DevModSection &sec, int32_t cidx) {
	ModChan c;
	if (sec.sub_off >= 0) {   // a plane of the section's own sub-image
		const DevSubPlane sp = plan.sub_planes[sec.sub_off + cidx];
		c.base = sp.ptr; c.stride = sp.w; c.gw = sp.w; c.gh = sp.h; c.meta = sp.meta; c.shifts = 0;
		return c;
	}
	if (sec.chan_off >= 0) {   // frames with channels of different sizes: an explicit rectangle
		const DevChanRect r = plan.chan_rects[sec.chan_off + cidx];
		const DevPlaneRef p = plan.planes[r.plane];
		c.base = p.ptr + (size_t) r.y0 * (size_t) p.w + (size_t) r.x0; c.stride = p.w; c.gw = r.w; c.gh = r.h; c.meta = p.meta; c.shifts = r.shifts;
		return c;
	}
	const DevPlaneRef p = plan.planes[sec.first_channel + cidx];
	c.meta = p.meta; c.stride = p.w; c.shifts = 0;
	// image channels: the section's rectangle; meta channels (palette): the whole plane
	const int32_t gx = c.meta ? 0 : sec.gx, gy = c.meta ? 0 : sec.gy;
	c.gw = c.meta ? p.w : sec.gw; c.gh = c.meta ? p.h : sec.gh;
	c.base = p.ptr + (size_t) gy * (size_t) c.stride + (size_t) gx;
	return c;
}

template <bool UNI, bool RING>
J40_DEV uint32_t decode_modular_section(const DevModPlan &plan, const ModTables &t, int32_t g) {
	// by value: references into HBM would be re-read after every sample store (the compiler cannot rule out aliasing)
	const DevModFrame f = *plan.frame;
	const DevModSection sec = plan.sections[g];
	const DevCodeSpec &spec = plan.spec[sec.spec_idx];
	DevBits b;
	bits_init<UNI>(b, plan.codestream, sec.byte_off, sec.size, sec.bit_off);
	DevCode code;
	int32_t *window = plan.lz_window ? plan.lz_window + (size_t) g * plan.lz_window_size : nullptr;
	code_init(code, spec, t.clusters, t.cluster_map, t.alias, t.prefix, window);
	// LZ77 distance multiplier: widest non-meta channel of this sub-image (j40.h:3841-3844)
	int32_t dist_mult = 0;
	if (sec.dist_mult_p1) dist_mult = sec.dist_mult_p1 - 1;   // (LfGlobal: the frame-wide image's, see plan.h)
	else for (int32_t cidx = 0; cidx < sec.num_channels; ++cidx) { const ModChan c = mod_channel(plan, sec, cidx); if (!c.meta) dist_mult = mod_max(dist_mult, c.gw); }
	dist_mult = mod_min(dist_mult, 1 << 21);
	ModWP wp;
	wp.on = sec.uses_wp; wp.width = sec.gw;
	wp.p1 = sec.wp[0]; wp.p2 = sec.wp[1];
	for (int i = 0; i < 5; ++i) wp.p3[i] = sec.wp[2 + i];
	for (int i = 0; i < 4; ++i) wp.w[i] = sec.wp[7 + i];
	wp.errors = nullptr;
	for (int i = 0; i < 5; ++i) wp.pred[i] = 0;
	wp.blend_err_w = wp.blend_err_n = wp.blend_err_nw = wp.blend_err_ne = 0;
	uint32_t err = 0;
	for (int32_t cidx = 0; cidx < sec.num_channels && !b.err && !err; ++cidx) {
		const ModChan chan = mod_channel(plan, sec, cidx);
		const int32_t stride = chan.stride, meta = chan.meta, gw = chan.gw, gh = chan.gh;
		if (gw <= 0 || gh <= 0) continue;
		int16_t *base = chan.base;
		wp.width = gw;
		constexpr bool ring = RING;
		if (wp.on) {
			if (RING) wp.errors = t.wp_errors; else wp.errors = plan.wp_scratch + (size_t) g * (size_t) (2 * f.max_width * 5);
			for (int32_t i = 0; i < 2 * gw * 5; ++i) wp.errors[i] = 0;
			for (int i = 0; i < 5; ++i) wp.pred[i] = 0;
			wp.blend_err_w = wp.blend_err_n = wp.blend_err_nw = wp.blend_err_ne = 0;
		}
		for (int32_t y = 0; y < gh && !b.err && !err; ++y) {
			int16_t *row = base + (size_t) y * (size_t) stride;
			int32_t *cur = t.rows + (y % 3) * t.rows_width;
			const int32_t *prev = t.rows + ((y + 2) % 3) * t.rows_width, *pprev = t.rows + ((y + 1) % 3) * t.rows_width;
			// RING: the neighbourhood slides along the row in registers -- of the eight neighbours only NN and the sample that
			// will be NEE of the next pixel are fetched per pixel (rows hold 3 spare entries, so x + 3 is always addressable;
			// entries outside the rectangle and the rows above the first are never selected)
			int32_t r_nww = 0, r_nw = 0, r_n = 0, r_ne = 0, r_nee = 0, c_w = 0, c_ww = 0;
			if (ring) { r_n = uni<UNI>(prev[0]); r_ne = uni<UNI>(prev[1]); r_nee = uni<UNI>(prev[2]); }
			for (int32_t x = 0; x < gw; ++x) {
				ModNeigh p;
				int32_t r_next = 0;
				if (ring) {
					const int32_t vnn = uni<UNI>(pprev[x]);
					r_next = uni<UNI>(prev[x + 3]);
					p.w = x > 0 ? c_w : y > 0 ? r_n : 0;
					p.n = y > 0 ? r_n : p.w;
					p.nw = x > 0 && y > 0 ? r_nw : p.w;
					p.ne = x + 1 < gw && y > 0 ? r_ne : p.n;
					p.nn = y > 1 ? vnn : p.n;
					p.nee = x + 2 < gw && y > 0 ? r_nee : p.ne;
					p.ww = x > 1 ? c_ww : p.w;
					p.nww = x > 1 && y > 0 ? r_nww : p.ww;
				} else p = mod_neighbours<UNI>(row, stride, gw, x, y);
				wp_before<UNI>(wp, x, y, p);
				const DevTreeNode *n = t.tree;
				DevTreeNode node;
				{ const int32_t *w4 = (const int32_t *) n; node.prop = uni<UNI>(w4[0]); node.value = uni<UNI>(w4[1]); node.a = uni<UNI>(w4[2]); node.b = uni<UNI>(w4[3]); }
				while (node.prop >= 0) {
					int32_t val;
					switch (node.prop) {
					case 0: val = cidx; break;
					case 1: val = sec.sidx; break;
					case 2: val = y; break;
					case 3: val = x; break;
					case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14:
						val = neighbour_property(node.prop, x, p.w, p.n, p.nw, p.ne, p.nn, p.ww, p.nww); break;   // (props_dev.h)
					case 15:
						val = wp.blend_err_w;   // the largest of the blend's errors at W, N, NW, NE (first wins a tie)
						if (mod_abs(val) < mod_abs(wp.blend_err_n)) val = wp.blend_err_n;
						if (mod_abs(val) < mod_abs(wp.blend_err_nw)) val = wp.blend_err_nw;
						if (mod_abs(val) < mod_abs(wp.blend_err_ne)) val = wp.blend_err_ne;
						break;
					default: {
						// "previous channel" properties: the r-th earlier channel of this sub-image with the same
						// geometry, nearest first (j40.h:4156-4165)
						int32_t r = (node.prop - 16) / 4, rc = -1;
						ModChan ref;
						for (int32_t k = cidx - 1; k >= 0; --k) {
							ref = mod_channel(plan, sec, k);
							if (ref.meta != meta || ref.gw != gw || ref.gh != gh || ref.shifts != chan.shifts) continue;
							if (r-- == 0) { rc = k; break; }
						}
						if (rc < 0) { err = ERR_TREC; val = 0; break; }
						const int32_t rstride = ref.stride;
						const int16_t *rrow = ref.base + (size_t) y * (size_t) rstride;
						val = uni<UNI>((int32_t) rrow[x]);
						if (node.prop & 2) {
							const int32_t rw = uni<UNI>(x > 0 ? (int32_t) rrow[x - 1] : 0);
							const int32_t rn = uni<UNI>(y > 0 ? (int32_t) rrow[x - rstride] : rw);
							const int32_t rnw = uni<UNI>(x > 0 && y > 0 ? (int32_t) rrow[x - 1 - rstride] : rw);
							val -= mod_gradient(rw, rn, rnw);
						}
						if (node.prop & 1) val = mod_abs(val);
					} }
					if (err) break;
					n += val > node.value ? node.a : node.b;
					{ const int32_t *w4 = (const int32_t *) n; node.prop = uni<UNI>(w4[0]); node.value = uni<UNI>(w4[1]); node.a = uni<UNI>(w4[2]); node.b = uni<UNI>(w4[3]); }
				}
				if (err) break;
				int32_t v = code_symbol<UNI>(b, code, node.value, dist_mult, plan.lz_window_size);
				v = ((v & 1) ? -(v / 2 + 1) : v / 2) * node.b + node.a;
				v += mod_predict(-1 - node.prop, wp, p, &err);
				if (v < -32768 || v > 32767) { err = ERR_POVF; break; }
				row[x] = (int16_t) v;
				if (ring) { cur[x] = v; c_ww = c_w; c_w = v; r_nww = r_nw; r_nw = r_n; r_n = r_ne; r_ne = r_nee; r_nee = r_next; }
				wp_after(wp, x, y, v);
				if (b.err) break;
			}
		}
	}
	if (!b.err && !err) code_finish<UNI>(b, code);
	if (!b.err && !err && f.check_section_end) bits_finish_section(b, f.single_declared_end);
	return b.err ? b.err : err;
}

// K4a: inverse RCT of one pixel (j40.h:4341-4393); values are int16 with wrap-around like the reference
J40_DEV void inverse_rct_pixel(int32_t type7, int16_t &a, int16_t &bb, int16_t &c) {
	const int16_t p0 = a, p1 = bb, p2 = c;
	switch (type7) {
	case 0: break;
	case 1: c = (int16_t) (p2 + p0); break;
	case 2: c = (int16_t) (p1 + p0); break;
	case 3: bb = (int16_t) (p1 + p0); c = (int16_t) (p2 + p0); break;
	case 4: bb = (int16_t) (p1 + (int16_t) (p0 / 2 + p2 / 2 + (p0 & p2 & 1))); break;
	case 5: bb = (int16_t) ((int32_t) p1 + p0 + (p2 >> 1)); c = (int16_t) (p2 + p0); break;
	default: {
		const int32_t tmp = (int32_t) p0 - ((int32_t) p2 >> 1);
		const int32_t q1 = (int32_t) p2 + tmp;
		const int32_t q2 = tmp - ((int32_t) p1 >> 1);
		a = (int16_t) (q2 + p1); bb = (int16_t) q1; c = (int16_t) q2;
	} }
}

// K4a': the RCTs a pass-group section lists in its own header, undone last to first over the section's
// rectangle (j40__inverse_transform on the group's sub-image, j40.h:7030, before it is pasted, j40.h:3688).
// The reference renames the three planes by the permutation type / 7; here the samples move.
J40_DEV void section_inverse_rcts(const DevModPlan &plan, int32_t s, int32_t lane, int32_t nlanes) {
	const DevModSection sec = plan.sections[s];
	if (sec.local_count <= 0) return;
	for (int32_t k = sec.local_count; k-- > 0; ) {
		const int32_t begin = plan.local_rct[2 * (sec.local_off + k)], type = plan.local_rct[2 * (sec.local_off + k) + 1];
		const ModChan ch[3] = {mod_channel(plan, sec, begin), mod_channel(plan, sec, begin + 1), mod_channel(plan, sec, begin + 2)};
		const int32_t perm = type / 7;
		// output channel PERM[perm][j] takes transformed channel j (j40.h:4395-4398)
		const int32_t d0 = perm == 0 || perm == 3 ? 0 : perm == 1 || perm == 4 ? 1 : 2;
		const int32_t d1 = perm == 0 || perm == 5 ? 1 : perm == 1 || perm == 3 ? 2 : 0;
		const int32_t d2 = 3 - d0 - d1;
		for (int32_t i = lane; i < ch[0].gw * ch[0].gh; i += nlanes) {
			const int32_t y = i / ch[0].gw, x = i - y * ch[0].gw;
			const size_t at = (size_t) y * (size_t) ch[0].stride + (size_t) x;   // the three channels are equal-sized planes of one image
			int16_t p[3] = {ch[0].base[at], ch[1].base[at], ch[2].base[at]};
			inverse_rct_pixel(type % 7, p[0], p[1], p[2]);
			ch[d0].base[at] = p[0]; ch[d1].base[at] = p[1]; ch[d2].base[at] = p[2];
		}
	}
}

// the spec's 72 palette delta triples; entry 2k is triple k, entry 2k + 1 its negation (j40.h:4275)
#ifdef __HIPCC__
__device__
#endif
static const int16_t DEV_PALETTE_DELTAS[72][3] = {
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

// K4b: palette look-up for output channel i of one pixel, before the optional delta prediction
// (j40.h:4446-4468)
J40_DEV int16_t palette_value(int16_t idx, int32_t i, const int16_t *palrow /* row i of the palette or null */, int32_t nb_colours, int32_t bpp) {
	int16_t val;
	if (idx < 0) {
		if (i < 3) {
			idx = (int16_t) (~idx % 143);
			const int32_t entry = idx + 1;
			val = DEV_PALETTE_DELTAS[entry >> 1][i];
			if (entry & 1) val = (int16_t) -val;
			if (bpp > 8) val = (int16_t) (val << ((bpp < 24 ? bpp : 24) - 8));
		} else val = 0;
	} else if (idx < nb_colours) {
		val = palrow[idx];
	} else {
		idx = (int16_t) (idx - nb_colours);
		if (idx < 64) {
			val = (int16_t) ((i < 3 ? idx >> (2 * i) : 0) * (((int32_t) 1 << bpp) - 1) / 4 + ((int32_t) 1 << (bpp - 3 > 0 ? bpp - 3 : 0)));
		} else {
			val = (int16_t) (idx - 64);
			for (int32_t j = 0; j < i; ++j) val = (int16_t) (val / 5);
			val = (int16_t) ((val % 5) * ((1 << bpp) - 1) / 4);
		}
	}
	return val;
}

// K5: one pixel of j40__render_to_u8x4_rgba (j40.h:7947-7953)
J40_DEV uint32_t pack_rgba8(int32_t r, int32_t g, int32_t b2, int32_t a, int32_t bpp) {
	const int32_t maxpixel = (1 << bpp) - 1, maxpixel2 = 1 << (bpp - 1);
	const int32_t v[4] = {r, g, b2, a};
	uint32_t out = 0;
	for (int i = 0; i < 4; ++i) {
		const int32_t p = v[i] < 0 ? 0 : v[i] > maxpixel ? maxpixel : v[i];
		out |= (uint32_t) ((p * 255 + maxpixel2) / maxpixel) << (8 * i);
	}
	return out;
}

} // namespace j40hip
