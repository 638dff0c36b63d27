// j40_amd/csrc/device/vardct_dev.h -- per-varblock geometry and the fused coefficient load
// (dequantise + chroma-from-luma + LLF substitution) shared by the coefficients -> pixels kernels
// (j40__dequant_hf, j40.h:7053; j40__combine_vardct_from_lf_group, j40.h:7099-7175).
#pragma once
#include "idct_dev.h"

namespace j40hip {

struct VbGeom {
	int32_t coeff_base;   // index of the block's first coefficient in plan.coeffs[c]
	int32_t llf_base;     // index of the block's first LLF coefficient in plan.llf[c]
	float mult[3];
	float kx_hf, kb_hf;
	int32_t px, py;       // top-left pixel in the frame
	int32_t effw, effh;   // visible size
};

J40_DEV VbGeom varblock_geometry(const DevPlan &plan, const DevVarblock &vb) {
	const DevFrame &f = *plan.frame;
	VbGeom g;
	g.coeff_base = vb.coeff_base; g.llf_base = vb.llf_base;
	g.mult[1] = vb.mult1; g.mult[0] = vb.mult1 * f.x_qm_mul; g.mult[2] = vb.mult1 * f.b_qm_mul;   // j40.h:7078-7080
	g.kx_hf = vb.kx_hf; g.kb_hf = vb.kb_hf;
	g.px = vb.px; g.py = vb.py; g.effw = vb.effw; g.effh = vb.effh;
	return g;
}

// loads coefficient `i` (canonical layout index) of all three channels from the dense planes (multi-pass frames):
// dequantised, chroma-from-luma applied, LLF corner substituted (j40.h:7086-7094, 7157-7172)
J40_DEV void load_coeff3(const DevPlan &plan, const VbGeom &g, const float *dq, int32_t dq_size, int32_t i, int32_t long_side, int32_t vh8, int32_t vw8, float out[3]) {
	const DevFrame &f = *plan.frame;
	const int32_t srow = i / long_side, scol = i - srow * long_side;
	if (srow < vh8 && scol < vw8) {
		const int32_t l = g.llf_base + srow * vw8 + scol;
		const float lx = plan.llf[0][l], ly = plan.llf[1][l], lb = plan.llf[2][l];
		out[0] = lx + ly * f.kx_lf; out[1] = ly; out[2] = lb + ly * f.kb_lf;
		return;
	}
	const float qx = dequant_coeff(plan.coeffs[0][g.coeff_base + i], f.quant_bias[0], f.quant_bias_num, g.mult[0], dq[i]);
	const float qy = dequant_coeff(plan.coeffs[1][g.coeff_base + i], f.quant_bias[1], f.quant_bias_num, g.mult[1], dq[dq_size + i]);
	const float qb = dequant_coeff(plan.coeffs[2][g.coeff_base + i], f.quant_bias[2], f.quant_bias_num, g.mult[2], dq[2 * dq_size + i]);
	out[0] = qx + qy * g.kx_hf; out[1] = qy; out[2] = qb + qy * g.kb_hf;
}

// ---- sparse coefficients (DevPlan::events) -> LDS tiles ----
// A tile holds one channel of one block; `TileMap` says where canonical index i lives in it. The three channel tiles are
// `cstride` floats apart. Steps, all cooperative over lanes `lane, lane + nlanes, ...` (tests: 0, 1):
//   1. the caller zeroes the tiles (and synchronises);  2. tile_scatter_events: dequantised non-zeros with their
//   chroma-from-luma contributions;  3. tile_fill_llf: the LLF corner (no event lands there).
// Same values as load_coeff3 computes per position: a zero coefficient dequantises to +0 and 0 + 0 * k = +0.
struct TileMap {
	int32_t rows, columns, pitch, linear;   // linear: the 8x8 special transforms keep canonical index i in row i / 8, column i % 8 (rows `pitch` apart)
	J40_DEVM int32_t at(int32_t i) const {
		if (linear) return (i >> 3) * pitch + (i & 7);
		const int32_t r = columns > rows ? i / columns : i % rows, c = columns > rows ? i % columns : i / rows;   // j40.h:5978-5985
		return r * pitch + c;
	}
};

// the same for tiles whose power-of-two sides are known at run time only (k_vardct_large): shifts, no division per event
struct TileMapLog {
	int32_t log_rows, log_columns, pitch;
	J40_DEVM int32_t at(int32_t i) const {
		const bool wide = log_columns > log_rows;
		const int32_t r = wide ? i >> log_columns : i & ((1 << log_rows) - 1), c = wide ? i & ((1 << log_columns) - 1) : i >> log_rows;
		return r * pitch + c;
	}
};

// `be`: the block's entry of DevPlan::block_events (first event, counts in emission order Y, X, B). `dq_scan`: the weights in
// scan order (DevFrame::dq_scan_off). Chroma-from-luma rides along: a Y coefficient also contributes kx * Y to X and kb * Y to B
// at its position, so X and B are accumulated (x + kx * y has two addends, and IEEE addition commutes, so the order in which
// the two arrive does not matter; +0 + v = v). On the device the accumulation is an LDS / global atomic add.
J40_DEV void tile_add(float *p, float v) {
#ifdef __HIPCC__
	atomicAdd(p, v);
#else
	*p += v;
#endif
}
// event `e` (0-based within the block) of the block whose entry of DevPlan::block_events is `be`
template <typename BE, typename GEOM, typename TILE, typename ORD, typename DQ, typename MAP>
J40_DEV void tile_scatter_one(const DevPlan &plan, const GEOM &g, const BE &be, uint32_t e, ORD order /* pass 0: [3][n] */, DQ dq_scan /* [3][n] */, int32_t n,
		const MAP &map, TILE tile, int32_t cstride, const float quant_bias[3], float quant_bias_num) {
	const uint32_t first = be[0], n0 = be[1], n1 = be[2];
	const int32_t c = e < n0 ? 1 : e < n0 + n1 ? 0 : 2;   // events come in the order the channels are coded: Y, X, B
	const CoeffEvent ev = plan.events[first + e];
	const int32_t pos = (int32_t) coeff_event_pos(ev);
	const int32_t at = map.at(order[c * n + pos]);
	const float v = dequant_coeff((float) coeff_event_value(ev), quant_bias[c], quant_bias_num, g.mult[c], dq_scan[c * n + pos]);
	if (c == 1) {
		tile[cstride + at] = v;
		tile_add(&tile[at], v * g.kx_hf);
		tile_add(&tile[2 * cstride + at], v * g.kb_hf);
	} else tile_add(&tile[c * cstride + at], v);
}
template <typename MAP>
J40_DEV void tile_scatter_events(const DevPlan &plan, const VbGeom &g, const uint32_t be[4], const uint16_t *order, const float *dq_scan, int32_t n,
		const MAP &map, float *tile, int32_t cstride, const float quant_bias[3], float quant_bias_num, int32_t lane, int32_t nlanes) {
	const uint32_t total = be[1] + be[2] + be[3];
	for (uint32_t e = (uint32_t) lane; e < total; e += (uint32_t) nlanes) tile_scatter_one(plan, g, be, e, order, dq_scan, n, map, tile, cstride, quant_bias, quant_bias_num);
}

// The pixel kernels' form: the events of all `nb` blocks of a workgroup as one list shared by every lane, so that small blocks
// with a dozen non-zeros do not cost a wavefront each. prefix[b] = events of blocks 0..b-1, prefix[b] = total for b >= nb
// (b <= NB, a power of two); tiles of consecutive blocks are `bstride` floats apart.
// dq_block: nullptr, or per block the offset to add to dq_scan (blocks of different transforms in one workgroup: the 8x8 specials)
template <int NB, typename ORD, typename DQ>
J40_DEV void tiles_scatter_events(const DevPlan &plan, const VbGeom *geom, const uint32_t (*be)[4], const uint32_t *prefix, ORD order, DQ dq_scan, const uint32_t *dq_block, int32_t n,
		const TileMap &map, float *tiles, int32_t bstride, int32_t cstride, const float quant_bias[3], float quant_bias_num, int32_t lane, int32_t nlanes) {
	const uint32_t total = prefix[NB];
	for (uint32_t e = (uint32_t) lane; e < total; e += (uint32_t) nlanes) {
		int32_t b = 0;
#pragma unroll
		for (int32_t step = NB >> 1; step >= 1; step >>= 1) if (prefix[b + step] <= e) b += step;
		tile_scatter_one(plan, geom[b], be[b], e - prefix[b], order, dq_block ? dq_scan + dq_block[b] : dq_scan, n, map, tiles + (size_t) b * (size_t) bstride, cstride, quant_bias, quant_bias_num);
	}
}
// LLF corners of all `nb` blocks, one lane per (block, LLF position)
J40_DEV void tiles_fill_llf(const DevPlan &plan, const VbGeom *geom, int32_t nb, int32_t long_side, int32_t vh8, int32_t vw8, const TileMap &map, float *tiles, int32_t bstride, int32_t cstride,
		float kx_lf, float kb_lf, int32_t lane, int32_t nlanes) {
	const int32_t per = vh8 * vw8;
	for (int32_t w = lane; w < nb * per; w += nlanes) {
		const int32_t b = w / per, k = w - b * per;
		const int32_t srow = k / vw8, scol = k - srow * vw8, at = map.at(srow * long_side + scol), l = geom[b].llf_base + k;
		const float lx = plan.llf[0][l], ly = plan.llf[1][l], lb = plan.llf[2][l];
		float *tile = tiles + (size_t) b * (size_t) bstride;
		tile[at] = lx + ly * kx_lf; tile[cstride + at] = ly; tile[2 * cstride + at] = lb + ly * kb_lf;
	}
}

template <typename MAP>
J40_DEV void tile_fill_llf(const DevPlan &plan, const VbGeom &g, int32_t long_side, int32_t vh8, int32_t vw8, const MAP &map, float *tile, int32_t cstride, float kx_lf, float kb_lf,
		int32_t lane, int32_t nlanes) {
	for (int32_t k = lane; k < vh8 * vw8; k += nlanes) {
		const int32_t srow = k / vw8, scol = k - srow * vw8, at = map.at(srow * long_side + scol), l = g.llf_base + k;
		const float lx = plan.llf[0][l], ly = plan.llf[1][l], lb = plan.llf[2][l];
		tile[at] = lx + ly * kx_lf; tile[cstride + at] = ly; tile[2 * cstride + at] = lb + ly * kb_lf;
	}
}

// ONE channel of a block at a time (k_vardct_large's 128x128 tiles: LDS holds one channel with its work buffer, not three): the
// channel's own events, and for X and B the Y events' chroma-from-luma contributions -- the same two addends per position as in
// tile_scatter_one, whichever arrives first. `tile`: the channel's tile, zeroed.
template <typename TILE, typename MAP>
J40_DEV void tile_scatter_channel(const DevPlan &plan, const VbGeom &g, const uint32_t be[4], int32_t ch, const uint16_t *order, const float *dq_scan, int32_t n,
		const MAP &map, TILE tile, const float quant_bias[3], float quant_bias_num, int32_t lane, int32_t nlanes) {
	const uint32_t first = be[0], n0 = be[1], n1 = be[2], total = be[1] + be[2] + be[3];
	// (Y, X, B in this order: Y's events are wanted by every channel, X's and B's by their own)
	const uint32_t end = ch == 1 ? n0 : ch == 0 ? n0 + n1 : total;
	for (uint32_t e = (uint32_t) lane; e < end; e += (uint32_t) nlanes) {
		const int32_t c = e < n0 ? 1 : e < n0 + n1 ? 0 : 2;
		if (c != ch && c != 1) continue;   // (B's pass walks over X's events)
		const CoeffEvent ev = plan.events[first + e];
		const int32_t pos = (int32_t) coeff_event_pos(ev);
		const int32_t at = map.at(order[c * n + pos]);
		const float v = dequant_coeff((float) coeff_event_value(ev), quant_bias[c], quant_bias_num, g.mult[c], dq_scan[c * n + pos]);
		if (ch == 1) tile[at] = v;
		else tile_add((float *) &tile[at], c == ch ? v : v * (ch == 0 ? g.kx_hf : g.kb_hf));
	}
}
template <typename TILE, typename MAP>
J40_DEV void tile_fill_llf_channel(const DevPlan &plan, const VbGeom &g, int32_t ch, int32_t long_side, int32_t vh8, int32_t vw8, const MAP &map, TILE tile, float kx_lf, float kb_lf,
		int32_t lane, int32_t nlanes) {
	for (int32_t k = lane; k < vh8 * vw8; k += nlanes) {
		const int32_t srow = k / vw8, scol = k - srow * vw8, at = map.at(srow * long_side + scol), l = g.llf_base + k;
		const float ly = plan.llf[1][l];
		tile[at] = ch == 1 ? ly : ch == 0 ? plan.llf[0][l] + ly * kx_lf : plan.llf[2][l] + ly * kb_lf;
	}
}

} // namespace j40hip
