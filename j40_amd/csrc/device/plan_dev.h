// j40_amd/csrc/device/plan_dev.h -- the LF-dependent half of a VarDCT frame's plan as device functions (plan_kernels.hip runs
// them; tests/hostsim compiles them for the CPU and compares their products with plan_build.cpp's, array by array).
//
// What they replace, given the decoded planes of every LfGroup section (LF integers, chroma-from-luma maps, the two rows of the
// varblock-info channel):
//   plan_place_lf_group   the varblock placement of j40__hf_metadata (j40.h:6634-6701): the next varblock goes to the first free
//                         cell in raster order; its coefficient offset, quantisation-field index and error checks -- one serial
//                         walk per LfGroup (the position of a block depends on every block before it)
//   plan_scan_frame       prefix sums: where each group's block list and each (DctSelect, LfGroup)'s run of work items start
//   plan_emit_varblock    per varblock: the LF index of its top-left cell (j40.h:6566-6570) and with it the three block contexts
//                         (j40.h:6951-6953) -> the K1 record (DevGroupBlock); geometry, multipliers and chroma-from-luma
//                         factors (j40.h:7078-7080, 7138-7143) -> the K2 record (DevVarblock)
// The host path (frame.cpp lf_group_finish + plan_build.cpp build_vardct_plan) computes the same arrays; in the pipeline it no
// longer runs.
#pragma once
#include "hf_dev.h"

namespace j40hip {

enum { PLAN_LOG_GSIZE8 = 5 };   // VarDCT frames: groups of 256 x 256 pixels = 32 x 32 cells (frame.cpp: group_size_shift stays 8)

// occ[x * stride]: the row below the lowest cell any placed block occupies in column x; grp_cnt[64 * stride], cls_cnt[28 * stride]:
// running counts. stride interleaves the scratch of the 64 LfGroups a wavefront places side by side (1 on the CPU).
J40_DEV void plan_place_lf_group(const DevPlanBuild &pb, int32_t g, uint16_t *occ, uint16_t *grp_cnt, uint32_t *cls_cnt, int32_t stride) {
	DevLfGroup &gg = pb.lf_groups[g];
	DevLfSlot &slot = pb.lf_slots[g];
	const int32_t w8 = gg.width8, h8 = gg.height8;
	const int32_t ggx = g % pb.ggcolumns, ggy = g / pb.ggcolumns;
	for (int32_t x = 0; x < w8; ++x) occ[x * stride] = 0;
	for (int32_t i = 0; i < 64; ++i) grp_cnt[i * stride] = 0;
	for (int32_t i = 0; i < 28; ++i) cls_cnt[i * stride] = 0;
	uint32_t err = slot.status, used = 0;
	int32_t voff = 0;
	if (!err) {
		const int32_t nbv = slot.nb_varblocks;
		const int16_t *info0 = pb.vbinfo + 2 * (size_t) gg.cell_base, *info1 = info0 + nbv;
		DevVbRec *recs = pb.vb_recs + gg.vb_base;
		const int32_t coeff_limit = w8 * h8 * 64;
		int32_t coeffoff = 0;
		for (int32_t y0 = 0; y0 < h8 && !err; ++y0) for (int32_t x0 = 0; x0 < w8; ++x0) {
			if ((int32_t) occ[x0 * stride] > y0) continue;
			if (voff >= nbv) { err = ERR_VBLK; break; }
			const int32_t dctsel = info0[voff];
			if (dctsel < 0 || dctsel >= 27) { err = ERR_DCTQ; break; }
			const int32_t log_rows = DEV_DCT_SELECT[dctsel][0], log_columns = DEV_DCT_SELECT[dctsel][1];
			const int32_t vw8 = 1 << (log_columns - 3), vh8 = 1 << (log_rows - 3), x1 = x0 + vw8 - 1, y1 = y0 + vh8 - 1;
			// the block must lie inside the LfGroup and inside one group (j40.h:6655-6656) ...
			if (!(x1 < w8 && (x0 >> PLAN_LOG_GSIZE8) == (x1 >> PLAN_LOG_GSIZE8)) || !(y1 < h8 && (y0 >> PLAN_LOG_GSIZE8) == (y1 >> PLAN_LOG_GSIZE8))) { err = ERR_VBLK; break; }
			// ... and its coefficients inside the LfGroup's arrays. The reference does not check this (its note at j40.h:6691): blocks
			// that overlap ones placed before can add up to more cells than the LfGroup has, and it then writes past its arrays.
			if (coeffoff + (1 << (log_rows + log_columns)) > coeff_limit) { err = ERR_VBLK; break; }
			for (int32_t j = 0; j < vw8; ++j) { uint16_t &o = occ[(x0 + j) * stride]; if ((int32_t) o < y1 + 1) o = (uint16_t) (y1 + 1); }
			const int32_t hfmul_m1 = info1[voff];
			int32_t qf = 0;
			for (int32_t j = 0; j < pb.nb_qf_thr; ++j) qf += hfmul_m1 >= pb.qf_thr[j];
			const int32_t grp = (y0 >> PLAN_LOG_GSIZE8) * 8 + (x0 >> PLAN_LOG_GSIZE8);
			DevVbRec r;
			r.coeffoff_qfidx = (uint32_t) (coeffoff + qf); r.hfmul_m1 = (int16_t) hfmul_m1; r.x8 = (uint8_t) x0; r.y8 = (uint8_t) y0;
			r.dctsel = (uint8_t) dctsel; r.grp = (uint8_t) grp;
			r.rank_in_group = grp_cnt[grp * stride]; grp_cnt[grp * stride] = (uint16_t) (r.rank_in_group + 1);
			r.rank_in_class = cls_cnt[dctsel * stride]; cls_cnt[dctsel * stride] = r.rank_in_class + 1;
			recs[voff] = r;
			used |= 1u << dctsel;
			coeffoff += 1 << (log_rows + log_columns);
			++voff;
		}
		if (!err && voff != nbv) err = ERR_VBLK;
	}
	slot.status = err; slot.placed = voff; slot.dct_used = used;
	gg.nb_varblocks = voff;
	// every group lies in exactly one LfGroup: plain stores
	for (int32_t gy = 0; gy < 8; ++gy) for (int32_t gx = 0; gx < 8; ++gx) {
		if ((gx << PLAN_LOG_GSIZE8) >= w8 || (gy << PLAN_LOG_GSIZE8) >= h8) continue;
		pb.group_count[(ggy * 8 + gy) * pb.gcolumns + ggx * 8 + gx] = grp_cnt[(gy * 8 + gx) * stride];
	}
	for (int32_t d = 0; d < 28; ++d) pb.class_count[g * 28 + d] = cls_cnt[d * stride];
}

// one thread per frame
J40_DEV void plan_scan_frame(const DevPlanBuild &pb) {
	uint32_t at = 0;
	for (int32_t g = 0; g < pb.num_groups; ++g) { pb.group_block_start[g] = at; at += pb.group_count[g]; }
	pb.group_block_start[pb.num_groups] = at;
	// work items of the pixel kernels: grouped by DctSelect, inside a class by (LfGroup, varblock) -- plan_build.cpp's order
	uint32_t k = 0;
	for (int32_t d = 0; d < 28; ++d) {
		pb.class_start[d] = (int32_t) k;
		for (int32_t g = 0; g < pb.num_lf_groups; ++g) { const uint32_t n = pb.class_count[g * 28 + d]; pb.class_count[g * 28 + d] = k; k += n; }
	}
}

J40_DEV void plan_emit_varblock(const DevPlanBuild &pb, int32_t g, int32_t v) {
	const DevLfGroup &gg = pb.lf_groups[g];
	const DevVbRec r = pb.vb_recs[gg.vb_base + v];
	const int32_t x8 = r.x8, y8 = r.y8, dctsel = r.dctsel;
	const int32_t ggx = g % pb.ggcolumns, ggy = g / pb.ggcolumns;
	const int32_t gid = (ggy * 8 + (r.grp >> 3)) * pb.gcolumns + ggx * 8 + (r.grp & 7);
	const uint32_t blk = pb.group_block_start[gid] + r.rank_in_group;
	const size_t cell = (size_t) gg.cell_base + (size_t) y8 * (size_t) gg.width8 + (size_t) x8;
	// LF index of the top-left cell: thresholds counted on the raw integers, 8-bit arithmetic like frame.cpp (j40.h:6566-6570)
	uint8_t lfidx = 0;
	{
		const int32_t vx = pb.lfraw[0][cell], vy = pb.lfraw[1][cell], vb = pb.lfraw[2][cell];
		for (int32_t t = 0; t < pb.nb_lf_thr[0]; ++t) lfidx = (uint8_t) (lfidx + (vx > pb.lf_thr[0][t]));
		lfidx = (uint8_t) (lfidx * (pb.nb_lf_thr[0] + 1));
		for (int32_t t = 0; t < pb.nb_lf_thr[2]; ++t) lfidx = (uint8_t) (lfidx + (vb > pb.lf_thr[2][t]));
		lfidx = (uint8_t) (lfidx * (pb.nb_lf_thr[2] + 1));
		for (int32_t t = 0; t < pb.nb_lf_thr[1]; ++t) lfidx = (uint8_t) (lfidx + (vy > pb.lf_thr[1][t]));
	}
	const int32_t log_rows = DEV_DCT_SELECT[dctsel][0], log_columns = DEV_DCT_SELECT[dctsel][1], order_idx = DEV_DCT_SELECT[dctsel][2];
	DevGroupBlock gb;
	gb.coeffoff_qfidx = r.coeffoff_qfidx;
	gb.pos_dct = (uint16_t) (((y8 & 31) * 32 + (x8 & 31)) | (dctsel << 10));
	{
		const uint8_t *map = pb.pool_u8 + pb.block_ctx_map_off;
		const int32_t nb_qf1 = pb.nb_qf_thr + 1, lfidx_size = pb.lfidx_size;
		const int32_t bctx0 = (order_idx * nb_qf1 + (int32_t) (r.coeffoff_qfidx & 15u)) * lfidx_size + lfidx;
		uint32_t b3 = 0;
		for (int32_t c_yxb = 0; c_yxb < 3; ++c_yxb) b3 |= (uint32_t) (map[bctx0 + 13 * nb_qf1 * lfidx_size * c_yxb] & 15) << (4 * c_yxb);
		gb.bctx3 = (uint16_t) b3;
	}
	pb.group_blocks[blk] = gb;
	DevVarblock dv;
	const int32_t coeffoff = (int32_t) (r.coeffoff_qfidx & ~15u);
	dv.coeff_base = gg.cell_base * 64 + coeffoff; dv.llf_base = gg.cell_base + (coeffoff >> 6);
	dv.mult1 = pb.mult_base * (1.0f / ((float) r.hfmul_m1 + 1.0f));   // j40.h:6699, 7078
	const size_t c64 = (size_t) gg.c64_base + (size_t) (y8 / 8) * (size_t) gg.width64 + (size_t) (x8 / 8);
	dv.kx_hf = pb.base_corr_x + pb.inv_colour_factor * (float) pb.xfromy[c64];   // j40.h:7138-7143, one factor per varblock
	dv.kb_hf = pb.base_corr_b + pb.inv_colour_factor * (float) pb.bfromy[c64];
	dv.px = gg.left + x8 * 8; dv.py = gg.top + y8 * 8;
	const int32_t eh = gg.height - y8 * 8, ew = gg.width - x8 * 8;
	dv.effh = (uint16_t) (eh < (1 << log_rows) ? eh : 1 << log_rows); dv.effw = (uint16_t) (ew < (1 << log_columns) ? ew : 1 << log_columns);
	dv.dctsel = (uint8_t) dctsel;
	dv.pad[0] = (uint8_t) g; dv.pad[1] = (uint8_t) (g >> 8); dv.pad[2] = (uint8_t) (g >> 16);
	dv.blk = (int32_t) blk;
	pb.vb_sorted[pb.class_count[g * 28 + dctsel] + r.rank_in_class] = dv;
}

// The frame's verdict: the first failing section in the order the reference reads them -- by offset, LfGroup and pass-group
// sections alike (j40.h:7840-7860; a dequantisation matrix that fails to load is the host's to add: it knows which ones do).
// Serial form (the kernel reduces the same keys in parallel): key = offset << 32 | code, smallest wins.
J40_DEV uint64_t plan_verdict_key(uint32_t status, uint32_t byte_off) { return status && status != (uint32_t) ERR_LFFB ? (uint64_t) byte_off << 32 | status : ~(uint64_t) 0; }

} // namespace j40hip
