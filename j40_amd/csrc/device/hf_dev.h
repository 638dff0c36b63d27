// j40_amd/csrc/device/hf_dev.h -- K1: HF coefficient decode of one group (all passes), sequential per
// lane. Restates j40__pass_group / j40__hf_coeffs (j40.h:7007-7041, 6888-7005) on the flat plan.
//
// Work item = one 256x256 group; its passes are decoded back to back by the same lane so the
// `coeffs[order[i]] += value` accumulation (j40.h:6989) needs no atomics.
#pragma once
#include "entropy_dev.h"

namespace j40hip {

// DctSelect -> log rows, log columns, coefficient order (spec table, cf. j40.h:4591)
#ifdef __HIPCC__
__device__
#endif
static const int8_t DEV_DCT_SELECT[27][3] = {
	{3, 3, 0}, {3, 3, 1}, {3, 3, 1}, {3, 3, 1}, {4, 4, 2}, {5, 5, 3}, {4, 3, 4}, {3, 4, 4}, {5, 3, 5}, {3, 5, 5}, {5, 4, 6}, {4, 5, 6}, {3, 3, 1}, {3, 3, 1},
	{3, 3, 1}, {3, 3, 1}, {3, 3, 1}, {3, 3, 1}, {6, 6, 7}, {6, 5, 8}, {5, 6, 8}, {7, 7, 9}, {7, 6, 10}, {6, 7, 10}, {8, 8, 11}, {8, 7, 12}, {7, 8, 12},
};

// coefficient context tables, pre-doubled (spec constants; cf. j40.h:6935-6947)
#ifdef __HIPCC__
__device__
#endif
static const int8_t DEV_FREQ_CTX2[64] = {
	-1, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 30, 32, 32, 34, 34, 36, 36, 38, 38, 40, 40, 42, 42, 44, 44,
	46, 46, 46, 46, 48, 48, 48, 48, 50, 50, 50, 50, 52, 52, 52, 52, 54, 54, 54, 54, 56, 56, 56, 56, 58, 58, 58, 58, 60, 60, 60, 60,
};
#ifdef __HIPCC__
__device__
#endif
static const int16_t DEV_NNZ_CTX2[64] = {
	0, 0, 62, 124, 124, 186, 186, 186, 186, 246, 246, 246, 246, 304, 304, 304, 304, 304, 304, 304, 304, 360, 360, 360, 360, 360, 360, 360, 360, 360, 360, 360,
	360, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412, 412,
};

J40_DEV int32_t unpack_signed_dev(int32_t x) { return (x / 2) ^ -(x & 1); }   // (x & 1) ? -(x / 2 + 1) : x / 2 for every x (-(h + 1) = ~h), without the branch the compiler made of it

// what one section decode reads besides the bitstream. The kernel stages these in LDS when they fit
// (the serial decoder then never waits on HBM for a table), otherwise they point into HBM.
struct HfTables {
	const DevCluster *clusters;      // of this pass' code spec
	const uint8_t *cluster_map;
	const uint64_t *alias;
	const int32_t *prefix;
	const uint8_t *block_ctx_map;
	const int16_t *nnz_ctx2;         // DEV_NNZ_CTX2
	const int8_t *freq_ctx2;         // DEV_FREQ_CTX2
	const DevGroupBlock *blocks;     // this group's varblocks in visiting order
	int32_t nblocks;
	int8_t *nonzeros;                // [32 * 32][3] scratch
	int32_t *window;                 // LZ77 window or nullptr
	uint32_t block_first;            // ordinal of blocks[0] in plan.group_blocks / plan.block_events
	uint32_t ev_first, ev_end;       // this group's region of plan.events (sparse coefficients)
};

// decodes one (pass, group) section. SCAN: single-pass frames append one event {scan position, value} per non-zero
// coefficient (DevPlan::events; the pixel kernels undo the order), otherwise accumulate at the canonical position of the
// dense planes like j40.h:6989.
template <bool SCAN, bool UNI>
J40_DEV uint32_t decode_hf_section(const DevPlan &plan, const DevFrame &f, const DevCodeSpec &spec, const HfTables &t, int32_t pass, const DevSection &sec) {
	const DevLfGroup &gg = plan.lf_groups[sec.ggidx];
	DevBits b;
	bits_init<UNI>(b, plan.codestream, sec.byte_off, sec.size, sec.bit_off);
	const uint32_t preset = bits_u<UNI>(b, f.preset_bits);
	if ((int32_t) preset >= f.num_hf_presets) bits_set_error(b, ERR_RNGE);
	const int32_t ctxoff = 495 * f.nb_block_ctx * (int32_t) preset;
	DevCode code;
	code_init(code, spec, t.clusters, t.cluster_map, t.alias, t.prefix, t.window);
	const int32_t gw8 = sec.gw8;
	const int32_t nb_block_ctx = f.nb_block_ctx;
	const size_t cell64 = (size_t) gg.cell_base * 64;
	uint32_t ev_at = t.ev_first;
	for (int32_t k = 0; k < t.nblocks && !b.err; ++k) {
		if (SCAN) plan.block_events[4 * (size_t) (t.block_first + (uint32_t) k)] = ev_at;
		DevGroupBlock gb;
		{ const uint32_t *p = (const uint32_t *) (t.blocks + k); gb.coeffoff_qfidx = uni<UNI>(p[0]); const uint32_t w = uni<UNI>(p[1]); gb.pos_dct = (uint16_t) w; gb.bctx3 = (uint16_t) (w >> 16); }
		const int32_t dctsel = gb.pos_dct >> 10, nzpos = ((gb.pos_dct >> 5) & 31) * gw8 + (gb.pos_dct & 31);
		const int32_t x8 = gb.pos_dct & 31, y8 = (gb.pos_dct >> 5) & 31;
		const int32_t log_rows = uni<UNI>((int32_t) DEV_DCT_SELECT[dctsel][0]), log_columns = uni<UNI>((int32_t) DEV_DCT_SELECT[dctsel][1]), order_idx = uni<UNI>((int32_t) DEV_DCT_SELECT[dctsel][2]);
		const int32_t log_size = log_rows + log_columns, shift = log_size - 6, size = 1 << log_size;
		const int32_t coeffoff = (int32_t) (gb.coeffoff_qfidx & ~15u);
		for (int32_t c_yxb = 0; c_yxb < 3 && !b.err; ++c_yxb) {
			const int32_t c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			float *coeffs = SCAN ? nullptr : plan.coeffs[c] + cell64 + coeffoff;
			const uint32_t chan_first = ev_at;
			const int32_t bctx = (gb.bctx3 >> (4 * c_yxb)) & 15;
			// number of non-zeros, predicted from the left / top blocks (j40.h:6959-6967)
			int32_t nz;
			if (x8 > 0) nz = y8 > 0 ? (t.nonzeros[(nzpos - 1) * 3 + c] + t.nonzeros[(nzpos - gw8) * 3 + c] + 1) >> 1 : t.nonzeros[(nzpos - 1) * 3 + c];
			else nz = y8 > 0 ? t.nonzeros[(nzpos - gw8) * 3 + c] : 32;
			nz = uni<UNI>(nz);
			const int32_t nzctx = ctxoff + bctx + (nz < 8 ? nz : 4 + nz / 2) * nb_block_ctx;
			nz = code_symbol<UNI>(b, code, nzctx, 0, plan.lz_window_size);
			if (nz > (63 << shift)) { bits_set_error(b, ERR_COEF); break; }
			const int32_t qnz = (nz + (1 << shift) - 1) >> shift;
			for (int32_t i = 0; i < (1 << (log_rows - 3)); ++i) for (int32_t j = 0; j < (1 << (log_columns - 3)); ++j)
				t.nonzeros[(nzpos + i * gw8 + j) * 3 + c] = (int8_t) qnz;
			const int32_t cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
			int32_t prev = nz <= (size >> 4);
			const uint16_t *order = SCAN ? nullptr : plan.pool_u16 + f.order_off[(pass * 13 + order_idx) * 3 + c];
			for (int32_t i = 1 << shift; nz > 0 && i < size; ++i) {
				const int32_t ctx = cctx + uni<UNI>((int32_t) t.nnz_ctx2[(nz + (1 << shift) - 1) >> shift]) + uni<UNI>((int32_t) t.freq_ctx2[i >> shift]) + prev;
				const int32_t ucoeff = code_symbol<UNI>(b, code, ctx, 0, plan.lz_window_size);
				if (ucoeff) {
					if (SCAN) {
						if (ev_at >= t.ev_end || !coeff_event_fits(unpack_signed_dev(ucoeff))) { bits_set_error(b, ERR_EVOF); break; }
						CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, unpack_signed_dev(ucoeff));
						plan.events[ev_at++] = ev;
					} else coeffs[uni<UNI>((uint32_t) order[i])] += (float) unpack_signed_dev(ucoeff);
				}
				prev = ucoeff != 0;
				nz -= prev;
				if (b.err) break;
			}
			if (SCAN) plan.block_events[4 * (size_t) (t.block_first + (uint32_t) k) + 1 + (size_t) c_yxb] = ev_at - chan_first;
			if (nz != 0) bits_set_error(b, ERR_COEF);
		}
	}
	if (!b.err) code_finish<UNI>(b, code);
	if (!b.err && f.check_section_end) bits_finish_section(b, f.single_declared_end);
	if (f.sections_have_trailer && plan.section_end_bit) plan.section_end_bit[&sec - plan.sections] = 8u * b.pos - (uint32_t) b.nbits;   // the extra channels' sub-image starts here
	return b.err;
}

// The same section decode as a flat state machine: ONE symbol-decode site per iteration, fed either
// with the context of a block's non-zero count or of its next coefficient. This is the form the
// throughput-oriented kernel runs with one section per LANE: the 64 lanes of a wavefront sit in
// different blocks and channels, but every iteration all of them decode one symbol together, so the
// expensive part (rANS / prefix step, hybrid integer, bit refill) never diverges; only the short
// per-phase prologue and epilogue run under partial exec masks. Same results as decode_hf_section.
template <bool SCAN>
J40_DEV uint32_t decode_hf_section_flat(const DevPlan &plan, const DevFrame &f, const DevCodeSpec &spec, const HfTables &t, int32_t pass, const DevSection &sec) {
	const DevLfGroup &gg = plan.lf_groups[sec.ggidx];
	DevBits b;
	bits_init<false>(b, plan.codestream, sec.byte_off, sec.size, sec.bit_off);
	const uint32_t preset = bits_u<false>(b, f.preset_bits);
	if ((int32_t) preset >= f.num_hf_presets) bits_set_error(b, ERR_RNGE);
	const int32_t ctxoff = 495 * f.nb_block_ctx * (int32_t) preset;
	DevCode code;
	code_init(code, spec, t.clusters, t.cluster_map, t.alias, t.prefix, t.window);
	const int32_t gw8 = sec.gw8;
	const int32_t nb_block_ctx = f.nb_block_ctx;
	const size_t cell64 = (size_t) gg.cell_base * 64;
	// iterator over (block, channel) and the coefficient loop state of the current one
	int32_t k = 0, c_yxb = 0;
	bool in_coeffs = false, done = t.nblocks == 0 || b.err != 0;
	int32_t x8 = 0, y8 = 0, nzpos = 0, log_rows = 3, log_columns = 3, order_idx = 0, shift = 0, size = 64, coeffoff = 0, bctx3 = 0;
	int32_t c = 1, bctx = 0, nz = 0, i = 0, prev = 0, cctx = 0;
	float *coeffs = nullptr;
	const uint16_t *order = nullptr;
	uint32_t ev_at = t.ev_first, chan_first = t.ev_first;
	while (!done) {
		int32_t ctx;
		if (!in_coeffs) {  // next symbol: number of non-zeros of (block k, channel c_yxb), j40.h:6959-6967
			if (c_yxb == 0) {
				const uint32_t *p = (const uint32_t *) (t.blocks + k);
				const uint32_t coeffoff_qfidx = p[0], w = p[1];
				const int32_t dctsel = (int32_t) ((w >> 10) & 31);
				if (SCAN) plan.block_events[4 * (size_t) (t.block_first + (uint32_t) k)] = ev_at;
				bctx3 = (int32_t) (w >> 16);
				x8 = (int32_t) (w & 31); y8 = (int32_t) ((w >> 5) & 31); nzpos = y8 * gw8 + x8;
				log_rows = DEV_DCT_SELECT[dctsel][0]; log_columns = DEV_DCT_SELECT[dctsel][1]; order_idx = DEV_DCT_SELECT[dctsel][2];
				shift = log_rows + log_columns - 6; size = 64 << shift;
				coeffoff = (int32_t) (coeffoff_qfidx & ~15u);
			}
			c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			bctx = (bctx3 >> (4 * c_yxb)) & 15;
			int32_t pnz;
			if (x8 > 0) pnz = y8 > 0 ? (t.nonzeros[(nzpos - 1) * 3 + c] + t.nonzeros[(nzpos - gw8) * 3 + c] + 1) >> 1 : t.nonzeros[(nzpos - 1) * 3 + c];
			else pnz = y8 > 0 ? t.nonzeros[(nzpos - gw8) * 3 + c] : 32;
			ctx = ctxoff + bctx + (pnz < 8 ? pnz : 4 + pnz / 2) * nb_block_ctx;
		} else {
			ctx = cctx + t.nnz_ctx2[(nz + (1 << shift) - 1) >> shift] + t.freq_ctx2[i >> shift] + prev;
		}
		const int32_t v = code_symbol<false>(b, code, ctx, 0, plan.lz_window_size);
		if (!in_coeffs) {
			nz = v;
			if (nz > (63 << shift)) { bits_set_error(b, ERR_COEF); break; }
			const int32_t qnz = (nz + (1 << shift) - 1) >> shift;
			for (int32_t r = 0; r < (1 << (log_rows - 3)); ++r) for (int32_t q = 0; q < (1 << (log_columns - 3)); ++q)
				t.nonzeros[(nzpos + r * gw8 + q) * 3 + c] = (int8_t) qnz;
			cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
			prev = nz <= (size >> 4);
			i = 1 << shift;
			if (!SCAN) {
				coeffs = (c == 0 ? plan.coeffs[0] : c == 1 ? plan.coeffs[1] : plan.coeffs[2]) + cell64 + coeffoff;
				order = plan.pool_u16 + f.order_off[(pass * 13 + order_idx) * 3 + c];
			}
			chan_first = ev_at;
			in_coeffs = nz > 0;
			if (SCAN && !in_coeffs) plan.block_events[4 * (size_t) (t.block_first + (uint32_t) k) + 1 + (size_t) c_yxb] = 0;
		} else {
			if (v) {
				if (SCAN) {
					if (ev_at >= t.ev_end || !coeff_event_fits(unpack_signed_dev(v))) bits_set_error(b, ERR_EVOF);
					else { CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, unpack_signed_dev(v)); plan.events[ev_at++] = ev; }
				} else coeffs[order[i]] += (float) unpack_signed_dev(v);
			}
			prev = v != 0;
			nz -= prev;
			++i;
			if (nz == 0) { in_coeffs = false; if (SCAN) plan.block_events[4 * (size_t) (t.block_first + (uint32_t) k) + 1 + (size_t) c_yxb] = ev_at - chan_first; }
			else if (i >= size) bits_set_error(b, ERR_COEF);   // ran out of coefficients with non-zeros left (j40.h:6996)
		}
		if (b.err) break;
		if (!in_coeffs && ++c_yxb == 3) { c_yxb = 0; done = ++k >= t.nblocks; }
	}
	if (!b.err) code_finish<false>(b, code);
	if (!b.err && f.check_section_end) bits_finish_section(b, f.single_declared_end);
	if (f.sections_have_trailer && plan.section_end_bit) plan.section_end_bit[&sec - plan.sections] = 8u * b.pos - (uint32_t) b.nbits;   // the extra channels' sub-image starts here
	return b.err;
}

// whole group, all passes, every table read straight from HBM (used by tests/hostsim and as the
// kernel's fallback when a frame's tables do not fit the LDS budget)
J40_DEV void decode_hf_group(const DevPlan &plan, int32_t g, bool flat = false) {
	const DevFrame &f = *plan.frame;
	HfTables t;
	t.block_ctx_map = plan.pool_u8 + plan.block_ctx_map_off;
	t.nnz_ctx2 = DEV_NNZ_CTX2; t.freq_ctx2 = DEV_FREQ_CTX2;
	t.blocks = plan.group_blocks + plan.group_block_start[g];
	t.nblocks = (int32_t) (plan.group_block_start[g + 1] - plan.group_block_start[g]);
	t.nonzeros = plan.nonzeros + (size_t) g * (32 * 32 * 3);
	t.window = plan.lz_window ? plan.lz_window + (size_t) g * plan.lz_window_size : nullptr;
	t.block_first = plan.group_block_start[g];
	t.ev_first = f.sparse_coeffs ? plan.ev_range[2 * g] : 0; t.ev_end = f.sparse_coeffs ? plan.ev_range[2 * g + 1] : 0;
	for (int32_t pass = 0; pass < f.num_passes; ++pass) {
		const DevCodeSpec &spec = plan.coeff_specs[pass];
		t.clusters = plan.clusters + spec.cluster_off; t.cluster_map = plan.pool_u8 + spec.cluster_map_off;
		t.alias = plan.pool_u64; t.prefix = plan.pool_i32;
		const DevSection &sec = plan.sections[pass * f.num_groups + g];
		uint32_t err;
		if (flat) err = f.sparse_coeffs ? decode_hf_section_flat<true>(plan, f, spec, t, pass, sec) : decode_hf_section_flat<false>(plan, f, spec, t, pass, sec);
		else err = f.sparse_coeffs ? decode_hf_section<true, false>(plan, f, spec, t, pass, sec) : decode_hf_section<false, false>(plan, f, spec, t, pass, sec);
		plan.status[pass * f.num_groups + g] = err;
	}
}

} // namespace j40hip
