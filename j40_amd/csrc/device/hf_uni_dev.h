// j40_amd/csrc/device/hf_uni_dev.h -- K1, latency form, fast path: ONE section per wavefront, every lane running the same decoder
// (decode_hf_section<true, true>, hf_dev.h, is the general form and the reference for results and status codes), specialised like
// the throughput form's fast path (hf_lanes_dev.h) for what VarDCT encoders write -- rANS, no LZ77, one pass -- and organised
// around the one thing that bounds it: the serial chain per symbol. A wavefront alone on its SIMD retires a dependent LDS read in
// ~90 cycles and a dependent scalar instruction in a few; the general form's coefficient symbol waits for four LDS reads one after
// the other (context tables -> context map -> cluster record -> alias entry). Here:
//   * the small tables live ACROSS THE LANES of vector registers (entry i in lane i: the two coefficient-context tables, the
//     DctSelect table, the clusters' hybrid-integer words) and are read with v_readlane -- no memory on the chain;
//   * the context of the NEXT coefficient depends on the current one only through `prev` (was it non-zero?), so both candidates
//     are formed while the current symbol is still being decoded and their clusters fetched by ONE LDS read (lane 0 reads the
//     context map for prev = 0, lane 1 for prev = 1); when the symbol is known its successor's cluster is a v_readlane away;
//   * that leaves one LDS read on a coefficient's chain: the alias entry (j40__ans_code, j40.h:2441-2461), addressed by cluster and state.
// Same packed tables as k_hf_lanes (DevCodeSpec::lane_cfg_off, AnsEntry); same bit reader and error order as the general form
// (entropy_dev.h). tests/hostsim compiles this for the CPU (lane tables as arrays) and compares it with the general form.
#pragma once
#include "hf_dev.h"

namespace j40hip {

#ifdef __HIPCC__
struct LaneReg { int32_t v; };   // entry i of a table of up to 64 words, held by lane i
J40_DEV int32_t lr_get(const LaneReg &r, int32_t i) { return __builtin_amdgcn_readlane(r.v, i); }
template <class F> J40_DEV void lr_fill(LaneReg &r, F f) { r.v = f((int32_t) (threadIdx.x & 63)); }
// the clusters of two candidate contexts with one LDS read: lane 0 (and every even lane) reads candidate a, the odd lanes b
struct SpecPair { int32_t v; };
J40_DEV void spec_issue(SpecPair &s, const J40_LDS uint8_t *ctx_map, int32_t a, int32_t b) { s.v = (int32_t) ctx_map[(threadIdx.x & 1) ? b : a]; }
J40_DEV uint32_t spec_take(const SpecPair &s, int32_t which) { return (uint32_t) __builtin_amdgcn_readlane(s.v, which); }
#else
struct LaneReg { int32_t v[64]; };
J40_DEV int32_t lr_get(const LaneReg &r, int32_t i) { return r.v[i & 63]; }
template <class F> J40_DEV void lr_fill(LaneReg &r, F f) { for (int32_t i = 0; i < 64; ++i) r.v[i] = f(i); }
struct SpecPair { int32_t a, b; };
J40_DEV void spec_issue(SpecPair &s, const uint8_t *ctx_map, int32_t a, int32_t b) { s.a = ctx_map[a]; s.b = ctx_map[b]; }
J40_DEV uint32_t spec_take(const SpecPair &s, int32_t which) { return (uint32_t) (which ? s.b : s.a); }
#endif

struct UniTables {
	const J40_LDS uint8_t *ctx_map;     // [num_dist] context -> cluster
	const J40_LDS uint64_t *alias;      // [cluster << log_alpha | bucket], AnsEntry (entropy.hpp)
	LaneReg cfg;                        // [cluster] split_exp | msb_in_token << 4 | lsb_in_token << 8 | max_token << 12 (at most 64 clusters)
	LaneReg nnz2, freq2, dct;           // DEV_NNZ_CTX2, DEV_FREQ_CTX2, DctSelect -> log_rows | log_columns << 8 | order_idx << 16
	int32_t log_alpha, log_bucket, num_dist;
};

// one symbol of cluster `cl`: rANS step (j40.h:2441-2466) + hybrid integer (j40.h:2313-2334); errors through `b` like code_symbol
template <bool UNI>
J40_DEV int32_t uni_symbol(DevBits &b, uint32_t &state, const UniTables &t, uint32_t cl) {
	if (state == 0) { state = bits_u<UNI>(b, 16); state |= bits_u<UNI>(b, 16) << 16; }
	const uint32_t idx = state & 0xfff, i = idx >> t.log_bucket, pos = idx & ((1u << t.log_bucket) - 1);
	const uint64_t e = uni64<UNI>(t.alias[(cl << t.log_alpha) + i]);
	const uint32_t m = (uint32_t) lr_get(t.cfg, (int32_t) cl);
	const bool aliased = pos >= (uint32_t) (e & 0xff);
	int32_t token = (int32_t) (aliased ? (uint32_t) (e >> 20) & 0xff : i);
	const uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
	state = d * (state >> 12) + offset + pos;
	if (state < (1u << 16)) state = (state << 16) | bits_u<UNI>(b, 16);
	// hybrid integer (hybrid_int_dev, entropy_dev.h)
	const int32_t split_exp = (int32_t) (m & 15), msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), max_token = (int32_t) (m >> 12);
	const int32_t split = 1 << split_exp;
	if (token < split) return token;
	if (token > max_token) { token = max_token; bits_set_error(b, ERR_IOVF); }
	const int32_t in_token = msb + lsb;
	const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
	const int32_t mid = (int32_t) bits_u<UNI>(b, midbits);
	const int32_t top = 1 << msb;
	const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
	return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
}

// the single-pass, sparse-coefficient section decode of decode_hf_section<true, UNI> (hf_dev.h) over the packed tables; `blocks`,
// `nonzeros` as there (HfTables). Returns the section's status.
template <bool UNI>
J40_DEV uint32_t decode_hf_section_fast(const DevPlan &plan, const DevFrame &f, const UniTables &t, const HfTables &h, const DevSection &sec) {
	DevBits b;
	bits_init<UNI>(b, plan.codestream, sec.byte_off, sec.size, sec.bit_off);
	const uint32_t preset = bits_u<UNI>(b, f.preset_bits);
	if ((int32_t) preset >= f.num_hf_presets) bits_set_error(b, ERR_RNGE);
	const int32_t ctxoff = 495 * f.nb_block_ctx * (int32_t) preset;
	const int32_t gw8 = sec.gw8, nb_block_ctx = f.nb_block_ctx, last_ctx = t.num_dist - 1;
	uint32_t state = 0;
	uint32_t ev_at = h.ev_first;
	for (int32_t k = 0; k < h.nblocks && !b.err; ++k) {
		plan.block_events[4 * (size_t) (h.block_first + (uint32_t) k)] = ev_at;
		const uint32_t w = uni<UNI>(((const uint32_t *) (h.blocks + k))[1]);   // DevGroupBlock: pos_dct | bctx3 << 16
		const int32_t pos_dct = (int32_t) (w & 0xffff), bctx3 = (int32_t) (w >> 16);
		const int32_t dctsel = pos_dct >> 10, x8 = pos_dct & 31, y8 = (pos_dct >> 5) & 31, nzpos = y8 * gw8 + x8;
		const int32_t di = lr_get(t.dct, dctsel);
		const int32_t log_rows = di & 255, log_columns = (di >> 8) & 255;
		const int32_t log_size = log_rows + log_columns, shift = log_size - 6, size = 1 << log_size, round = (1 << shift) - 1;
		for (int32_t c_yxb = 0; c_yxb < 3 && !b.err; ++c_yxb) {
			const int32_t c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			const uint32_t chan_first = ev_at;
			const int32_t bctx = (bctx3 >> (4 * c_yxb)) & 15;
			// number of non-zeros, predicted from the left / top blocks (j40.h:6959-6967)
			int32_t nz;
			if (x8 > 0) nz = y8 > 0 ? (h.nonzeros[(nzpos - 1) * 3 + c] + h.nonzeros[(nzpos - gw8) * 3 + c] + 1) >> 1 : h.nonzeros[(nzpos - 1) * 3 + c];
			else nz = y8 > 0 ? h.nonzeros[(nzpos - gw8) * 3 + c] : 32;
			nz = uni<UNI>(nz);
			const int32_t nzctx = ctxoff + bctx + (nz < 8 ? nz : 4 + nz / 2) * nb_block_ctx;
			nz = uni_symbol<UNI>(b, state, t, uni<UNI>((uint32_t) t.ctx_map[nzctx]));
			if (nz > (63 << shift)) { bits_set_error(b, ERR_COEF); break; }
			const int32_t qnz = (nz + round) >> shift;
			for (int32_t i = 0; i < (1 << (log_rows - 3)); ++i) for (int32_t j = 0; j < (1 << (log_columns - 3)); ++j)
				h.nonzeros[(nzpos + i * gw8 + j) * 3 + c] = (int8_t) qnz;
			const int32_t cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
			int32_t prev = nz <= (size >> 4);
			int32_t i = 1 << shift;
			if (nz > 0 && i < size) {
				int32_t ctx = cctx + lr_get(t.nnz2, (nz + round) >> shift) + lr_get(t.freq2, i >> shift) + prev;
				uint32_t cl = uni<UNI>((uint32_t) t.ctx_map[ctx]);
				for (;;) {
					// the two contexts the next coefficient can have (prev = 0: nz stays; prev = 1: one non-zero fewer), fetched now
					SpecPair next;
					{
						const int32_t fq = lr_get(t.freq2, ((i + 1) >> shift) & 63);
						int32_t ca = cctx + lr_get(t.nnz2, (nz + round) >> shift) + fq, cb = cctx + lr_get(t.nnz2, (nz - 1 + round) >> shift) + fq + 1;
						ca = ca > last_ctx ? last_ctx : ca; cb = cb > last_ctx ? last_ctx : cb;   // (past the block's last position: never used)
						spec_issue(next, t.ctx_map, ca, cb);
					}
					const int32_t ucoeff = uni_symbol<UNI>(b, state, t, cl);
					if (ucoeff) {
						if (ev_at >= h.ev_end || !coeff_event_fits(unpack_signed_dev(ucoeff))) { bits_set_error(b, ERR_EVOF); break; }
						CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, unpack_signed_dev(ucoeff));
						plan.events[ev_at++] = ev;
					}
					prev = ucoeff != 0;
					nz -= prev;
					if (b.err) break;
					++i;
					if (!(nz > 0 && i < size)) break;
					cl = spec_take(next, prev);
				}
			}
			plan.block_events[4 * (size_t) (h.block_first + (uint32_t) k) + 1 + (size_t) c_yxb] = ev_at - chan_first;
			if (nz != 0) bits_set_error(b, ERR_COEF);
		}
	}
	if (!b.err) {   // code_finish, j40.h:2884
		if (state) { if (state != 0x130000) bits_set_error(b, ERR_ANS); }
		else { if (bits_u<UNI>(b, 16) != 0x0000) bits_set_error(b, ERR_ANS); if (bits_u<UNI>(b, 16) != 0x0013) bits_set_error(b, ERR_ANS); }
	}
	if (!b.err && f.check_section_end) bits_finish_section(b, f.single_declared_end);
	if (f.sections_have_trailer && plan.section_end_bit) plan.section_end_bit[&sec - plan.sections] = 8u * b.pos - (uint32_t) b.nbits;   // the extra channels' sub-image starts here
	return b.err;
}

} // namespace j40hip
