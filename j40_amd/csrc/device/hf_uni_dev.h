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
//   * the bit reader is the lane decoder's (absolute position, reads that never fail, one refill point per symbol, renormalisation
//     and extra bits always taken with length 0 when they do not apply) on scalar registers: a symbol is straight-line scalar code
//     with a handful of branches (the general reader's refill logic, inlined four times per symbol, was most of its instructions).
// Same packed tables as k_hf_lanes (DevCodeSpec::lane_cfg_off, AnsEntry) and the lane decoder's error rules. tests/hostsim compiles this for the CPU (lane tables as arrays) and compares it with the general form.
#pragma once
#include "hf_lanes_dev.h"

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

// the two candidate contexts' bases, even lanes the one for "the coefficient was zero", odd lanes the other: a per-lane vector that
// changes only when a non-zero coefficient changes the count of those left; per symbol the frequency context is added and the
// context map read (spec_issue_base)
#ifdef __HIPCC__
struct SpecBase { int32_t v; };
J40_DEV void spec_base(SpecBase &s, int32_t a, int32_t b) { s.v = (threadIdx.x & 1) ? b : a; }
J40_DEV void spec_issue_base(SpecPair &p, const J40_LDS uint8_t *ctx_map, const SpecBase &s, int32_t fq, int32_t last_ctx) { const int32_t c = s.v + fq; p.v = (int32_t) ctx_map[c > last_ctx ? last_ctx : c]; }
#else
struct SpecBase { int32_t a, b; };
J40_DEV void spec_base(SpecBase &s, int32_t a, int32_t b) { s.a = a; s.b = b; }
J40_DEV void spec_issue_base(SpecPair &p, const uint8_t *ctx_map, const SpecBase &s, int32_t fq, int32_t last_ctx) { const int32_t ca = s.a + fq, cb = s.b + fq; p.a = ctx_map[ca > last_ctx ? last_ctx : ca]; p.b = ctx_map[cb > last_ctx ? last_ctx : cb]; }
#endif

struct UniTables {
	const J40_LDS uint8_t *ctx_map;     // [num_dist] context -> cluster
	const J40_LDS uint64_t *alias;      // [cluster << log_alpha | bucket], AnsEntry (entropy.hpp)
	LaneReg cfg;                        // [cluster] split_exp | msb_in_token << 4 | lsb_in_token << 8 | max_token << 12 (at most 64 clusters)
	LaneReg nnz2, freq2, dct;           // DEV_NNZ_CTX2, DEV_FREQ_CTX2, DctSelect -> log_rows | log_columns << 8 | order_idx << 16
	int32_t log_alpha, log_bucket, num_dist;
};

// LSB-first bit window like LaneBits (hf_lanes_dev.h) with wave-uniform state: the window, its fill and the position live in scalar
// registers; only the word requested ahead sits in a vector register until it is appended. Absolute position of the next unread bit
// = 8 * pos - nbits; reads never fail (the codestream buffer is padded), the caller compares the position with the section's end.
struct UBits {
	const J40_GLOBAL uint8_t *base;
	uint64_t bits;
	int32_t nbits;
	uint32_t pos;
	uint32_t ahead;
};
// Round 6, three things about how the compiler lays this decoder out on the scalar unit, each behind a switch for
// the A/B runs (tools/r06_l.sh):
//   J40_UNI_SLOAD  the codestream's words come through the SCALAR cache (s_load_dword: the address is wave-uniform, the buffer is not
//                  written while the kernel runs). As a vector load the word asked for ahead lived in a vector register across the
//                  symbol loop, the loop's back edge copied it, and the copy waited (s_waitcnt vmcnt(0)) for the load -- and for every
//                  event store before it: once per SYMBOL. With scalar loads nothing of the loop waits on the vector memory counter.
//   J40_UNI_SMUL   d * (state >> 12) as s_mul_i32: told that both factors are below 2^24 the compiler picks the 24-bit multiply, which
//                  exists on the vector unit only (two moves, v_mad_u32_u24, a nop and a v_readfirstlane per symbol).
//   J40_UNI_PREV   "was the coefficient non-zero" stays a scalar condition (s_cselect between the two candidate clusters' lanes) instead
//                  of becoming a lane index by way of a vector select and a v_readfirstlane.
#ifndef J40_UNI_SLOAD
#define J40_UNI_SLOAD 1
#endif
#ifndef J40_UNI_SMUL
#define J40_UNI_SMUL 1
#endif
#ifndef J40_UNI_PREV
#define J40_UNI_PREV 1
#endif
//   J40_UNI_FLOW   errors leave the symbol and the coefficient loop where they are found (a scalar compare and a branch each) instead of
//                  travelling as values: `nonzero = v != 0 && e2 == 0` and `e2 ? e2 : nz != 0 && i >= size ? "coef" : 0` came out as
//                  64-bit lane masks put together with s_cselect_b64 / s_and_b64 (some twenty-five scalar instructions a symbol); and the
//                  renormalisation is a branch taken once in four or five symbols instead of a 16-or-0-bit read every time.
#ifndef J40_UNI_FLOW
#define J40_UNI_FLOW 1
#endif
//   J40_UNI_TIGHT  the coefficient symbol written out inside its loop (decode_hf_section_fast): an error is `err = ...; break` -- a compare
//                  and a branch to the loop's exit, no value that says "all went well" merged over three paths --; the two candidate
//                  contexts keep what a zero coefficient does not change (the non-zero count's share: two table reads, seven scalar
//                  instructions) in a per-lane vector that only a non-zero coefficient rebuilds; the alias entry's address is a bit-field
//                  extract, a shift and a shift-add.
#ifndef J40_UNI_TIGHT
#define J40_UNI_TIGHT 1
#endif
template <bool UNI> J40_DEV uint32_t ub_load32(const J40_GLOBAL uint8_t *base, uint32_t pos) {
#ifdef __HIPCC__
	if (UNI && J40_UNI_SLOAD) return *(const __attribute__((address_space(4))) uint32_t *) (uintptr_t) (base + pos);
#endif
	return lane_load32(base, pos);
}
template <bool UNI> J40_DEV uint32_t ub_mul(uint32_t a, uint32_t b) {
#ifdef __HIPCC__
	if (UNI && J40_UNI_SMUL) { uint32_t r; asm("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b)); return r; }
#endif
	return a * b;
}
template <bool UNI> J40_DEV void ub_init(UBits &b, const J40_GLOBAL uint8_t *base, uint32_t start_bit) {
	b.base = base;
	const uint32_t pos0 = (start_bit >> 3) & ~3u, skip = start_bit - 8u * pos0;   // skip < 32
	b.bits = (uint64_t) (uni<UNI>(ub_load32<UNI>(base, pos0)) >> skip);
	b.nbits = 32 - (int32_t) skip;
	b.pos = pos0 + 4;
	b.ahead = ub_load32<UNI>(base, b.pos);
}
template <bool UNI> J40_DEV void ub_refill(UBits &b) {   // > 32 bits buffered afterwards
	if (b.nbits <= 32) {
		b.bits |= (uint64_t) uni<UNI>(b.ahead) << b.nbits;
		b.nbits += 32;
		b.pos += 4u;
		b.ahead = ub_load32<UNI>(b.base, b.pos);
	}
}
J40_DEV uint32_t ub_take(UBits &b, int32_t n) {   // 0 <= n <= 31, n <= nbits
	const uint32_t v = (uint32_t) b.bits & ((1u << n) - 1u);
	b.bits >>= n; b.nbits -= n;
	return v;
}
J40_DEV uint32_t ub_position(const UBits &b) { return 8u * b.pos - (uint32_t) b.nbits; }

// one symbol of cluster `cl`: rANS step (j40.h:2441-2466) + hybrid integer (j40.h:2313-2334), the branch-light form of lane_symbol
// (hf_lanes_dev.h) on scalars: renormalisation and extra bits are always taken, with length 0 when they do not apply; *err receives
// the error the reference would have raised first ("shrt" while renormalising, then "iovf", then "shrt" in the extra bits).
// The window holds > 32 bits on entry (ub_refill).
template <bool UNI>
J40_DEV int32_t uni_symbol(UBits &b, uint32_t &state, const UniTables &t, uint32_t cl, uint32_t end_bit, uint32_t *err) {
	if (state == 0) {   // first symbol of the section (j40.h:2445-2449)
		state = ub_take(b, 16); state |= ub_take(b, 16) << 16;
		ub_refill<UNI>(b);
	}
	const uint32_t idx = state & 0xfff, i = idx >> t.log_bucket, pos = idx & ((1u << t.log_bucket) - 1);
	const uint64_t e = uni64<UNI>(t.alias[(cl << t.log_alpha) + i]);
	const uint32_t m = (uint32_t) lr_get(t.cfg, (int32_t) cl);
	const uint32_t elo = (uint32_t) e, ehi = (uint32_t) (e >> 32);
	const bool aliased = pos >= (elo & 0xff);
	const int32_t token = (int32_t) (aliased ? (elo >> 20) & 0xff : i);
	const uint32_t offset = aliased ? (elo >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (ehi >> 9) & 0x1fff;
	state = ub_mul<UNI>(d, state >> 12) + offset + pos;
	if (J40_UNI_FLOW) {
		*err = 0;
		if (state < (1u << 16)) {
			const uint32_t low = ub_take(b, 16);
			state = (state << 16) | low;
			if (ub_position(b) > end_bit) { *err = ERR_SHRT; return 0; }
		}
		const int32_t split_exp = (int32_t) (m & 15), split = 1 << split_exp;
		if (token < split) return token;   // (most coefficient tokens are literal)
		const int32_t mt = (int32_t) (m >> 12);
		if (token > mt) { *err = ERR_IOVF; return 0; }
		const int32_t msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), in_token = msb + lsb;
		const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
		if (midbits > b.nbits) ub_refill<UNI>(b);   // rare: more than ~17 extra bits
		const int32_t mid = (int32_t) ub_take(b, midbits);
		if (ub_position(b) > end_bit) { *err = ERR_SHRT; return 0; }
		const int32_t top = 1 << msb;
		const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
		return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
	}
	const bool renorm = state < (1u << 16);
	const uint32_t low = ub_take(b, renorm ? 16 : 0);
	state = renorm ? (state << 16) | low : state;
	const bool short1 = ub_position(b) > end_bit;
	const int32_t split_exp = (int32_t) (m & 15), split = 1 << split_exp;
	if (token < split) { *err = short1 ? (uint32_t) ERR_SHRT : 0u; return token; }   // (a scalar branch: most coefficient tokens are literal)
	const int32_t mt = (int32_t) (m >> 12);
	const bool iovf = token > mt;
	const int32_t tok = iovf ? mt : token;
	const int32_t msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), in_token = msb + lsb;
	const int32_t midbits = split_exp - in_token + ((tok - split) >> in_token);
	if (midbits > b.nbits) ub_refill<UNI>(b);   // rare: more than ~17 extra bits
	const int32_t mid = (int32_t) ub_take(b, midbits);
	const bool short2 = ub_position(b) > end_bit;
	const int32_t top = 1 << msb;
	const int32_t lo = tok & ((1 << lsb) - 1), hi = (tok >> lsb) & (top - 1);
	*err = short1 ? (uint32_t) ERR_SHRT : iovf ? (uint32_t) ERR_IOVF : short2 ? (uint32_t) ERR_SHRT : 0u;
	return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
}

// The single-pass, sparse-coefficient section decode: the loop nest of decode_hf_section<true, UNI> (hf_dev.h) with the error rules of
// the lane decoder (decode_hf_sections_lane, hf_lanes_dev.h: the first error of a symbol ends the section), which tests/hostsim holds
// against the general decoder and the reference on damaged streams. `blocks`, `nonzeros` as in HfTables. Returns the section's status.
template <bool UNI>
J40_DEV uint32_t decode_hf_section_fast(const DevPlan &plan, const DevFrame &f, const UniTables &t, const HfTables &h, const DevSection &sec) {
	const uint32_t start_bit = 8u * sec.byte_off + sec.bit_off, end_bit = 8u * (sec.byte_off + sec.size);
	UBits b;
	ub_init<UNI>(b, (const J40_GLOBAL uint8_t *) plan.codestream, start_bit);
	ub_refill<UNI>(b);
	uint32_t err = 0;
	const uint32_t preset = ub_take(b, f.preset_bits);
	if (ub_position(b) > end_bit) err = ERR_SHRT;
	else if ((int32_t) preset >= f.num_hf_presets) err = ERR_RNGE;
	const int32_t ctxoff = 495 * f.nb_block_ctx * (int32_t) preset;
	const int32_t gw8 = sec.gw8, nb_block_ctx = f.nb_block_ctx, last_ctx = t.num_dist - 1;
	uint32_t state = 0;
	uint32_t ev_at = h.ev_first;
	for (int32_t k = 0; k < h.nblocks && !err; ++k) {
		const uint32_t blk_first = ev_at;
		uint32_t counts[3] = {0, 0, 0};
		const uint32_t w = uni<UNI>(((const uint32_t *) (h.blocks + k))[1]);   // DevGroupBlock: pos_dct | bctx3 << 16
		const int32_t pos_dct = (int32_t) (w & 0xffff), bctx3 = (int32_t) (w >> 16);
		const int32_t dctsel = pos_dct >> 10, x8 = pos_dct & 31, y8 = (pos_dct >> 5) & 31, nzpos = y8 * gw8 + x8;
		const int32_t di = lr_get(t.dct, dctsel);
		const int32_t log_rows = di & 255, log_columns = (di >> 8) & 255;
		const int32_t log_size = log_rows + log_columns, shift = log_size - 6, size = 1 << log_size, round = (1 << shift) - 1;
		for (int32_t c_yxb = 0; c_yxb < 3 && !err; ++c_yxb) {
			const int32_t c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			const uint32_t chan_first = ev_at;
			const int32_t bctx = (bctx3 >> (4 * c_yxb)) & 15;
			// number of non-zeros, predicted from the left / top blocks (j40.h:6959-6967)
			int32_t nz;
			if (x8 > 0) nz = y8 > 0 ? (h.nonzeros[(nzpos - 1) * 3 + c] + h.nonzeros[(nzpos - gw8) * 3 + c] + 1) >> 1 : h.nonzeros[(nzpos - 1) * 3 + c];
			else nz = y8 > 0 ? h.nonzeros[(nzpos - gw8) * 3 + c] : 32;
			nz = uni<UNI>(nz);
			const int32_t nzctx = ctxoff + bctx + (nz < 8 ? nz : 4 + nz / 2) * nb_block_ctx;
			uint32_t e2;
			ub_refill<UNI>(b);
			nz = uni_symbol<UNI>(b, state, t, uni<UNI>((uint32_t) t.ctx_map[nzctx]), end_bit, &e2);
			e2 = e2 ? e2 : nz > (63 << shift) ? (uint32_t) ERR_COEF : 0u;
			if (e2) { err = e2; break; }
			const int32_t qnz = (nz + round) >> shift;
			for (int32_t i = 0; i < (1 << (log_rows - 3)); ++i) for (int32_t j = 0; j < (1 << (log_columns - 3)); ++j)
				h.nonzeros[(nzpos + i * gw8 + j) * 3 + c] = (int8_t) qnz;
			const int32_t cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
			int32_t prev = nz <= (size >> 4);
			int32_t i = 1 << shift;
			if (nz > 0) {   // (i < size: the first coefficient position is 1 << shift < 64 << shift)
				if (J40_UNI_TIGHT) {
					int32_t nn_a = lr_get(t.nnz2, (nz + round) >> shift), nn_b = lr_get(t.nnz2, (nz - 1 + round) >> shift);
					uint32_t cl = uni<UNI>((uint32_t) t.ctx_map[cctx + nn_a + lr_get(t.freq2, i >> shift) + prev]);
					SpecBase sb;
					spec_base(sb, cctx + nn_a, cctx + nn_b + 1);
					const uint32_t pos_mask = (1u << t.log_bucket) - 1u, la3 = (uint32_t) t.log_alpha + 3u;
					const J40_LDS uint8_t *alias_bytes = (const J40_LDS uint8_t *) t.alias;
					for (;;) {
						SpecPair next;
						spec_issue_base(next, t.ctx_map, sb, lr_get(t.freq2, ((i + 1) >> shift) & 63), last_ctx);
						ub_refill<UNI>(b);
						if (state == 0) {   // (j40.h:2445-2449; a stream whose state comes to exactly zero mid-section)
							state = ub_take(b, 16); state |= ub_take(b, 16) << 16;
							ub_refill<UNI>(b);
						}
						// the symbol: rANS step (j40.h:2441-2466) ...
						const uint32_t bucket = (state >> t.log_bucket) & ((1u << t.log_alpha) - 1u), pos = state & pos_mask;
						const uint64_t e = uni64<UNI>(*(const J40_LDS uint64_t *) (alias_bytes + ((cl << la3) + (bucket << 3))));
						const uint32_t m = (uint32_t) lr_get(t.cfg, (int32_t) cl);
						const uint32_t elo = (uint32_t) e;
						const bool aliased = pos >= (elo & 0xff);
						const int32_t token = (int32_t) (aliased ? (elo >> 20) & 0xff : bucket);
						const uint32_t offset = aliased ? (elo >> 8) & 0xfff : 0u;
						const uint32_t d = (uint32_t) (e >> (aliased ? 28 : 41)) & 0x1fff;
						state = ub_mul<UNI>(d, state >> 12) + offset + pos;
						if (state < (1u << 16)) {
							const uint32_t low = ub_take(b, 16);
							state = (state << 16) | low;
							if (ub_position(b) > end_bit) { err = ERR_SHRT; break; }
						}
						// ... and hybrid integer (j40.h:2313-2334): most coefficient tokens are literal
						int32_t v = token;
						const int32_t split_exp = (int32_t) (m & 15), split = 1 << split_exp;
						if (token >= split) {
							if (token > (int32_t) (m >> 12)) { err = ERR_IOVF; break; }
							const int32_t msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), in_token = msb + lsb;
							const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
							if (midbits > b.nbits) ub_refill<UNI>(b);   // rare: more than ~17 extra bits
							const int32_t mid = (int32_t) ub_take(b, midbits);
							if (ub_position(b) > end_bit) { err = ERR_SHRT; break; }
							const int32_t top = 1 << msb;
							const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
							v = ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
						}
						const uint32_t c0 = spec_take(next, 0), c1 = spec_take(next, 1);
						if (v != 0) {
							const int32_t sv = unpack_signed_dev(v);
							if (ev_at >= h.ev_end || !coeff_event_fits(sv)) { err = ERR_EVOF; break; }
							CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, sv); plan.events[ev_at++] = ev;
							if (--nz == 0) break;
							cl = c1;
							nn_a = nn_b; nn_b = lr_get(t.nnz2, (nz - 1 + round) >> shift);
							spec_base(sb, cctx + nn_a, cctx + nn_b + 1);
						} else cl = c0;
						if (++i >= size) { err = ERR_COEF; break; }   // non-zeros left but no coefficient left (j40.h:6996)
					}
					counts[c_yxb] = ev_at - chan_first;
					continue;
				}
				uint32_t cl = uni<UNI>((uint32_t) t.ctx_map[cctx + lr_get(t.nnz2, (nz + round) >> shift) + lr_get(t.freq2, i >> shift) + prev]);
				for (;;) {
					// the two contexts the next coefficient can have (prev = 0: nz stays; prev = 1: one non-zero fewer), fetched now
					SpecPair next;
					{
						const int32_t fq = lr_get(t.freq2, ((i + 1) >> shift) & 63);
						int32_t ca = cctx + lr_get(t.nnz2, (nz + round) >> shift) + fq, cb = cctx + lr_get(t.nnz2, (nz - 1 + round) >> shift) + fq + 1;
						ca = ca > last_ctx ? last_ctx : ca; cb = cb > last_ctx ? last_ctx : cb;   // (past the block's last position: never used)
						spec_issue(next, t.ctx_map, ca, cb);
					}
					ub_refill<UNI>(b);
					const int32_t v = uni_symbol<UNI>(b, state, t, cl, end_bit, &e2);
					if (J40_UNI_FLOW) {
						if (e2) { err = e2; break; }
						const uint32_t c0 = spec_take(next, 0), c1 = spec_take(next, 1);
						if (v != 0) {
							const int32_t sv = unpack_signed_dev(v);
							if (ev_at >= h.ev_end || !coeff_event_fits(sv)) { err = ERR_EVOF; break; }
							CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, sv); plan.events[ev_at++] = ev;
							if (--nz == 0) break;
							cl = c1;
						} else cl = c0;
						if (++i >= size) { err = ERR_COEF; break; }   // non-zeros left but no coefficient left (j40.h:6996)
						continue;
					}
					const bool nonzero = v != 0 && e2 == 0;
					if (nonzero) {
						const int32_t sv = unpack_signed_dev(v);
						if (ev_at >= h.ev_end || !coeff_event_fits(sv)) e2 = ERR_EVOF;
						else { CoeffEvent ev; ev.packed = coeff_event_pack((uint32_t) i, sv); plan.events[ev_at++] = ev; }
					}
					if (J40_UNI_PREV) nz = v != 0 ? nz - 1 : nz;
					else { prev = v != 0; nz -= prev; }
					++i;
					e2 = e2 ? e2 : nz != 0 && i >= size ? (uint32_t) ERR_COEF : 0u;   // non-zeros left but no coefficient left (j40.h:6996)
					if (e2) { err = e2; break; }
					if (nz == 0) break;
					if (J40_UNI_PREV) { const uint32_t c0 = spec_take(next, 0), c1 = spec_take(next, 1); cl = v != 0 ? c1 : c0; }
					else cl = spec_take(next, prev);
				}
			}
			counts[c_yxb] = ev_at - chan_first;
		}
		if (!err) {   // the block's entry of DevPlan::block_events in one piece, like the lane decoder (a section that fails leaves its last block's entry as it was)
			uint32_t *be = plan.block_events + 4 * (size_t) (h.block_first + (uint32_t) k);
			be[0] = blk_first; be[1] = counts[0]; be[2] = counts[1]; be[3] = counts[2];
		}
	}
	if (!err) {   // j40.h:2884-2893: the final state, or the untouched initial state, must be 0x130000
		if (state == 0) { ub_refill<UNI>(b); state = ub_take(b, 16); state |= ub_take(b, 16) << 16; if (ub_position(b) > end_bit) err = ERR_SHRT; }
		if (!err && state != 0x130000) err = ERR_ANS;
	}
	if (!err && f.check_section_end) {   // single-section frames: zero padding up to the byte boundary, then no byte of the section left (j40.h:8203, 7796)
		const uint32_t at = ub_position(b), padn = (0u - at) & 7u;
		if (padn > (uint32_t) b.nbits) ub_refill<UNI>(b);
		if (ub_take(b, (int32_t) padn)) err = ERR_PAD0;
		else if (at + padn != 8u * f.single_declared_end) err = at + padn < 8u * f.single_declared_end ? (uint32_t) ERR_SHRT : (uint32_t) ERR_EXCS;
	}
	if (f.sections_have_trailer && plan.section_end_bit) plan.section_end_bit[&sec - plan.sections] = ub_position(b);   // the extra channels' sub-image starts here
	return err;
}

} // namespace j40hip
